/*
 * recording.hip — ONE recording demodulated on many lanes as overlapped tiles.
 *
 * Native counterpart of meteor_demod_amd/recording.py (same scheme, same
 * decisions, byte-identical result: tests/test_gpu_recording.py).  The reference
 * runs a recording as one serial recurrence (main.c:303-316); see DESIGN.md §3.1
 * for why tiles cannot equal it bit for bit and what is exact instead:
 *
 *   pilot   head of the recording as one stream from power-on state until the
 *           carrier loop has locked and converged  -> the reference's own bytes
 *   pass 1  every tile starts `pre` samples early from the pilot's end state
 *   match   rotation (Costas lock is 4-fold ambiguous) and one-symbol seam
 *           disagreement of each tile against its predecessor, measured on the
 *           samples both demodulated
 *   pass 2  stream i continues exactly from its pass-1 end state, turned into
 *           the pilot's rotation, with tile i+1
 *
 * All sample arithmetic is done by the demodulator kernels through the public
 * C-ABI of this library; the kernels here only compare and move int8 symbols.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <vector>

#include "meteor_demod_amd.h"

namespace {

struct TailPair {            /* what match_kernel compares for one tile */
	const int8_t *a; uint32_t a_cnt;          /* A: reference demodulation (predecessor / pass 2) */
	const int8_t *b; uint32_t b_cnt;          /* B: this tile's demodulation of the same samples  */
	int32_t b_rot;                            /* quarter turns applied to B before comparing      */
	int32_t force_weak;                       /* no overlap at all                                */
};

struct TileCopy {            /* what assemble_kernel moves for one tile */
	const int8_t *src; uint32_t keep; int32_t rot;
	const int8_t *head; int32_t head_rot;     /* optional symbol inserted in front (seam gap)     */
	uint64_t dst;                             /* symbol index in the output                       */
};

__device__ __forceinline__ void
rot_pair(int i, int q, int k, int &ri, int &rq)    /* (i + jq) * j^k */
{
	switch (k & 3) {
	case 0: ri = i; rq = q; break;
	case 1: ri = -q; rq = i; break;
	case 2: ri = -i; rq = -q; break;
	default: ri = q; rq = -i; break;
	}
}

/* One block per tile.  Tails are the last K+1 symbols of A and of B; three alignments:
 *   shift  0: a[1..K] vs b[1..K]     +1: a[0..K-1] vs b[1..K]     -1: a[1..K] vs b[0..K-1]
 * score(r) = Re( sum a * conj(b) * j^-r ): r=0 re, 1 im, 2 -re, 3 -im.  First maximum wins. */
__global__ void
match_kernel(const TailPair *pairs, int K, int32_t *shift_out, int32_t *rot_out, int32_t *weak_out)
{
	const TailPair p = pairs[blockIdx.x];
	__shared__ long long acc[7][64];
	long long s[7] = { 0, 0, 0, 0, 0, 0, 0 };
	auto load = [&](const int8_t *base, uint32_t cnt, int x, int rot, int &i, int &q) {
		const long long idx = (long long)cnt - (K + 1) + x;
		if (idx < 0) { i = 0; q = 0; return; }
		const int ri = base[2 * idx], rq = base[2 * idx + 1];
		rot_pair(ri, rq, rot, i, q);
	};
	for (int j = threadIdx.x; j < K; j += blockDim.x) {
		int a0i, a0q, a1i, a1q, b0i, b0q, b1i, b1q;
		load(p.a, p.a_cnt, j, 0, a0i, a0q);      load(p.a, p.a_cnt, j + 1, 0, a1i, a1q);
		load(p.b, p.b_cnt, j, p.b_rot, b0i, b0q); load(p.b, p.b_cnt, j + 1, p.b_rot, b1i, b1q);
		s[0] += a1i * b1i + a1q * b1q;  s[1] += a1q * b1i - a1i * b1q;      /* shift 0  */
		s[2] += a0i * b1i + a0q * b1q;  s[3] += a0q * b1i - a0i * b1q;      /* shift +1 */
		s[4] += a1i * b0i + a1q * b0q;  s[5] += a1q * b0i - a1i * b0q;      /* shift -1 */
		s[6] += a1i * a1i + a1q * a1q;                                      /* energy of a[1..K] */
	}
	for (int k = 0; k < 7; k++) acc[k][threadIdx.x] = s[k];
	__syncthreads();
	if (threadIdx.x == 0) {
		long long t[7];
		for (int k = 0; k < 7; k++) { t[k] = 0; for (unsigned l = 0; l < blockDim.x; l++) t[k] += acc[k][l]; }
		const int shifts[3] = { 0, 1, -1 };
		long long best = 0; int bs = 0, br = 0; bool have = false;
		for (int c = 0; c < 3; c++) {
			const long long re = t[2 * c], im = t[2 * c + 1];
			const long long sc4[4] = { re, im, -re, -im };
			int r = 0;
			for (int k = 1; k < 4; k++) if (sc4[k] > sc4[r]) r = k;
			if (!have || sc4[r] > best) { best = sc4[r]; bs = shifts[c]; br = r; have = true; }
		}
		const bool weak = (best * 2 < t[6]) || p.force_weak;
		shift_out[blockIdx.x] = weak ? 0 : bs;
		rot_out[blockIdx.x] = weak ? 0 : br;
		weak_out[blockIdx.x] = weak ? 1 : 0;
	}
}

/* OQPSK (see recording.py:match_rails): the rails are correlated separately because a tile locked +-90 degrees pairs
 * them one symbol apart.  Tails of K+2 symbols aligned at their ends; a = A[1..K]; for every quarter turn r the rails of
 * B * j^r are tried at shifts -1, 0, +1 each.  mode 1 = this; mode 2 = heads (both start on the same sample, complex
 * symbols, shifts 0 / +1 / -1 like match_kernel). */
__global__ void
match_rails_kernel(const TailPair *pairs, int K, int32_t *rot_out, int32_t *weak_out)
{
	const TailPair p = pairs[blockIdx.x];
	__shared__ long long acc[25][64];                      /* [r][rail][d] = 24 sums + energy */
	long long s[25];
	for (int k = 0; k < 25; k++) s[k] = 0;
	auto load = [&](const int8_t *base, uint32_t cnt, int x, int &i, int &q) {     /* x in 0..K+1 from the tail start */
		const long long idx = (long long)cnt - (K + 2) + x;
		if (idx < 0) { i = 0; q = 0; return; }
		i = base[2 * idx]; q = base[2 * idx + 1];
	};
	for (int j = threadIdx.x; j < K; j += blockDim.x) {
		int ai, aq; load(p.a, p.a_cnt, j + 1, ai, aq);
		for (int d = 0; d < 3; d++) {                       /* shift d-1 */
			int bi, bq; load(p.b, p.b_cnt, j + d, bi, bq);  /* b index 1 + (d-1) + j */
			const int mI[4] = { bi, -bq, -bi, bq }, mQ[4] = { bq, bi, -bq, -bi };
			for (int r = 0; r < 4; r++) { s[r * 6 + d] += ai * mI[r]; s[r * 6 + 3 + d] += aq * mQ[r]; }
		}
		s[24] += ai * ai + aq * aq;
	}
	for (int k = 0; k < 25; k++) acc[k][threadIdx.x] = s[k];
	__syncthreads();
	if (threadIdx.x == 0) {
		long long t[25];
		for (int k = 0; k < 25; k++) { t[k] = 0; for (unsigned l = 0; l < blockDim.x; l++) t[k] += acc[k][l]; }
		long long best = 0; int br = 0;
		for (int r = 0; r < 4; r++) {
			long long sI = t[r * 6], sQ = t[r * 6 + 3];
			for (int d = 1; d < 3; d++) { if (t[r * 6 + d] > sI) sI = t[r * 6 + d]; if (t[r * 6 + 3 + d] > sQ) sQ = t[r * 6 + 3 + d]; }
			if (r == 0 || sI + sQ > best) { best = sI + sQ; br = r; }
		}
		const bool weak = (best * 2 < t[24]) || p.force_weak;
		rot_out[blockIdx.x] = weak ? 0 : br;
		weak_out[blockIdx.x] = weak ? 1 : 0;
	}
}

__global__ void
match_heads_kernel(const TailPair *pairs, int K, int32_t *shift_out, int32_t *rot_out, int32_t *weak_out)
{
	const TailPair p = pairs[blockIdx.x];
	__shared__ long long acc[7][64];
	long long s[7] = { 0, 0, 0, 0, 0, 0, 0 };
	auto load = [&](const int8_t *base, uint32_t cnt, int x, int &i, int &q) {
		if ((uint32_t)x >= cnt) { i = 0; q = 0; return; }
		i = base[2 * x]; q = base[2 * x + 1];
	};
	for (int j = threadIdx.x; j < K; j += blockDim.x) {
		int a0i, a0q, a1i, a1q, b0i, b0q, b1i, b1q;
		load(p.a, p.a_cnt, j, a0i, a0q); load(p.a, p.a_cnt, j + 1, a1i, a1q);
		load(p.b, p.b_cnt, j, b0i, b0q); load(p.b, p.b_cnt, j + 1, b1i, b1q);
		s[0] += a0i * b0i + a0q * b0q;  s[1] += a0q * b0i - a0i * b0q;      /* shift 0:  a[0..K-1] vs b[0..K-1] */
		s[2] += a1i * b0i + a1q * b0q;  s[3] += a1q * b0i - a1i * b0q;      /* shift +1: a[1..K]   vs b[0..K-1] */
		s[4] += a0i * b1i + a0q * b1q;  s[5] += a0q * b1i - a0i * b1q;      /* shift -1: a[0..K-1] vs b[1..K]   */
		s[6] += a0i * a0i + a0q * a0q;
	}
	for (int k = 0; k < 7; k++) acc[k][threadIdx.x] = s[k];
	__syncthreads();
	if (threadIdx.x == 0) {
		long long t[7];
		for (int k = 0; k < 7; k++) { t[k] = 0; for (unsigned l = 0; l < blockDim.x; l++) t[k] += acc[k][l]; }
		const int shifts[3] = { 0, 1, -1 };
		long long best = 0; int bs = 0, br = 0; bool have = false;
		for (int c = 0; c < 3; c++) {
			const long long re = t[2 * c], im = t[2 * c + 1];
			const long long sc4[4] = { re, im, -re, -im };
			int r = 0;
			for (int k = 1; k < 4; k++) if (sc4[k] > sc4[r]) r = k;
			if (!have || sc4[r] > best) { best = sc4[r]; bs = shifts[c]; br = r; have = true; }
		}
		const bool weak = best * 2 < t[6];
		shift_out[blockIdx.x] = weak ? 0 : bs;
		rot_out[blockIdx.x] = weak ? 0 : br;
		weak_out[blockIdx.x] = weak ? 1 : 0;
	}
}

/* ---- per-tile carrier estimate (Doppler): z^4 of the samples has a line at 4x the carrier offset -------- */

template <int FMT> struct RawIQ;
template <> struct RawIQ<16> { typedef int16_t t; __device__ static float2 get(const void *p, uint64_t i) { const int16_t *q = static_cast<const int16_t *>(p) + 2 * i; return make_float2((float)q[0], (float)q[1]); } };
template <> struct RawIQ<8>  { typedef uint8_t t; __device__ static float2 get(const void *p, uint64_t i) { const uint8_t *q = static_cast<const uint8_t *>(p) + 2 * i; return make_float2((float)((int)q[0] - 128), (float)((int)q[1] - 128)); } };
template <> struct RawIQ<32> { typedef float t;   __device__ static float2 get(const void *p, uint64_t i) { const float *q = static_cast<const float *>(p) + 2 * i; return make_float2(q[0], q[1]); } };

/* One block (1024 threads) per tile, the whole estimate in one kernel and in LDS (no FFT library: hipFFT compiles its
 * kernels at run time, 1.6 s in every new process - more than a whole recording takes):
 *   1. mean of the window's nwin = NF * D samples;
 *   2. z^4 of the mean-free samples, summed in groups of D (boxcar decimation: the line sits within +-4 * fmax, far
 *      inside the decimated band; the boxcar's droop there is < 1 dB), Hann window, stored bit-reversed;
 *   3. in-place radix-2 FFT of NF <= 16384 points (128 KB of the CU's 160 KB LDS);
 *   4. largest magnitude among bins -kmax+1 .. kmax-1, parabolic interpolation -> rad per NCO step; quality = peak / mean
 *      magnitude of the searched band (noise alone: 3-4; a 12 dB signal: 40-50). */
template <int FMT>
__global__ void __launch_bounds__(1024)
carrier_line_kernel(const void *iq, uint64_t n_samples, const uint64_t *starts, int log2_nf, int decim, int kmax,
                    float hz_per_bin_over4, float rad_per_hz, float *freq_out, float *quality_out)
{
	extern __shared__ float2 spec[];                      /* NF complex floats */
	__shared__ float red[3][1024];
	__shared__ int redi[1024];
	const int NF = 1 << log2_nf, nwin = NF * decim, tid = threadIdx.x, nth = blockDim.x;
	const uint64_t s0 = starts[blockIdx.x];
	auto sample = [&](int k) {
		const uint64_t i = s0 + (uint64_t)k < n_samples ? s0 + (uint64_t)k : n_samples - 1;
		return RawIQ<FMT>::get(iq, i);
	};
	float sr = 0.0f, si = 0.0f;
	for (int k = tid; k < nwin; k += nth) { const float2 v = sample(k); sr += v.x; si += v.y; }
	red[0][tid] = sr; red[1][tid] = si;
	__syncthreads();
	for (int o = nth / 2; o > 0; o >>= 1) {
		if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
		__syncthreads();
	}
	const float mr = red[0][0] / nwin, mi = red[1][0] / nwin;
	__syncthreads();
	const float wstep = 2.0f / (float)(NF - 1);
	for (int m = tid; m < NF; m += nth) {
		float ar = 0.0f, ai = 0.0f;
		for (int d = 0; d < decim; d++) {
			float2 v = sample(m * decim + d);
			v.x -= mr; v.y -= mi;
			const float2 z2 = make_float2(v.x * v.x - v.y * v.y, 2.0f * v.x * v.y);
			ar += z2.x * z2.x - z2.y * z2.y; ai += 2.0f * z2.x * z2.y;
		}
		const float w = (0.5f - 0.5f * cospif(wstep * (float)m)) * 1e-12f;   /* Hann; the scale keeps |z|^4 of full-scale s16 far from overflow */
		spec[__brev((unsigned)m) >> (32 - log2_nf)] = make_float2(ar * w, ai * w);
	}
	__syncthreads();
	for (int st = 0; st < log2_nf; st++) {                /* decimation in time, natural-order output */
		const int half = 1 << st;
		for (int b = tid; b < NF / 2; b += nth) {
			const int j = b & (half - 1), i0 = ((b >> st) << (st + 1)) + j, i1 = i0 + half;
			float sn, cs;
			sincospif(-(float)j / (float)half, &sn, &cs);
			const float2 u = spec[i0], v = spec[i1];
			const float2 t = make_float2(v.x * cs - v.y * sn, v.x * sn + v.y * cs);
			spec[i0] = make_float2(u.x + t.x, u.y + t.y);
			spec[i1] = make_float2(u.x - t.x, u.y - t.y);
		}
		__syncthreads();
	}
	auto mag = [&](int k) { const float2 v = spec[(k + NF) & (NF - 1)]; return v.x * v.x + v.y * v.y; };
	float best = -1.0f, sum = 0.0f; int bidx = 0;
	for (int k = -kmax + tid; k <= kmax; k += nth) {
		const float m = mag(k);
		sum += sqrtf(m);
		if (m > best && k > -kmax && k < kmax) { best = m; bidx = k; }
	}
	red[0][tid] = best; redi[tid] = bidx; red[2][tid] = sum;
	__syncthreads();
	for (int o = nth / 2; o > 0; o >>= 1) {
		if (tid < o) {
			const float ov = red[0][tid + o]; const int oi = redi[tid + o];
			if (ov > red[0][tid] || (ov == red[0][tid] && oi < redi[tid])) { red[0][tid] = ov; redi[tid] = oi; }
			red[2][tid] += red[2][tid + o];
		}
		__syncthreads();
	}
	if (tid == 0) {
		const int k = redi[0];
		const float a = sqrtf(mag(k - 1)), b = sqrtf(mag(k)), c = sqrtf(mag(k + 1));
		const float den = a - 2.0f * b + c;
		const float delta = den != 0.0f ? 0.5f * (a - c) / den : 0.0f;
		freq_out[blockIdx.x] = ((float)k + delta) * hz_per_bin_over4 * rad_per_hz;
		quality_out[blockIdx.x] = b / (red[2][0] / (float)(2 * kmax + 1) + 1e-30f);
	}
}

/* One block per window: sample power (mean |z - mean|^2) of lens[w] samples from starts[w] (AGC seeds). */
template <int FMT>
__global__ void
window_power_kernel(const void *iq, const uint64_t *starts, const uint32_t *lens, float *power_out)
{
	const uint64_t s0 = starts[blockIdx.x];
	const uint32_t n = lens[blockIdx.x];
	__shared__ float red[2][256];
	float sr = 0.0f, si = 0.0f;
	for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) { const float2 v = RawIQ<FMT>::get(iq, s0 + k); sr += v.x; si += v.y; }
	red[0][threadIdx.x] = sr; red[1][threadIdx.x] = si;
	__syncthreads();
	for (int o = 128; o > 0; o >>= 1) {
		if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
		__syncthreads();
	}
	const float mr = n ? red[0][0] / n : 0.0f, mi = n ? red[1][0] / n : 0.0f;
	__syncthreads();
	float pw = 0.0f;
	for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
		float2 v = RawIQ<FMT>::get(iq, s0 + k);
		v.x -= mr; v.y -= mi;
		pw += v.x * v.x + v.y * v.y;
	}
	red[0][threadIdx.x] = pw;
	__syncthreads();
	for (int o = 128; o > 0; o >>= 1) {
		if ((int)threadIdx.x < o) red[0][threadIdx.x] += red[0][threadIdx.x + o];
		__syncthreads();
	}
	if (threadIdx.x == 0) power_out[blockIdx.x] = n ? red[0][0] / n : 0.0f;
}

__global__ void
assemble_kernel(const TileCopy *tiles, int8_t *out)
{
	const TileCopy t = tiles[blockIdx.x];
	int8_t *dst = out + 2 * t.dst;
	if (t.head && threadIdx.x == 0) {
		int i, q; rot_pair(t.head[0], t.head[1], t.head_rot, i, q);
		dst[0] = (int8_t)i; dst[1] = (int8_t)q;
	}
	if (t.head) dst += 2;
	for (uint32_t k = threadIdx.x; k < t.keep; k += blockDim.x) {
		int i, q; rot_pair(t.src[2 * k], t.src[2 * k + 1], t.rot, i, q);
		dst[2 * k] = (int8_t)i; dst[2 * k + 1] = (int8_t)q;
	}
}

/* ---- host side ----------------------------------------------------------------------------- */

struct DevMem {                      /* frees everything on scope exit */
	std::vector<void *> p;
	~DevMem() { for (void *q : p) (void)hipFree(q); }
	template <typename T> int alloc(T **out, size_t n) {
		void *q = nullptr;
		if (hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return MDEMOD_ERR_NOMEM;
		p.push_back(q); *out = static_cast<T *>(q); return MDEMOD_OK;
	}
};

struct Ctx {                         /* owns a demodulator context */
	mdemod_ctx *c = nullptr;
	~Ctx() { if (c) mdemod_destroy(c); }
};

#define TRY(expr) do { int rc_ = (expr); if (rc_ < 0) return rc_; } while (0)
#define HTRY(expr) do { if ((expr) != hipSuccess) { fprintf(stderr, "meteor_demod_amd: %s failed (%s:%d)\n", #expr, __FILE__, __LINE__); return MDEMOD_ERR_HIP; } } while (0)

template <typename T>
int
upload(DevMem &m, const std::vector<T> &h, T **dev, hipStream_t st)
{
	TRY(m.alloc(dev, h.size()));
	if (!h.empty()) HTRY(hipMemcpyAsync(*dev, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st));
	return MDEMOD_OK;
}

int
counts_of(mdemod_ctx *c, uint32_t n, std::vector<uint32_t> &out, hipStream_t st, std::vector<mdemod_status> *status_out = nullptr)
{
	std::vector<mdemod_status> s(n);
	TRY(mdemod_get_status(c, 0, n, s.data(), st));
	out.resize(n);
	for (uint32_t i = 0; i < n; i++) {
		if (s[i].overflow) return MDEMOD_ERR_OVERFLOW;
		out[i] = s[i].symbols_this_call;
	}
	if (status_out) status_out->swap(s);
	return MDEMOD_OK;
}

int
run_match(DevMem &m, const std::vector<TailPair> &pairs, int K, std::vector<int32_t> &shift, std::vector<int32_t> &rot,
          std::vector<int32_t> &weak, hipStream_t st, int mode = 0)
{
	const size_t T = pairs.size();
	if (T == 0) { shift.clear(); rot.clear(); weak.clear(); return MDEMOD_OK; }
	TailPair *d_pairs; int32_t *d_out;
	TRY(upload(m, pairs, &d_pairs, st));
	TRY(m.alloc(&d_out, 3 * T));
	HTRY(hipMemsetAsync(d_out, 0, 3 * T * sizeof(int32_t), st));
	if (mode == 1) hipLaunchKernelGGL(match_rails_kernel, dim3((unsigned)T), dim3(64), 0, st, d_pairs, K, d_out + T, d_out + 2 * T);
	else if (mode == 2) hipLaunchKernelGGL(match_heads_kernel, dim3((unsigned)T), dim3(64), 0, st, d_pairs, K, d_out, d_out + T, d_out + 2 * T);
	else hipLaunchKernelGGL(match_kernel, dim3((unsigned)T), dim3(64), 0, st, d_pairs, K, d_out, d_out + T, d_out + 2 * T);
	HTRY(hipGetLastError());
	std::vector<int32_t> h(3 * T);
	HTRY(hipMemcpyAsync(h.data(), d_out, 3 * T * sizeof(int32_t), hipMemcpyDeviceToHost, st));
	HTRY(hipStreamSynchronize(st));
	shift.assign(h.begin(), h.begin() + T); rot.assign(h.begin() + T, h.begin() + 2 * T); weak.assign(h.begin() + 2 * T, h.end());
	return MDEMOD_OK;
}

/* One window of the reference's AGC in closed form (recording.py:_agc_step; agc.c:13-25). */
double
agc_step(double g, double c, double power, double nsym)
{
	const double gstar = c / std::sqrt(std::max(power, 1e-30));
	return gstar + (g - gstar) * std::exp(-std::min(50.0, 1e-4 * 190.0 / std::max(gstar, 1e-30) * nsym));
}

/* recording.py:fit_agc_calibration */
double
fit_agc_calibration(const std::vector<double> &gains, const std::vector<double> &powers, const std::vector<double> &nsyms)
{
	const size_t J = gains.size() - 1;
	const double c0 = gains[J] * std::sqrt(std::max(powers[J], 1e-30));
	const size_t j0 = J > 8 ? J - 8 : 0;
	if (J == j0 || !std::isfinite(c0) || c0 <= 0) return c0;
	auto model = [&](double c) {
		double g = gains[j0];
		for (size_t j = j0 + 1; j <= J; j++) g = agc_step(g, c, powers[j], nsyms[j]);
		return g;
	};
	double lo = c0 / 8, hi = c0 * 8;
	if (!(model(lo) <= gains[J] && gains[J] <= model(hi))) return c0;
	for (int it = 0; it < 50; it++) {
		const double mid = 0.5 * (lo + hi);
		if (model(mid) < gains[J]) lo = mid; else hi = mid;
	}
	return 0.5 * (lo + hi);
}

struct PilotBlock { uint64_t start; uint32_t len; double gain_after; uint64_t symbols_after; };

} /* namespace */

/* Window geometry of the carrier estimator: the z^4 line sits within +-4 * 0.33 rad/symbol; decimate by D (boxcar, in the
 * kernel) as far as that band stays inside 80 % of the decimated one, keep the transform within the 16384 points that fit
 * in LDS.  window_samples is rounded down to a power of two in [4096, 2^17] (and further if the band forbids decimation). */
static void
carrier_window(const mdemod_params *params, uint32_t window_samples, int *nwin, int *decim, int *log2_nf, int *kmax)
{
	const double symrate = params->symrate, fs = params->samplerate;
	int n = 4096;
	while (n * 2 <= static_cast<int>(std::min<uint32_t>(window_samples, 1u << 17))) n *= 2;
	const double band_hz = 4 * 0.33 * symrate / (2 * 3.141592653589793);
	int d = 1;
	while (d < 16 && fs / (2.0 * (d * 2)) >= 1.25 * band_hz) d *= 2;
	while (n / d > 16384) n /= 2;                        /* only when the band forbids more decimation */
	while (d > 1 && n / d < 4096) d /= 2;
	int l2 = 0;
	while ((1 << l2) < n / d) l2++;
	*nwin = n; *decim = d; *log2_nf = l2;
	*kmax = std::min(n / d / 2 - 2, static_cast<int>(band_hz / fs * n) + 2);
}

extern "C" uint32_t
mdemod_carrier_window_samples(const mdemod_params *params, uint32_t window_samples)
{
	if (!params || params->samplerate <= 0 || params->symrate <= 0) return 0;
	int nwin, decim, l2, kmax;
	carrier_window(params, window_samples, &nwin, &decim, &l2, &kmax);
	return static_cast<uint32_t>(nwin);
}

extern "C" int
mdemod_estimate_carrier(const mdemod_params *params, const void *iq_dev, uint64_t n_samples,
                        const uint64_t *starts_dev, uint32_t n_windows, uint32_t window_samples,
                        float *freq_dev, float *quality_dev, void *hip_stream)
{
	if (!params || !iq_dev || !starts_dev || !freq_dev || !quality_dev || n_samples == 0) return MDEMOD_ERR_PARAM;
	if (params->samplerate <= 0 || params->symrate <= 0 || (params->bps != 8 && params->bps != 16 && params->bps != 32)) return MDEMOD_ERR_PARAM;
	if (n_windows == 0) return MDEMOD_OK;
	hipStream_t st = static_cast<hipStream_t>(hip_stream);
	int nwin, decim, log2_nf, kmax;
	carrier_window(params, window_samples, &nwin, &decim, &log2_nf, &kmax);
	const double symrate = params->symrate, fs = params->samplerate;
	const dim3 grid(n_windows);
	const size_t lds = (static_cast<size_t>(nwin) / decim) * sizeof(float2);
	const float hz_per_bin_over4 = static_cast<float>(fs / nwin / 4.0);
	const float rad_per_hz = static_cast<float>(2 * 3.141592653589793 / (symrate * (params->oqpsk ? 2 : 1)));   /* OQPSK: NCO steps twice a symbol */
#define LAUNCH_LINE(F) do { \
		HTRY(hipFuncSetAttribute(reinterpret_cast<const void *>(carrier_line_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds))); \
		hipLaunchKernelGGL(carrier_line_kernel<F>, grid, dim3(1024), lds, st, iq_dev, n_samples, starts_dev, log2_nf, decim, kmax, \
		                   hz_per_bin_over4, rad_per_hz, freq_dev, quality_dev); } while (0)
	switch (params->bps) {
	case 16: LAUNCH_LINE(16); break;
	case 8:  LAUNCH_LINE(8); break;
	default: LAUNCH_LINE(32); break;
	}
#undef LAUNCH_LINE
	HTRY(hipGetLastError());
	return MDEMOD_OK;
}

extern "C" void
mdemod_recording_default_opts(mdemod_recording_opts *o)
{
	if (!o) return;
	o->tile_samples = 0;                     /* 0 = 20 536 symbols worth of samples (65 600 at 72k in 230 kS/s), kept off powers of two */
	o->pre_samples = 0xFFFFFFFFu;            /* 0xFFFFFFFF = 5 129 symbols worth of samples (16 384 at 72k in 230 kS/s) */
	o->pilot_block = 65536; o->pilot_margin_symbols = 20000;
	o->max_pilot_samples = 1ull << 22; o->match_symbols = 192; o->refine = 1; o->carrier_seed = 1; o->reserved = 0;
}

extern "C" int
mdemod_demodulate_recording(const mdemod_params *params, const mdemod_recording_opts *opts_in,
                            const void *iq_dev, uint64_t n_samples,
                            int8_t *soft_dev, uint64_t soft_cap_symbols,
                            mdemod_recording_report *rep, void *hip_stream)
{
	if (!params || !iq_dev || !soft_dev || !rep) return MDEMOD_ERR_PARAM;
	mdemod_recording_opts o;
	if (opts_in) o = *opts_in; else mdemod_recording_default_opts(&o);
	if (params->oqpsk && !o.refine) return MDEMOD_ERR_PARAM;     /* OQPSK needs the state rotation pass: DESIGN.md 3.1 */
	{	/* same rule as recording.py:default_tiling() */
		const double osf = static_cast<double>(params->samplerate) / static_cast<double>(params->symrate);
		if (o.tile_samples == 0) {
			o.tile_samples = std::max<uint32_t>(4096, static_cast<uint32_t>(20536 * osf) / 64 * 64);
			if ((o.tile_samples & (o.tile_samples - 1)) == 0) o.tile_samples += 64;
		}
		/* OQPSK: twice the warm-up (its carrier loop has half the bandwidth and may still be re-locking after 5 129 symbols;
		   recording.py:default_tiling, profiles/r01_rotation_jump_cases.md) */
		if (o.pre_samples == 0xFFFFFFFFu) o.pre_samples = static_cast<uint32_t>((params->oqpsk ? 10258 : 5129) * osf);
	}
	if (!o.tile_samples || !o.pilot_block || !o.match_symbols) return MDEMOD_ERR_PARAM;
	hipStream_t st = static_cast<hipStream_t>(hip_stream);
	memset(rep, 0, sizeof(*rep));
	rep->first_lock_symbol = -1;
	const size_t sb = 2 * static_cast<size_t>(params->bps) / 8;
	const unsigned char *iq = static_cast<const unsigned char *>(iq_dev);
	const int K = static_cast<int>(o.match_symbols);

	const auto t_start = std::chrono::steady_clock::now();
	auto seconds_since = [](std::chrono::steady_clock::time_point t0) {
		return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	};
	/* ---- pilot: the reference's own serial run of the head -------------------------------- */
	mdemod_params pp = *params; pp.n_streams = 1;
	Ctx pilot;
	TRY(mdemod_create(&pp, &pilot.c));
	uint64_t pos = 0, nsym = 0; bool have_lock = false; uint64_t locked_at = 0;
	std::vector<PilotBlock> pilot_blocks;                /* AGC calibration (carrier_seed == 1) */
	mdemod_stream_state seed;
	memset(&seed, 0, sizeof(seed));
	TRY(mdemod_get_state(pilot.c, 0, &seed, st));
	while (pos < n_samples) {
		const uint32_t b = static_cast<uint32_t>(std::min<uint64_t>(o.pilot_block, n_samples - pos));
		/* the block may produce up to one symbol per sample (mdemod_max_symbols); the caller's buffer only has to hold what
		   it does produce: the kernel checks the capacity it is given and reports an overflow */
		const uint64_t cap = std::min<uint64_t>(mdemod_max_symbols(pilot.c, b), soft_cap_symbols - std::min(nsym, soft_cap_symbols));
		if (cap == 0) return MDEMOD_ERR_OVERFLOW;
		TRY(mdemod_process_device_uniform(pilot.c, iq + pos * sb, 0, b, soft_dev + 2 * nsym, cap, static_cast<uint32_t>(cap), st));
		mdemod_status s1;
		TRY(mdemod_get_status(pilot.c, 0, 1, &s1, st));
		if (s1.overflow) return MDEMOD_ERR_OVERFLOW;
		nsym += s1.symbols_this_call;
		TRY(mdemod_get_state(pilot.c, 0, &seed, st));
		pilot_blocks.push_back(PilotBlock{pos, b, seed.agc_gain, seed.n_symbols});
		pos += b;
		if (seed.pll_locked && !have_lock) { have_lock = true; locked_at = seed.n_symbols; }
		if (!seed.pll_locked) have_lock = false;
		/* ... and not before the reference's AGC has settled: its step is absolute (agc.c:13-25), 6 time constants =
		   6 * gain / (1e-4 * 190) symbols - nothing for s16-scale input, ~200 k symbols for float input around +-1
		   (recording.py:agc_settle_symbols) */
		const double agc_settle = 6.0 * static_cast<double>(seed.agc_gain) / (1e-4 * 190.0);
		if (have_lock && seed.n_symbols - locked_at >= o.pilot_margin_symbols && static_cast<double>(seed.n_symbols) >= agc_settle) break;
		if (pos >= o.max_pilot_samples) break;
	}
	rep->pilot_samples = pos; rep->pilot_symbols = seed.n_symbols;
	rep->pilot_locked = seed.pll_locked; rep->first_lock_symbol = seed.first_lock_symbol;
	rep->samples_demodulated = pos;
	const uint64_t n_pilot_sym = nsym;
	rep->pilot_seconds = seconds_since(t_start);
	const auto t_tiles = std::chrono::steady_clock::now();

	/* ---- plan ------------------------------------------------------------------------------ */
	std::vector<uint64_t> starts, lens, pres;
	for (uint64_t s0 = pos; s0 < n_samples; s0 += o.tile_samples) {
		starts.push_back(s0);
		lens.push_back(std::min<uint64_t>(o.tile_samples, n_samples - s0));
		pres.push_back(std::min<uint64_t>(o.pre_samples, s0));
	}
	const size_t T = starts.size();
	rep->n_tiles = static_cast<uint32_t>(T);
	if (T == 0) { rep->n_symbols = n_pilot_sym; return MDEMOD_OK; }

	std::vector<float> seed_hist(2 * static_cast<size_t>(mdemod_history_len(pilot.c)));
	TRY(mdemod_get_history(pilot.c, 0, seed_hist.data(), st));

	mdemod_params bp = *params; bp.n_streams = static_cast<uint32_t>(T);
	Ctx bank;
	TRY(mdemod_create(&bp, &bank.c));
	DevMem mem;
	const uint64_t cap_pre = std::max<uint64_t>(1, mdemod_max_symbols(bank.c, *std::max_element(pres.begin(), pres.end())));
	const uint64_t cap = mdemod_max_symbols(bank.c, *std::max_element(lens.begin(), lens.end()));
	int8_t *soft_pre, *soft1, *soft2 = nullptr;
	TRY(mem.alloc(&soft_pre, T * cap_pre * 2));
	TRY(mem.alloc(&soft1, T * cap * 2));

	auto launch = [&](const std::vector<uint64_t> &off, const std::vector<uint64_t> &cnt, int8_t *soft, uint64_t stride,
	                  std::vector<uint32_t> &produced, std::vector<mdemod_status> *status_out = nullptr) -> int {
		std::vector<uint32_t> c32(cnt.begin(), cnt.end());
		uint64_t *d_off; uint32_t *d_cnt;
		TRY(upload(mem, off, &d_off, st));
		TRY(upload(mem, c32, &d_cnt, st));
		TRY(mdemod_process_device(bank.c, iq_dev, d_off, d_cnt, soft, stride, static_cast<uint32_t>(stride), st));
		TRY(counts_of(bank.c, static_cast<uint32_t>(T), produced, st, status_out));
		for (uint64_t c : cnt) rep->samples_demodulated += c;
		return MDEMOD_OK;
	};

	/* ---- pass 1 ---------------------------------------------------------------------------- */
	TRY(mdemod_set_state_all(bank.c, &seed, st));
	std::vector<uint64_t> off_pre(T);
	for (size_t i = 0; i < T; i++) off_pre[i] = starts[i] - pres[i];
	if (o.carrier_seed == 1) {
		/* Doppler: every tile starts from its own carrier estimate (see recording.py:carrier_estimates) */
		const int nfft = static_cast<int>(mdemod_carrier_window_samples(params, static_cast<uint32_t>(std::min<uint64_t>(
		                     static_cast<uint64_t>(o.tile_samples) + o.pre_samples, 1u << 17))));   /* window in samples */
		float consts[8];
		TRY(mdemod_get_loop_constants(bank.c, consts));
		const float fmax = consts[2];
		const double symrate = params->symrate, fs = params->samplerate;
		std::vector<float> fmid(T), qual(T);
		uint64_t *d_starts; float *d_freq, *d_qual;
		/* windows that would run past the end of the recording are moved back (the last tiles) */
		std::vector<uint64_t> wstart(T);
		for (size_t i = 0; i < T; i++)
			wstart[i] = std::min<uint64_t>(off_pre[i], n_samples >= static_cast<uint64_t>(nfft) ? n_samples - nfft : 0);
		TRY(upload(mem, wstart, &d_starts, st));
		TRY(mem.alloc(&d_freq, T));
		TRY(mem.alloc(&d_qual, T));
		TRY(mdemod_estimate_carrier(params, iq_dev, n_samples, d_starts, static_cast<uint32_t>(T), static_cast<uint32_t>(nfft), d_freq, d_qual, st));
		HTRY(hipMemcpyAsync(fmid.data(), d_freq, T * sizeof(float), hipMemcpyDeviceToHost, st));
		HTRY(hipMemcpyAsync(qual.data(), d_qual, T * sizeof(float), hipMemcpyDeviceToHost, st));
		HTRY(hipStreamSynchronize(st));
		/* AGC gain seeds (recording.py: window_power, fit_agc_calibration, agc_trajectory): g* = c / sqrt(sample power),
		   c fitted on the pilot's last blocks, then the closed-form recursion over the tiles' bodies */
		{
			const size_t nb = std::min<size_t>(10, pilot_blocks.size());
			const size_t b0 = pilot_blocks.size() - nb;
			std::vector<uint64_t> ws; std::vector<uint32_t> wl;
			for (size_t j = b0; j < pilot_blocks.size(); j++) { ws.push_back(pilot_blocks[j].start); wl.push_back(pilot_blocks[j].len); }
			for (size_t i = 0; i < T; i++) { ws.push_back(starts[i]); wl.push_back(static_cast<uint32_t>(lens[i])); }
			uint64_t *d_ws; uint32_t *d_wl; float *d_wp;
			TRY(upload(mem, ws, &d_ws, st));
			TRY(upload(mem, wl, &d_wl, st));
			TRY(mem.alloc(&d_wp, ws.size()));
			const dim3 grid(static_cast<unsigned>(ws.size()));
			switch (params->bps) {
			case 16: hipLaunchKernelGGL(window_power_kernel<16>, grid, dim3(256), 0, st, iq_dev, d_ws, d_wl, d_wp); break;
			case 8:  hipLaunchKernelGGL(window_power_kernel<8>, grid, dim3(256), 0, st, iq_dev, d_ws, d_wl, d_wp); break;
			default: hipLaunchKernelGGL(window_power_kernel<32>, grid, dim3(256), 0, st, iq_dev, d_ws, d_wl, d_wp); break;
			}
			HTRY(hipGetLastError());
			std::vector<float> wp(ws.size());
			HTRY(hipMemcpyAsync(wp.data(), d_wp, wp.size() * sizeof(float), hipMemcpyDeviceToHost, st));
			HTRY(hipStreamSynchronize(st));
			std::vector<double> blk_gain(nb), blk_power(nb), blk_syms(nb);
			for (size_t j = 0; j < nb; j++) {
				const PilotBlock &pb = pilot_blocks[b0 + j];
				const uint64_t before = (b0 + j) ? pilot_blocks[b0 + j - 1].symbols_after : 0;
				blk_gain[j] = pb.gain_after; blk_power[j] = wp[j]; blk_syms[j] = static_cast<double>(pb.symbols_after - before);
			}
			const double c = fit_agc_calibration(blk_gain, blk_power, blk_syms);
			double g = seed.agc_gain;
			std::vector<float> gains(T);
			for (size_t i = 0; i < T; i++) {
				gains[i] = static_cast<float>(g);
				g = agc_step(g, c, wp[nb + i], static_cast<double>(lens[i]) * symrate / fs);
			}
			float *d_gain;
			TRY(upload(mem, gains, &d_gain, st));
			TRY(mdemod_set_gain_seeds(bank.c, d_gain, st));
		}
		/* tiles without a clear line (fade, interference) take their good neighbours' estimate, interpolated over the
		   tile index; with no good tile at all, the pilot's frequency (recording.py:fill_weak_estimates) */
		{
			const float min_quality = 8.0f;
			std::vector<size_t> good;
			for (size_t i = 0; i < T; i++) if (qual[i] >= min_quality) good.push_back(i);
			if (good.empty()) {
				for (size_t i = 0; i < T; i++) fmid[i] = seed.pll_freq;
			} else if (good.size() < T) {
				size_t g = 0;
				for (size_t i = 0; i < T; i++) {
					if (qual[i] >= min_quality) continue;
					while (g + 1 < good.size() && good[g + 1] < i) g++;
					if (i < good.front()) fmid[i] = fmid[good.front()];
					else if (i > good.back()) fmid[i] = fmid[good.back()];
					else {
						const size_t lo = good[g], hi = good[g + 1];
						fmid[i] = static_cast<float>(fmid[lo] + (static_cast<double>(fmid[hi]) - fmid[lo]) * (i - lo) / static_cast<double>(hi - lo));
					}
				}
				rep->weak_carrier_tiles = static_cast<uint32_t>(T - good.size());
			}
		}
		const double dt_sym = static_cast<double>(o.tile_samples) * symrate / fs;
		std::vector<float> f0(T); std::vector<int32_t> ud(T);
		for (size_t i = 0; i < T; i++) {
			double slope = 0.0;
			if (T > 2) {
				const size_t j = std::min(std::max<size_t>(i, 1), T - 2);
				slope = (static_cast<double>(fmid[j + 1]) - fmid[j - 1]) / (2 * dt_sym);
			}
			/* the estimate belongs to the middle of the window actually used */
			double f = fmid[i] - slope * (static_cast<double>(wstart[i]) + nfft / 2 - static_cast<double>(off_pre[i])) * symrate / fs;
			f = std::max<double>(-fmax, std::min<double>(fmax, f));
			f0[i] = static_cast<float>(f); ud[i] = slope >= 0 ? 1 : -1;
		}
		float *d_f0; int32_t *d_ud;
		TRY(upload(mem, f0, &d_f0, st));
		TRY(upload(mem, ud, &d_ud, st));
		TRY(mdemod_set_carrier_seeds(bank.c, d_f0, d_ud, st));
	}
	std::vector<uint32_t> cnt_pre, cnt1, cnt2;
	TRY(launch(off_pre, pres, soft_pre, cap_pre, cnt_pre));
	std::vector<mdemod_status> status_body;          /* of the launch whose symbols are emitted: first lock when the pilot has none */
	TRY(launch(starts, lens, soft1, cap, cnt1, &status_body));
	/* stream count at the start of every tile's EMITTED body, and the stream that ran it (identity without pass 2) */
	std::vector<uint64_t> n_start(T); std::vector<size_t> body_stream(T);
	for (size_t i = 0; i < T; i++) { n_start[i] = seed.n_symbols + cnt_pre[i]; body_stream[i] = i; }

	/* ---- rotation + seam of every tile against its predecessor (tile 0: against the pilot) ---- */
	std::vector<TailPair> pairs(T);
	for (size_t i = 0; i < T; i++) {
		TailPair &p = pairs[i];
		if (i == 0) { p.a = soft_dev; p.a_cnt = static_cast<uint32_t>(std::min<uint64_t>(n_pilot_sym, 0xFFFFFFFFu)); if (n_pilot_sym > 0xFFFFFFFFull) { p.a = soft_dev + 2 * (n_pilot_sym - 0xFFFFFFFFull); } }
		else { p.a = soft1 + (i - 1) * cap * 2; p.a_cnt = cnt1[i - 1]; }
		p.b = soft_pre + i * cap_pre * 2; p.b_cnt = cnt_pre[i];
		p.b_rot = 0; p.force_weak = pres[i] == 0;
	}
	std::vector<int32_t> shift, rot, weak;
	TRY(run_match(mem, pairs, K, shift, rot, weak, st, params->oqpsk ? 1 : 0));
	std::vector<int32_t> R(T);
	int32_t accr = 0;
	for (size_t i = 0; i < T; i++) { accr = (accr + rot[i]) & 3; R[i] = accr; rep->weak_seams += weak[i]; }

	std::vector<TileCopy> copies(T);
	std::vector<int32_t> seam(T, 0);
	if (params->oqpsk) {
		/* ---- OQPSK pass 2: state rotation (carrier + half-symbol clock), body, then a look-ahead into the next tile ---- */
		std::vector<int32_t> q(T);
		for (size_t i = 0; i < T; i++) q[i] = (4 - R[i]) & 3;
		int32_t *d_q;
		TRY(upload(mem, q, &d_q, st));
		TRY(mdemod_rotate_carrier(bank.c, d_q, st));
		TRY(mdemod_set_state(bank.c, static_cast<uint32_t>(T - 1), &seed, st));
		TRY(mdemod_set_history(bank.c, static_cast<uint32_t>(T - 1), seed_hist.data(), st));
		const uint64_t post = 4096;
		std::vector<uint64_t> starts2(T), lens2(T), ends2(T), post2(T);
		for (size_t i = 0; i < T; i++) {
			starts2[i] = starts[(i + 1) % T]; lens2[i] = lens[(i + 1) % T];
			ends2[i] = starts2[i] + lens2[i]; post2[i] = std::min<uint64_t>(post, n_samples - ends2[i]);
		}
		TRY(mem.alloc(&soft2, T * cap * 2));
		std::vector<uint32_t> cnt2s, cnt_posts;
		TRY(launch(starts2, lens2, soft2, cap, cnt2s, &status_body));
		for (size_t i = 0; i < T; i++) {
			body_stream[i] = (i + T - 1) % T;
			n_start[i] = i ? seed.n_symbols + cnt_pre[i - 1] + cnt1[i - 1] : seed.n_symbols;
		}
		const uint64_t cap_post = std::max<uint64_t>(1, mdemod_max_symbols(bank.c, *std::max_element(post2.begin(), post2.end())));
		int8_t *soft_post;
		TRY(mem.alloc(&soft_post, T * cap_post * 2));
		TRY(launch(ends2, post2, soft_post, cap_post, cnt_posts));
		auto stream_of = [&](size_t tile) { return (tile + T - 1) % T; };
		std::vector<TailPair> heads(T > 1 ? T - 1 : 0);
		for (size_t i = 0; i + 1 < T; i++) {                       /* seam i | i+1 */
			heads[i].a = soft_post + stream_of(i) * cap_post * 2; heads[i].a_cnt = cnt_posts[stream_of(i)];
			heads[i].b = soft2 + stream_of(i + 1) * cap * 2;      heads[i].b_cnt = cnt2s[stream_of(i + 1)];
			heads[i].b_rot = 0; heads[i].force_weak = 0;
		}
		std::vector<int32_t> sh, r2, w2;
		TRY(run_match(mem, heads, K, sh, r2, w2, st, 2));
		for (size_t i = 0; i < T; i++) {
			seam[i] = i ? -sh[i - 1] : 0;
			if (i) rep->weak_seams += w2[i - 1];
			if (i && !w2[i - 1] && (r2[i - 1] & 3)) rep->rotation_jumps++;
			copies[i].src = soft2 + stream_of(i) * cap * 2; copies[i].rot = 0; copies[i].keep = cnt2s[stream_of(i)];
			copies[i].head = (i && seam[i] == -1) ? soft_post + stream_of(i - 1) * cap_post * 2 : nullptr;
			copies[i].head_rot = 0;
		}
	} else if (!o.refine) {
		for (size_t i = 0; i < T; i++) {
			seam[i] = shift[i];
			copies[i].src = soft1 + i * cap * 2; copies[i].rot = R[i]; copies[i].keep = cnt1[i];
			copies[i].head = (shift[i] == -1 && cnt_pre[i] > 0) ? soft_pre + (i * cap_pre + cnt_pre[i] - 1) * 2 : nullptr;
			copies[i].head_rot = R[i];
		}
	} else {
		/* ---- pass 2: stream i := exact continuation of its pass-1 end state, in rotation 0, on tile i+1;
		 *      stream T-1 takes over from the pilot and runs tile 0 ---- */
		std::vector<int32_t> q(T);
		for (size_t i = 0; i < T; i++) q[i] = (4 - R[i]) & 3;
		int32_t *d_q;
		TRY(upload(mem, q, &d_q, st));
		TRY(mdemod_rotate_carrier(bank.c, d_q, st));
		TRY(mdemod_set_state(bank.c, static_cast<uint32_t>(T - 1), &seed, st));
		TRY(mdemod_set_history(bank.c, static_cast<uint32_t>(T - 1), seed_hist.data(), st));
		std::vector<uint64_t> starts2(T), lens2(T);
		for (size_t i = 0; i < T; i++) { starts2[i] = starts[(i + 1) % T]; lens2[i] = lens[(i + 1) % T]; }
		TRY(mem.alloc(&soft2, T * cap * 2));
		std::vector<uint32_t> cnt2s;
		TRY(launch(starts2, lens2, soft2, cap, cnt2s, &status_body));
		for (size_t i = 0; i < T; i++) {
			body_stream[i] = (i + T - 1) % T;
			n_start[i] = i ? seed.n_symbols + cnt_pre[i - 1] + cnt1[i - 1] : seed.n_symbols;
		}
		auto stream_of = [&](size_t tile) { return (tile + T - 1) % T; };     /* tile i was run by stream i-1 */
		cnt2.resize(T);
		for (size_t i = 0; i < T; i++) cnt2[i] = cnt2s[stream_of(i)];

		/* seam i|i+1: tile i+1 continued from tile i's PASS-1 trajectory, tile i's PASS-2 body is what is emitted */
		for (size_t i = 0; i < T; i++) {
			TailPair &p = pairs[i];
			p.a = soft2 + stream_of(i) * cap * 2; p.a_cnt = cnt2[i];
			p.b = soft1 + i * cap * 2; p.b_cnt = cnt1[i]; p.b_rot = R[i]; p.force_weak = 0;
		}
		std::vector<int32_t> shift2, rot2, weak2;
		TRY(run_match(mem, pairs, K, shift2, rot2, weak2, st));
		for (size_t i = 0; i + 1 < T; i++) { rep->weak_seams += weak2[i]; if (!weak2[i] && (rot2[i] & 3)) rep->rotation_jumps++; }
		for (size_t i = 0; i < T; i++) {
			seam[i] = i ? shift2[i - 1] : 0;
			copies[i].src = soft2 + stream_of(i) * cap * 2; copies[i].rot = 0; copies[i].keep = cnt2[i];
			copies[i].head = (i && seam[i] == -1 && cnt1[i - 1] > 0) ? soft1 + ((i - 1) * cap + cnt1[i - 1] - 1) * 2 : nullptr;
			copies[i].head_rot = i ? R[i - 1] : 0;
		}
	}

	/* ---- concatenate: pilot ++ tiles, with the seam fixes ---- */
	uint64_t out_pos = n_pilot_sym - (seam[0] == 1 ? 1 : 0);
	if (seam[0] == 1 && rep->pilot_symbols) rep->pilot_symbols--;       /* the pilot's last symbol was a duplicate of tile 0's first: the exact prefix is one shorter */
	for (size_t i = 0; i < T; i++) {
		const uint32_t drop = (i + 1 < T && seam[i + 1] == 1) ? 1 : 0;
		copies[i].keep = copies[i].keep > drop ? copies[i].keep - drop : 0;
		if (seam[i] == -1 && !copies[i].head) seam[i] = 0;
		copies[i].dst = out_pos;
		out_pos += copies[i].keep + (copies[i].head ? 1 : 0);
		if (seam[i]) rep->seam_fixes++;
	}
	if (out_pos > soft_cap_symbols) return MDEMOD_ERR_OVERFLOW;
	if (rep->first_lock_symbol < 0) {
		/* The pilot never locked (a recording that starts before the signal does): the lock gate (main.c:308-315) opens
		   at the first tile whose stream reports a first lock - inside its emitted body, or before it (then the whole
		   body counts).  Approximate to the tiles' own acquisition, which is faster than the serial sweep. */
		for (size_t i = 0; i < T; i++) {
			const int64_t fl = status_body[body_stream[i]].first_lock_symbol;
			if (fl < 0) continue;
			const uint64_t inside = static_cast<uint64_t>(fl) > n_start[i] ? static_cast<uint64_t>(fl) - n_start[i] : 0;
			rep->first_lock_symbol = static_cast<int64_t>(copies[i].dst + std::min<uint64_t>(inside, copies[i].keep));
			break;
		}
	}
	TileCopy *d_copies;
	TRY(upload(mem, copies, &d_copies, st));
	hipLaunchKernelGGL(assemble_kernel, dim3(static_cast<unsigned>(T)), dim3(256), 0, st, d_copies, soft_dev);
	HTRY(hipGetLastError());
	HTRY(hipStreamSynchronize(st));
	rep->n_symbols = out_pos;
	rep->tiles_seconds = seconds_since(t_tiles);
	return MDEMOD_OK;
}

/* Host-buffer convenience (PCIe inclusive): what the C CLI's --tiled mode calls. */
extern "C" int
mdemod_demodulate_recording_host(const mdemod_params *params, const mdemod_recording_opts *opts,
                                 const void *iq_host, uint64_t n_samples,
                                 int8_t *soft_host, uint64_t soft_cap_symbols,
                                 mdemod_recording_report *rep)
{
	if (!params || !iq_host || !soft_host || !rep) return MDEMOD_ERR_PARAM;
	if (hipSetDevice(params->device) != hipSuccess) return MDEMOD_ERR_HIP;
	const size_t sb = 2 * static_cast<size_t>(params->bps) / 8;
	DevMem mem;
	unsigned char *d_iq; int8_t *d_soft;
	TRY(mem.alloc(&d_iq, static_cast<size_t>(n_samples) * sb));
	TRY(mem.alloc(&d_soft, static_cast<size_t>(soft_cap_symbols) * 2));
	if (n_samples) HTRY(hipMemcpy(d_iq, iq_host, static_cast<size_t>(n_samples) * sb, hipMemcpyHostToDevice));
	TRY(mdemod_demodulate_recording(params, opts, d_iq, n_samples, d_soft, soft_cap_symbols, rep, nullptr));
	if (rep->n_symbols) HTRY(hipMemcpy(soft_host, d_soft, static_cast<size_t>(rep->n_symbols) * 2, hipMemcpyDeviceToHost));
	return MDEMOD_OK;
}
