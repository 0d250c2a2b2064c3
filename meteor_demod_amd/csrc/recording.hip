/*
 * recording.hip — ONE recording demodulated on many lanes as overlapped tiles.
 *
 * The reference runs a recording as one serial recurrence (main.c:303-316); see
 * NOTEBOOK.md §3.1 for why tiles cannot equal it bit for bit (a 1-LSB change of one
 * input sample leaves 0.2 % of the reference's own symbols more than 1 LSB away,
 * for ever) and what is done instead:
 *
 *   pilot    head of the recording as one stream from power-on state until the
 *            carrier loop has locked and converged  -> the reference's own bytes
 *   acquire  every tile starts early from the pilot's loop state and its own
 *            carrier / gain estimate; after the acquisition the two loop
 *            integrators are put back on their seeds
 *   frame    rotation of every tile against its predecessor by dead reckoning of
 *            the NCO phase (Costas lock is 4-fold ambiguous), undone in the state
 *   settle   in the serial run's rotation, then the body (emitted)
 *   seams    residual rotation (repair) and one-symbol disagreement of each tile
 *            against its predecessor, measured on samples both demodulated
 *
 * All sample arithmetic is done by the demodulator kernels through the public
 * C-ABI of this library; the kernels here estimate carriers and compare and move
 * int8 symbols.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <functional>
#include <thread>
#include <vector>

#include "mdemod_internal_api.h"

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
	__shared__ long long acc[8][64];
	long long s[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
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
		s[7] += b1i * b1i + b1q * b1q;                                      /* ... and of b[1..K]  */
	}
	for (int k = 0; k < 8; k++) acc[k][threadIdx.x] = s[k];
	__syncthreads();
	if (threadIdx.x == 0) {
		long long t[8];
		for (int k = 0; k < 8; k++) { t[k] = 0; for (unsigned l = 0; l < blockDim.x; l++) t[k] += acc[k][l]; }
		const int shifts[3] = { 0, 1, -1 };
		long long best = 0; int bs = 0, br = 0; bool have = false;
		for (int c = 0; c < 3; c++) {
			const long long re = t[2 * c], im = t[2 * c + 1];
			const long long sc4[4] = { re, im, -re, -im };
			int r = 0;
			for (int k = 1; k < 4; k++) if (sc4[k] > sc4[r]) r = k;
			if (!have || sc4[r] > best) { best = sc4[r]; bs = shifts[c]; br = r; have = true; }
		}
		/* less than half of a perfect match, amplitudes taken out: the two runs may sit at different AGC gains (float input:
		   the reference's AGC needs seconds), which says nothing about their rotation */
		const bool weak = (best <= 0) || (4.0 * (double)best * (double)best < (double)t[6] * (double)t[7]) || p.force_weak;
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
	__shared__ long long acc[26][64];                      /* [r][rail][d] = 24 sums + the two energies */
	long long s[26];
	for (int k = 0; k < 26; k++) s[k] = 0;
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
		{ int bi, bq; load(p.b, p.b_cnt, j + 1, bi, bq); s[25] += bi * bi + bq * bq; }
	}
	for (int k = 0; k < 26; k++) acc[k][threadIdx.x] = s[k];
	__syncthreads();
	if (threadIdx.x == 0) {
		long long t[26];
		for (int k = 0; k < 26; k++) { t[k] = 0; for (unsigned l = 0; l < blockDim.x; l++) t[k] += acc[k][l]; }
		long long best = 0; int br = 0;
		for (int r = 0; r < 4; r++) {
			long long sI = t[r * 6], sQ = t[r * 6 + 3];
			for (int d = 1; d < 3; d++) { if (t[r * 6 + d] > sI) sI = t[r * 6 + d]; if (t[r * 6 + 3 + d] > sQ) sQ = t[r * 6 + 3 + d]; }
			if (r == 0 || sI + sQ > best) { best = sI + sQ; br = r; }
		}
		const bool weak = (best <= 0) || (4.0 * (double)best * (double)best < (double)t[24] * (double)t[25]) || p.force_weak;
		rot_out[blockIdx.x] = weak ? 0 : br;
		weak_out[blockIdx.x] = weak ? 1 : 0;
	}
}

__global__ void
match_heads_kernel(const TailPair *pairs, int K, int32_t *shift_out, int32_t *rot_out, int32_t *weak_out)
{
	const TailPair p = pairs[blockIdx.x];
	__shared__ long long acc[8][64];
	long long s[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
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
		s[7] += b0i * b0i + b0q * b0q;
	}
	for (int k = 0; k < 8; k++) acc[k][threadIdx.x] = s[k];
	__syncthreads();
	if (threadIdx.x == 0) {
		long long t[8];
		for (int k = 0; k < 8; k++) { t[k] = 0; for (unsigned l = 0; l < blockDim.x; l++) t[k] += acc[k][l]; }
		const int shifts[3] = { 0, 1, -1 };
		long long best = 0; int bs = 0, br = 0; bool have = false;
		for (int c = 0; c < 3; c++) {
			const long long re = t[2 * c], im = t[2 * c + 1];
			const long long sc4[4] = { re, im, -re, -im };
			int r = 0;
			for (int k = 1; k < 4; k++) if (sc4[k] > sc4[r]) r = k;
			if (!have || sc4[r] > best) { best = sc4[r]; bs = shifts[c]; br = r; have = true; }
		}
		const bool weak = (best <= 0) || (4.0 * (double)best * (double)best < (double)t[6] * (double)t[7]);
		shift_out[blockIdx.x] = weak ? 0 : bs;
		rot_out[blockIdx.x] = weak ? 0 : br;
		weak_out[blockIdx.x] = weak ? 1 : 0;
	}
}

/* ---- per-tile carrier estimate (NOTEBOOK.md 3.1, Doppler): z^4 of the samples has a line at 4x the carrier offset -------- */

template <int FMT> struct RawIQ;
template <> struct RawIQ<16> { typedef int16_t t; __device__ static float2 get(const void *p, uint64_t i) { uint32_t w; __builtin_memcpy(&w, static_cast<const int16_t *>(p) + 2 * i, 4); return make_float2((float)(int16_t)(w & 0xFFFFu), (float)((int32_t)w >> 16)); } };   /* one load per pair */
template <> struct RawIQ<8>  { typedef uint8_t t; __device__ static float2 get(const void *p, uint64_t i) { const uint8_t *q = static_cast<const uint8_t *>(p) + 2 * i; return make_float2((float)((int)q[0] - 128), (float)((int)q[1] - 128)); } };
template <> struct RawIQ<32> { typedef float t;   __device__ static float2 get(const void *p, uint64_t i) { const float *q = static_cast<const float *>(p) + 2 * i; return make_float2(q[0], q[1]); } };

/* In-place radix-2 FFT of 2^log2_nf points in LDS by the whole block: input stored bit-reversed, output in natural order.
 * Ends with a barrier. */
__device__ __forceinline__ void
lds_fft(float2 *spec, int log2_nf, int tid, int nth)
{
	const int NF = 1 << log2_nf;
	for (int st = 0; st < log2_nf; st++) {                /* decimation in time */
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
}

/* Position of a Hann-windowed line from the largest bin k and its two neighbours (magnitudes a, b, c): the ratio of the two
 * largest gives it (Grandke); a parabola through three magnitudes is biased by up to 0.03 bin. */
__device__ __forceinline__ float
grandke_offset(float a, float b, float c)
{
	if (!(b > 0.0f)) return 0.0f;
	const float al = (c >= a ? c : a) / b;
	float delta = (2.0f * al - 1.0f) / (al + 1.0f);
	delta = fminf(fmaxf(delta, 0.0f), 0.5f);
	return c < a ? -delta : delta;
}

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
carrier_line_kernel(const void *iq, uint64_t n_samples, const uint64_t *starts, const float *chirp, float chirp_scale, int log2_nf, int decim, int pre, int kmax,
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
	/* de-chirp: the carrier phase is 0.5 * c * t^2 around the middle of the window (c in turns per sample^2 here), z^4 has four times that */
	const float c4 = chirp ? chirp[blockIdx.x] * chirp_scale : 0.0f;
	for (int m = tid; m < NF; m += nth) {
		float ar = 0.0f, ai = 0.0f;
		for (int d = 0; d < decim; d += pre) {
			/* `pre` samples are averaged BEFORE the 4th power: at 14 samples per symbol the noise of the whole sampled band goes
			   into z^4 otherwise (its line-to-floor ratio falls with the 4th power of the per-sample SNR), while the signal
			   lives in the lowest ~1/6 of it; a boxcar of pre <= osf/3 samples takes 6 dB (pre = 4) of that noise away first */
			float2 v = make_float2(0.0f, 0.0f);
			for (int e = 0; e < pre; e++) { const float2 u = sample(m * decim + d + e); v.x += u.x; v.y += u.y; }
			v.x -= mr * (float)pre; v.y -= mi * (float)pre;
			const float2 z2 = make_float2(v.x * v.x - v.y * v.y, 2.0f * v.x * v.y);
			float2 z4 = make_float2(z2.x * z2.x - z2.y * z2.y, 2.0f * z2.x * z2.y);
			if (c4 != 0.0f) {
				const double t = (double)(m * decim + d) + 0.5 * (double)(pre - 1) - 0.5 * (double)nwin;
				const double turns = -(double)c4 * t * t;                 /* -4 * 0.5 * c * t^2 in turns */
				float sn, cs;
				sincospif(2.0f * (float)(turns - floor(turns)), &sn, &cs);
				z4 = make_float2(z4.x * cs - z4.y * sn, z4.x * sn + z4.y * cs);
			}
			ar += z4.x; ai += z4.y;
		}
		const float w = (0.5f - 0.5f * cospif(wstep * (float)m)) * (1e-12f / (float)(pre * pre * pre * pre));   /* Hann; the scale keeps |z|^4 of full-scale s16 far from overflow */
		spec[__brev((unsigned)m) >> (32 - log2_nf)] = make_float2(ar * w, ai * w);
	}
	__syncthreads();
	lds_fft(spec, log2_nf, tid, nth);
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
		/* (a parabola's 0.03 bin would be 0.1 rad over a tile of dead reckoning) */
		const float delta = grandke_offset(a, b, c);
		freq_out[blockIdx.x] = ((float)k + delta) * hz_per_bin_over4 * rad_per_hz;
		quality_out[blockIdx.x] = b / (red[2][0] / (float)(2 * kmax + 1) + 1e-30f);
	}
}

/* Feed-forward symbol-clock estimate, one block per window (same windows as the carrier estimate).  QPSK: |z|^2 of an RRC-shaped
 * signal has a spectral line at the symbol rate.  OQPSK: there the two rails' lines cancel (half a symbol apart); z^2 carries
 * them instead, at twice the carrier +- the symbol rate - their distance is twice the symbol rate whatever the carrier.
 * The line is moved to zero with the nominal rate (phase in double), summed in groups of D samples (boxcar: the searched band
 * is +-1/4096 of the rate, timing.c:80-86), Hann window, 4096-point FFT in LDS, Grandke peak.  Result: rad per interpolated
 * step as timing.c:14 `freq` holds it.  12 dB, 262 144 samples: 1e-7 of the rate; the reference's own loop wanders by 3e-6. */
template <int FMT>
__global__ void __launch_bounds__(1024)
clock_line_kernel(const void *iq, uint64_t n_samples, const uint64_t *starts, const float *carrier, const float *chirp,
                  double carrier_scale, double chirp_scale, int oqpsk, int log2_nf, int decim, int kmax, int knoise,
                  double f_nom, double t_freq_scale, float *t_freq_out, float *quality_out)
{
	extern __shared__ float2 spec[];                      /* NF (QPSK) or 2 * NF (OQPSK) complex floats */
	__shared__ float red[3][1024];
	__shared__ int redi[1024];
	const int NF = 1 << log2_nf, nwin = NF * decim, tid = threadIdx.x, nth = blockDim.x;
	const uint64_t s0 = starts[blockIdx.x];
	auto sample = [&](int k) {
		const uint64_t i = s0 + (uint64_t)k < n_samples ? s0 + (uint64_t)k : n_samples - 1;
		return RawIQ<FMT>::get(iq, i);
	};
	/* No mean removal and no normalisation (one pass over the samples): a DC offset m adds 2 Re(z conj(m)) to |z|^2, and z has no
	   energy at the symbol rate (the RRC band ends at 0.8 of it); the power term of |z|^2 sits at DC, a rate away from the line,
	   and does not add up coherently over the D samples of a boxcar as the line does.  Quality is a ratio. */
	const float wstep = 2.0f / (float)(NF - 1);
	/* OQPSK: lines of z^2 at 2 fc +- f_nom (cycles per sample); the chirp of z^2 is twice the carrier's */
	const double fc2 = (oqpsk && carrier) ? 2.0 * (double)carrier[blockIdx.x] * carrier_scale : 0.0;
	const double c2 = (oqpsk && chirp) ? (double)chirp[blockIdx.x] * chirp_scale : 0.0;
	const int nline = oqpsk ? 2 : 1;
	/* Thread t takes samples t, t + 1024, ...: a wave reads 64 consecutive samples (one or more whole groups of D), the D lanes
	   of a group add up with xor shuffles and the first lane stores the decimated point.  (One thread per point and D strided
	   samples per thread was 3x slower at D = 64: 64 lanes on 64 different cache lines with every load; 4 samples per thread
	   4x slower still, for the same reason.)  The rotation of each thread's samples advances by a constant per step; it is
	   set from the double-precision phase every 64 steps. */
	const int lg = __ffs(decim) - 1;                                   /* decim is a power of two */
	double fl[2];
	float2 rot[2], stp[2];
	for (int l = 0; l < nline; l++) {
		fl[l] = oqpsk ? fc2 + (l == 0 ? f_nom : -f_nom) : f_nom;
		double sp = (double)nth * fl[l]; sp -= floor(sp);
		float sn, cs;
		sincospif(-2.0f * (float)sp, &sn, &cs); stp[l] = make_float2(cs, sn);
		rot[l] = make_float2(1.0f, 0.0f);
	}
	for (int it = 0; it < nwin / nth; it++) {
		const int n = it * nth + tid;
		if ((it & 63) == 0)
			for (int l = 0; l < nline; l++) {
				double ph = (double)n * fl[l]; ph -= floor(ph);
				float sn, cs;
				sincospif(-2.0f * (float)ph, &sn, &cs); rot[l] = make_float2(cs, sn);
			}
		const float2 v = sample(n);
		float2 u;
		if (oqpsk) {
			u = make_float2(v.x * v.x - v.y * v.y, 2.0f * v.x * v.y);
			if (c2 != 0.0) {
				const double t = (double)n - 0.5 * (double)nwin;
				const double turns = -c2 * t * t;
				float sn, cs;
				sincospif(2.0f * (float)(turns - floor(turns)), &sn, &cs);
				u = make_float2(u.x * cs - u.y * sn, u.x * sn + u.y * cs);
			}
		} else {
			u = make_float2(v.x * v.x + v.y * v.y, 0.0f);
		}
		const unsigned br = __brev((unsigned)(n >> lg)) >> (32 - log2_nf);
		for (int l = 0; l < nline; l++) {
			float ax = u.x * rot[l].x - u.y * rot[l].y, ay = u.x * rot[l].y + u.y * rot[l].x;
			rot[l] = make_float2(rot[l].x * stp[l].x - rot[l].y * stp[l].y, rot[l].x * stp[l].y + rot[l].y * stp[l].x);
			for (int off = decim >> 1; off > 0; off >>= 1) { ax += __shfl_xor(ax, off); ay += __shfl_xor(ay, off); }
			if ((tid & (decim - 1)) == 0) spec[l * NF + br] = make_float2(ax, ay);
		}
	}
	__syncthreads();
	for (int m = tid; m < NF; m += nth) {                               /* Hann window, once per point */
		const float w = 0.5f - 0.5f * cospif(wstep * (float)m);
		const unsigned br = __brev((unsigned)m) >> (32 - log2_nf);
		for (int l = 0; l < nline; l++) { float2 &v = spec[l * NF + br]; v = make_float2(v.x * w, v.y * w); }
	}
	__syncthreads();
	float pos[2] = { 0.0f, 0.0f }, qual[2] = { 0.0f, 0.0f };
	for (int l = 0; l < nline; l++) {
		float2 *sp2 = spec + l * NF;
		lds_fft(sp2, log2_nf, tid, nth);
		auto mag = [&](int k) { const float2 v = sp2[(k + NF) & (NF - 1)]; return v.x * v.x + v.y * v.y; };
		float best = -1.0f, sum = 0.0f; int bidx = 0;
		for (int k = -knoise + tid; k <= knoise; k += nth) {
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
		const int k = redi[0];
		const float a = sqrtf(mag(k - 1)), b = sqrtf(mag(k)), c = sqrtf(mag(k + 1));
		pos[l] = (float)k + grandke_offset(a, b, c);
		qual[l] = b / (red[2][0] / (float)(2 * knoise + 1) + 1e-30f);
		__syncthreads();
	}
	if (tid == 0) {
		/* QPSK: the line sits pos[0] bins above the nominal rate; OQPSK: half the distance of the two lines is the rate */
		const double dev = (oqpsk ? 0.5 * ((double)pos[0] - (double)pos[1]) : (double)pos[0]) / (double)nwin;
		t_freq_out[blockIdx.x] = (float)((f_nom + dev) * t_freq_scale);
		quality_out[blockIdx.x] = oqpsk ? fminf(qual[0], qual[1]) : qual[0];
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

/* 8 symbols (16 bytes) of a tile, rotated by k quarter turns: (i + jq) * j^k on int8 lanes (|values| <= 127: negation is exact) */
__device__ __forceinline__ uint4
rot_group(uint4 v, int k)
{
	k &= 3;
	if (k == 0) return v;
	uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
	for (int j = 0; j < 4; j++) {
		uint32_t o = 0;
#pragma unroll
		for (int h = 0; h < 2; h++) {
			const int i = (int)(int8_t)(w[j] >> (16 * h)), q = (int)(int8_t)(w[j] >> (16 * h + 8));
			int ri, rq;
			rot_pair(i, q, k, ri, rq);
			o |= ((uint32_t)(ri & 0xFF) | ((uint32_t)(rq & 0xFF) << 8)) << (16 * h);
		}
		w[j] = o;
	}
	return make_uint4(w[0], w[1], w[2], w[3]);
}

/* One block per (tile, slice): the tile's kept symbols go to their place in the output, 16 bytes per thread and step (neither
 * side need be more than 2-byte aligned: a seam fix starts a row one symbol in).  gridDim.y slices share a tile. */
__global__ void
assemble_kernel(const TileCopy *tiles, int8_t *out)
{
	const TileCopy t = tiles[blockIdx.x];
	int8_t *dst = out + 2 * t.dst;
	if (t.head && threadIdx.x == 0 && blockIdx.y == 0) {
		int i, q; rot_pair(t.head[0], t.head[1], t.head_rot, i, q);
		dst[0] = (int8_t)i; dst[1] = (int8_t)q;
	}
	if (t.head) dst += 2;
	const uint32_t groups = t.keep / 8;
	for (uint32_t g = blockIdx.y * blockDim.x + threadIdx.x; g < groups; g += gridDim.y * blockDim.x) {
		uint4 raw;
		__builtin_memcpy(&raw, t.src + 16 * (size_t)g, 16);             /* a seam fix starts a row one symbol in: 2-byte aligned */
		const uint4 v = rot_group(raw, t.rot);
		__builtin_memcpy(dst + 16 * (size_t)g, &v, 16);
	}
	if (blockIdx.y == 0)
		for (uint32_t k = groups * 8 + threadIdx.x; k < t.keep; k += blockDim.x) {
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
#define HTRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { mdm_note_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); (void)hipGetLastError(); return e_ == hipErrorOutOfMemory ? MDEMOD_ERR_NOMEM : MDEMOD_ERR_HIP; } } while (0)

template <typename T>
int
upload(DevMem &m, const std::vector<T> &h, T **dev, hipStream_t st)
{
	TRY(m.alloc(dev, h.size()));
	if (!h.empty()) {
		HTRY(hipMemcpyAsync(*dev, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st));
		HTRY(hipStreamSynchronize(st));          /* h is pageable and may die with the caller's scope: small tables, wait here */
	}
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
 * in LDS.  window_samples is rounded down to a power of two in [4096, 2^18] (and further if the band forbids decimation). */
static void
carrier_window(const mdemod_params *params, uint32_t window_samples, int *nwin, int *decim, int *log2_nf, int *kmax)
{
	const double symrate = params->symrate, fs = params->samplerate;
	int n = 4096;
	while (n * 2 <= static_cast<int>(std::min<uint32_t>(window_samples, 1u << 18))) n *= 2;
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
mdemod_estimate_carrier_chirp(const mdemod_params *params, const void *iq_dev, uint64_t n_samples,
                              const uint64_t *starts_dev, const float *chirp_dev, uint32_t n_windows, uint32_t window_samples,
                              float *freq_dev, float *quality_dev, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!params || !iq_dev || !starts_dev || !freq_dev || !quality_dev || n_samples == 0) return MDEMOD_ERR_PARAM;
	if (params->samplerate <= 0 || params->symrate <= 0 || (params->bps != 8 && params->bps != 16 && params->bps != 32)) return MDEMOD_ERR_PARAM;
	if (n_windows == 0) return MDEMOD_OK;
	hipStream_t st = static_cast<hipStream_t>(hip_stream);
	int nwin, decim, log2_nf, kmax;
	carrier_window(params, window_samples, &nwin, &decim, &log2_nf, &kmax);
	const double symrate = params->symrate, fs = params->samplerate;
	const int nco = params->oqpsk ? 2 : 1;                       /* OQPSK: the NCO steps twice a symbol (pll.c:77,93) */
	const dim3 grid(n_windows);
	/* samples averaged in front of the 4th power: the largest power of two up to a third of a symbol (and a divisor of decim) */
	int pre = 1;
	while (pre * 2 <= decim && pre * 2 * 3.0 <= fs / symrate) pre *= 2;
	const size_t lds = (static_cast<size_t>(nwin) / decim) * sizeof(float2);
	const float hz_per_bin_over4 = static_cast<float>(fs / nwin / 4.0);
	const float rad_per_hz = static_cast<float>(2 * 3.141592653589793 / (symrate * nco));
	/* chirp in rad per NCO step per sample -> carrier rad per sample^2 (x nco * symrate / fs) -> turns of z^4 per sample^2 (/ pi) */
	const float chirp_scale = static_cast<float>(nco * symrate / fs / 3.141592653589793);
#define LAUNCH_LINE(F) do { \
		HTRY(hipFuncSetAttribute(reinterpret_cast<const void *>(carrier_line_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds))); \
		hipLaunchKernelGGL(carrier_line_kernel<F>, grid, dim3(1024), lds, st, iq_dev, n_samples, starts_dev, chirp_dev, chirp_scale, log2_nf, decim, pre, kmax, \
		                   hz_per_bin_over4, rad_per_hz, freq_dev, quality_dev); } while (0)
	switch (params->bps) {
	case 16: LAUNCH_LINE(16); break;
	case 8:  LAUNCH_LINE(8); break;
	default: LAUNCH_LINE(32); break;
	}
#undef LAUNCH_LINE
	HTRY(hipGetLastError());
	return MDEMOD_OK;
} MDEMOD_API_CATCH

extern "C" int
mdemod_estimate_carrier(const mdemod_params *params, const void *iq_dev, uint64_t n_samples,
                        const uint64_t *starts_dev, uint32_t n_windows, uint32_t window_samples,
                        float *freq_dev, float *quality_dev, void *hip_stream)
try { MDEMOD_API_ENTER
	return mdemod_estimate_carrier_chirp(params, iq_dev, n_samples, starts_dev, nullptr, n_windows, window_samples, freq_dev, quality_dev, hip_stream);
} MDEMOD_API_CATCH

extern "C" int
mdemod_estimate_clock(const mdemod_params *params, const void *iq_dev, uint64_t n_samples,
                      const uint64_t *starts_dev, const float *carrier_dev, const float *chirp_dev,
                      uint32_t n_windows, uint32_t window_samples, float *t_freq_dev, float *quality_dev, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!params || !iq_dev || !starts_dev || !t_freq_dev || !quality_dev || n_samples == 0) return MDEMOD_ERR_PARAM;
	if (params->samplerate <= 0 || params->symrate <= 0 || params->interp_factor <= 0 || (params->bps != 8 && params->bps != 16 && params->bps != 32)) return MDEMOD_ERR_PARAM;
	if (n_windows == 0) return MDEMOD_OK;
	hipStream_t st = static_cast<hipStream_t>(hip_stream);
	const double symrate = params->symrate, fs = params->samplerate;
	int nwin = 4096;
	while (nwin * 2 <= static_cast<int>(std::min<uint32_t>(window_samples, 1u << 18))) nwin *= 2;
	const int log2_nf = 12, decim = nwin >> log2_nf;
	const double f_nom = symrate / fs;                             /* cycles per sample */
	const int nco = params->oqpsk ? 2 : 1;
	/* the reference's loop keeps its rate within 1/4096 of the nominal one (timing.c:80-86); OQPSK: each line also carries twice
	   the error of the carrier it was moved by (an estimate: a few bins at most) */
	const int kmax = std::min((1 << log2_nf) / 2 - 2, static_cast<int>(f_nom / 4096.0 * nwin) + (params->oqpsk ? 12 : 3));
	const int knoise = std::min((1 << log2_nf) / 2 - 2, std::max(128, 2 * kmax));
	const double carrier_scale = symrate * nco / fs / (2 * 3.141592653589793);          /* rad per NCO step -> cycles per sample */
	/* chirp in rad per NCO step per sample -> carrier turns per sample^2 = x nco * symrate / fs / 2 pi; phase 0.5 c t^2, doubled by z^2 */
	const double chirp_scale = nco * symrate / fs / (2 * 3.141592653589793);
	const double t_freq_scale = 2 * 3.141592653589793 / params->interp_factor;
	const size_t lds = (static_cast<size_t>(1) << log2_nf) * sizeof(float2) * (params->oqpsk ? 2 : 1);
#define LAUNCH_CLK(F) do { \
		HTRY(hipFuncSetAttribute(reinterpret_cast<const void *>(clock_line_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds))); \
		hipLaunchKernelGGL(clock_line_kernel<F>, dim3(n_windows), dim3(1024), lds, st, iq_dev, n_samples, starts_dev, carrier_dev, chirp_dev, \
		                   carrier_scale, chirp_scale, params->oqpsk ? 1 : 0, log2_nf, decim, kmax, knoise, f_nom, t_freq_scale, t_freq_dev, quality_dev); } while (0)
	switch (params->bps) {
	case 16: LAUNCH_CLK(16); break;
	case 8:  LAUNCH_CLK(8); break;
	default: LAUNCH_CLK(32); break;
	}
#undef LAUNCH_CLK
	HTRY(hipGetLastError());
	return MDEMOD_OK;
} MDEMOD_API_CATCH

extern "C" void
mdemod_recording_default_opts(mdemod_recording_opts *o)
{
	if (!o) return;
	o->tile_samples = 0;                     /* automatic, see the header */
	o->acquire_samples = o->frame_samples = o->settle_samples = 0xFFFFFFFFu;
	o->pilot_block = 65536; o->pilot_margin_symbols = 0xFFFFFFFFu;
	o->max_pilot_samples = 0xFFFFFFFFFFFFFFFFull; o->match_symbols = 192; o->repair = 1; o->carrier_seed = 1; o->clock_seed = 0;
	o->debug = 0; o->debug_tile = -1;
}

namespace {

constexpr double kPi = 3.141592653589793;

/* piecewise-linear f(t) through (centre[i], value[i]), extrapolated with the end slopes */
double
interp_at(const std::vector<double> &cx, const std::vector<double> &v, double t)
{
	const size_t n = cx.size();
	if (n == 0) return 0.0;
	if (n == 1) return v[0];
	size_t j = static_cast<size_t>(std::upper_bound(cx.begin(), cx.end(), t) - cx.begin());
	j = std::min(std::max<size_t>(j, 1), n - 1);
	const double w = (t - cx[j - 1]) / (cx[j] - cx[j - 1]);
	return v[j - 1] + (v[j] - v[j - 1]) * w;
}

/* time (in interpolated steps from the start of the recording) of the stream's last NCO step, from its symbol clock:
 * the clock's phase has advanced t_phase since the last symbol (timing.c:32-57; OQPSK: since the last full symbol, and its
 * I rail fired at pi when the next firing is the Q rail's) */
double
last_nco_time(const mdemod_stream_state &s, double pos_samples, int interp, int oqpsk)
{
	double ph = s.t_phase;
	if (oqpsk && s.t_dual_state == 2) ph -= kPi;
	return pos_samples * interp - ph / static_cast<double>(s.t_freq);
}

/* quarter turns of stream b against stream a by dead reckoning: theta_b - (theta_a + f * steps) */
int
frame_between(double th_a, double t_a, double th_b, double t_b, double f_mean, double steps_per_nco, double *residual)
{
	const double n = std::nearbyint((t_b - t_a) / steps_per_nco);
	double d = std::fmod(th_b - th_a - f_mean * n, 2 * kPi);
	if (d < 0) d += 2 * kPi;
	const int r = static_cast<int>(std::nearbyint(d / (kPi / 2))) & 3;
	double res = d - std::nearbyint(d / (kPi / 2)) * (kPi / 2);
	*residual = res;
	return r;
}

} /* namespace */

/* What the serial head leaves behind. */
struct PilotOut {
	uint64_t P = 0, nsym = 0;            /* samples demodulated serially, symbols written to the output */
	mdemod_stream_state seed;            /* loop state at P */
	std::vector<PilotBlock> blocks;      /* AGC calibration */
	std::vector<float> hist;             /* filter history at P */
	double seconds = 0.0;
};

/* ---- pilot: the reference's own serial run of the head, from the power-on state until the PLL has locked and stayed locked for
 * the margin (the reference's own symbols into soft_dev, first-lock index included) ---- */
static int
run_pilot(const mdemod_params *params, const mdemod_recording_opts &o, const void *iq_dev, uint64_t n_samples,
          int8_t *soft_dev, uint64_t soft_cap_symbols, hipStream_t st, const std::function<void(uint64_t)> *need, PilotOut &po)
{
	const size_t sb = 2 * static_cast<size_t>(params->bps) / 8;
	const unsigned char *iq = static_cast<const unsigned char *>(iq_dev);
	const auto t_start = std::chrono::steady_clock::now();
	mdemod_params pp = *params; pp.n_streams = 1;
	Ctx pilot;
	TRY(mdemod_create(&pp, &pilot.c));
	uint64_t pos = 0, nsym = 0; bool have_lock = false, lost_lock = false; uint64_t locked_at = 0;
	DevMem est_mem; unsigned char *d_est = nullptr; unsigned false_checks = 0; bool waiting_for_genuine = false;
	std::vector<PilotBlock> &pilot_blocks = po.blocks;
	mdemod_stream_state &seed = po.seed;
	memset(&seed, 0, sizeof(seed));
	TRY(mdemod_get_state(pilot.c, 0, &seed, st));
	while (pos < n_samples) {
		/* once the lock has been seen the hand-over is a known number of symbols away: shorter blocks from there on, so that the
		   serial head does not run up to 65 535 samples past it (9 ms on average at 3.5 MS/s) */
		const uint32_t blk = have_lock ? std::min<uint32_t>(o.pilot_block, 8192) : o.pilot_block;
		const uint32_t b = static_cast<uint32_t>(std::min<uint64_t>(blk, n_samples - pos));
		/* the block may produce up to one symbol per sample (mdemod_max_symbols); the caller's buffer only has to hold what
		   it does produce: the kernel checks the capacity it is given and reports an overflow */
		const uint64_t cap = std::min<uint64_t>(mdemod_max_symbols(pilot.c, b), soft_cap_symbols - std::min(nsym, soft_cap_symbols));
		if (cap == 0) return MDEMOD_ERR_OVERFLOW;
		if (need) (*need)(pos + b);
		TRY(mdemod_process_device_uniform(pilot.c, iq + pos * sb, 0, b, soft_dev + 2 * nsym, cap, static_cast<uint32_t>(cap), st));
		mdemod_status s1;
		TRY(mdemod_get_status(pilot.c, 0, 1, &s1, st));
		if (s1.overflow) return MDEMOD_ERR_OVERFLOW;
		nsym += s1.symbols_this_call;
		TRY(mdemod_get_state(pilot.c, 0, &seed, st));
		pilot_blocks.push_back(PilotBlock{pos, b, seed.agc_gain, seed.n_symbols});
		pos += b;
		if (seed.pll_locked && !have_lock) {
			have_lock = true;
			/* the margin counts from the lock itself when this is the first one (the status has its symbol), otherwise from the
			   end of the block that saw it */
			locked_at = (!lost_lock && seed.first_lock_symbol >= 0) ? static_cast<uint64_t>(seed.first_lock_symbol) : seed.n_symbols;
		}
		if (!seed.pll_locked) { if (have_lock) lost_lock = true; have_lock = false; }
		/* ... and not before the reference's AGC has settled: its step is absolute (agc.c:13-25), 6 time constants =
		   6 * gain / (1e-4 * 190) symbols - nothing for s16-scale input, ~200 k symbols for float input around +-1 */
		const double agc_settle = 6.0 * static_cast<double>(seed.agc_gain) / (1e-4 * 190.0);
		if (have_lock && seed.n_symbols - locked_at >= o.pilot_margin_symbols && static_cast<double>(seed.n_symbols) >= agc_settle) {
			/* ... and not on a lock that is none: the reference's detector (pll.c:117-123: mean |e| < 85) also fires far from the
			   carrier - always while a float recording's AGC is still coming up (small |e| because everything is small), and for
			   good on every other OQPSK recording.  As long as the loop's carrier word is more than 100 Hz from the signal's own
			   4th-power line here, the head goes on (it IS the reference: whatever that does, these are its bytes); looked at every
			   fourth block, given up at max_pilot_samples like the wait for a lock. */
			bool genuine = true;
			if (o.carrier_seed == 1 && pos < n_samples) {
				if ((false_checks++ & 3) == 0) {
					if (!d_est) TRY(est_mem.alloc(&d_est, 32));
					const uint32_t win = mdemod_carrier_window_samples(params, static_cast<uint32_t>(std::min(262144.0, 20536.0 * params->samplerate / params->symrate)));
					const uint64_t w0 = pos > win ? pos - win : 0;
					HTRY(hipMemcpyAsync(d_est, &w0, sizeof(w0), hipMemcpyHostToDevice, st));
					HTRY(hipStreamSynchronize(st));
					float *d_f = reinterpret_cast<float *>(d_est + 8), *d_q = reinterpret_cast<float *>(d_est + 16);
					TRY(mdemod_estimate_carrier(params, iq_dev, std::min<uint64_t>(n_samples, pos), reinterpret_cast<const uint64_t *>(d_est), 1, win, d_f, d_q, st));
					float fq[4];
					HTRY(hipMemcpyAsync(fq, d_est + 8, sizeof(fq), hipMemcpyDeviceToHost, st));
					HTRY(hipStreamSynchronize(st));
					const double thr = 2 * kPi * 100.0 / (static_cast<double>(params->symrate) * (params->oqpsk ? 2 : 1));     /* genuine locks are within 40 Hz by then, false ones start at 160 */
					genuine = !(fq[2] >= 8.0f && std::fabs(static_cast<double>(seed.pll_freq) - static_cast<double>(fq[0])) > thr);
					waiting_for_genuine = !genuine;
				} else {
					genuine = !waiting_for_genuine;         /* between two looks: what the last look said */
				}
			}
			if (genuine) break;
		}
		if (pos >= o.max_pilot_samples) break;
	}
	po.P = pos; po.nsym = nsym;
	po.hist.resize(2 * static_cast<size_t>(mdemod_history_len(pilot.c)));
	TRY(mdemod_get_history(pilot.c, 0, po.hist.data(), st));
	po.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
	return MDEMOD_OK;
}

/* The entry proper: one object per call, one method per stage of NOTEBOOK.md 3.1 (they used to be one 635-line function).  `need(upto)`,
 * when given, returns once samples [0, upto) of iq_dev are there: the host-buffer entry copies the recording in behind the serial
 * head. */
struct Stitcher {
	/* the call */
	const mdemod_params *params; const mdemod_recording_opts *opts_in; const void *iq_dev; uint64_t n_samples;
	int8_t *soft_dev; uint64_t soft_cap_symbols; mdemod_recording_report *rep; hipStream_t st; const std::function<void(uint64_t)> *need;
	mdemod_recording_opts o;
	double osf = 0, symrate = 0, fs = 0; int nco = 1, K = 32, interp = 1; bool dbg = false, settle_auto = false;
	std::chrono::steady_clock::time_point t_tiles;
	/* the serial head */
	PilotOut po; mdemod_stream_state seed; uint64_t P = 0, n_pilot_sym = 0;
	/* the plan: tile i emits [E_i, E_i + len_i), its stream starts at s0_i and runs acq_i + frm_i + stl_i samples before that */
	uint64_t B = 0, A = 0, KP = 0, WS = 0; size_t T = 0;
	std::vector<uint64_t> E, len, s0, acq, frm, stl, q;
	mdemod_params bp; Ctx bank, saved; DevMem mem; float consts[8]; double fmax = 0, tau_pll = 0;
	/* estimates: carrier (centre, fbar, slope) and clock lines (ge_*) on grids of their own, taken while the head runs; per tile
	   the clock seed and the carrier's local slope */
	std::vector<double> tclk, centre, fbar, slope, slope_tile; int nfft = 0; float min_quality = 8.0f; std::vector<uint64_t> wstart; double f_pilot_target = 0;
	std::vector<double> ge_cx; std::vector<float> ge_th, ge_cq; uint32_t ge_wc = 0; bool ge_no_carrier = true, ge_clock_deferred = false;
	/* recorded on the caller's stream at entry: the estimators read iq_dev on a stream of their own and must come after whatever
	   the caller queued on hip_stream to produce it.  Declared BEFORE `est`: members are destroyed in reverse order of declaration,
	   so on every way out of run_all() the thread is joined first and the event it waits on is destroyed after that. */
	struct InputReady { hipEvent_t ev = nullptr; ~InputReady() { if (ev) (void)hipEventDestroy(ev); } } input_ready;
	struct EstThread { std::thread t; int rc = MDEMOD_OK; ~EstThread() { if (t.joinable()) t.join(); } } est;
	/* seeds */
	std::vector<float> f0, tf, gains; std::vector<int32_t> ud; float *d_f0 = nullptr, *d_tf = nullptr, *d_gain = nullptr; int32_t *d_ud = nullptr;
	/* the bank's buffers and what the launches leave in them */
	uint64_t tail_samples = 0, cap_lead = 0, cap = 0, cap_post = 0; int8_t *soft_pre = nullptr, *soft1 = nullptr, *soft_post = nullptr; int32_t *d_rot = nullptr;
	std::vector<uint32_t> cnt_tmp, cnt_pre, cnt1, cnt_post; std::vector<mdemod_status> status_body, status_settle; std::vector<int32_t> R, shift, rot, weak;
	std::vector<uint64_t> stl_off, ends, post_len;
	/* rotations: taken out of the state before settling, left for the output, expected by the second round */
	std::vector<char> run, jump_at, merged; std::vector<int32_t> state_rot, out_rot, expect; bool second_round = false;
	std::vector<const int8_t *> src_of;

	void mark(const char *what)
	{
		if (!dbg) return;
		(void)hipStreamSynchronize(st);
		fprintf(stderr, "[recording] %8.2f ms  %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_tiles).count() * 1e3, what);
	}
	double f_at(double t) const { return interp_at(centre, fbar, t); }
	/* the carrier word the SERIAL loop has at sample t of tile i's lead (see seed_tiles) */
	double f_seed(size_t i, double t) const
	{
		const double lag = slope_tile[i] * osf / nco * tau_pll;
		/* before the hand-over (a lead that starts inside the pilot) the serial run was further away still: same exponential, back to
		   where the pilot's margin began at most, and by no more than one time constant (OQPSK's loop has a quarter of QPSK's: 74x
		   the pilot's offset is not a seed) */
		const double back = -std::min(1.0, static_cast<double>(o.pilot_margin_symbols) * nco / std::max(tau_pll, 1.0));   /* never past the lock, never more than one time constant */
		const double gone = tau_pll > 0 ? std::exp(-std::max(back, (t - static_cast<double>(P)) / osf * nco / tau_pll)) : 0.0;
		const double f = f_at(t) - lag + (o.carrier_seed == 1 ? (static_cast<double>(seed.pll_freq) - f_pilot_target) * gone : 0.0);
		return std::max(-fmax, std::min(fmax, f));
	}
	int put_carrier_seeds(bool at_acquired)
	{
		for (size_t i = 0; i < T; i++) {
			f0[i] = i == 0 ? seed.pll_freq : static_cast<float>(f_seed(i, static_cast<double>(s0[i] + (at_acquired ? acq[i] : 0))));
			ud[i] = i == 0 ? seed.pll_updown : (slope_tile[i] >= 0 ? 1 : -1);
		}
		HTRY(hipMemcpyAsync(d_f0, f0.data(), T * sizeof(float), hipMemcpyHostToDevice, st));
		HTRY(hipMemcpyAsync(d_ud, ud.data(), T * sizeof(int32_t), hipMemcpyHostToDevice, st));
		HTRY(hipStreamSynchronize(st));
		return mdemod_set_carrier_seeds(bank.c, d_f0, d_ud, st);
	}
	/* ---- launches ------------------------------------------------------------------------------------------- */
	/* discard: the streams run and their state advances, nothing is written (capacity 0; the kernels then raise the overflow flag,
	   which means nothing here) - acquisition, frame and most of the settling are only run for their end state */
	int launch(const std::vector<uint64_t> &off, const std::vector<uint64_t> &cnt, int8_t *soft, uint64_t stride,
	                  std::vector<uint32_t> &produced, std::vector<mdemod_status> *status_out = nullptr, bool discard = false)
	{
		std::vector<uint32_t> c32(cnt.begin(), cnt.end());
		uint64_t *d_off; uint32_t *d_cnt;
		TRY(upload(mem, off, &d_off, st));
		TRY(upload(mem, c32, &d_cnt, st));
			TRY(mdemod_process_device(bank.c, iq_dev, d_off, d_cnt, soft, stride, discard ? 0u : static_cast<uint32_t>(stride), st));
		if (discard) HTRY(hipStreamSynchronize(st));
		else TRY(counts_of(bank.c, static_cast<uint32_t>(T), produced, st, status_out));
		for (uint64_t c : cnt) rep->samples_demodulated += c;
		return MDEMOD_OK;
	}

	/* options and the constants everything else derives from */
	int prepare()
	{
		if (!params || !iq_dev || !soft_dev || !rep) return MDEMOD_ERR_PARAM;
		if (params->samplerate <= 0 || params->symrate <= 0) return MDEMOD_ERR_PARAM;
		if (opts_in) o = *opts_in; else mdemod_recording_default_opts(&o);
		osf = static_cast<double>(params->samplerate) / static_cast<double>(params->symrate);
		symrate = params->symrate; fs = params->samplerate;
		nco = params->oqpsk ? 2 : 1;
		if (o.acquire_samples == 0xFFFFFFFFu) o.acquire_samples = static_cast<uint32_t>(2000 * osf);
		if (o.frame_samples == 0xFFFFFFFFu) o.frame_samples = static_cast<uint32_t>(1500 * osf);
		settle_auto = o.settle_samples == 0xFFFFFFFFu;
		if (settle_auto) o.settle_samples = static_cast<uint32_t>(24000 * osf);
		/* symbols between the reference's first lock and the hand-over: the tiles right after it are seeded with a model of the serial
		   loop's remaining convergence, which holds once the lock is this old (measured: the first tiles lose 5 % of their +-1 LSB
		   agreement with 10 000 symbols less; OQPSK's loop, at twice the bandwidth, wanders more and wants 30 000) */
		if (o.pilot_margin_symbols == 0xFFFFFFFFu) o.pilot_margin_symbols = (o.clock_seed || !o.carrier_seed) ? (params->oqpsk ? 30000 : 20000)      /* tiles that start from the pilot's own omega / carrier word: those need longer */
			                                                                                 : (params->oqpsk ? 20000 : 15000);
		if (o.max_pilot_samples == 0xFFFFFFFFFFFFFFFFull) o.max_pilot_samples = static_cast<uint64_t>(1.5e6 * osf);
		if (!o.pilot_block || !o.match_symbols) return MDEMOD_ERR_PARAM;
		memset(rep, 0, sizeof(*rep));
		rep->first_lock_symbol = -1;
		K = static_cast<int>(std::max<uint32_t>(o.match_symbols, 32));     /* fewer symbols cannot tell 4 rotations x 3 shifts apart at 12 dB */
		dbg = o.debug != 0;
		min_quality = 8.0f;                   /* spectral line over the band's mean: below this a window has no line to speak of */
		return MDEMOD_OK;
	}

	/* pilot: the reference's own serial run of the head (run_pilot) */
	int run_head()
	{
		/* ---- pilot ---- */
		TRY(run_pilot(params, o, iq_dev, n_samples, soft_dev, soft_cap_symbols, st, need, po));
		seed = po.seed;
		const uint64_t nsym = po.nsym;
		P = po.P;
		if (need) (*need)(n_samples);                          /* everything after the head reads all over the recording */
		rep->pilot_samples = P; rep->pilot_symbols = seed.n_symbols;
		rep->pilot_locked = seed.pll_locked; rep->first_lock_symbol = seed.first_lock_symbol;
		rep->samples_demodulated = P;
		n_pilot_sym = nsym;
		rep->exact_symbols = n_pilot_sym;
		rep->pilot_seconds = po.seconds;
		t_tiles = std::chrono::steady_clock::now();
		return MDEMOD_OK;
	}

	/* tile grid behind the head */
	int plan_tiles()
	{
		/* ---- plan: tile i emits [E_i, E_i + len_i); its stream starts `lead` samples early -------------------------- */
		if (P >= n_samples) { rep->n_symbols = n_pilot_sym; T = 0; return MDEMOD_OK; }
		if (o.tile_samples == 0) {
			/* as short as fills the lanes (latency of a small recording is the samples ONE lane runs: lead + tile), as long as the
			   lead stays a small part of the work once the GPU is full; kept off powers of two (lanes read at base + l * tile) */
			const double rest_sym = static_cast<double>(n_samples - P) / osf;
			/* Up to ~1000 tiles run one per WAVE (latency kernel, 1.9x a lane's rate with a SIMD to itself): a recording of up to
			   1000 x 41 072 symbols is cut into that many; a longer one into tiles for one residency round of the lane kernels
			   (131 072 lanes, with slack). */
			double b_sym = rest_sym / 1000.0 <= 41072.0 ? std::max(8192.0, rest_sym / 1000.0)
			                                             : std::min(41072.0, std::max(8192.0, rest_sym / 126976.0));
			/* Between 1 000 and 2 048 such tiles (2^27 .. 2^28 samples at configs[1]) the lane kernels would run ~5 000 .. 10 000 tiles
			   of 8 192 symbols on a GPU they do not fill, each lane its lead + tile at a lane's rate; 2 048 waves still run one or two
			   to a SIMD where the library picks the wave kernel for a bank that size (r05: wants_latency_kernel; configs[1] 2^27 tile
			   phase 72 -> 50 ms, 2^28 72 -> 66 ms), with bodies of 20 000 .. 41 072 symbols like the 1 000-tile regime's. */
			if (rest_sym / 1000.0 > 41072.0 && rest_sym / 2048.0 <= 41072.0) {
				mdemod_params probe = *params;
				probe.n_streams = 2048;
				char which[96] = "";
				if (mdemod_plan_kernel(&probe, which, sizeof(which), nullptr, nullptr) == MDEMOD_OK && strstr(which, "demod_kernel_lat"))
					b_sym = rest_sym / 2048.0;
			}
			o.tile_samples = std::max<uint32_t>(4096, (static_cast<uint32_t>(b_sym * osf) + 63) / 64 * 64);
			if ((o.tile_samples & (o.tile_samples - 1)) == 0) o.tile_samples += 64;
		}
		else o.tile_samples = (o.tile_samples + 7) / 8 * 8;       /* a user's tile: whole groups of 8 samples, so that the rows of the bank's buffers (pitch = samples + 8 symbols) start 16-byte aligned for the assembly kernel's 16-byte reads */
		rep->tile_samples = o.tile_samples;
		B = o.tile_samples; A = o.acquire_samples; KP = o.frame_samples; WS = o.settle_samples;
		T = static_cast<size_t>((n_samples - P + B - 1) / B);
		rep->n_tiles = static_cast<uint32_t>(T);
		E.assign(T, 0); len.assign(T, 0); s0.assign(T, 0); acq.assign(T, 0); frm.assign(T, 0); stl.assign(T, 0); q.assign(T, 0);
		/* Round 5 (tools/tile_tail.py, tools/converged_pairs.py): what a tile's +-1 LSB agreement with the serial run hangs on is how
		   far its clock word (timing.c:84, a float: one ulp = 0.076 ppm at configs[1]) sits from the serial run's at the same symbol:
		   0 / 1 / 2 / 3 / 4 / 5 ulps apart agree on 0.9984 / 0.9975 / 0.9970 / 0.9964 / 0.9955 / 0.9943.  Two converged runs of the
		   reference sit 0 / 1 / 2 / 3 / >= 4 ulps apart 50 / 31 / 12 / 4.5 / 2.1 % of the time; a tile that has settled for 24 000
		   symbols starts its body at 36 / 36 / 16 / 7 / 4.7 %, one that has settled for 32 000 at 41 / 36 / 14 / 5.4 / 3 %, and more
		   does not change it.  With bodies of 23 000 symbols (the ~1000-tile regime) the first thousands of symbols are a small part
		   and the share of 4096-symbol windows below 0.99 is 0.16 % either way (the reference's own pairs: 0.054 %); with bodies of
		   8 192 symbols (the lane-kernel regime) it is 0.5 % at 24 000 and 0.2 % at 32 000.  So: tiles settle for 32 000 where
		   the caller does not say (all of them: see the next comment) - EXCEPT behind the hand-over, where the serial run itself is still drifting in with its carrier loop's
		   16 200-symbol pole and a tile that has had longer to converge on the carrier agrees with it LESS (windows of 0.07 .. 0.35 in
		   tiles 1 .. 3 with 32 000 and 40 000; the seed model of seed_tiles covers one time constant back, not two). */
		/* (measured on the SAME windows as 47 converged twins of the serial run - recording.tiled_vs_twins, where the bad windows are
		   mostly the signal's: 8 192-symbol bodies 0.41 % of windows below 0.99 against the twins' 0.16 % with 24 000, 0.17 % with 32 000;
		   23 000-symbol bodies 0.18 % against 0.054 % with 24 000 - so every tile takes the 32 000, not only the short ones) */
		const uint64_t WS_far = settle_auto ? static_cast<uint64_t>(32000 * osf) : WS;
		const uint64_t far_from = P + static_cast<uint64_t>(50000 * osf);
		for (size_t i = 0; i < T; i++) {
			E[i] = P + i * B; len[i] = std::min<uint64_t>(B, n_samples - E[i]);
			if (i == 0) { s0[i] = q[i] = E[i]; acq[i] = frm[i] = stl[i] = 0; continue; }     /* tile 0: the pilot's exact continuation */
			const uint64_t lead = std::min<uint64_t>(A + KP + (E[i] >= far_from ? WS_far : WS), E[i]);
			s0[i] = E[i] - lead;
			acq[i] = std::min<uint64_t>(A, lead); frm[i] = std::min<uint64_t>(KP, lead - acq[i]); stl[i] = lead - acq[i] - frm[i];
			q[i] = s0[i] + acq[i] + frm[i];
		}

		bp = *params; bp.n_streams = static_cast<uint32_t>(T);
		mark("plan");
		TRY(mdemod_create(&bp, &bank.c));
		mark("bank created");
		TRY(mdemod_get_loop_constants(bank.c, consts));
		const double pll_alpha = consts[0], pll_beta = consts[1];
		fmax = consts[2];
		tau_pll = pll_beta > 0 ? pll_alpha / pll_beta : 0.0;     /* slow pole of the (overdamped) carrier loop, in NCO steps: pll.c:133-140 */
		interp = params->interp_factor;
		return MDEMOD_OK;
	}

	/* ---- symbol-rate line (a pass moves the clock with the carrier: 20 ppm and more between the head and the far end; the loop's
	   integrator needs 8 000 symbols per e-fold to make that up).  The clock is smooth: every tile reads a straight line through
	   the estimates around it (estimate_clocks).  OQPSK's line pair sits at twice the carrier +- the rate: it is looked for around
	   the carrier curve, or - without one - around `carrier_word` (the head's). ---- */
	int clock_lines(hipStream_t es, DevMem &emem, float carrier_word)
	{
		if (o.clock_seed == 0 && n_samples >= 4096) {
			uint32_t wc = 4096;
			while (wc * 2 <= std::min<uint64_t>(n_samples, 1u << 18)) wc *= 2;
			std::vector<uint64_t> cst;
			for (uint64_t a = 0; ; a += wc) {
				if (a + wc >= n_samples) { cst.push_back(n_samples - wc); break; }
				cst.push_back(a);
			}
			const size_t Tc = cst.size();
			const bool curve = o.carrier_seed == 1 && !ge_no_carrier;
			std::vector<float> cf(Tc, carrier_word), cc(Tc, 0.0f);
			ge_cx.assign(Tc, 0.0);
			for (size_t k = 0; k < Tc; k++) {
				ge_cx[k] = static_cast<double>(cst[k]) + 0.5 * wc;
				if (curve) {
					cf[k] = static_cast<float>(interp_at(centre, fbar, std::min(std::max(ge_cx[k], centre.front()), centre.back())));     /* OQPSK: where its two lines are */
					cc[k] = static_cast<float>(interp_at(centre, slope, std::min(std::max(ge_cx[k], centre.front()), centre.back())));
				}
			}
			uint64_t *d_cst; float *d_cf, *d_cc, *d_tfq, *d_cq;
			TRY(upload(emem, cst, &d_cst, es)); TRY(upload(emem, cf, &d_cf, es)); TRY(upload(emem, cc, &d_cc, es));
			TRY(emem.alloc(&d_tfq, Tc)); TRY(emem.alloc(&d_cq, Tc));
			TRY(mdemod_estimate_clock(params, iq_dev, n_samples, d_cst, d_cf, d_cc, static_cast<uint32_t>(Tc), wc, d_tfq, d_cq, es));
			ge_th.assign(Tc, 0.0f); ge_cq.assign(Tc, 0.0f);
			HTRY(hipMemcpyAsync(ge_th.data(), d_tfq, Tc * sizeof(float), hipMemcpyDeviceToHost, es));
			HTRY(hipMemcpyAsync(ge_cq.data(), d_cq, Tc * sizeof(float), hipMemcpyDeviceToHost, es));
			HTRY(hipStreamSynchronize(es));
			ge_wc = wc;
		}
		return MDEMOD_OK;
	}

	/* ---- carrier and symbol clock of the recording, on grids of their own (nothing here needs the head: this runs on a second
	 * host thread and stream while the head does) ------------------------------------------------------------------------------
	 * carrier: 4th-power line (mdemod_estimate_carrier_chirp) of windows side by side over the whole recording, about 20 000
	 * symbols each (65 536 samples at 72k in 230 kS/s, 262 144 at 1 MS/s): the frames are dead-reckoned from tile to tile with the
	 * carrier read off this curve, which has to be good to a fraction of a radian over a tile.  clock: symbol-rate line
	 * (mdemod_estimate_clock) of the estimator's longest windows (2^18 samples: 5e-8 of the rate), side by side as well. */
	int estimate_grid()
	{
		(void)hipGetLastError(); HTRY(hipSetDevice(params->device));      /* (what an earlier call left pending is not this call's: demod_api.cpp select_device) */
		struct OwnStream { hipStream_t s = nullptr; ~OwnStream() { if (s) (void)hipStreamDestroy(s); } } own;
		{
			int least = 0, greatest = 0;                          /* behind the head's launches in the queues */
			(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
			HTRY(hipStreamCreateWithPriority(&own.s, hipStreamNonBlocking, least));
		}
		hipStream_t es = own.s;
		if (input_ready.ev) HTRY(hipStreamWaitEvent(es, input_ready.ev, 0));
		DevMem emem;
		if (need) (*need)(n_samples);                          /* the host-buffer entry is still copying the recording in */
		nfft = static_cast<int>(mdemod_carrier_window_samples(params, static_cast<uint32_t>(std::min(262144.0, 20536.0 * osf))));
		const uint64_t W = static_cast<uint64_t>(nfft);
		wstart.clear();
		if (n_samples <= W) wstart.push_back(0);
		else {
			for (uint64_t a = 0; a + W <= n_samples; a += W) wstart.push_back(a);
			if (wstart.back() + W < n_samples) wstart.push_back(n_samples - W);      /* the tail: one more window, flush with the end */
		}
		const size_t G = wstart.size();
		centre.assign(G, 0.0); fbar.assign(G, 0.0); slope.assign(G, 0.0);         /* rad per NCO step at centre[i]; slope in rad per NCO step per sample */
		for (size_t i = 0; i < G; i++) centre[i] = static_cast<double>(wstart[i]) + nfft / 2;
		ge_no_carrier = true;
		if (o.carrier_seed == 1 && n_samples >= 4096) {
			uint64_t *d_starts, *d_starts_sel; float *d_freq, *d_qual, *d_chirp;
			TRY(upload(emem, wstart, &d_starts, es));
			TRY(emem.alloc(&d_freq, G)); TRY(emem.alloc(&d_qual, G)); TRY(emem.alloc(&d_chirp, G)); TRY(emem.alloc(&d_starts_sel, G));
			std::vector<float> fh(G), qh(G), chirp(G, 0.0f), fsel(G), qsel(G), csel(G);
			std::vector<uint64_t> wsel(G); std::vector<size_t> sel;
			/* a ramp that moves the line by less than a tenth of a bin across the window smears nothing (the Hann window's main lobe is
			   four bins wide): such a window keeps the estimate it has.  Without a Doppler ramp the local slopes are the estimates' own
			   noise, a few hundredths of a bin, and the plain pass is the only one. */
			const double bin = (fs / nfft / 4.0) * 2 * kPi / (symrate * nco);
			for (int pass = 0; pass < 3; pass++) {
				/* pass 0: plain; passes 1, 2: with the local slope taken out of the window (a Doppler ramp smears the line) */
				if (pass) {
					for (size_t i = 0; i < G; i++) {      /* robust local slope: median of up to five neighbouring finite differences */
						double v[5]; int m = 0;
						for (size_t k = i >= 2 ? i - 2 : 0; k <= std::min(G - 1, i + 2); k++) v[m++] = slope[k];
						std::sort(v, v + m);
						chirp[i] = static_cast<float>(v[m / 2]);
					}
					sel.clear();
					for (size_t i = 0; i < G; i++) if (std::fabs(chirp[i]) * nfft > 0.1 * bin) sel.push_back(i);
					if (sel.empty()) break;
					for (size_t k = 0; k < sel.size(); k++) { wsel[k] = wstart[sel[k]]; csel[k] = chirp[sel[k]]; }
					HTRY(hipMemcpyAsync(d_chirp, csel.data(), sel.size() * sizeof(float), hipMemcpyHostToDevice, es));
					HTRY(hipMemcpyAsync(d_starts_sel, wsel.data(), sel.size() * sizeof(uint64_t), hipMemcpyHostToDevice, es));
					HTRY(hipStreamSynchronize(es));
					TRY(mdemod_estimate_carrier_chirp(params, iq_dev, n_samples, d_starts_sel, d_chirp, static_cast<uint32_t>(sel.size()),
					                                  static_cast<uint32_t>(nfft), d_freq, d_qual, es));
					HTRY(hipMemcpyAsync(fsel.data(), d_freq, sel.size() * sizeof(float), hipMemcpyDeviceToHost, es));
					HTRY(hipMemcpyAsync(qsel.data(), d_qual, sel.size() * sizeof(float), hipMemcpyDeviceToHost, es));
					HTRY(hipStreamSynchronize(es));
					for (size_t k = 0; k < sel.size(); k++) { fh[sel[k]] = fsel[k]; qh[sel[k]] = qsel[k]; }
				} else {
					TRY(mdemod_estimate_carrier_chirp(params, iq_dev, n_samples, d_starts, nullptr, static_cast<uint32_t>(G),
					                                  static_cast<uint32_t>(nfft), d_freq, d_qual, es));
					HTRY(hipMemcpyAsync(fh.data(), d_freq, G * sizeof(float), hipMemcpyDeviceToHost, es));
					HTRY(hipMemcpyAsync(qh.data(), d_qual, G * sizeof(float), hipMemcpyDeviceToHost, es));
					HTRY(hipStreamSynchronize(es));
				}
				if (dbg) fprintf(stderr, "[recording] carrier pass %d: %zu of %zu windows of %d\n", pass, pass ? sel.size() : G, G, nfft);
				/* windows without a clear line (fade, interference) take their good neighbours' estimate, interpolated over time;
				   with no good window at all, the pilot's frequency (put in once the head is through) */
				std::vector<size_t> good;
				for (size_t i = 0; i < G; i++) if (qh[i] >= min_quality) good.push_back(i);
				rep->weak_carrier_tiles = static_cast<uint32_t>(G - good.size());
				ge_no_carrier = good.empty();
				if (good.empty()) break;
				{
					std::vector<double> gx, gv;
					for (size_t g : good) { gx.push_back(centre[g]); gv.push_back(fh[g]); }
					for (size_t i = 0; i < G; i++)
						fbar[i] = qh[i] >= min_quality ? static_cast<double>(fh[i]) : interp_at(gx, gv, std::min(std::max(centre[i], gx.front()), gx.back()));
				}
				/* one estimate far off the line through its neighbours (a spur, a burst: 1 in 1e5 windows) would put every later tile
				   in the wrong frame: the carrier is smooth, the neighbours decide */
				if (G >= 5) {
					std::vector<double> fixed = fbar;
					for (size_t i = 1; i + 1 < G; i++) {
						const double w = (centre[i] - centre[i - 1]) / (centre[i + 1] - centre[i - 1]);
						const double pred = fbar[i - 1] + (fbar[i + 1] - fbar[i - 1]) * w;
						if (std::fabs(fbar[i] - pred) > 6e-6 / nco) {
							/* which of the three is the odd one?  the one whose own neighbours agree with each other without it */
							const size_t a = i >= 2 ? i - 2 : i - 1, b = std::min(G - 1, i + 2);
							const double wa = (centre[i] - centre[a]) / (centre[b] - centre[a]);
							const double pred2 = fbar[a] + (fbar[b] - fbar[a]) * wa;
							if (std::fabs(pred - pred2) < std::fabs(fbar[i] - pred2)) fixed[i] = pred;
						}
					}
					fbar = fixed;
				}
				for (size_t i = 0; i < G; i++) {
					const size_t lo = i ? i - 1 : 0, hi = std::min(G - 1, i + 1);
					slope[i] = centre[hi] > centre[lo] ? (fbar[hi] - fbar[lo]) / (centre[hi] - centre[lo]) : 0.0;
				}
			}
		}
		/* the symbol-rate lines, unless they have to wait for the head's carrier word (OQPSK without a carrier curve) */
		ge_cx.clear(); ge_th.clear(); ge_cq.clear(); ge_wc = 0;
		ge_clock_deferred = params->oqpsk && (o.carrier_seed != 1 || ge_no_carrier);
		if (!ge_clock_deferred) TRY(clock_lines(es, emem, 0.0f));
		return MDEMOD_OK;
	}

	/* the carrier curve as the tiles read it: joined with the estimator thread; without a line anywhere, the head's own word */
	int estimate_carriers()
	{
		if (est.t.joinable()) est.t.join();
		TRY(est.rc);
		mark("estimates joined");
		if (o.carrier_seed != 1 || ge_no_carrier) {
			centre.assign(2, 0.0); centre[1] = static_cast<double>(std::max<uint64_t>(n_samples, 1));
			fbar.assign(2, static_cast<double>(seed.pll_freq)); slope.assign(2, 0.0);
		}
		/* the carrier's local slope where tile i is dead-reckoned: over [q_{i-1}, q_i] */
		slope_tile.assign(T, 0.0);
		for (size_t i = 0; i < T; i++) {
			const double c = i == 0 ? static_cast<double>(P) : 0.5 * (static_cast<double>(i == 1 ? P : q[i - 1]) + static_cast<double>(q[i]));
			slope_tile[i] = interp_at(centre, slope, std::min(std::max(c, centre.front()), centre.back()));
		}
		mark("carrier lines");
		return MDEMOD_OK;
	}

	/* every tile's clock seed: a straight line through the clock estimates around it; the pilot's hand-over target */
	int estimate_clocks()
	{
		tclk.assign(T, static_cast<double>(seed.t_freq));   /* symbol clock seeds, rad per interpolated step */
		if (ge_clock_deferred && T > 1) TRY(clock_lines(st, mem, seed.pll_freq));
		if (o.clock_seed == 0 && T > 1 && ge_wc) {
			const std::vector<double> &cx = ge_cx; const std::vector<float> &th = ge_th, &cq = ge_cq;
			const size_t Tc = cx.size(); const uint32_t wc = ge_wc;
			size_t lo = 0, hi = 0, weak = 0;
			const double span = 2.5 * wc;                            /* five windows: the clock moves by < 1e-6 of the rate per second */
			for (size_t i = 0; i < T; i++) {
				const double t = static_cast<double>(s0[i]);
				while (lo + 1 < Tc && cx[lo] < t - span) lo++;
				while (hi + 1 < Tc && cx[hi + 1] <= t + span) hi++;
				double sw = 0, sx = 0, sy = 0, sxx = 0, sxy = 0;
				for (size_t j = lo; j <= hi; j++) {
					if (cq[j] < min_quality) continue;
					const double x = cx[j] - t, y = static_cast<double>(th[j]) - static_cast<double>(seed.t_freq);
					sw += 1; sx += x; sy += y; sxx += x * x; sxy += x * y;
				}
				if (sw < 1) { weak++; continue; }                       /* no line anywhere near: the pilot's omega stays */
				const double det = sw * sxx - sx * sx;
				const double at_t = (sw >= 3 && det > 1e-6 * sw * sxx) ? (sy * sxx - sx * sxy) / det : sy / sw;
				const double lim = static_cast<double>(consts[6]);   /* timing.c:80-86 keeps the loop within this of its centre */
				tclk[i] = static_cast<double>(consts[5]) + std::max(-lim, std::min(lim, static_cast<double>(seed.t_freq) + at_t - static_cast<double>(consts[5])));
			}
			rep->weak_clock_tiles = static_cast<uint32_t>(weak);
			if (dbg) fprintf(stderr, "[recording] clock seeds from %zu windows of %u: pilot %.9g, tiles %.9g .. %.9g (weak %zu)\n", Tc, wc, static_cast<double>(seed.t_freq), tclk[T > 1 ? 1 : 0], tclk[T - 1], weak);
		}
		mark("carrier estimates");
		/* The carrier loop is heavily overdamped: its frequency word follows a moving carrier with the lag slope * tau (pll.c:115
		   integrates beta * e, the phase term alpha * e does the tracking), and right after the pilot's hand-over the serial run is
		   still converging with the same time constant.  A tile settles for less than tau, so it is seeded with the frequency the
		   SERIAL loop has at that point, not with the carrier: estimate - lag + what is left of the pilot's own offset. */
		f_pilot_target = f_at(static_cast<double>(P)) - (T > 1 ? slope_tile[0] * osf / nco * tau_pll : 0.0);
		/* a lock the reference declares far from the carrier (its OQPSK loop does on about half of all recordings with an offset, and
		   never leaves it) is reported: the tiles demodulate the signal, the reference from there on does not */
		if (o.carrier_seed == 1 && T > 1 && seed.pll_locked && !ge_no_carrier && rep->weak_carrier_tiles < centre.size() / 2 &&
		    std::fabs(static_cast<double>(seed.pll_freq) - f_pilot_target) > 2 * kPi * 100.0 / (symrate * nco)) rep->pilot_locked = 2;
		return MDEMOD_OK;
	}

	/* loop state, gains and carrier words of every stream */
	int seed_tiles()
	{
		/* ---- seeds ------------------------------------------------------------------------------------------------ */
		TRY(mdemod_set_state_all(bank.c, &seed, st));
		TRY(mdemod_set_state(bank.c, 0, &seed, st));
		TRY(mdemod_set_history(bank.c, 0, po.hist.data(), st));
		f0.assign(T, 0.0f); tf.assign(T, seed.t_freq); gains.assign(T, seed.agc_gain); ud.assign(T, 0);
		for (size_t i = 1; i < T; i++) tf[i] = static_cast<float>(tclk[i]);
		TRY(mem.alloc(&d_f0, T)); TRY(mem.alloc(&d_tf, T)); TRY(mem.alloc(&d_gain, T)); TRY(mem.alloc(&d_ud, T));
		/* the reference's AGC moves by 1e-4 * 190 / gain of itself per symbol (agc.c:13-25): with s16-scale input (gain ~ 0.03) it is
		   there within a few symbols whatever it starts from; only a slow one (float input around +-1: tens of thousands of symbols)
		   needs a seed per tile, and only then is the recording read once more for its power */
		const bool slow_agc = 6.0 * static_cast<double>(seed.agc_gain) / (1e-4 * 190.0) > 0.25 * static_cast<double>(A) / osf;
		if (T > 1 && !slow_agc) {
			HTRY(hipMemcpyAsync(d_gain, gains.data(), T * sizeof(float), hipMemcpyHostToDevice, st));
			HTRY(hipMemcpyAsync(d_tf, tf.data(), T * sizeof(float), hipMemcpyHostToDevice, st));
			HTRY(hipStreamSynchronize(st));
			TRY(put_carrier_seeds(false));
		}
		if (T > 1 && slow_agc) {
			/* AGC gain seeds: g* = c / sqrt(sample power), c fitted on the pilot's last blocks, then the reference's AGC in closed
			   form over the tiles' powers (agc.c:13-25) */
			const size_t nb = std::min<size_t>(10, po.blocks.size());
			const size_t b0 = po.blocks.size() - nb;
			std::vector<uint64_t> ws; std::vector<uint32_t> wl;
			for (size_t j = b0; j < po.blocks.size(); j++) { ws.push_back(po.blocks[j].start); wl.push_back(po.blocks[j].len); }
			for (size_t i = 0; i < T; i++) { ws.push_back(E[i]); wl.push_back(static_cast<uint32_t>(len[i])); }
			uint64_t *d_ws; uint32_t *d_wl; float *d_wp;
			TRY(upload(mem, ws, &d_ws, st));
			TRY(upload(mem, wl, &d_wl, st));
			TRY(mem.alloc(&d_wp, ws.size()));
			HTRY(hipStreamSynchronize(st));
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
				const PilotBlock &pb = po.blocks[b0 + j];
				const uint64_t before = (b0 + j) ? po.blocks[b0 + j - 1].symbols_after : 0;
				blk_gain[j] = pb.gain_after; blk_power[j] = wp[j]; blk_syms[j] = static_cast<double>(pb.symbols_after - before);
			}
			const double c = nb ? fit_agc_calibration(blk_gain, blk_power, blk_syms) : 0.0;
			/* gE[k]: the serial run's gain at E_k; stream i starts its lead before E_i and takes the gain of the boundary at or
			   before its start */
			std::vector<double> gE(T + 1);
			gE[0] = seed.agc_gain;
			for (size_t k = 0; k < T; k++) gE[k + 1] = agc_step(gE[k], c, wp[nb + k], static_cast<double>(len[k]) * symrate / fs);
			for (size_t i = 1; i < T; i++) {
				const size_t back = static_cast<size_t>((E[i] - s0[i] + B - 1) / B);
				gains[i] = static_cast<float>(gE[i > back ? i - back : 0]);
			}
			HTRY(hipMemcpyAsync(d_gain, gains.data(), T * sizeof(float), hipMemcpyHostToDevice, st));
			HTRY(hipMemcpyAsync(d_tf, tf.data(), T * sizeof(float), hipMemcpyHostToDevice, st));
			HTRY(hipStreamSynchronize(st));
			TRY(mdemod_set_gain_seeds(bank.c, d_gain, st));
			TRY(put_carrier_seeds(false));
		}
		return MDEMOD_OK;
	}

	/* acquire, re-seed the integrators, frame by dead reckoning, checkpoint */
	int acquire_and_frame()
	{
		mark("seeds");
		/* of the lead only the last symbols are kept: what the seam check compares with the predecessor's tail */
		tail_samples = static_cast<uint64_t>((K + 24) * osf * 1.05) + 16;
		cap_lead = std::max<uint64_t>(8, mdemod_max_symbols(bank.c, tail_samples));
		cap = mdemod_max_symbols(bank.c, B);
		TRY(mem.alloc(&soft_pre, T * cap_lead * 2));
		TRY(mem.alloc(&soft1, T * cap * 2));
		mark("output buffers allocated");
		cnt_tmp.clear(); cnt_pre.assign(T, 0); cnt1.assign(T, 0);
		status_body.clear();
		R.assign(T, 0);
		if (T > 1) {
			/* acquire, then the two integrators back on their seeds (the gain keeps what it found) */
			TRY(launch(s0, acq, soft_pre, cap_lead, cnt_tmp, nullptr, true));
			mark("acquire");
			TRY(put_carrier_seeds(true));
			TRY(mdemod_set_clock_seeds(bank.c, d_tf, st));
			std::vector<uint64_t> off(T);
			for (size_t i = 0; i < T; i++) off[i] = s0[i] + acq[i];
			TRY(launch(off, frm, soft_pre, cap_lead, cnt_tmp, nullptr, true));
			mark("frame");

			/* ---- frames by dead reckoning along the chain pilot -> tile 1 -> tile 2 ... ---------------------------------- */
			std::vector<mdemod_stream_state> qs(T);
			TRY(mdemod_get_states(bank.c, 0, static_cast<uint32_t>(T), qs.data(), st));
			double th_prev = seed.pll_phase, t_prev = last_nco_time(seed, static_cast<double>(P), interp, params->oqpsk);
			size_t i_prev = 0;                                     /* the chain's last trusted link */
			int32_t acc_prev = 0; double res2 = 0.0; size_t hung = 0;
			for (size_t i = 1; i < T; i++) {
				const double th = qs[i].pll_phase, tt = last_nco_time(qs[i], static_cast<double>(q[i]), interp, params->oqpsk);
				const double t_mid = 0.5 * (t_prev + tt) / interp;
				double res;
				/* NCO steps between the two: the symbol period of THIS stretch of the recording (a pass moves the clock: 50 ppm over a
				   41 072-symbol tile would be two steps with the pilot's period) */
				const double steps_per_nco = 2 * kPi / (0.5 * (tclk[i_prev] + tclk[i])) / nco;
				const int32_t accr = (acc_prev + frame_between(th_prev, t_prev, th, tt, f_at(t_mid), steps_per_nco, &res)) & 3;
				R[i] = accr; res2 += res * res;
				/* Two streams that are on the symbols are a whole number of steps apart (seen: +-0.02).  One in 1e5 tiles is still hung
				   up between two symbols after acquire + frame (the Mueller-Mueller detector's unstable equilibrium): its own phase
				   says little (best guess kept: the seam check will see), and the chain must not go through it - its successor is
				   reckoned from the last stream that was on the symbols.  (Found on the 6.5 G-sample recording: tile 121 764 was
				   0.44 of a step off, both its links had a residual of -0.7 rad, one rounded the wrong way and 5 184 tiles behind
				   it were repaired for it.) */
				const double steps = (tt - t_prev) / steps_per_nco, off_grid = std::fabs(steps - std::nearbyint(steps));
				const bool trusted = off_grid <= 0.2 || i - i_prev > 4;
				if (dbg && o.debug_tile >= 0 && std::llabs(static_cast<long long>(i) - static_cast<long long>(o.debug_tile)) <= 2)
					fprintf(stderr, "[recording]   trace %zu (from %zu): theta %.5f t %.3f (dt %.3f steps = %.4f nco) f %.9g res %.4f R %d t_freq %.9g t_phase %.5f locked %d pll_freq %.9g\n", i, i_prev, th, tt, tt - t_prev,
					        steps, f_at(t_mid), res, accr, static_cast<double>(qs[i].t_freq), static_cast<double>(qs[i].t_phase), qs[i].pll_locked, static_cast<double>(qs[i].pll_freq));
				if (dbg && (std::fabs(res) > 0.5 || !trusted)) fprintf(stderr, "[recording]   frame %zu: dead-reckoning residual %.3f rad, %.3f of a step off the symbols%s\n", i, res, off_grid, trusted ? "" : " (not chained through)");
				if (trusted) { th_prev = th; t_prev = tt; i_prev = i; acc_prev = accr; } else hung++;
			}
			if (dbg) fprintf(stderr, "[recording] %zu tiles not on the symbols when their frame was taken\n", hung);
			rep->frame_residual_rms = static_cast<float>(std::sqrt(res2 / static_cast<double>(T - 1)));
			/* checkpoint of the bank before any rotation: what a repair starts from */
			if (o.repair) {
				TRY(mdemod_create(&bp, &saved.c));
				TRY(mdemod_copy_state(saved.c, bank.c, st));
			}
			mark("frames dead-reckoned, checkpoint");
		}
		TRY(mem.alloc(&d_rot, T));
		shift.assign(T, 0); rot.assign(T, 0); weak.assign(T, 0);
		stl_off.assign(T, 0);
		for (size_t i = 0; i < T; i++) stl_off[i] = q[i];
		const uint64_t post = 4096;                               /* OQPSK: look-ahead into the next tile for the seam check */
		soft_post = nullptr; cap_post = 0;
		cnt_post.assign(T, 0);
		ends.assign(T, 0); post_len.assign(T, 0);
		for (size_t i = 0; i < T; i++) { ends[i] = E[i] + len[i]; post_len[i] = std::min<uint64_t>(post, n_samples - ends[i]); }
		if (params->oqpsk) {
			cap_post = std::max<uint64_t>(8, mdemod_max_symbols(bank.c, post));
			TRY(mem.alloc(&soft_post, T * cap_post * 2));
		}
		return MDEMOD_OK;
	}

	/* settle + body (+ OQPSK look-ahead), every seam, one repair round for odd tiles */
	int settle_body_seams()
	{
		/* settle + body (+ OQPSK look-ahead) of the streams in `run` (all of them the first time), then every seam again */
		run.assign(T, 1);
		state_rot = R;                    /* output rotation taken out of each stream's state before it settles */
		out_rot.assign(T, 0); expect.assign(T, 0);     /* rotation left for the output; what the second round should find */
		jump_at.assign(T, 0);                       /* seams that still show a rotation after the repair */
		second_round = false;
		for (int round = 0; round < 2; round++) {
			std::vector<int32_t> qt(T);
			for (size_t i = 0; i < T; i++) qt[i] = run[i] ? (4 - state_rot[i]) & 3 : 0;
			HTRY(hipMemcpyAsync(d_rot, qt.data(), T * sizeof(int32_t), hipMemcpyHostToDevice, st));
			HTRY(hipStreamSynchronize(st));
			TRY(mdemod_rotate_carrier(bank.c, d_rot, st));
			auto masked = [&](const std::vector<uint64_t> &c) { std::vector<uint64_t> m(T); for (size_t i = 0; i < T; i++) m[i] = run[i] ? c[i] : 0; return m; };
			std::vector<mdemod_status> stat;
			{
				std::vector<uint64_t> stl_a(T), stl_b(T), off_b(T);
				for (size_t i = 0; i < T; i++) { stl_b[i] = std::min<uint64_t>(stl[i], tail_samples); stl_a[i] = stl[i] - stl_b[i]; off_b[i] = q[i] + stl_a[i]; }
				TRY(launch(stl_off, masked(stl_a), soft_pre, cap_lead, cnt_tmp, nullptr, true));
				std::vector<mdemod_status> after_settle;
				TRY(launch(off_b, masked(stl_b), soft_pre, cap_lead, cnt_tmp, o.debug >= 3 ? &after_settle : nullptr));
				if (o.debug >= 3) { if (status_settle.empty()) status_settle = after_settle; for (size_t i = 0; i < T; i++) if (run[i]) status_settle[i] = after_settle[i]; }
			}
			mark("settle");
			for (size_t i = 0; i < T; i++) if (run[i]) cnt_pre[i] = cnt_tmp[i];
			TRY(launch(E, masked(len), soft1, cap, cnt_tmp, &stat));
			mark("body");
			if (status_body.empty()) status_body = stat;
			for (size_t i = 0; i < T; i++) if (run[i]) { cnt1[i] = cnt_tmp[i]; status_body[i] = stat[i]; }
			if (params->oqpsk) {
				TRY(launch(ends, masked(post_len), soft_post, cap_post, cnt_tmp));
				for (size_t i = 0; i < T; i++) if (run[i]) cnt_post[i] = cnt_tmp[i];
			}

			/* ---- seams: tile i's settled tail against its predecessor's body tail (tile 1: against tile 0 = the serial run) ---- */
			std::vector<TailPair> pairs(T > 1 ? T - 1 : 0);
			for (size_t i = 1; i < T; i++) {
				TailPair &p = pairs[i - 1];
				p.a = soft1 + (i - 1) * cap * 2; p.a_cnt = cnt1[i - 1];
				p.b = soft_pre + i * cap_lead * 2; p.b_cnt = cnt_pre[i];
				p.b_rot = 0; p.force_weak = stl[i] == 0;
			}
			std::vector<int32_t> sh, ro, we;
			TRY(run_match(mem, pairs, K, sh, ro, we, st, params->oqpsk ? 1 : 0));
			mark("seams");
			for (size_t i = 1; i < T; i++) { shift[i] = params->oqpsk ? 0 : sh[i - 1]; rot[i] = we[i - 1] ? 0 : (ro[i - 1] & 3); weak[i] = we[i - 1]; }
			/* rotation each stream's output still needs to sit in the serial run's frame: b * j^rot matches a, summed along the chain */
			std::vector<int32_t> C(T, 0);
			for (size_t i = 1; i < T; i++) C[i] = (C[i - 1] + rot[i]) & 3;
			bool odd = false;
			for (size_t i = 1; i < T; i++) odd = odd || (C[i] & 1);
			if (dbg) {
				fprintf(stderr, "[recording] round %d: T=%zu tile=%u lead=%u+%u+%u\n", round, T, o.tile_samples, o.acquire_samples, o.frame_samples, o.settle_samples);
				for (size_t i = 1; i < T; i++)
					if (rot[i] || weak[i] || shift[i] || run[i] != 1 || o.debug >= 2)
						fprintf(stderr, "[recording]   seam %zu: rot %d weak %d shift %d C %d run %d R %d cnt_pre %u cnt1 %u locked %d f0 %.6f q %.1f\n", i, rot[i], weak[i], shift[i], C[i], (int)run[i], R[i],
						        cnt_pre[i], cnt1[i], status_body.size() > i ? status_body[i].locked : -1, f_at(static_cast<double>(E[i])), 0.0);
			}
			if (round == 0) {
				for (size_t i = 1; i < T; i++) rep->frame_misses += rot[i] ? 1 : 0;
				out_rot = C;
				if (!odd) break;
				size_t n_odd = 0;
				for (size_t i = 1; i < T; i++) n_odd += (C[i] & 1);
				/* A repair is one more settle + body for the lanes concerned - as long as for all of them.  It pays when dead reckoning
				   failed broadly; one odd tile in 10^5 (its decisions are exact once the output is turned, only its soft values sit on
				   the other rail's timing noise: 2 % of them off by more than an LSB) does not move the result by 1e-6. */
				/* OQPSK has no such choice: a tile a quarter turn off pairs its rails one symbol apart (demod.c:66-76), which turning
				   the output cannot undo */
				if (!o.repair || !saved.c || (!params->oqpsk && n_odd * 200 < T)) {
					for (size_t i = 1; i < T; i++) rep->rotation_jumps += (!o.repair && (rot[i] & 1)) ? 1 : 0;
					rep->odd_tiles_kept = static_cast<uint32_t>(n_odd);
					break;
				}
				/* repair: a stream an odd number of quarter turns off has settled on the other rail's noise (timing.c:65-66).  It
				   starts again from the checkpoint with the measured rotation taken out of its state; streams that are 0 or 180
				   degrees off keep their run (180 degrees is exact on the output).  The checkpoint restores every stream, the ones
				   that keep their run are simply not launched again. */
				TRY(mdemod_copy_state(bank.c, saved.c, st));
				for (size_t i = 0; i < T; i++) {
					run[i] = (C[i] & 1) ? 1 : 0;
					if (run[i]) { state_rot[i] = (R[i] + C[i]) & 3; expect[i] = 0; rep->repaired_tiles++; }
					else expect[i] = C[i];
				}
			} else {
				out_rot = C;
				second_round = true;
				/* what the second round was not expected to find.  A re-run tile that sits half a turn off on BOTH its seams is no jump:
				   its output is turned (exact) and continuous with its neighbours */
				std::vector<int32_t> D(T);
				for (size_t i = 0; i < T; i++) D[i] = (C[i] - expect[i]) & 3;
				for (size_t i = 1; i < T; i++) {
					if (D[i] == D[i - 1]) continue;
					if (!((D[i] - D[i - 1]) & 1) && (i + 1 >= T || D[i + 1] == D[i - 1])) { i++; continue; }
					jump_at[i] = 1;
				}
			}
		}
		return MDEMOD_OK;
	}

	/* tiles the repair did not cure run as the tail of their predecessors' streams */
	int hand_over_uncured()
	{
		/* ---- tiles the repair did not cure --------------------------------------------------------------------------------------
		   A stream that slips a quarter turn on its way in one run and not in the other (seen on OQPSK at low resolution / under a
		   ramp: 2 of 260 soak recordings) comes out odd again, the other way round, when it is re-run with its state turned.  Its
		   samples are then demodulated by its PREDECESSOR instead: that stream runs again from the checkpoint through its own tile
		   (the same bytes as before) and on through the next one - no seam, no rotation to get wrong. ---- */
		merged.assign(T, 0);                         /* tile i is the second half of stream i-1's long run */
		src_of.assign(T, nullptr);
		if (second_round && saved.c) {
			std::vector<size_t> preds;
			for (size_t i = 1; i < T; i++)
				if (run[i] && (out_rot[i] & 1) && !(out_rot[i - 1] & 1) && (i + 1 >= T || !(out_rot[i + 1] & 1)) && !merged[i - 1] && len[i] > 0) { merged[i] = 1; preds.push_back(i - 1); }
			const uint64_t cap_fix = mdemod_max_symbols(bank.c, 2 * B);
			int8_t *soft_fix = nullptr;
			if (!preds.empty()) {
				/* rows are indexed by stream (the bank's launches have one pitch for all): rows up to the last stream that runs
				   are enough.  No room for them: the tiles stay as they are, a quarter turn off, and are REPORTED (rotation_jumps,
				   below) - better than failing a recording whose every other symbol is done. */
				const size_t rows = preds.back() + 1;
				if (mem.alloc(&soft_fix, rows * cap_fix * 2) != MDEMOD_OK) {
					for (size_t j : preds) merged[j + 1] = 0;
					preds.clear();
					soft_fix = nullptr;
				}
			}
			if (!preds.empty()) {
				TRY(mdemod_copy_state(bank.c, saved.c, st));
				std::vector<int32_t> qt(T, 0);
				std::vector<uint64_t> c_stl(T, 0), c_body(T, 0), c_post(T, 0), o_post(T, 0);
				for (size_t j : preds) { qt[j] = (4 - state_rot[j]) & 3; c_stl[j] = stl[j]; c_body[j] = len[j] + len[j + 1]; o_post[j] = ends[j + 1]; c_post[j] = post_len[j + 1]; }
				HTRY(hipMemcpyAsync(d_rot, qt.data(), T * sizeof(int32_t), hipMemcpyHostToDevice, st));
				HTRY(hipStreamSynchronize(st));
				TRY(mdemod_rotate_carrier(bank.c, d_rot, st));
				std::vector<uint32_t> cnt_fix;
				TRY(launch(stl_off, c_stl, soft_pre, cap_lead, cnt_tmp, nullptr, true));
				TRY(launch(E, c_body, soft_fix, cap_fix, cnt_fix));
				if (params->oqpsk) {
					TRY(launch(o_post, c_post, soft_post, cap_post, cnt_tmp));
					for (size_t j : preds) cnt_post[j] = cnt_tmp[j];
				}
				/* the seam behind the long run: its tail against the next tile's settled tail */
				std::vector<TailPair> pairs;
				std::vector<size_t> at;
				for (size_t j : preds) {
					const size_t i = j + 1;
					src_of[j] = soft_fix + j * cap_fix * 2; cnt1[j] = cnt_fix[j]; cnt1[i] = 0;
					out_rot[i] = out_rot[j]; shift[i] = 0; weak[i] = 0; jump_at[i] = 0;
					rep->repaired_tiles++;
					if (i + 1 < T) {
						TailPair p;
						p.a = src_of[j]; p.a_cnt = cnt1[j]; p.b = soft_pre + (i + 1) * cap_lead * 2; p.b_cnt = cnt_pre[i + 1]; p.b_rot = 0; p.force_weak = stl[i + 1] == 0;
						pairs.push_back(p); at.push_back(i + 1);
					}
				}
				std::vector<int32_t> sh, ro, we;
				TRY(run_match(mem, pairs, K, sh, ro, we, st, params->oqpsk ? 1 : 0));
				for (size_t k = 0; k < at.size(); k++) {
					const size_t n = at[k];
					if (!params->oqpsk) shift[n] = sh[k];
					weak[n] = we[k];
					/* measured: tile n needs ro[k] quarter turns against the long run; it was given out_rot[n] against the chain */
					jump_at[n] = (!we[k] && ((out_rot[n - 1] + ro[k] - out_rot[n]) & 3)) ? 1 : 0;
				}
				if (dbg) fprintf(stderr, "[recording] %zu tiles handed to their predecessors' streams\n", preds.size());
				mark("merged");
			}
		}
		for (size_t i = 1; i < T; i++) rep->rotation_jumps += jump_at[i];
		if (params->oqpsk && T > 1) {
			/* rails come from firings half a symbol apart: the one-symbol disagreement is looked for on heads, tile i-1's
			   look-ahead past its end against tile i's body (both start on the same sample) */
			std::vector<TailPair> heads(T - 1);
			for (size_t i = 1; i < T; i++) {
				const size_t pr = merged[i - 1] ? i - 2 : i - 1;       /* the stream that ran up to this tile's first sample */
				heads[i - 1].a = soft_post + pr * cap_post * 2;     heads[i - 1].a_cnt = cnt_post[pr];
				heads[i - 1].b = soft1 + i * cap * 2;               heads[i - 1].b_cnt = cnt1[i];
				heads[i - 1].b_rot = 0; heads[i - 1].force_weak = merged[i] ? 1 : 0;
			}
			std::vector<int32_t> sh2, r2, w2;
			TRY(run_match(mem, heads, K, sh2, r2, w2, st, 2));
			for (size_t i = 1; i < T; i++) { if (merged[i]) continue; shift[i] = w2[i - 1] ? 0 : -sh2[i - 1]; if (w2[i - 1]) weak[i] = 1; }
		}
		for (size_t i = 0; i < T; i++) rep->weak_seams += weak[i];
		return MDEMOD_OK;
	}

	/* concatenate pilot ++ tiles with the seam fixes */
	int assemble()
	{
		/* ---- concatenate: pilot ++ tiles, with the seam fixes and what rotation is left on the output ---- */
		std::vector<TileCopy> copies(T);
		for (size_t i = 0; i < T; i++) {
			copies[i].src = src_of[i] ? src_of[i] : soft1 + i * cap * 2; copies[i].rot = out_rot[i]; copies[i].keep = cnt1[i];
			copies[i].head = nullptr; copies[i].head_rot = 0;
			if (i && shift[i] == -1) {
				/* the symbol straddling the seam is missing on both sides: take it from tile i's own settled run (its last symbol
				   before the body; OQPSK: from the predecessor's look-ahead) */
				const size_t pr = merged[i - 1] ? i - 2 : i - 1;
				if (params->oqpsk) { if (cnt_post[pr] > 0) { copies[i].head = soft_post + pr * cap_post * 2; copies[i].head_rot = out_rot[pr]; } }
				else if (cnt_pre[i] > 0) { copies[i].head = soft_pre + (i * cap_lead + cnt_pre[i] - 1) * 2; copies[i].head_rot = out_rot[i]; }
			}
		}
		uint64_t out_pos = n_pilot_sym;
		for (size_t i = 0; i < T; i++) {
			const size_t nx = (i + 1 < T && merged[i + 1]) ? i + 2 : i + 1;      /* (a merged tile is part of this one's run) */
			const uint32_t drop = (!merged[i] && nx < T && shift[nx] == 1) ? 1 : 0;      /* the successor emits this tile's last symbol too */
			copies[i].keep = copies[i].keep > drop ? copies[i].keep - drop : 0;
			if (shift[i] == -1 && !copies[i].head) shift[i] = 0;
			copies[i].dst = out_pos;
			out_pos += copies[i].keep + (copies[i].head ? 1 : 0);
			if (shift[i]) rep->seam_fixes++;
			if (i == 0) rep->exact_symbols = out_pos;
		}
		if (out_pos > soft_cap_symbols) return MDEMOD_ERR_OVERFLOW;
		if (rep->first_lock_symbol < 0) {
			/* The pilot never locked (a recording that starts before the signal does): the lock gate (main.c:308-315) opens
			   at the first tile whose stream reports a first lock - inside its emitted body, or before it (then the whole
			   body counts).  Approximate to the tiles' own acquisition, which is faster than the serial sweep. */
			for (size_t i = 0; i < T; i++) {
				const int64_t fl = status_body[i].first_lock_symbol;
				if (fl < 0) continue;
				const uint64_t n_start = status_body[i].n_symbols - cnt1[i];
				const uint64_t inside = static_cast<uint64_t>(fl) > n_start ? static_cast<uint64_t>(fl) - n_start : 0;
				rep->first_lock_symbol = static_cast<int64_t>(copies[i].dst + std::min<uint64_t>(inside, copies[i].keep));
				break;
			}
		}
		TileCopy *d_copies;
		TRY(upload(mem, copies, &d_copies, st));
		/* few tiles: several blocks per tile, so that the copy still fills the GPU */
		const unsigned slices = static_cast<unsigned>(std::min<uint64_t>(64, std::max<uint64_t>(1, 4096 / T)));
		hipLaunchKernelGGL(assemble_kernel, dim3(static_cast<unsigned>(T), slices), dim3(256), 0, st, d_copies, soft_dev);
		HTRY(hipGetLastError());
		HTRY(hipStreamSynchronize(st));
		mark("assembled");
		rep->n_symbols = out_pos;
		rep->tiles_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_tiles).count();
		return MDEMOD_OK;
	}

	int run_all()
	{
		TRY(prepare());
		HTRY(hipEventCreateWithFlags(&input_ready.ev, hipEventDisableTiming));
		HTRY(hipEventRecord(input_ready.ev, st));
		try {
			/* (a thread's exception is nobody's to catch: std::terminate.  The function-try-blocks of the entries cover the calling thread only) */
			est.t = std::thread([this]() { try { est.rc = estimate_grid(); } catch (...) { est.rc = MDEMOD_ERR_NOMEM; } });    /* joined by estimate_carriers, or by ~EstThread on an early return */
		} catch (...) {
			est.rc = estimate_grid();                                       /* no thread to be had: before the head, then */
		}
		TRY(run_head());
		TRY(plan_tiles());
		if (T == 0) return MDEMOD_OK;                        /* the head was the whole recording */
		TRY(estimate_carriers());
		TRY(estimate_clocks());
		TRY(seed_tiles());
		TRY(acquire_and_frame());
		TRY(settle_body_seams());
		TRY(hand_over_uncured());
		if (o.debug >= 3) dump_tiles();
		return assemble();
	}

	/* debug >= 3: one line per tile for tools/tile_tail.py (what was the tile given, where did it end up) */
	void dump_tiles() const
	{
		for (size_t i = 0; i < T; i++)
			fprintf(stderr, "[tile] %zu E %llu len %llu s0 %llu acq %llu frm %llu stl %llu seed_tf %.9g seed_f0 %.9g start_tf %.9g start_f0 %.9g end_tf %.9g end_f0 %.9g end_gain %.9g cnt %u out_rot %d\n", i,
			        (unsigned long long)E[i], (unsigned long long)len[i], (unsigned long long)s0[i], (unsigned long long)acq[i], (unsigned long long)frm[i],
			        (unsigned long long)stl[i], static_cast<double>(tf.size() > i ? tf[i] : 0.0f), static_cast<double>(f0.size() > i ? f0[i] : 0.0f),
			        static_cast<double>(status_settle.size() > i ? status_settle[i].omega : 0.0f), static_cast<double>(status_settle.size() > i ? status_settle[i].pll_freq : 0.0f),
			        static_cast<double>(status_body.size() > i ? status_body[i].omega : 0.0f), static_cast<double>(status_body.size() > i ? status_body[i].pll_freq : 0.0f),
			        static_cast<double>(status_body.size() > i ? status_body[i].gain : 0.0f), cnt1.size() > i ? cnt1[i] : 0u, out_rot.size() > i ? out_rot[i] : 0);
	}
};

static int
demodulate_recording_impl(const mdemod_params *params, const mdemod_recording_opts *opts_in,
                          const void *iq_dev, uint64_t n_samples,
                          int8_t *soft_dev, uint64_t soft_cap_symbols,
                          mdemod_recording_report *rep, void *hip_stream, const std::function<void(uint64_t)> *need)
{
	if (!params || !iq_dev || !soft_dev || !rep) return MDEMOD_ERR_PARAM;
	if (params->samplerate <= 0 || params->symrate <= 0) return MDEMOD_ERR_PARAM;
	Stitcher s;
	s.params = params; s.opts_in = opts_in; s.iq_dev = iq_dev; s.n_samples = n_samples; s.soft_dev = soft_dev;
	s.soft_cap_symbols = soft_cap_symbols; s.rep = rep; s.st = static_cast<hipStream_t>(hip_stream); s.need = need;
	return s.run_all();
}

extern "C" int
mdemod_demodulate_recording(const mdemod_params *params, const mdemod_recording_opts *opts_in,
                            const void *iq_dev, uint64_t n_samples,
                            int8_t *soft_dev, uint64_t soft_cap_symbols,
                            mdemod_recording_report *rep, void *hip_stream)
try { MDEMOD_API_ENTER
	return demodulate_recording_impl(params, opts_in, iq_dev, n_samples, soft_dev, soft_cap_symbols, rep, hip_stream, nullptr);
} MDEMOD_API_CATCH

/* Host-buffer convenience (PCIe inclusive): what the C CLI's --tiled mode calls.  The head goes in first; the rest of the
 * recording is copied by a second thread on its own stream while the serial head runs (one wave, ~0.1 s: about what 1 GB of
 * pageable memory takes). */
extern "C" int
mdemod_demodulate_recording_host(const mdemod_params *params, const mdemod_recording_opts *opts,
                                 const void *iq_host, uint64_t n_samples,
                                 int8_t *soft_host, uint64_t soft_cap_symbols,
                                 mdemod_recording_report *rep)
try { MDEMOD_API_ENTER
	if (!params || !iq_host || !soft_host || !rep) return MDEMOD_ERR_PARAM;
	const bool dbg = opts && opts->debug != 0;
	const auto t_in = std::chrono::steady_clock::now();
	auto mark = [&](const char *what) {
		if (dbg) fprintf(stderr, "[recording host] %8.2f ms  %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_in).count() * 1e3, what);
	};
	(void)hipGetLastError(); HTRY(hipSetDevice(params->device));      /* (what an earlier call left pending is not this call's: demod_api.cpp select_device) */
	mark("device set");
	const size_t sb = 2 * static_cast<size_t>(params->bps) / 8;
	DevMem mem;
	unsigned char *d_iq; int8_t *d_soft;
	TRY(mem.alloc(&d_iq, static_cast<size_t>(n_samples) * sb));
	TRY(mem.alloc(&d_soft, static_cast<size_t>(soft_cap_symbols) * 2));
	mark("buffers allocated");
	const uint64_t head = std::min<uint64_t>(n_samples, 1u << 20);
	if (head) HTRY(hipMemcpy(d_iq, iq_host, static_cast<size_t>(head) * sb, hipMemcpyHostToDevice));
	mark("head copied");
	std::atomic<uint64_t> there{head};
	std::atomic<int> copy_failed{0};
	struct Joiner { std::thread t; ~Joiner() { if (t.joinable()) t.join(); } } copier;
	if (head < n_samples) {
		const int device = params->device;
		copier.t = std::thread([&, device]() { try {
			hipStream_t cs = nullptr;
			if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { copy_failed = 1; there = n_samples; return; }
			const uint64_t chunk = 1u << 23;                  /* samples per copy: progress in steps of 8-64 MB */
			for (uint64_t at = head; at < n_samples; at += chunk) {
				const uint64_t c = std::min<uint64_t>(chunk, n_samples - at);
				if (hipMemcpyAsync(d_iq + at * sb, static_cast<const unsigned char *>(iq_host) + at * sb, static_cast<size_t>(c) * sb, hipMemcpyHostToDevice, cs) != hipSuccess ||
				    hipStreamSynchronize(cs) != hipSuccess) { copy_failed = 1; break; }
				there = at + c;
			}
			mark("recording copied in");
			there = n_samples;                                /* also after a failure: nobody may wait for ever */
			(void)hipStreamDestroy(cs);
		} catch (...) { copy_failed = 1; there = n_samples; } });          /* (an exception in a thread would be std::terminate) */
	}
	const std::function<void(uint64_t)> need = [&](uint64_t upto) {
		while (there.load() < std::min(upto, n_samples)) std::this_thread::sleep_for(std::chrono::microseconds(50));
	};
	/* a stream of this call's own: on the null stream the calls of several host threads (the C host's --jobs) would run one
	   kernel after the other, serial heads included */
	struct OwnStream { hipStream_t s = nullptr; ~OwnStream() { if (s) (void)hipStreamDestroy(s); } } own;
	HTRY(hipStreamCreateWithFlags(&own.s, hipStreamNonBlocking));
	const int rc = demodulate_recording_impl(params, opts, d_iq, n_samples, d_soft, soft_cap_symbols, rep, own.s, &need);
	if (copier.t.joinable()) copier.t.join();
	mark("demodulated");
	if (copy_failed.load()) return MDEMOD_ERR_HIP;
	TRY(rc);
	if (rep->n_symbols) {
		HTRY(hipMemcpyAsync(soft_host, d_soft, static_cast<size_t>(rep->n_symbols) * 2, hipMemcpyDeviceToHost, own.s));
		HTRY(hipStreamSynchronize(own.s));
	}
	mark("symbols copied out");
	return MDEMOD_OK;
} MDEMOD_API_CATCH
