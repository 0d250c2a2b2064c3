/*
 * demod_kernel.hip — the fused LRPT demodulator kernel for gfx950 (MI355X).
 *
 * One LANE = one stream (a recording, or a tile of one treated as a stream).
 * The path is a strictly serial recurrence per stream (each symbol's sampling
 * instant, gain and rotation depend on the previous symbol: demod.c:24-48), so
 * bit-exact parallelism exists only ACROSS streams.  Every lane runs the whole
 * chain  polyphase RRC (filter.c:46-65) -> AGC (agc.c:13-25) -> NCO mix
 * (pll.c:51-97, sincos.c) -> timing update (timing.c:60-87) -> Costas update
 * (pll.c:100-130) -> int8 quantise (main.c:305-306)  for its own stream.
 *
 * Data movement (the part that is designed for CDNA4):
 *   - Each wave keeps a ring of raw input granules in LDS, laid out
 *     [granule][lane][4 samples]: lane l only ever touches column l, so every
 *     ds_read_b128 / ds_write_b128 is bank-conflict free whatever position each
 *     lane's window is at.  A granule is 4 consecutive IQ samples (16 B for
 *     s16, 8 B for u8, 32 B for f32) kept in the INPUT format; conversion to
 *     float happens in registers at use.
 *   - The ring is refilled wave-synchronously in chunks: all 64 lanes fetch the
 *     same granule indices of their own streams (each lane one 16-B
 *     global_load_dwordx4; 8 consecutive refills consume a lane's 128-B line),
 *     one iteration ahead of use (register-staged, committed to LDS at the top
 *     of the next iteration) so HBM latency hides behind a full FIR.
 *   - The RRC taps live in LDS as [alignment 0..3][bank] rows, zero padded, so a
 *     lane whose window starts at sample offset a inside a granule reads row
 *     (a, bank) with aligned ds_read_b128 and multiplies the pad slots by 0
 *     (exact: acc + (+-0) == acc because acc can never be -0).  Row stride is
 *     19 x 16 B so that the rows a 16-lane group may touch fall on distinct LDS
 *     bank slots.
 *   - The FIR itself is the reference's sequential oldest->newest sum, one
 *     unfused multiply and one add per component per tap; no MFMA, no FMA: the
 *     rounding sequence IS the specification.
 *
 * Lanes run in near lock-step (same nominal samples/symbol); a lane that gets
 * ahead of the loaded data simply idles one iteration, a lane that lags gates
 * the next refill, so no cross-lane re-synchronisation is ever needed.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "demod_internal.h"
#include "demod_device.h"

#pragma clang fp contract(off)

namespace {

/* ---- input formats (wavfile.c:58-69) -------------------------------------- */

template <int FMT> struct Format;

template <> struct Format<16> {
	typedef uint32_t sample_t;                 /* int16 I | int16 Q << 16 */
	static constexpr int kSampleBytes = 4;
	static constexpr uint32_t kZero = 0u;
	__device__ static __forceinline__ cf32 decode(uint32_t w) {
		cf32 r;
		r.re = (float)(int)(int16_t)(w & 0xFFFFu);
		r.im = (float)((int)w >> 16);
		return r;
	}
};

template <> struct Format<8> {
	typedef uint16_t sample_t;                 /* u8 I | u8 Q << 8 */
	static constexpr int kSampleBytes = 2;
	static constexpr uint16_t kZero = 0x8080u;
	__device__ static __forceinline__ cf32 decode(uint16_t w) {
		cf32 r;
		r.re = (float)((int)(w & 0xFFu) - 128);
		r.im = (float)((int)(w >> 8) - 128);
		return r;
	}
};

template <> struct Format<32> {
	typedef float2 sample_t;
	static constexpr int kSampleBytes = 8;
	__device__ static __forceinline__ cf32 decode(float2 w) {
		cf32 r; r.re = w.x; r.im = w.y; return r;
	}
};

template <int FMT> struct alignas(Format<FMT>::kSampleBytes >= 4 ? 16 : 8) Granule { typename Format<FMT>::sample_t s[4]; };

/* Unaligned-tolerant global load of one granule (4 samples). */
template <int FMT>
__device__ __forceinline__ Granule<FMT>
load_granule(const typename Format<FMT>::sample_t *p)
{
	Granule<FMT> g;
	__builtin_memcpy(&g, p, sizeof(g));
	return g;
}

/* ---- the kernel ------------------------------------------------------------ */

/*
 * NGW  > 0: compile-time number of window granules (fully unrolled FIR)
 * NGW == 0: taken from DemodConsts at run time (generic configurations)
 */
/* Blocks are at most 256 threads and the 24 KB-per-wave LDS ring already limits this kernel to <= 2
 * waves per SIMD, so the register allocator may use up to 256 VGPRs.  Without the bound hipcc must
 * assume 1024-thread blocks (4 waves/SIMD), caps at 128 VGPRs and spills 300-800 bytes per lane to
 * scratch (= HBM): measured 2x slower. */
/* GTAB: the coefficient table stays in global memory (it does not fit the LDS next to the rings: -O 64 with 129 and more taps, ...):
 * slow, but such a configuration is demodulated instead of refused. */
template <int FMT, int OQPSK, int NGW, int CHUNK, int GTAB = 0>
__global__ void __launch_bounds__(256, 2)
demod_kernel(const DemodLaunch L)
{
	typedef Format<FMT> F;
	typedef typename F::sample_t sample_t;
	constexpr int GB = 4 * F::kSampleBytes;          /* bytes per granule */

	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

	const DemodConsts &C = L.c;
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	const int waves_per_block = blockDim.x >> 6;
	const uint32_t stream = blockIdx.x * blockDim.x + threadIdx.x;
	const bool valid = stream < L.n_streams;

	/* LDS carve-up: [ctab][tanh lut][ring wave 0][ring wave 1]... */
	float *ctab = reinterpret_cast<float *>(lds);
	float *lut = GTAB ? ctab : ctab + L.ctab_floats;
	const int ring_bytes = C.ring_granules * 64 * GB;
	unsigned char *ring = reinterpret_cast<unsigned char *>(lut + 32) + wave * ring_bytes;
	(void)waves_per_block;

	if (!GTAB) for (uint32_t i = threadIdx.x; i < L.ctab_floats; i += blockDim.x) ctab[i] = L.ctab[i];
	if (threadIdx.x < 32) lut[threadIdx.x] = L.tanh_lut[threadIdx.x];

	/* ---- per-stream geometry ---- */
	uint32_t n = 0;
	const sample_t *src = nullptr;
	if (valid) {
		n = L.n_samples_arr ? L.n_samples_arr[stream] : L.n_samples;
		const uint64_t off = L.iq_offset ? L.iq_offset[stream] : (uint64_t)stream * L.iq_stride;
		src = reinterpret_cast<const sample_t *>(L.iq) + off;
	}
	const int hpad = C.hpad;
	const int back = C.taps - 1;
	const int G = C.ring_granules;
	const int v_end = hpad + (int)n;                         /* virtual stream = history ++ block */

	/* ---- load state ---- */
	float gain = 1.0f, bias_re = 0.0f, bias_im = 0.0f;
	PllState pll = { 0.0f, 0.0f, 1000.0f, 0, 0, 1 };
	float t_phase = 0.0f, t_freq = C.t_center, t_prev = 0.0f, inphase = 0.0f;
	int dual_state = 1;
	uint64_t n_symbols = 0;
	int64_t first_lock = -1;
	if (valid) {
		gain = L.st.agc_gain[stream]; bias_re = L.st.agc_bias_re[stream]; bias_im = L.st.agc_bias_im[stream];
		pll.phase = L.st.pll_phase[stream]; pll.freq = L.st.pll_freq[stream]; pll.err = L.st.pll_err[stream];
		const int fl = L.st.flags[stream];
		pll.locked = (fl & MDEMOD_FLAG_LOCKED) ? 1 : 0;
		pll.locked_once = (fl & MDEMOD_FLAG_LOCKED_ONCE) ? 1 : 0;
		pll.updown = (fl & MDEMOD_FLAG_UPDOWN_POS) ? 1 : -1;
		dual_state = (fl >> MDEMOD_FLAG_DUAL_SHIFT) & 3;
		t_phase = L.st.t_phase[stream]; t_freq = L.st.t_freq[stream]; t_prev = L.st.t_prev[stream];
		inphase = L.st.inphase[stream];
		n_symbols = L.st.n_symbols[stream];
		first_lock = L.st.first_lock[stream];
	}

	/* ---- history -> ring granules [0, hpad/4) ---- */
	unsigned char *col = ring + lane * GB;                  /* this lane's column */
	/* Every ring slot the FIR can touch must hold a FINITE value: the aligned coefficient rows multiply slots outside
	 * the lane's window by zero, and for float input stale LDS bits can be NaN/Inf (0 * NaN = NaN; found by
	 * tools/config_fuzz.py).  Granules not loaded yet are therefore cleared once per launch. */
	for (int g = hpad >> 2; g < C.ring_granules; g++)
		__builtin_memset(col + g * 64 * GB, 0, GB);
	{
		const sample_t *hist = reinterpret_cast<const sample_t *>(L.st.hist);
		for (int k = 0; k < hpad; k++) {
			sample_t s;
			if (valid) s = hist[(size_t)k * L.n_streams + stream];
			else __builtin_memset(&s, 0, sizeof(s));
			*reinterpret_cast<sample_t *>(col + (k >> 2) * 64 * GB + (k & 3) * F::kSampleBytes) = s;
		}
	}
	__syncthreads();                                         /* ctab/lut visible; only barrier in the kernel */

	/* ---- main loop state ---- */
	int g_hi = hpad >> 2;                                    /* granules [.., g_hi) are in the ring (wave-uniform) */
	int s_hi = g_hi % G;                                     /* ring slot granule g_hi will occupy (wave-uniform)  */
	int v_cur = hpad - 1;                                    /* newest sample pushed so far                         */
	int isub = 0;                                            /* next interpolation sub-step                         */
	int fire_sub = 0;
	bool fired = false;                                      /* a firing is waiting for its data                    */
	bool done = !valid || n == 0;
	uint32_t sym_call = 0, ev_call = 0;
	int v_last_emit = -1;                                    /* sample of the last emitted symbol            */
	int overflow = 0;

	/* wave-uniform end of data: max over lanes of needed granules */
	int g_need = done ? 0 : ((v_end + 3) >> 2);
	for (int o = 32; o > 0; o >>= 1) {
		const int other = __shfl_xor(g_need, o);
		g_need = other > g_need ? other : g_need;
	}

	Granule<FMT> stage[CHUNK];
	bool staged = false;                                     /* wave-uniform */

	int8_t *soft_out = L.soft + (size_t)stream * L.soft_stride * 2;
	const float thr_q = MD_TWO_PI_F;

	/* Watchdog (as in rotwin_body.h): a wave needs at most a few iterations per interpolated step of its longest
	 * stream; g_need is the wave-uniform number of granules of that stream. */
	const uint64_t guard64 = 16ull * (uint64_t)(g_need + 2) * (uint64_t)C.interp + 4096ull;
	uint32_t guard = guard64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)guard64;

	while (true) {
		if (guard-- == 0) { overflow = 1; break; }
		/* (1) commit the chunk fetched during the previous iteration */
		if (staged) {
#pragma unroll
			for (int c = 0; c < CHUNK; c++) {
				*reinterpret_cast<Granule<FMT> *>(col + s_hi * 64 * GB) = stage[c];
				s_hi = (s_hi + 1 == G) ? 0 : s_hi + 1;
			}
			g_hi += CHUNK;
			staged = false;
		}

		/* (2) step the symbol clock to the next firing (timing.c:32-57).  Fast path as in the register-window
		 * kernel: k_safe blind adds that provably cannot fire (checked per lane), then four checked
		 * steps (the increment is positive, so "reached thr" is monotone); generic loop otherwise. */
		if (!fired && !done) {
			const float thr = OQPSK ? (float)dual_state * MD_PI_F : thr_q;
			const int k_safe = C.step_safe;
			const int steps_left = (v_end - 1 - v_cur) * C.interp + (isub ? C.interp - isub : 0);
			if ((t_phase < thr - (float)k_safe * C.step_fmax - 1e-3f) && (steps_left >= k_safe + 4)) {
				float p = t_phase;
				int k = 0;
				for (; k + 8 <= k_safe; k += 8) {
					p = p + t_freq; p = p + t_freq; p = p + t_freq; p = p + t_freq;
					p = p + t_freq; p = p + t_freq; p = p + t_freq; p = p + t_freq;
				}
				for (; k < k_safe; k++) p = p + t_freq;
				const float p1 = p + t_freq, p2 = p1 + t_freq, p3 = p2 + t_freq, p4 = p3 + t_freq;
				const bool c1 = p1 >= thr, c2 = p2 >= thr, c3 = p3 >= thr, c4 = p4 >= thr;
				const int m = k_safe + 1 + (c1 ? 0 : 1) + (c2 ? 0 : 1) + (c3 ? 0 : 1);
				float ph = c3 ? p3 : p4;
				ph = c2 ? p2 : ph;
				ph = c1 ? p1 : ph;
				t_phase = ph;
				const uint32_t w = (uint32_t)(isub + m);
				const uint32_t qd = (C.interp == 1) ? w : __umulhi(w, C.interp_magic);   /* floor(w / interp) */
				const int isub_new = (int)(w - qd * (uint32_t)C.interp);
				v_cur += (int)qd + (isub_new > 0 ? 1 : 0) - (isub > 0 ? 1 : 0);
				fire_sub = (isub_new == 0) ? C.interp - 1 : isub_new - 1;
				isub = isub_new;
				fired = c4;
			}
			while (!fired && !done) {
				if (isub == 0) {
					if (v_cur + 1 >= v_end) { done = true; break; }
					v_cur++;                                  /* filter_fwd_sample, filter.c:39-43 */
				}
				t_phase = t_phase + t_freq;
				fire_sub = isub;
				isub = (isub + 1 == C.interp) ? 0 : isub + 1;
				if (t_phase >= thr) fired = true;
			}
		}
		if (md_all(done)) break;

		/* (3) refill decision (wave-uniform) and fetch, one iteration ahead */
		if (g_hi < g_need) {
			const int v_low = done ? 0x3FFFFFFF : (v_cur - hpad);
			const bool room = (g_hi + CHUNK - G) <= (v_low >> 2);
			if (md_all(room)) {
#pragma unroll
				for (int c = 0; c < CHUNK; c++) {
					const int m0 = ((g_hi + c) << 2) - hpad;   /* first block sample of the granule */
					if (m0 + 3 < (int)n) {
						stage[c] = load_granule<FMT>(src + m0);
					} else {
#pragma unroll
						for (int u = 0; u < 4; u++) {
							if (m0 + u < (int)n) stage[c].s[u] = src[m0 + u];
							else __builtin_memset(&stage[c].s[u], 0, sizeof(sample_t));
						}
					}
				}
				staged = true;
			}
		}

		/* (4) process the firing if its window is resident */
		const bool go = fired && (v_cur < (g_hi << 2));
		if (go) {
			fired = false;
			const int w0 = v_cur - back;                     /* oldest sample of the window */
			const int a = w0 & 3;
			/* ring slot of the window's first granule, relative to the wave-uniform
			 * head: granule g_hi - d sits d slots behind s_hi (1 <= d <= G). */
			int gq = s_hi - (g_hi - (w0 >> 2));
			gq = (gq < 0) ? gq + G : gq;
			const int bank = C.interp - 1 - fire_sub;       /* filter.c:52 */
			const float *row;
			if constexpr (GTAB) row = L.ctab + (a * C.interp + bank) * C.ctab_row_stride;
			else row = ctab + (a * C.interp + bank) * C.ctab_row_stride;

			/* filter.c:55-62: sequential, oldest first, unfused */
			float acc_re = 0.0f, acc_im = 0.0f;
			const int ngw = NGW ? NGW : C.win_granules;
#pragma unroll
			for (int q = 0; q < ngw; q++) {
				/* every 4th granule: make the next addresses depend on the running sums, or hipcc hoists
				 * all 2*NGW LDS reads of the unrolled loop to the top and spills (264 VGPRs at 129 taps) */
				if ((q & 3) == 3) asm volatile("" : "+v"(gq) : "v"(acc_re), "v"(acc_im));
				const Granule<FMT> g = *reinterpret_cast<const Granule<FMT> *>(col + gq * 64 * GB);
				const float4 h = *reinterpret_cast<const float4 *>(row + 4 * q);
				gq = (gq + 1 == G) ? 0 : gq + 1;
				const cf32 s0 = F::decode(g.s[0]), s1 = F::decode(g.s[1]);
				const cf32 s2 = F::decode(g.s[2]), s3 = F::decode(g.s[3]);
				acc_re = acc_re + s0.re * h.x;  acc_im = acc_im + s0.im * h.x;
				acc_re = acc_re + s1.re * h.y;  acc_im = acc_im + s1.im * h.y;
				acc_re = acc_re + s2.re * h.z;  acc_im = acc_im + s2.im * h.z;
				acc_re = acc_re + s3.re * h.w;  acc_im = acc_im + s3.im * h.w;
			}

			cf32 y = { acc_re, acc_im };
			y = md_agc(y, gain, bias_re, bias_im);

			/* pll.c:51-97 */
			const float sn = md_fast_sin(-pll.phase);
			const float cs = md_fast_cos(-pll.phase);
			bool emit = true;
			float out_re, out_im;
			if (OQPSK) {
				if (dual_state == 1) {                        /* demod.c:66-71 */
					inphase = y.re * cs - y.im * sn;
					emit = false;
				}
				out_re = inphase;
				out_im = y.re * sn + y.im * cs;               /* demod.c:76 */
				dual_state = (dual_state % 2) + 1;            /* timing.c:52 */
			} else {
				out_re = y.re * cs - y.im * sn;
				out_im = y.re * sn + y.im * cs;
			}
			md_nco_advance(pll.phase, pll.freq);

			if (emit) {
				/* only the LAST symbol fired inside one input sample is kept (demod.c:33-47, 62-90; see rotwin_body.h) */
				if (v_cur == v_last_emit) { sym_call--; n_symbols--; }
				v_last_emit = v_cur;
				md_timing_update(t_phase, t_freq, t_prev, C.t_alpha, C.t_beta, C.t_center, C.t_maxdev, out_im);
				int first = 0;
				const int changed = md_pll_update(pll, lut, C.pll_alpha, C.pll_beta, C.pll_fmax,
				                                  out_re, out_im, first);
				if (first) first_lock = (int64_t)n_symbols;
				if (changed) {
					if (ev_call < MDEMOD_MAX_LOCK_EVENTS) {
						mdemod_lock_event ev;
						ev.symbol = n_symbols; ev.locked = pll.locked; ev.pad = 0;
						L.st.events[(size_t)stream * MDEMOD_MAX_LOCK_EVENTS + ev_call] = ev;
					}
					ev_call++;
				}
				if (sym_call < L.soft_cap) {
					const int qi = md_quantise(out_re), qq = md_quantise(out_im);
					*reinterpret_cast<uint16_t *>(soft_out + 2 * (size_t)sym_call) =
					    (uint16_t)((qi & 0xFF) | ((qq & 0xFF) << 8));
				} else {
					overflow = 1;
				}
				sym_call++;
				n_symbols++;
			}
		}
	}

	/* ---- store state ---- */
	if (valid) {
		L.st.agc_gain[stream] = gain; L.st.agc_bias_re[stream] = bias_re; L.st.agc_bias_im[stream] = bias_im;
		L.st.pll_phase[stream] = pll.phase; L.st.pll_freq[stream] = pll.freq; L.st.pll_err[stream] = pll.err;
		L.st.flags[stream] = (pll.locked ? MDEMOD_FLAG_LOCKED : 0) | (pll.locked_once ? MDEMOD_FLAG_LOCKED_ONCE : 0) |
		                     (pll.updown > 0 ? MDEMOD_FLAG_UPDOWN_POS : 0) | (dual_state << MDEMOD_FLAG_DUAL_SHIFT);
		L.st.t_phase[stream] = t_phase; L.st.t_freq[stream] = t_freq; L.st.t_prev[stream] = t_prev;
		L.st.inphase[stream] = inphase;
		L.st.n_samples[stream] += n;
		L.st.n_symbols[stream] = n_symbols;
		L.st.first_lock[stream] = first_lock;
		L.st.sym_this_call[stream] = sym_call;
		L.st.ev_this_call[stream] = ev_call;
		L.st.overflow[stream] = overflow;

		/* new history = last hpad samples of (old history ++ block); ascending k
		 * makes the in-place shift safe (reads index n+k >= k). */
		sample_t *hist = reinterpret_cast<sample_t *>(L.st.hist);
		for (int k = 0; k < hpad; k++) {
			const int64_t idx = (int64_t)n + k;
			const sample_t s = (idx < hpad) ? hist[(size_t)idx * L.n_streams + stream]
			                                : src[idx - hpad];
			hist[(size_t)k * L.n_streams + stream] = s;
		}
	}
}

} /* namespace */

/* ---- host-callable launcher -------------------------------------------------- */

template <int FMT, int OQPSK, int NGW, int CHUNK, int GTAB = 0>
static hipError_t
launch_one(const DemodLaunch &L, int block, size_t lds_bytes, hipStream_t stream)
{
	const uint32_t blocks = (L.n_streams + block - 1) / block;
	auto kfn = demod_kernel<FMT, OQPSK, NGW, CHUNK, GTAB>;
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(kfn, dim3(blocks), dim3(block), lds_bytes, stream, L);
	return hipGetLastError();
}

template <int FMT, int OQPSK>
static hipError_t
launch_ngw(const DemodLaunch &L, int block, int gtab, size_t lds_bytes, hipStream_t stream)
{
	if (gtab) return launch_one<FMT, OQPSK, 0, 2, 1>(L, block, lds_bytes, stream);
	switch (L.c.win_granules) {
	case 17: return launch_one<FMT, OQPSK, 17, 2>(L, block, lds_bytes, stream);   /* -f 32: 65 taps */
	case 33: return launch_one<FMT, OQPSK, 33, 2>(L, block, lds_bytes, stream);   /* -f 64: 129 taps */
	default: return launch_one<FMT, OQPSK, 0, 2>(L, block, lds_bytes, stream);
	}
}

template <int FMT>
static hipError_t
launch_mode(const DemodLaunch &L, int block, int gtab, size_t lds_bytes, hipStream_t stream)
{
	return L.c.oqpsk ? launch_ngw<FMT, 1>(L, block, gtab, lds_bytes, stream)
	                 : launch_ngw<FMT, 0>(L, block, gtab, lds_bytes, stream);
}

/* Called by demod_api.cpp. */
hipError_t
mdemod_launch_demod(const DemodLaunch &L, int fmt, int block, int global_table, size_t lds_bytes, hipStream_t stream)
{
	if (L.c.chunk_granules != 2) return hipErrorInvalidValue;
	switch (fmt) {
	case 16: return launch_mode<16>(L, block, global_table, lds_bytes, stream);
	case 8:  return launch_mode<8>(L, block, global_table, lds_bytes, stream);
	case 32: return launch_mode<32>(L, block, global_table, lds_bytes, stream);
	default: return hipErrorInvalidValue;
	}
}
