/*
 * demod_kernel_lat.hip — latency kernel: ONE stream per wavefront.
 *
 * The throughput kernels run one stream per lane: 64 streams per wave, every lane a serial recurrence at ~0.5 M symbols/s.
 * That is the right shape for 10^5 streams and the wrong one for the serial head of a recording (the "pilot" of
 * csrc/recording.hip: one stream, the reference's own lock acquisition, demod.c:24-91) or for a few thousand tiles: 63 of 64
 * lanes idle and the one busy lane pays the full instruction count of the FIR (filter.c:46-65: 65 taps x 4 unfused ops).
 *
 * Here the 64 lanes of a wave work on the SAME stream:
 *
 *   farm    The FIR of a firing depends only on WHERE the symbol clock fires (input sample, polyphase bank), not on any
 *           loop state.  Where the next ~21 firings fall is predictable to +-1 interpolated step (the clock word moves by
 *           2^-12 at most, timing.c:84; the per-symbol correction alpha*e is ~1e-3 of a step).  Lane l computes the FIR of
 *           firing l/3 at predicted step + (l%3 - 1): 63 complete FIRs for the price of one, each one the reference's
 *           sequential oldest-first unfused sum.  Samples (converted floats) and the plain polyphase table live in LDS.
 *   serial  The scalar recurrence (symbol clock, AGC, NCO, timing and Costas updates, lock detector: demod_device.h, the
 *           same functions as the other kernels) then runs firing by firing with wave-uniform values; the exact clock tells
 *           which candidate was the right one and its FIR value is fetched with v_readlane.  A firing outside the three
 *           candidates (transients) computes its FIR on the spot and ends the batch.
 *
 * Results are bit-identical to the other kernels and to the reference (tests run every golden through this kernel).
 * State and history live in the context's own layout, so a context may use this kernel for one call and a throughput
 * kernel for the next.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "demod_internal.h"
#include "demod_device.h"

#pragma clang fp contract(off)

namespace {

constexpr int kCand = 3;                 /* candidates per firing: predicted step -1, 0, +1 */
constexpr int kFire = 21;                /* firings per batch: 63 lanes                      */
constexpr int kMaxChunks = 20;           /* 64-sample chunks prefetched per batch            */
constexpr int kMirror = 264;             /* ring entries mirrored behind its end: >= the longest window (hpad <= 256, taps <= hpad + 1) */
constexpr int kMaxHpad = 256;            /* mdemod_lat_geometry refuses longer histories: the mirror above and the four rounds of 64 lanes that save the history */
static_assert(kMirror >= kMaxHpad + 1 && kMirror % 8 == 0, "a FIR window must fit the mirrored run");

template <int FMT> struct LFmt;
template <> struct LFmt<16> {
	typedef uint32_t sample_t;
	__device__ static __forceinline__ float2 decode(uint32_t w) { return make_float2((float)(int)(int16_t)(w & 0xFFFFu), (float)((int)w >> 16)); }
};
template <> struct LFmt<8> {
	typedef uint16_t sample_t;
	__device__ static __forceinline__ float2 decode(uint16_t w) { return make_float2((float)((int)(w & 0xFFu) - 128), (float)((int)(w >> 8) - 128)); }
};
template <> struct LFmt<32> {
	typedef float2 sample_t;
	__device__ static __forceinline__ float2 decode(float2 w) { return w; }
};

__device__ __forceinline__ float uni(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

/* KSAFE: the number of blind symbol-clock steps when it is one of the two everyday values (14: QPSK 72k at 230 kS/s -O 5; 6: OQPSK
 * 80k), so that the adds are straight-line code; 0 = read it from the launch constants. */
template <int FMT, int OQPSK, int KSAFE>
__global__ void __launch_bounds__(64)
demod_kernel_lat(const DemodLaunch L, const float *rrc, int ring_size, int span, int float_history)
{
	typedef LFmt<FMT> F;
	typedef typename F::sample_t sample_t;
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	const DemodConsts &C = L.c;
	const int lane = threadIdx.x;
	const uint32_t stream = blockIdx.x;
	const int interp = C.interp, taps = C.taps, hpad = C.hpad;
	const int mask = ring_size - 1;

	float2 *ring = reinterpret_cast<float2 *>(lds);                       /* ring[v & mask]: sample v of (history ++ block) */
	/* ... and ring[ring_size + i] = ring[i] for i < kMirror (>= taps): a FIR window that starts anywhere in the ring is ONE linear run
	   of LDS words, no index arithmetic per tap (round 5: the farm's 65 taps cost 9 instructions each, 3.5 of them the `& mask`) */
	float *coef = reinterpret_cast<float *>(ring + ring_size + kMirror);  /* [bank][taps], filter.c:18-22 */
	auto ring_put = [&](int v, float2 s) {
		const int i = v & mask;
		ring[i] = s;
		if (i < kMirror) ring[i + ring_size] = s;
	};
	float *lut = coef + interp * taps;
	float2 *obuf = reinterpret_cast<float2 *>(lut + 32);                  /* demodulated symbols of this batch, quantised when flushed */

	for (int i = lane; i < interp * taps; i += 64) coef[i] = rrc[i];
	if (lane < 32) lut[lane] = L.tanh_lut[lane];

	const int n = (int)(L.n_samples_arr ? L.n_samples_arr[stream] : L.n_samples);
	const uint64_t off = L.iq_offset ? L.iq_offset[stream] : (uint64_t)stream * L.iq_stride;
	const sample_t *src = reinterpret_cast<const sample_t *>(L.iq) + off;
	const int v_end = hpad + n;

	/* ---- state (wave-uniform) ---- */
	float gain = L.st.agc_gain[stream], bias_re = L.st.agc_bias_re[stream], bias_im = L.st.agc_bias_im[stream];
	PllWord pll;
	pll.phase = L.st.pll_phase[stream]; pll.freq = L.st.pll_freq[stream]; pll.err = L.st.pll_err[stream];
	const int fl_in = L.st.flags[stream];
	uint32_t fl = (uint32_t)fl_in & 7u;                 /* bit 0 locked, 1 locked_once, 2 updown > 0: kept packed (md_pll_update_packed) */
	int dual_state = (fl_in >> MDEMOD_FLAG_DUAL_SHIFT) & 3;
	float t_phase = L.st.t_phase[stream], t_freq = L.st.t_freq[stream], t_prev = L.st.t_prev[stream];
	float inphase = L.st.inphase[stream];
	const uint64_t nsym0 = L.st.n_symbols[stream];
	int overflow = 0, ev_call = 0, first_lock_call = -1, last_v = -1;
	uint32_t sym_call = 0;                   /* symbols emitted in this call, including the ones still in obuf */

	/* ---- history -> ring ---- */
	for (int k = lane; k < hpad; k += 64) {
		float2 h;
		if (float_history) h = reinterpret_cast<const float2 *>(L.st.hist)[(size_t)stream * hpad + k];
		else h = F::decode(reinterpret_cast<const sample_t *>(L.st.hist)[(size_t)k * L.n_streams + stream]);
		ring_put(k, h);
	}
	int r_hi = hpad;                         /* samples [.., r_hi) are in the ring (wave-uniform) */
	auto load_chunk = [&](int v0) -> float2 {      /* sample v0 + lane of the virtual stream (zeros past the end) */
		const int m = v0 + lane - hpad;
		float2 s = make_float2(0.0f, 0.0f);
		if (m < n) s = F::decode(src[m]);
		return s;
	};
	/* two spans ahead before the first batch */
	while (r_hi < v_end && r_hi < hpad + 2 * span) {
		const float2 s = load_chunk(r_hi);
		if (r_hi + lane < v_end) ring_put(r_hi + lane, s);
		r_hi = min(v_end, r_hi + 64);
	}
	__syncthreads();

	int v_cur = hpad - 1, isub = 0;
	bool done = n == 0;
	uint32_t out_base = 0; int out_cnt = 0;  /* obuf[0..out_cnt) = symbols out_base.. of this call */
	const float U = OQPSK ? MD_PI_F : MD_TWO_PI_F;
	const int k_safe = KSAFE ? KSAFE : C.step_safe;
	const float f_hi = C.step_fmax;
	const uint32_t magic = C.interp_magic;
	uint64_t guard = 4ull * (uint64_t)(n + hpad) * (uint64_t)interp + 4096ull;     /* 64 bits: the careful path spends one per interpolated step, and 2^30 samples x 64 steps do not fit 32 */
	const int n_chunks = min(kMaxChunks, (span + 63) / 64 + 1);
	int since_emit = 1 << 28;                /* interpolated steps since the last emitted symbol (same-sample rule below) */

	/* position after `w - isub_at_origin` steps from (v_org, isub_org): samples pushed = ceil(w / interp) - (isub_org > 0) */
	auto locate = [&](int v_org, int isub_org, int steps, int &v, int &isub_new) {
		const uint32_t w = (uint32_t)(isub_org + steps);
		const uint32_t q = (interp == 1) ? w : __umulhi(w, magic);
		isub_new = (int)(w - q * (uint32_t)interp);
		v = v_org + (int)q + (isub_new > 0 ? 1 : 0) - (isub_org > 0 ? 1 : 0);
	};
	/* FIR of a firing on sample v, bank b, by every lane at once (filter.c:55-62: sequential, oldest first, unfused) */
	auto fir_here = [&](int v, int bank) -> cf32 {
		while (r_hi <= v) {                                                  /* beyond what has been loaded: extend the ring first */
			const float2 s = load_chunk(r_hi);
			if (r_hi + lane < v_end) ring_put(r_hi + lane, s);
			r_hi = min(v_end, r_hi + 64);
			__syncthreads();
		}
		int p = (v - taps + 1) & mask;
		const float *h = coef + bank * taps;
		float ar = 0.0f, ai = 0.0f;
		for (int k = 0; k < taps; k++) {
			const float2 x = ring[p];
			const float hk = h[k];
			ar = ar + x.x * hk;
			ai = ai + x.y * hk;
			p = (p + 1) & mask;
		}
		cf32 y; y.re = ar; y.im = ai;
		return y;
	};
	/* everything after the FIR: agc.c:13-25, pll.c:51-97, demod.c:33-47 / 62-90, timing.c:60-87, pll.c:100-130, main.c:305-306.
	 * `same_sample`: this symbol fired on the input sample the previous one fired on. */
	auto scalar_stage = [&](cf32 y, bool same_sample) {
		y = md_agc(y, gain, bias_re, bias_im);
		/* fast_sin(-phase) and fast_cos(-phase) = fast_sin((float)(-phase + pi/2)) (sincos.c:37-40): the same instructions on two
		   arguments, lane 1 takes the cosine's */
		const float a0 = -pll.phase;
		const float a1 = (float)((double)a0 + MD_HALF_PI_D);
		const float sc = md_fast_sin<false>(lane == 1 ? a1 : a0);      /* |phase| < 2pi + fmax, fmax < 6: mdemod_create only picks this kernel then */
		const float sn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc), 0));
		const float cs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc), 1));
		bool emit = true;
		float out_re, out_im;
		if (OQPSK) {
			if (dual_state == 1) { inphase = y.re * cs - y.im * sn; emit = false; }   /* demod.c:66-71 */
			out_re = inphase;
			out_im = y.re * sn + y.im * cs;                                          /* demod.c:76    */
			dual_state = (dual_state % 2) + 1;                                       /* timing.c:52   */
		} else {
			out_re = y.re * cs - y.im * sn;
			out_im = y.re * sn + y.im * cs;
		}
		md_nco_advance<true>(pll.phase, pll.freq);
		if (emit) {
			/* only the LAST symbol fired inside one input sample survives (demod.c:33-47: `*sample` and `ret` are overwritten) */
			if (__builtin_expect(same_sample, 0)) {
				sym_call--;
				if (out_cnt > 0) out_cnt--; else out_base--;
			}
			md_timing_update(t_phase, t_freq, t_prev, C.t_alpha, C.t_beta, C.t_center, C.t_maxdev, out_im);
			uint32_t first = 0;
			/* the lock detector's flag word is wave-uniform, but kept in a VECTOR register: its bit arithmetic then stays on the vector
			   unit next to the comparisons that feed it, instead of crossing to the scalar unit and back a dozen times per symbol
			   (a lone wave waits out every crossing; round 5: configs[1] 4.70 -> 4.91 MS/s, the other two unchanged) */
			asm volatile("" : "+v"(fl));
			const uint32_t changed = md_pll_update_packed<true>(pll, fl, lut, C.pll_alpha, C.pll_beta, C.pll_fmax, out_re, out_im, first);
			if (__builtin_expect(first != 0, 0)) first_lock_call = (int)sym_call;
			if (__builtin_expect(changed != 0, 0)) {
				if (ev_call < MDEMOD_MAX_LOCK_EVENTS && lane == 0) {
					mdemod_lock_event ev;
					ev.symbol = nsym0 + sym_call; ev.locked = (int)(fl & 1u); ev.pad = 0;
					L.st.events[(size_t)stream * MDEMOD_MAX_LOCK_EVENTS + ev_call] = ev;
				}
				ev_call++;
			}
			obuf[out_cnt] = make_float2(out_re, out_im);                 /* every lane, same value: no exec-mask detour */
			out_cnt++;
			sym_call++;
			since_emit = 0;
		}
	};
	auto flush = [&]() {
		__syncthreads();
		if (out_cnt > 0) {
			if (lane < out_cnt) {
				const uint32_t pos = out_base + (uint32_t)lane;
				const float2 o = obuf[lane];                              /* main.c:305-306, one symbol per lane */
				const uint32_t sym = (uint32_t)(md_quantise(o.x) & 0xFF) | ((uint32_t)(md_quantise(o.y) & 0xFF) << 8);
				if (pos < L.soft_cap) reinterpret_cast<uint16_t *>(L.soft + (size_t)stream * L.soft_stride * 2)[pos] = (uint16_t)sym;
			}
			if (out_base + (uint32_t)out_cnt > L.soft_cap) overflow = 1;
			out_base += (uint32_t)out_cnt;
			out_cnt = 0;
		}
		__syncthreads();
	};
	/* One firing the careful way (timing.c:32-57 step by step, FIR on the spot): the last samples of a block, and any firing the
	 * prediction did not cover.  Returns false when the block ends before the clock fires. */
	auto careful_firing = [&]() -> bool {
		const float thr = OQPSK ? (float)dual_state * MD_PI_F : MD_TWO_PI_F;
		int fire_sub = 0, v_before = v_cur;
		bool fired = false;
		while (!fired && guard) {
			guard--;
			if (isub == 0) {
				if (v_cur + 1 >= v_end) { done = true; return false; }
				v_cur++;
			}
			t_phase = t_phase + t_freq;
			fire_sub = isub;
			isub = (isub + 1 == interp) ? 0 : isub + 1;
			since_emit++;
			if (t_phase >= thr) fired = true;
		}
		if (!fired) { done = true; return false; }
		(void)v_before;
		const cf32 y = fir_here(v_cur, interp - 1 - fire_sub);
		const bool same = (v_cur == last_v);
		scalar_stage(y, same && (!OQPSK || dual_state == 2));
		if (since_emit == 0) last_v = v_cur;
		return true;
	};

#ifdef LAT_EXP_TIMING
	/* experiment builds only (tools/build_exp_lat.sh x -DLAT_EXP_TIMING): where a batch's time goes, by s_memtime, printed by stream 0 */
	unsigned long long tm_pf = 0, tm_farm = 0, tm_serial = 0, tm_flush = 0, tm_commit = 0, tm_batches = 0, tm_fired = 0;
	unsigned long long tm_c[3] = { 0, 0, 0 }, tm_miss = 0;
#define LAT_TM(acc) do { const unsigned long long now_ = __builtin_readcyclecounter(); acc += now_ - tm_last; tm_last = now_; } while (0)
#else
#define LAT_TM(acc) do { } while (0)
#endif
	while (!done && guard) {
		/* ---- the tail of the block (and blocks shorter than a batch): firing by firing ---- */
		if (v_end - 1 - v_cur <= span + 8) {
			for (int j = 0; j < kFire && !done && guard; j++) careful_firing();
			flush();
			continue;
		}
#ifdef LAT_EXP_TIMING
		unsigned long long tm_last = __builtin_readcyclecounter();
		tm_batches++;
#endif
		/* ---- (0) prefetch: the chunks that extend the ring by one span, committed after the farm ---- */
		/* the RAW samples: converting them here would wait for the loads (a batch's prefetch stage stood for 1 100 cycles at configs[1],
		   2 600 at configs[3] - one trip to HBM - before the farm could start; r05, -DLAT_EXP_TIMING); they are converted when they are
		   committed, a whole serial stage later */
		sample_t pend[kMaxChunks];
		const int r_hi0 = r_hi;
		const int v0 = v_cur, isub0 = isub;
		/* chunks c with c < n_chunks, r_hi0 + 64 c < v_end and r_hi0 + 64 c < v0 + 1 + 2 span: the first n_valid of them.  pend[] has to
		   stay in registers (static indices: the chain is written out), but the everyday rates need three or four chunks, not
		   twenty tests of three conditions each: groups of four, the later groups behind one test (round 5) */
		const int pf_lim = min(v_end, v0 + 1 + 2 * span) - r_hi0;
		const int n_valid = pf_lim <= 0 ? 0 : min(n_chunks, (pf_lim + 63) >> 6);
#define LAT_PF(c) if ((c) < n_valid) { const int m_ = r_hi0 + 64 * (c) + lane - hpad; if (m_ < n) pend[c] = src[m_]; }
#define LAT_PF4(c) LAT_PF(c) LAT_PF((c) + 1) LAT_PF((c) + 2) LAT_PF((c) + 3)
		LAT_PF4(0)
		if (n_valid > 4) { LAT_PF4(4) if (n_valid > 8) { LAT_PF4(8) if (n_valid > 12) { LAT_PF4(12) if (n_valid > 16) { LAT_PF4(16) } } } }
#undef LAT_PF4
#undef LAT_PF

		LAT_TM(tm_pf);
		/* ---- (1) farm: lane -> (firing j, candidate c); lane j also keeps the prediction of firing j for the serial part ---- */
		t_phase = uni(t_phase); t_freq = uni(t_freq);                    /* wave-uniform by construction: pin them to scalars */
		const float phase0 = t_phase, inv_f0 = 1.0f / t_freq;
		const int k0 = OQPSK ? dual_state : 1;                         /* index of the first threshold ahead, in units of U */
		auto predict = [&](int j) -> int {                             /* steps from the batch start to firing j (a prediction: +-1) */
			const int s = (int)ceilf(((float)(k0 + j) * U - phase0) * inv_f0);
			return s < 1 ? 1 : s;
		};
		const int pred_mine = predict(lane);
		float yr = 0.0f, yi = 0.0f;
		bool cand_ok = false;
		{
			const int j = lane / kCand, c = lane % kCand - 1;
			const int steps = predict(j) + c;
			if (lane < kCand * kFire && steps >= 1) {
				int v, isub_new;
				locate(v0, isub0, steps, v, isub_new);
				const int fire_sub = isub_new == 0 ? interp - 1 : isub_new - 1;
				const int bank = interp - 1 - fire_sub;                    /* filter.c:52 */
				if (v < r_hi) {
					cand_ok = true;
					/* filter.c:55-62, sequential, oldest first, unfused: both rails of a tap in one packed multiply and one packed add
					   (the same four roundings), the window one linear run of the mirrored ring */
					typedef float v2f __attribute__((ext_vector_type(2)));
					const v2f *w = reinterpret_cast<const v2f *>(ring + ((v - taps + 1) & mask));
					const float *h = coef + bank * taps;
					v2f acc = { 0.0f, 0.0f };
					/* eight taps' LDS reads in flight before the first is used (the sums stay in tap order): one tap per trip paid
					   the LDS latency 65 times per batch, a fifth of a lone wave's time (round 5) */
					int k = 0;
					for (; k + 8 <= taps; k += 8) {
						v2f x[8]; float c[8];
#pragma unroll
						for (int u = 0; u < 8; u++) { x[u] = w[k + u]; c[u] = h[k + u]; }
#pragma unroll
						for (int u = 0; u < 8; u++) { const v2f hh = { c[u], c[u] }; acc = acc + x[u] * hh; }
					}
					for (; k < taps; k++) {
						const float hk = h[k];
						const v2f hh = { hk, hk };
						acc = acc + w[k] * hh;
					}
					yr = acc.x; yi = acc.y;
				}
			}
		}
		const uint64_t ok_mask = __ballot(cand_ok);
		/* steps that may be taken blindly before the block's end needs looking at (timing.c:32-38 stops pushing samples there) */
		/* (the closed-form runs may take a few steps more; either rail of an OQPSK symbol may have a schedule without the other) */
		int k_most = k_safe;
		if (!KSAFE && C.jump[0].nb > 0) k_most = max(k_most, C.jump[0].max_steps);
		if (!KSAFE && C.jump[1].nb > 0) k_most = max(k_most, C.jump[1].max_steps);
		const long long steps_room = (long long)(v_end - 1 - v0) * interp - isub0 - (k_most + 4) - interp;     /* 2^30 samples x 64 steps: not an int */
		const int steps_limit = steps_room > 0x3FFFFFFF ? 0x3FFFFFFF : (int)steps_room;

		LAT_TM(tm_farm);
		/* ---- (2) serial: firing by firing, wave-uniform; sample positions are only worked out when they matter ---- */
		int steps_done = 0;                                              /* interpolated steps since the batch start */
		int emit_steps = -1;                                             /* steps_done at the last symbol emitted in this batch */
		bool miss = false;
		auto sample_of_last_emit = [&]() -> int {
			if (emit_steps < 0) return last_v;                           /* emitted in an earlier batch */
			int v, is;
			locate(v0, isub0, emit_steps, v, is);
			return v;
		};
		int j = 0;
		for (; j < kFire; j++) {
			const float thr = OQPSK ? (float)dual_state * MD_PI_F : MD_TWO_PI_F;
			/* long runs (sample rates from about 1.8 MS/s): the schedule of closed-form jumps the v3 kernels use (clock_jump.h) */
			const cj_sched &J = C.jump[OQPSK ? dual_state - 1 : 0];
			const bool jump = !KSAFE && J.nb > 0;
			const bool fast = (jump ? (t_phase > J.floor && t_phase < J.hi) : (t_phase < thr - (float)k_safe * f_hi - 1e-3f)) && (steps_done < steps_limit);
			bool regular = false;
			int m = 0;
			float ph = t_phase;
			if (__builtin_expect(fast, 1)) {
				float p = t_phase;
				int k_done = k_safe;
				if (KSAFE) {
#pragma unroll
					for (int k = 0; k < KSAFE; k++) p = p + t_freq;
				} else if (jump) {
					k_done = clock_jump_run(p, t_freq, thr, f_hi, C.step_inv, J);
				} else {
					for (int k = 0; k < k_safe; k++) p = p + t_freq;
				}
				const float p1 = p + t_freq, p2 = p1 + t_freq, p3 = p2 + t_freq, p4 = p3 + t_freq;
				const bool c1 = p1 >= thr, c2 = p2 >= thr, c3 = p3 >= thr, c4 = p4 >= thr;
				m = k_done + 1 + (c1 ? 0 : 1) + (c2 ? 0 : 1) + (c3 ? 0 : 1);
				ph = c3 ? p3 : p4;
				ph = c2 ? p2 : ph;
				ph = c1 ? p1 : ph;
				regular = c4;
			}
			const int pj = __builtin_amdgcn_readlane(pred_mine, j);
			const int cidx = steps_done + m - pj;                         /* -1, 0, +1 when the prediction holds */
			const int idx = kCand * j + cidx + 1;
			const bool hit = regular && cidx >= -1 && cidx <= 1 && ((ok_mask >> idx) & 1ull);
#ifdef LAT_EXP_TIMING
			if (hit) tm_c[cidx + 1]++; else tm_miss++;
#endif
			if (__builtin_expect(!hit, 0)) { miss = true; break; }                             /* irregular firing: handled after the loop, the batch ends */
			t_phase = ph;
			steps_done += m;
			since_emit += m;
			cf32 y;
			y.re = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yr), idx));
			y.im = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yi), idx));
			/* two firings on one input sample have fewer than interp steps between them: only then look at the positions */
			const bool emits = !OQPSK || dual_state == 2;
			bool same = false;
			if (__builtin_expect(emits && since_emit < interp, 0)) {
				int v, is;
				locate(v0, isub0, steps_done, v, is);
				same = (v == sample_of_last_emit());
			}
			scalar_stage(y, same);
			if (emits) emit_steps = steps_done;
		}
		guard = guard > (uint64_t)j ? guard - (uint64_t)j : 0;          /* the watchdog, once per batch: j firings done (64-bit arithmetic per firing was 7 instructions) */
		last_v = sample_of_last_emit();
		locate(v0, isub0, steps_done, v_cur, isub);
		/* an irregular firing (clock outside the blind window, block end near, candidate missing): with the position on the table
		   it is done the careful way; the prediction is stale after it, so the batch ended there */
		if (miss) careful_firing();

		LAT_TM(tm_serial);
#ifdef LAT_EXP_TIMING
		tm_fired += (unsigned long long)j;
#endif
		/* ---- (3) flush the batch's symbols: lane i writes symbol out_base + i ---- */
		flush();
		LAT_TM(tm_flush);
		/* ---- (4) commit the prefetched chunks ---- */
		/* (a chunk the careful way has loaded by itself meanwhile is not written again) */
#define LAT_CM(c) if ((c) < n_valid && r_hi0 + 64 * (c) >= r_hi) { \
			if (r_hi0 + 64 * (c) + lane < v_end) ring_put(r_hi0 + 64 * (c) + lane, F::decode(pend[c])); \
			r_hi = min(v_end, r_hi0 + 64 * (c) + 64); }
#define LAT_CM4(c) LAT_CM(c) LAT_CM((c) + 1) LAT_CM((c) + 2) LAT_CM((c) + 3)
		LAT_CM4(0)
		if (n_valid > 4) { LAT_CM4(4) if (n_valid > 8) { LAT_CM4(8) if (n_valid > 12) { LAT_CM4(12) if (n_valid > 16) { LAT_CM4(16) } } } }
#undef LAT_CM4
#undef LAT_CM
		__syncthreads();
		LAT_TM(tm_commit);
	}
#ifdef LAT_EXP_TIMING
	if (lane == 0 && stream == 0 && tm_batches)
		printf("[lat] %llu batches, %llu firings in them: ticks per batch prefetch %.0f farm %.0f serial %.0f (%.1f per firing) flush %.0f commit %.0f\n", tm_batches, tm_fired,
		       (double)tm_pf / tm_batches, (double)tm_farm / tm_batches, (double)tm_serial / tm_batches, (double)tm_serial / (double)(tm_fired ? tm_fired : 1),
		       (double)tm_flush / tm_batches, (double)tm_commit / tm_batches);
	if (lane == 0 && stream == 0 && tm_batches)
		printf("[lat] candidate taken: one step early %llu, as predicted %llu, one step late %llu; misses %llu\n", tm_c[0], tm_c[1], tm_c[2], tm_miss);
#endif
#undef LAT_TM
	if (guard == 0 && !done) overflow = 1;                                /* watchdog fired: reported as overflow */

	/* ---- store state ---- */
	if (lane == 0) {
		L.st.agc_gain[stream] = gain; L.st.agc_bias_re[stream] = bias_re; L.st.agc_bias_im[stream] = bias_im;
		L.st.pll_phase[stream] = pll.phase; L.st.pll_freq[stream] = pll.freq; L.st.pll_err[stream] = pll.err;
		L.st.flags[stream] = (int)(fl & 7u) | (dual_state << MDEMOD_FLAG_DUAL_SHIFT);
		L.st.t_phase[stream] = t_phase; L.st.t_freq[stream] = t_freq; L.st.t_prev[stream] = t_prev;
		L.st.inphase[stream] = inphase;
		L.st.n_samples[stream] += (uint64_t)n;
		L.st.n_symbols[stream] = nsym0 + sym_call;
		if (first_lock_call >= 0) L.st.first_lock[stream] = (int64_t)(nsym0 + (uint32_t)first_lock_call);
		L.st.sym_this_call[stream] = sym_call;
		L.st.ev_this_call[stream] = (uint32_t)ev_call;
		L.st.overflow[stream] = overflow;
	}
	/* history := last hpad samples of (old history ++ block), in the context's layout; ascending k is in-place safe */
	if (float_history) {
		float2 *hist = reinterpret_cast<float2 *>(L.st.hist) + (size_t)stream * hpad;
		float2 keep[4];                                                   /* hpad <= 256: four rounds of 64 lanes */
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const int k = r * 64 + lane, idx = n + k;
			if (k < hpad) keep[r] = idx < hpad ? hist[idx] : F::decode(src[idx - hpad]);
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < 4; r++) { const int k = r * 64 + lane; if (k < hpad) hist[k] = keep[r]; }
	} else {
		sample_t *hist = reinterpret_cast<sample_t *>(L.st.hist);
		sample_t keep[4];
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const int k = r * 64 + lane, idx = n + k;
			if (k < hpad) keep[r] = idx < hpad ? hist[(size_t)idx * L.n_streams + stream] : src[idx - hpad];
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < 4; r++) { const int k = r * 64 + lane; if (k < hpad) hist[(size_t)k * L.n_streams + stream] = keep[r]; }
	}
}

template <int FMT>
hipError_t
launch_lat(const DemodLaunch &L, const float *rrc, int ring_size, int span, int float_history, size_t lds_bytes, hipStream_t stream)
{
	auto kfn = L.c.oqpsk ? (L.c.step_safe == 6 ? demod_kernel_lat<FMT, 1, 6> : demod_kernel_lat<FMT, 1, 0>)
	                     : (L.c.step_safe == 14 ? demod_kernel_lat<FMT, 0, 14> : demod_kernel_lat<FMT, 0, 0>);
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(kfn, dim3(L.n_streams), dim3(64), lds_bytes, stream, L, rrc, ring_size, span, float_history);
	return hipGetLastError();
}

} /* namespace */

/* Geometry of the latency kernel for a configuration: samples one batch of kFire firings can advance (+ slack), the LDS ring
 * that holds history + two such spans, and the LDS bytes.  Returns false when the configuration does not fit (huge tables). */
bool
mdemod_lat_geometry(const DemodConsts &c, double samples_per_firing, int *ring_size, int *span, size_t *lds_bytes)
{
	if (c.hpad > kMaxHpad) return false;
	const int sp = static_cast<int>(kFire * samples_per_firing * 1.01) + 8;
	if ((sp + 63) / 64 + 1 > kMaxChunks) return false;
	int ring = 256;
	while (ring < c.hpad + 3 * sp + 128) ring *= 2;
	const size_t bytes = static_cast<size_t>(ring + kMirror) * 8 + (static_cast<size_t>(c.interp) * c.taps + 32) * 4 + 64 * 8 + 64;
	if (bytes > 64 * 1024) return false;
	*ring_size = ring; *span = sp; *lds_bytes = bytes;
	return true;
}

hipError_t
mdemod_launch_demod_lat(const DemodLaunch &L, int fmt, const float *rrc_dev, int ring_size, int span, int float_history, size_t lds_bytes, hipStream_t stream)
{
	switch (fmt) {
	case 16: return launch_lat<16>(L, rrc_dev, ring_size, span, float_history, lds_bytes, stream);
	case 8:  return launch_lat<8>(L, rrc_dev, ring_size, span, float_history, lds_bytes, stream);
	case 32: return launch_lat<32>(L, rrc_dev, ring_size, span, float_history, lds_bytes, stream);
	default: return hipErrorInvalidValue;
	}
}
