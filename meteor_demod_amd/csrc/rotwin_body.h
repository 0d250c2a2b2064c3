/*
 * rotwin_body.h — the demodulator kernel body shared by the rotating-window kernels (demod_kernel_rot.hip: converted float
 * window, std geometry; demod_kernel_rotp.hip: packed raw window, wide / mid / far geometries).
 *
 * One lane = one stream, the reference's serial recurrence bit for bit (citations in demod_device.h).  The FIR window lives in
 * registers that belong to hand-written assembly and never moves: a window policy class W (one object per lane; the register
 * windows have no members, the hybrid one keeps its LDS address) supplies
 *     setup(lds_addr)                  the LDS behind the body's own (0 bytes unless the policy asks the host for more)
 *     put_history(hist, valid, c)      SLIDE history samples (float pairs) -> chunk c of the window at rotation 0
 *     put_init(granules, c)            SLIDE raw input samples             -> chunk c of the window at rotation 0
 *     put(granules, q)                 the slide: SLIDE raw input samples become the newest chunk, chunk q (the oldest) goes
 *     fir(ctab_addr, a, bank, C, rot, re, im)      filter.c:46-65 for a lane whose taps start `a` slots into the window
 * and the geometry (kTaps, kBack, NW, SLIDE, AMAX, MAXSL, BLOCK, ROTN: rotations before the window is where it started,
 * RING: symbols of the per-lane output ring).  Everything else - symbol clock, AGC, NCO, loops, lock
 * detector, output ring, state and history hand-over - is here, compiled for the registers the policy leaves to hipcc.
 */
#ifndef MDEMOD_ROTWIN_BODY_H
#define MDEMOD_ROTWIN_BODY_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "demod_internal.h"
#include "demod_device.h"
#include "clock_jump.h"

#pragma clang fp contract(off)

#ifndef ROT_OQ_SYNC
#define ROT_OQ_SYNC 1            /* OQPSK: the lanes of a wave take their I-rail and Q-rail firings in the same loop iterations */
#endif
/* fast_sin's parabola as a table of 16 385 floats behind the output rings (demod_device.h: md_sin_from_code_lut): the kernel instances
 * compiled for the BASELINE settings take it (template parameter LUT; the host adds MDEMOD_SIN_LUT_BYTES to their LDS), measured
 * +1.4 % / +1.2 % / +0.9 % on configs[1] / [2] / [3] - 12 VALU instructions less per evaluation, two evaluations per firing */
typedef float rot_sinlut_t;

namespace {

template <int FMT> struct RFmt;
template <> struct RFmt<16> { typedef uint32_t sample_t; enum { GDW = 4 }; };    /* dwords per granule of 4 samples */
template <> struct RFmt<8>  { typedef uint16_t sample_t; enum { GDW = 2 }; };
template <> struct RFmt<32> { typedef float2   sample_t; enum { GDW = 8 }; };

template <int FMT> struct RGran { uint32_t w[RFmt<FMT>::GDW]; };

/* 4 consecutive samples starting at block sample m0 (zero SAMPLES past the end) */
template <int FMT>
__device__ __forceinline__ RGran<FMT>
rot_fetch(const typename RFmt<FMT>::sample_t *src, int m0, int n)
{
	typedef typename RFmt<FMT>::sample_t sample_t;
	RGran<FMT> g;
	if (m0 + 3 < n) {
		__builtin_memcpy(&g, src + m0, sizeof(g));
	} else {
		sample_t s[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			if (m0 + u < n) s[u] = src[m0 + u];
			else __builtin_memset(&s[u], 0, sizeof(s[u]));          /* never inside a firing's 65 samples: only ever multiplied by padding */
		}
		__builtin_memcpy(&g, s, sizeof(g));
	}
	return g;
}

/* a * b + c for a, b < 2^24 (b wave-uniform): v_mad_u32_u24 by hand - hipcc turns __mul24 of values it cannot bound into
 * v_mul_lo_u32 plus a v_bfe_i32 */
__device__ __forceinline__ uint32_t
rot_mad24(uint32_t a, uint32_t b, uint32_t c)
{
	uint32_t r;
	asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
	return r;
}

template <int K>
__device__ __forceinline__ float
blind_steps(float p, float f)
{
#pragma unroll
	for (int k = 0; k < K; k++) p = p + f;
	return p;
}

/* The symbol clock's way to the next firing, branch-free (timing.c:32-57): K blind float adds that provably cannot reach the
 * threshold, then four checked ones (the increment is positive, so "reached" is monotone); every add is the reference's add
 * (k * freq != k rounded adds).  Does nothing when the lane is too close to the threshold or to the end of its block: the
 * caller's generic loop steps those. */
struct RotClockConsts { int k_safe, steps_need, interp; float f_hi; uint32_t magic; float inv; const cj_sched *jump; };   /* jump: the launch arguments' two schedules, read where they are used (scalar loads) rather than held in registers */
template <int KS>          /* KS > 0: the number of blind steps is known at compile time (the launcher checks it), 0: any */
__device__ __forceinline__ void
rot_clock_fast(const RotClockConsts &K, float thr, int v_end, float &t_phase, float t_freq, int &isub, int &v_cur, int &fire_sub, bool &fired, int jidx = -1)
{
	int k_safe = KS ? KS : K.k_safe;
	const int interp = K.interp;
	/* any rate with a long run of steps per firing: the run's schedule of closed-form jumps (clock_jump.h), wave-uniform; jidx: which
	 * run this is (0: from 0, 1: the second rail of an OQPSK symbol, from pi; -1: the caller's lanes are not all on the same one) */
	const bool jump = KS == 0 && jidx >= 0 && K.jump[jidx < 0 ? 0 : jidx].nb > 0;
	const cj_sched &J = K.jump[jidx < 0 ? 0 : jidx];
	/* enough input left for k_safe + 4 steps (the part of the current sample still to be stepped is ignored: conservative) */
	const bool fast = (KS == 109 ? (t_phase > CJ109_P_LO && t_phase < CJ109_P_HI)
	                   : jump    ? (t_phase > J.floor && t_phase < J.hi)
	                             : (t_phase < thr - (float)k_safe * K.f_hi - 1e-3f))
	                  && (v_cur + (jump ? J.need : K.steps_need) < v_end);
	if (fast) {
		float p = t_phase;
		if (KS == 109) k_safe = clock_jump_109(p, t_freq, thr, K.inv);   /* configs[3]: 30 real additions and three binades in closed form (clock_jump.h) */
		else if (KS) p = blind_steps<KS>(p, t_freq);                  /* 14: QPSK 72k @ 230 kS/s, -O 5; 6: OQPSK 80k @ 230 kS/s */
		else if (jump) k_safe = clock_jump_run(p, t_freq, thr, K.f_hi, K.inv, J);
		else {
			int k = k_safe;
			for (; k >= 16; k -= 16) p = blind_steps<16>(p, t_freq);
			if (k & 8) p = blind_steps<8>(p, t_freq);
			if (k & 4) p = blind_steps<4>(p, t_freq);
			if (k & 2) p = blind_steps<2>(p, t_freq);
			if (k & 1) p = p + t_freq;
		}
		const float p1 = p + t_freq, p2 = p1 + t_freq, p3 = p2 + t_freq, p4 = p3 + t_freq;
		const bool c1 = p1 >= thr, c2 = p2 >= thr, c3 = p3 >= thr, c4 = p4 >= thr;
		const int m = k_safe + 1 + (c1 ? 0 : 1) + (c2 ? 0 : 1) + (c3 ? 0 : 1);
		float ph = c3 ? p3 : p4;
		ph = c2 ? p2 : ph;
		ph = c1 ? p1 : ph;
		t_phase = ph;
		const uint32_t w = (uint32_t)(isub + m);
		const uint32_t q = (interp == 1) ? w : __umulhi(w, K.magic);     /* floor(w / interp); the magic of 1 does not fit 32 bits */
		const int isub_new = (int)(w - q * (uint32_t)interp);
		v_cur += (int)q + (isub_new > 0 ? 1 : 0) - (isub > 0 ? 1 : 0);   /* samples pushed: ceil(w/interp) - (isub>0) */
		fire_sub = (isub_new == 0) ? interp - 1 : isub_new - 1;
		isub = isub_new;
		fired = c4;
	}
}

/* ---- the kernel body ------------------------------------------------------------- */

template <class W, int FMT, int OQPSK, int KS, int LUT = 0>
__device__ __forceinline__ void
rotwin_demod(const DemodLaunch &L)
{
	typedef typename RFmt<FMT>::sample_t sample_t;
	constexpr int kBack = W::kBack, SLIDE = W::SLIDE, AMAX = W::AMAX, BLOCK = W::BLOCK, NCH = W::NW / W::SLIDE, GPS = W::SLIDE / 4;
	constexpr int NST = GPS * W::DEPTH;                /* granules staged ahead of the window: DEPTH slides' worth (>= MAXSL) */
	static_assert(W::DEPTH >= W::MAXSL, "a loop iteration may slide MAXSL times");
	constexpr int ROTN = W::ROTN, RING = W::RING, RGR = RING / 8;      /* output ring: RGR groups of 8 symbols (16 bytes) per lane */
	/* GATHER: no window at all - every firing loads its own taps from memory (sample rates at which the taps of a firing are a
	 * minority of the samples that pass: demod_kernel_gat.hip) */
	constexpr bool GATHER = W::GATHER;
	static_assert(RING == 32 || RING == 16, "output ring");
	/* per-lane loop state that is only touched once per firing lives in LDS slots [field][lane] unless the window policy has
	 * registers to spare for it (W::REGSLOTS: bit 0 err, 1 t_prev, 2 flags, 3 sample index of the last symbol) */
	constexpr int RS = W::REGSLOTS;
#ifndef ROT_PRIO
#define ROT_PRIO 1
#endif
#ifndef ROT_PRIO_LEVEL
#define ROT_PRIO_LEVEL 2
#endif
	constexpr int PRIO = ROT_PRIO;               /* the scalar stage of a firing (and the symbol clock inside it) at raised wave priority, the FIR and the slide at 0 (NOTEBOOK.md 5.0: +5 % on configs[1] when round 2 found it) */

	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	float *ctab = reinterpret_cast<float *>(lds);
	float *lut = ctab + L.ctab_floats;
	enum { S_GAIN, S_BIAS_RE, S_BIAS_IM, S_PHASE, S_FREQ, S_ERR, S_FLAGS, S_TPREV, S_INPHASE, S_EVCALL, S_FIRSTLOCK, S_LASTV, S_COUNT };
	static_assert(S_COUNT == MDEMOD_RW_STATE_SLOTS, "host LDS sizing");
	float *sl = lut + 32 + (threadIdx.x >> 6) * (S_COUNT * 64) + (threadIdx.x & 63);
	int *sli = reinterpret_cast<int *>(sl);
	uint4 *stage = reinterpret_cast<uint4 *>(lut + 32 + (BLOCK / 64) * (S_COUNT * 64)) + threadIdx.x;
	const uint32_t ctab_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;
	W win;
	rot_sinlut_t *sintab = reinterpret_cast<rot_sinlut_t *>(stage - threadIdx.x + BLOCK * RGR);
	if constexpr (LUT) md_sin_lut_fill(sintab, (int)threadIdx.x, BLOCK);
	win.setup(ctab_addr + (uint32_t)(reinterpret_cast<unsigned char *>(stage - threadIdx.x + BLOCK * RGR) - lds) + (LUT ? MDEMOD_SIN_LUT_BYTES : 0));

	const DemodConsts &C = L.c;
	const uint32_t stream = blockIdx.x * blockDim.x + threadIdx.x;
	const bool valid = stream < L.n_streams;

	for (uint32_t i = threadIdx.x; i < L.ctab_floats; i += blockDim.x) ctab[i] = L.ctab[i];
	if (threadIdx.x < 32) lut[threadIdx.x] = L.tanh_lut[threadIdx.x];

	int n = 0;
	const sample_t *src = nullptr;
	if (valid) {
		n = (int)(L.n_samples_arr ? L.n_samples_arr[stream] : L.n_samples);
		const uint64_t off = L.iq_offset ? L.iq_offset[stream] : (uint64_t)stream * L.iq_stride;
		src = reinterpret_cast<const sample_t *>(L.iq) + off;
	}
	const int v_end = kBack + n;
	const int interp = C.interp;

	float t_phase = 0.0f, t_freq = C.t_center;
	int dual_state = 1;
	float r_gain = 1.0f, r_bias_re = 0.0f, r_bias_im = 0.0f, r_phase = 0.0f, r_freq = 0.0f;
	float r_err = 1000.0f, r_tprev = 0.0f;
	int r_flags = 0, r_lastv = -1;
	auto ld_err = [&]() -> float { return (RS & 1) ? r_err : sl[S_ERR * 64]; };
	auto st_err = [&](float v) { if (RS & 1) r_err = v; else sl[S_ERR * 64] = v; };
	auto ld_tprev = [&]() -> float { return (RS & 2) ? r_tprev : sl[S_TPREV * 64]; };
	auto st_tprev = [&](float v) { if (RS & 2) r_tprev = v; else sl[S_TPREV * 64] = v; };
	auto ld_flags = [&]() -> int { return (RS & 4) ? r_flags : sli[S_FLAGS * 64]; };
	auto st_flags = [&](int v) { if (RS & 4) r_flags = v; else sli[S_FLAGS * 64] = v; };
	auto ld_lastv = [&]() -> int { return (RS & 8) ? r_lastv : sli[S_LASTV * 64]; };
	auto st_lastv = [&](int v) { if (RS & 8) r_lastv = v; else sli[S_LASTV * 64] = v; };
	{
		float err = 1000.0f, t_prev = 0.0f, inph = 0.0f;
		int fl = MDEMOD_FLAG_UPDOWN_POS | (1 << MDEMOD_FLAG_DUAL_SHIFT);
		if (valid) {
			r_gain = L.st.agc_gain[stream]; r_bias_re = L.st.agc_bias_re[stream]; r_bias_im = L.st.agc_bias_im[stream];
			r_phase = L.st.pll_phase[stream]; r_freq = L.st.pll_freq[stream]; err = L.st.pll_err[stream];
			fl = L.st.flags[stream];
			t_phase = L.st.t_phase[stream]; t_freq = L.st.t_freq[stream]; t_prev = L.st.t_prev[stream];
			inph = L.st.inphase[stream];
		}
		dual_state = (fl >> MDEMOD_FLAG_DUAL_SHIFT) & 3;
		st_err(err);
		st_flags(fl & 7);
		st_tprev(t_prev); sl[S_INPHASE * 64] = inph;
		sli[S_EVCALL * 64] = 0; sli[S_FIRSTLOCK * 64] = -1; st_lastv(-1);
	}

	/* ---- window: the first kBack slots = history ([stream][kBack] float pairs), the rest = the first granules of the block ---- */
	const float2 *hist_in = reinterpret_cast<const float2 *>(L.st.hist) + (size_t)(valid ? stream : 0) * kBack;
	if constexpr (!GATHER) {
		const float2 *hist = hist_in;
#pragma unroll 1
		for (int c = 0; c < kBack / SLIDE; c++) win.put_history(hist + c * SLIDE, valid, __builtin_amdgcn_readfirstlane(c));
#pragma unroll 1
		for (int c = 0; c < NCH - kBack / SLIDE; c++) {
			RGran<FMT> g[GPS];
#pragma unroll
			for (int i = 0; i < GPS; i++) g[i] = rot_fetch<FMT>(src, SLIDE * c + 4 * i, n);
			win.put_init(g, __builtin_amdgcn_readfirstlane(kBack / SLIDE + c));
		}
	}
	int g_load = (W::NW - kBack) / 4;                      /* next block granule to fetch (wave-uniform) */
	RGran<FMT> stg[GATHER ? 1 : NST];
	if constexpr (!GATHER) {
#pragma unroll
		for (int i = 0; i < NST; i++) stg[i] = rot_fetch<FMT>(src, 4 * (g_load + i), n);
		g_load += NST;
	}

	__syncthreads();                                       /* coefficient rows + LUT visible */
#ifdef ROT_EXP_PHASE             /* experiment (r06, NOTEBOOK R6.3): the waves of a block selected by ROT_EXP_PHASE_MASK start ROT_EXP_PHASE x 64 cycles late -
                                    do two waves of a SIMD that are out of step (one in its FIR while the other is in its scalar stage) fill each other's bubbles?
                                    Measured: no - offsets of 900 / 1 800 / 3 600 cycles on waves 4-7, odd waves or waves 2-3, 6-7: +-0.3 % on configs[1], -1 % on configs[2] (NOTEBOOK R6.3) */
	if ((threadIdx.x >> 6) & ROT_EXP_PHASE_MASK) __builtin_amdgcn_s_sleep(ROT_EXP_PHASE);
#endif

	int rot = 0;                                           /* physical chunk that is logical chunk 0 (wave-uniform) */
	/* OQPSK: demod.c:62-84 does half the work on the I-rail firing (state 1: one mixer product, no timing or Costas update, no
	 * symbol).  The lanes of a wave are independent streams and would take their rails in any order; here a loop iteration serves
	 * ONE rail (wave-uniform `slot`, alternating), a lane whose pending firing is on the other rail waits an iteration - so the
	 * `emit` branches below are scalar and the I-rail iterations skip that code instead of running it masked off. */
	int slot = 1;
	int base = 0;
	int v_cur = kBack - 1;
	int isub = 0, fire_sub = 0;
	bool fired = false;
	bool done = !valid || n == 0;
	uint32_t sym_call = 0;
	RotClockConsts K;
	K.k_safe = C.step_safe; K.f_hi = C.step_fmax; K.magic = C.interp_magic; K.interp = C.interp; K.inv = C.step_inv;
	K.steps_need = ((KS == 109 ? CJ109_MAX_STEPS : C.step_safe) + 4 + C.interp - 1) / C.interp;      /* samples that hold k_safe + 4 steps */
	K.jump = C.jump;

	int n_wave_max = n;
	for (int o = 32; o > 0; o >>= 1) {
		const int other = __shfl_xor(n_wave_max, o);
		n_wave_max = other > n_wave_max ? other : n_wave_max;
	}
	n_wave_max = __builtin_amdgcn_readfirstlane(n_wave_max);
	const uint64_t guard64 = 4ull * (uint64_t)(n_wave_max + kBack) * (uint64_t)interp + 4096ull;
	uint32_t guard = guard64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)guard64;
#ifdef ROT_EXP_TIMING            /* experiment: where a wave's time goes (s_memtime between the stages; block 100, wave 0 prints) */
	uint64_t tacc[5] = { 0, 0, 0, 0, 0 }, tlast = clock64();
	uint32_t n_iter = 0, n_fir = 0, n_slide = 0, n_wslow = 0;
#define ROT_TICK(i) do { const uint64_t t_ = clock64(); tacc[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define ROT_TICK(i) do { } while (0)
#endif
	do {
		ROT_TICK(4);
		/* ---- (1) symbol clock (timing.c:32-57) of the lanes without a pending firing: the first iteration, block ends,
		 * lanes that ran ahead of the window.  A lane that fires does its next clock inside the firing (below). ---- */
		if (!fired && !done) {
			const float thr = OQPSK ? (float)dual_state * MD_PI_F : MD_TWO_PI_F;
			rot_clock_fast<KS>(K, thr, v_end, t_phase, t_freq, isub, v_cur, fire_sub, fired);
			while (!fired && !done) {
				if (isub == 0) {
					if (v_cur + 1 >= v_end) { done = true; break; }
					v_cur++;
				}
				t_phase = t_phase + t_freq;
				fire_sub = isub;
				isub = (isub + 1 == interp) ? 0 : isub + 1;
				if (t_phase >= thr) fired = true;
			}
		}
		if (md_all(done)) break;
		ROT_TICK(0);
#ifdef ROT_EXP_TIMING
		n_iter++;
#endif

		/* ---- (2) slide: when nobody needs logical chunk 0 any more it becomes the newest chunk ---- */
		if constexpr (!GATHER)
#pragma unroll
		for (int r = 0; r < W::MAXSL; r++) {
			const int a_now = v_cur - kBack - base;
			if (md_all(done || a_now >= SLIDE)) {
				RGran<FMT> g[GPS];
#pragma unroll
				for (int i = 0; i < GPS; i++) g[i] = stg[i];
				win.put(g, __builtin_amdgcn_readfirstlane(rot));
				rot = __builtin_amdgcn_readfirstlane((rot == ROTN - 1) ? 0 : rot + 1);      /* wave-uniform: keep them in SGPRs */
				base = __builtin_amdgcn_readfirstlane(base + SLIDE);
#pragma unroll
				for (int i = 0; i + GPS < NST; i++) stg[i] = stg[i + GPS];
				const int m_new = 4 * g_load;
				if (md_all(m_new + SLIDE - 1 < n)) {
#pragma unroll
					for (int i = 0; i < GPS; i++) __builtin_memcpy(&stg[NST - GPS + i], src + m_new + 4 * i, sizeof(RGran<FMT>));
				} else {
#pragma unroll
					for (int i = 0; i < GPS; i++) stg[NST - GPS + i] = rot_fetch<FMT>(src, m_new + 4 * i, n);
				}
				g_load += GPS;
			}
		}

		ROT_TICK(1);
		/* ---- (3) the firing, if its taps are inside the window ---- */
		const int a = v_cur - kBack - base;
		if (fired && (GATHER || a <= AMAX) && (!(OQPSK && ROT_OQ_SYNC) || dual_state == slot)) {
			fired = false;
			const int bank = interp - 1 - fire_sub;                     /* filter.c:52 */
			cf32 y;
			if constexpr (GATHER) win.fir_gather(ctab_addr, src, hist_in, v_cur, n, bank, C, y.re, y.im);
			else win.fir(ctab_addr, a, bank, C, __builtin_amdgcn_readfirstlane(rot), y.re, y.im);
			if (PRIO) __builtin_amdgcn_s_setprio(ROT_PRIO_LEVEL);
			ROT_TICK(2);
#ifdef ROT_EXP_TIMING
			n_fir++;
#endif

			y = md_agc<(KS == 14 && !OQPSK && LUT)>(y, r_gain, r_bias_re, r_bias_im);        /* (the configs[1] instance: the short cabsf, demod_device.h) */
			uint32_t fl = (uint32_t)ld_flags();          /* bit 0 locked, 1 locked_once, 2 updown > 0, 3 overflow */
			PllWord pll;
			pll.phase = r_phase; pll.freq = r_freq;
			pll.err = ld_err();

			float sn, cs;
			if constexpr (LUT) {
				sn = md_sin_from_code_lut(sintab, md_turn_code<false>(-pll.phase));
				cs = md_sin_from_code_lut(sintab, md_turn_code<false>((float)((double)(-pll.phase) + MD_HALF_PI_D)));   /* sincos.c:37-40 */
			} else {
				sn = md_fast_sin<false>(-pll.phase);
				cs = md_fast_cos<false>(-pll.phase);
			}
			bool emit = true;
			float out_re, out_im;
			if (OQPSK) {
				float inphase = sl[S_INPHASE * 64];
				const int rail = ROT_OQ_SYNC ? slot : dual_state;
				if (rail == 1) { inphase = y.re * cs - y.im * sn; emit = false; sl[S_INPHASE * 64] = inphase; }   /* demod.c:66-71 */
				out_re = inphase;
				out_im = y.re * sn + y.im * cs;                                          /* demod.c:76    */
				dual_state = 3 - rail;                                                   /* timing.c:52   */
			} else {
				out_re = y.re * cs - y.im * sn;
				out_im = y.re * sn + y.im * cs;
			}
			md_nco_advance<true>(pll.phase, pll.freq);

			if (emit) {
				/* demod.c:33-47: only the LAST symbol fired inside one input sample survives (found by fuzzing in round 2: NOTEBOOK.md 5.0) */
				const bool again = (v_cur == ld_lastv());
				st_lastv(v_cur);
				if (__builtin_expect(md_any(again), 0)) { if (again) sym_call--; }
				float t_prev = ld_tprev();
				md_timing_update(t_phase, t_freq, t_prev, C.t_alpha, C.t_beta, C.t_center, C.t_maxdev, out_im);
				st_tprev(t_prev);
			}
			/* the clock's way to the NEXT firing starts here: a chain of ~20 dependent adds that needs nothing but the timing
			 * update, next to the Costas update, the AGC's square root and the quantiser, which need nothing from it */
			rot_clock_fast<KS>(K, OQPSK ? (float)dual_state * MD_PI_F : MD_TWO_PI_F, v_end, t_phase, t_freq, isub, v_cur, fire_sub, fired,
			                   OQPSK ? (ROT_OQ_SYNC ? 2 - slot : -1) : 0);
#ifdef ROT_PRIO_SPLIT                /* experiment: what follows is off the path to the next FIR */
			if (PRIO) __builtin_amdgcn_s_setprio(ROT_PRIO_SPLIT);
#endif
#ifdef ROT_EXP_TIMING
			if (!fired) n_slide++;                     /* (experiment: lanes the fast clock left to the stepping loop) */
			if (md_any(!fired)) n_wslow++;
#endif
			if (emit) {
				uint32_t first = 0;
				const uint32_t changed = md_pll_update_packed(pll, fl, lut, C.pll_alpha, C.pll_beta, C.pll_fmax, out_re, out_im, first);
				if (__builtin_expect(changed != 0, 0)) {          /* a plain divergent branch: skipped when no lane of the wave takes it */
					if (first) sli[S_FIRSTLOCK * 64] = (int)sym_call;
					const int ev_call = sli[S_EVCALL * 64];
					if (ev_call < MDEMOD_MAX_LOCK_EVENTS) {
						mdemod_lock_event ev;
						ev.symbol = L.st.n_symbols[stream] + sym_call; ev.locked = (int)(fl & 1u); ev.pad = 0;
						L.st.events[(size_t)stream * MDEMOD_MAX_LOCK_EVENTS + ev_call] = ev;
					}
					sli[S_EVCALL * 64] = ev_call + 1;
				}
				const uint32_t sym = (uint32_t)(md_quantise(out_re) & 0xFF) | ((uint32_t)(md_quantise(out_im) & 0xFF) << 8);
				/* RING-symbol ring per lane in LDS, [16-byte group][thread]; every RING symbols one full run (64 bytes for 32) goes out */
				const uint32_t k = sym_call & (uint32_t)(RING - 1);
				reinterpret_cast<uint16_t *>(stage + (k >> 3) * BLOCK)[k & 7u] = (uint16_t)sym;
				sym_call++;
				if (__builtin_expect((sym_call & (uint32_t)(RING - 1)) == 0, 0)) {
					int8_t *soft_out = L.soft + (size_t)stream * L.soft_stride * 2;
					if (__builtin_expect(sym_call <= L.soft_cap, 1)) {
						/* one 16-byte group in flight at a time: four would not fit next to the live registers of the wide geometry
						 * and spill (once per 32 symbols: the three extra LDS round trips do not show) */
						/* memcpy, not a uint4 lvalue: the caller's buffer and pitch need not be 16-byte aligned (the hardware's
						 * unaligned access mode takes the same global_store_dwordx4 either way) */
						int8_t *dst = soft_out + 2 * (size_t)(sym_call - RING);
#pragma unroll
						for (int g4 = 0; g4 < RGR; g4++) {
							const uint4 qv = stage[g4 * BLOCK];
							__builtin_memcpy(dst + 16 * g4, &qv, 16);
							asm volatile("" ::: "memory");
						}
					} else {
						fl |= 8;
						for (uint32_t i = 0; i < (uint32_t)RING && sym_call - RING + i < L.soft_cap; i++)
							*reinterpret_cast<uint16_t *>(soft_out + 2 * (size_t)(sym_call - RING + i)) =
							    reinterpret_cast<const uint16_t *>(stage + (i >> 3) * BLOCK)[i & 7u];
					}
				}
			}
			r_phase = pll.phase; r_freq = pll.freq;
			st_err(pll.err);
			st_flags((int)fl);
			if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
			ROT_TICK(3);
		}
		if (OQPSK && ROT_OQ_SYNC) slot = __builtin_amdgcn_readfirstlane(3 - slot);      /* (keeps it in an SGPR: scalar branches) */
	} while (--guard);
#ifdef ROT_EXP_TIMING
	for (int o = 32; o > 0; o >>= 1) n_slide += __shfl_xor(n_slide, o);
	if (blockIdx.x == 100 && threadIdx.x == 0)
		printf("ROT_TIMING iters %u firs %u (lanes left to the stepping loop: %u of the wave's, in %u firings) cycles: clock %llu slide %llu fir %llu scalar %llu latch %llu\n", n_iter, n_fir, n_slide, n_wslow,
		       (unsigned long long)tacc[0], (unsigned long long)tacc[1], (unsigned long long)tacc[2], (unsigned long long)tacc[3], (unsigned long long)tacc[4]);
#endif

	if (guard == 0) st_flags(ld_flags() | 8);                        /* watchdog fired: reported as overflow */

	uint32_t stream_e = stream;
	asm volatile("" : "+v"(stream_e));
	int8_t *soft_e = L.soft + (size_t)stream_e * L.soft_stride * 2;
	int overflow = (ld_flags() >> 3) & 1;

	/* ---- flush the ring ---- */
	if (valid) {
		const uint32_t r = sym_call & (uint32_t)(RING - 1), sb = sym_call - r;
		for (uint32_t g = 0; g < (r >> 3); g++) {
			if (sb + 8 * g + 8 <= L.soft_cap) { const uint4 qv = stage[g * BLOCK]; __builtin_memcpy(soft_e + 2 * (size_t)(sb + 8 * g), &qv, 16); }
			else for (uint32_t i = 0; i < 8; i++) {
				if (sb + 8 * g + i < L.soft_cap) *reinterpret_cast<uint16_t *>(soft_e + 2 * (size_t)(sb + 8 * g + i)) = reinterpret_cast<const uint16_t *>(stage + g * BLOCK)[i];
				else overflow = 1;
			}
		}
		for (uint32_t i = r & ~7u; i < r; i++) {
			if (sb + i < L.soft_cap) *reinterpret_cast<uint16_t *>(soft_e + 2 * (size_t)(sb + i)) = reinterpret_cast<const uint16_t *>(stage + (i >> 3) * BLOCK)[i & 7u];
			else overflow = 1;
		}
	}

	/* ---- store state ---- */
	if (valid) {
		const int fl = ld_flags();
		L.st.agc_gain[stream_e] = r_gain; L.st.agc_bias_re[stream_e] = r_bias_re; L.st.agc_bias_im[stream_e] = r_bias_im;
		L.st.pll_phase[stream_e] = r_phase; L.st.pll_freq[stream_e] = r_freq; L.st.pll_err[stream_e] = ld_err();
		L.st.flags[stream_e] = (fl & 7) | (dual_state << MDEMOD_FLAG_DUAL_SHIFT);
		L.st.t_phase[stream_e] = t_phase; L.st.t_freq[stream_e] = t_freq; L.st.t_prev[stream_e] = ld_tprev();
		L.st.inphase[stream_e] = sl[S_INPHASE * 64];
		const uint64_t nsym0 = L.st.n_symbols[stream_e];
		const int first_lock_call = sli[S_FIRSTLOCK * 64];
		L.st.n_samples[stream_e] += (uint64_t)n;
		L.st.n_symbols[stream_e] = nsym0 + sym_call;
		if (first_lock_call >= 0) L.st.first_lock[stream_e] = (int64_t)(nsym0 + (uint32_t)first_lock_call);
		L.st.sym_this_call[stream_e] = sym_call;
		L.st.ev_this_call[stream_e] = (uint32_t)sli[S_EVCALL * 64];
		L.st.overflow[stream_e] = overflow;

		/* history := last 64 samples of (old history ++ block), as floats; ascending k is in-place safe */
		float2 *hist = reinterpret_cast<float2 *>(L.st.hist) + (size_t)stream_e * kBack;     /* [stream][kBack] */
		for (int k = 0; k < kBack; k++) {
			const int idx = n + k;
			float2 h;
			if (idx < kBack) h = hist[idx];
			else {
				const sample_t raw = src[idx - kBack];
				if constexpr (FMT == 16) h = make_float2((float)(int)(int16_t)(raw & 0xFFFFu), (float)((int)raw >> 16));
				else if constexpr (FMT == 8) h = make_float2((float)((int)(raw & 0xFFu) - 128), (float)((int)(raw >> 8) - 128));
				else h = raw;
			}
			hist[k] = h;
		}
	}
}


} /* namespace */

#endif
