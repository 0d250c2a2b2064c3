/*
 * demod_kernel_rw.hip — v2 "register window" demodulator kernel for gfx950.
 *
 * Same contract as demod_kernel.hip (one lane = one stream, bit-exact serial
 * recurrence per stream, reference citations in demod_device.h), different data
 * movement, chosen from measurements on MI355X (profiles/r01_v1_baseline.md,
 * tools/ubench/valu_rates.hip):
 *
 *   - v1 was latency bound at 6 waves/CU because its 24 KB-per-wave LDS ring
 *     capped occupancy, and its FIR paid 2 SDWA converts (4.2 cycles each) plus
 *     packed f32 ops (6.3 cycles each) per tap.
 *   - Here the FIR window lives in VGPRs (the register file is 3.2x the LDS) as
 *     ALREADY CONVERTED floats: NW = 80 samples x (re, im).  One tap is two
 *     scalar v_mul_f32 + two v_add_f32 (2.3 cycles each): 9.2 cycles instead of
 *     17.6, and LDS holds only the coefficient rows.
 *   - All lanes of a wave slide their windows TOGETHER, by 8 slots (two
 *     granules) at a time, with plain register moves: 144 v_mov_b32 every ~2.5
 *     symbols (~130 cycles per symbol against ~740 for the FIR).  Per-lane
 *     position differences are absorbed by the coefficient row: a lane whose
 *     65 samples start `a` slots into the register window reads row (a, bank),
 *     zero padded (exact: acc + (+-0) == acc and acc is never -0).  A lane that
 *     runs ahead of the window idles one iteration, a lane that lags gates the
 *     slide.  (A rotate-by-renaming variant with 19 unrolled FIR copies was
 *     tried first: hipcc merged the copies behind PHI moves and spilled.)
 *   - Coefficient rows are read with ds_read_b64 and an odd 8-byte row stride:
 *     the <= 32 consecutive rows a wave uses map to distinct LDS bank slots.
 *   - The symbol clock is stepped branch-free: K blind float adds that provably
 *     cannot fire, then a few predicated checked steps; a generic loop remains
 *     for block ends and unusual states.  Every add is the reference's add.
 *   - Soft symbols are collected 8 per lane and written as one 16-byte store.
 *   - The 160 window registers leave ~95 VGPRs at 2 waves/SIMD (the minimum that
 *     reaches full VALU issue rate); the compiler needed 293 and spilled to scratch
 *     (= HBM: 140 GB of spill traffic per launch).  The per-symbol loop state
 *     (AGC, PLL, lock flags, counters: 11 values) therefore lives in per-lane LDS
 *     slots [field][lane] (conflict free) and is only in registers while the
 *     scalar part of a symbol runs.
 *   - Filters shorter than 65 taps are embedded as 65-tap filters with leading
 *     zero coefficients (exact for the same reason as the padding).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "demod_internal.h"
#include "demod_device.h"

#pragma clang fp contract(off)

#ifndef MDEMOD_RW_3WAVES_NW
#define MDEMOD_RW_3WAVES_NW 96        /* packed windows up to this many slots are built for 3 waves/SIMD (<= 168 VGPRs): the mid geometry spills ~30 dwords there and is still 7 % faster than at 2 waves */
#endif
#ifndef MDEMOD_RW_PART
#define MDEMOD_RW_PART 0              /* 0 = both geometries in one object, 1 = std only, 2 = wide only */
#endif
#ifndef MDEMOD_RW_REGSTATE
#define MDEMOD_RW_REGSTATE 1          /* AGC / NCO state of the float std variant in VGPRs instead of LDS slots */
#endif
#ifndef MDEMOD_RW_SETPRIO
#define MDEMOD_RW_SETPRIO 1           /* the scalar stage of a firing runs at raised wave priority, see the main loop */
#endif
#ifndef MDEMOD_RW_STAGE
#define MDEMOD_RW_STAGE 2             /* soft symbols leave in 64-byte runs: 2 = a 32-symbol ring per lane in LDS; 1 = groups of 8 collected in registers, three of them staged in LDS; 0 = 16-byte stores */
#endif
#ifndef MDEMOD_RW_PREFETCH
#define MDEMOD_RW_PREFETCH 2          /* FIR coefficient prefetch distance (chunks) of the float std variant */
#endif

namespace {

/* Window geometry.  Std: filters up to 65 taps, <= 3.6 samples per firing (the LRPT rates at
 * ~230 kS/s).  Wide: up to 129 taps and 15 samples per firing (1 MS/s recordings): the window
 * only fits as packed raw samples, the lanes of a wave are spread over a whole symbol period
 * (up to 15 samples) so 24 alignments are needed, and the per-alignment coefficient rows
 * (24 x interp x 154 floats) no longer fit in LDS: the table is stored once per bank, zero
 * padded on both sides, in two copies shifted by one float so that every (alignment, slot
 * pair) is an aligned 8-byte read ("compact"). */
template <int KT_, int NW_, int SLIDE_, int MAXSL_, bool COMPACT_, int BLOCK_>
struct Geo {
	static constexpr int SLIDE = SLIDE_;      /* slots per slide (whole granules)        */
	static constexpr int KT = KT_;            /* embedded filter length                  */
	static constexpr int KB = KT_ - 1;        /* history length                          */
	static constexpr int NW = NW_;            /* window slots                            */
	static constexpr int MAXSL = MAXSL_;      /* slides of 8 slots per loop iteration    */
	static constexpr bool COMPACT = COMPACT_;
	static constexpr int BLOCK = BLOCK_;
};
typedef Geo<65, 80, 8, 1, false, MDEMOD_RW_BLOCK> GeoStd;
typedef Geo<129, MDEMOD_RW_WIDE_NW, MDEMOD_RW_WIDE_SLIDE, MDEMOD_RW_WIDE_MAXSL, true, MDEMOD_RW_WIDE_BLOCK> GeoWide;
/* Mid: the short filter at a high sample rate: wide's lane spread (32 alignments, 16-slot slides, compact table) on a
 * 96-slot packed window */
typedef Geo<65, MDEMOD_RW_MID_NW, 16, 1, true, MDEMOD_RW_BLOCK> GeoMid;
/* Far: the same at 2 MS/s-class rates (up to 30 samples per firing): 48 alignments, two slides per iteration */
typedef Geo<65, MDEMOD_RW_FAR_NW, 16, 2, true, MDEMOD_RW_BLOCK> GeoFar;

template <int FMT> struct Fmt;
template <> struct Fmt<16> {
	typedef uint32_t sample_t;
	__device__ static __forceinline__ cf32 decode(uint32_t w) {
		cf32 r; r.re = (float)(int)(int16_t)(w & 0xFFFFu); r.im = (float)((int)w >> 16); return r;
	}
};
template <> struct Fmt<8> {
	typedef uint16_t sample_t;
	__device__ static __forceinline__ cf32 decode(uint16_t w) {
		cf32 r; r.re = (float)((int)(w & 0xFFu) - 128); r.im = (float)((int)(w >> 8) - 128); return r;
	}
};
template <> struct Fmt<32> {
	typedef float2 sample_t;
	__device__ static __forceinline__ cf32 decode(float2 w) { cf32 r; r.re = w.x; r.im = w.y; return r; }
};

template <int FMT> struct Gran { typename Fmt<FMT>::sample_t s[4]; };

/* Window element: either the raw sample (PACKED: one VGPR for s16/u8, converted at every use)
 * or the converted float pair (two VGPRs, converted once). */
template <int FMT, bool PACKED> struct Win;
template <int FMT> struct Win<FMT, true> {
	typedef typename Fmt<FMT>::sample_t elem_t;
	__device__ static __forceinline__ elem_t pack(typename Fmt<FMT>::sample_t raw) { return raw; }
	__device__ static __forceinline__ cf32 get(elem_t e) { return Fmt<FMT>::decode(e); }
	__device__ static __forceinline__ elem_t from_float(float2 h);
};
template <> __device__ __forceinline__ uint32_t Win<16, true>::from_float(float2 h) {
	return ((uint32_t)(int)h.x & 0xFFFFu) | ((uint32_t)(int)h.y << 16);
}
template <> __device__ __forceinline__ uint16_t Win<8, true>::from_float(float2 h) {
	return (uint16_t)((((int)h.x + 128) & 0xFF) | ((((int)h.y + 128) & 0xFF) << 8));
}
template <> __device__ __forceinline__ float2 Win<32, true>::from_float(float2 h) { return h; }
template <int FMT> struct Win<FMT, false> {
	typedef float2 elem_t;
	__device__ static __forceinline__ elem_t pack(typename Fmt<FMT>::sample_t raw) {
		const cf32 s = Fmt<FMT>::decode(raw); return make_float2(s.re, s.im);
	}
	__device__ static __forceinline__ cf32 get(elem_t e) { cf32 r; r.re = e.x; r.im = e.y; return r; }
	__device__ static __forceinline__ elem_t from_float(float2 h) { return h; }
};

/* 4 consecutive samples starting at block sample m0 (zeros past the end). */
template <int FMT>
__device__ __forceinline__ Gran<FMT>
fetch_granule(const typename Fmt<FMT>::sample_t *src, int m0, int n)
{
	Gran<FMT> g;
	if (m0 + 3 < n) {
		__builtin_memcpy(&g, src + m0, sizeof(g));
	} else {
#pragma unroll
		for (int u = 0; u < 4; u++) {
			if (m0 + u < n) g.s[u] = src[m0 + u];
			else __builtin_memset(&g.s[u], 0, sizeof(g.s[u]));
		}
	}
	return g;
}

/* ---- FIR over the rotated register window (filter.c:46-65) --------------------- */

/* filter.c:55-62: sequential, oldest first, unfused.  Coefficients arrive in chunks of CH
 * taps, one chunk ahead of the arithmetic; the scheduling barrier between chunks keeps the
 * scheduler from hoisting every ds_read to the top (80 live VGPRs of coefficients would
 * spill the window). */
/* Coefficient fetch.  `volatile` keeps hipcc from fusing two ds_read_b64 into one
 * ds_read2_b64: the fused form is serviced in 16-lane groups over 16 bank slots and measured
 * 2.7x bank-conflict cycles on the ~40 distinct rows a wave uses; plain b64 has 32 slots. */
typedef float coef2_t __attribute__((ext_vector_type(2)));
typedef float coef4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float2
ld_coef(const float *row, int s)
{
	/* explicit LDS address space: address-space inference does not look through volatile */
	typedef const volatile coef2_t __attribute__((address_space(3))) *lds_coef_ptr;
	const coef2_t v = *(lds_coef_ptr)(row + s);
	return make_float2(v.x, v.y);
}
/* Four coefficients with one ds_read_b128 (rows 16-byte aligned): half the LDS instructions of the b64 form at the
 * same LDS bandwidth; serviced in 16-lane groups over 16 slots of 16 bytes. */
__device__ __forceinline__ void
ld_coef4(const float *row, int s, float2 &lo, float2 &hi)
{
	typedef const volatile coef4_t __attribute__((address_space(3))) *lds_coef4_ptr;
	const coef4_t v = *(lds_coef4_ptr)(row + s);
	lo = make_float2(v.x, v.y); hi = make_float2(v.z, v.w);
}
/* one chunk of 8 coefficients into h[4] */
template <bool WIDE_LOADS>
__device__ __forceinline__ void
ld_chunk(const float *row, int s0, float2 (&h)[4])
{
	if (WIDE_LOADS) {
		ld_coef4(row, s0, h[0], h[1]);
		ld_coef4(row, s0 + 4, h[2], h[3]);
	} else {
#pragma unroll
		for (int j = 0; j < 4; j++) h[j] = ld_coef(row, s0 + 2 * j);
	}
}

/* One chunk of CH taps: filter.c:55-62, sequential, oldest first, unfused. */
template <int NW, typename W, int CH>
__device__ __forceinline__ void
fir_chunk(const typename W::elem_t (&win)[NW], int c, const float2 (&h)[CH / 2], float &ar, float &ai)
{
#pragma unroll
	for (int j = 0; j < CH / 2; j++) {
		const int s = c * CH + 2 * j;
		const cf32 x0 = W::get(win[s]), x1 = W::get(win[s + 1]);
		ar = ar + x0.re * h[j].x;
		ai = ai + x0.im * h[j].x;
		ar = ar + x1.re * h[j].y;
		ai = ai + x1.im * h[j].y;
	}
}

/*
 * FIR over the register window.  Chunks [c_lo, c_hi) of 8 slots are evaluated (the caller
 * drops an edge chunk whose coefficients are zero for every lane of the wave).  Coefficients
 * are fetched two chunks ahead; the empty asm ties the fetch address to the accumulator so
 * the compiler cannot hoist all 40 reads to the top (80 live VGPRs of coefficients was what
 * capped the kernel at 2 waves/SIMD).
 */
template <int NW, typename W, bool WIDE_LOADS, int PF>
__device__ __forceinline__ void
fir_window(const typename W::elem_t (&win)[NW], const float *row, bool skip_first, bool skip_last,
           float &out_re, float &out_im)
{
	constexpr int CH = 8, NCH = NW / CH;
	static_assert(NW % CH == 0, "window is a whole number of chunks");
	static_assert(CH == 8, "ld_chunk loads 8 coefficients");
	float ar = 0.0f, ai = 0.0f;

	if (PF == 1) {
		float2 h0[CH / 2], h1[CH / 2];
		if (!skip_first) {                                   /* wave-uniform */
			ld_chunk<WIDE_LOADS>(row, 0, h0);
			fir_chunk<NW, W, CH>(win, 0, h0, ar, ai);
		}
		ld_chunk<WIDE_LOADS>(row, CH, h0);
#pragma unroll
		for (int c = 1; c < NCH - 1; c++) {
			int tie = 0;                                             /* opaque zero: keeps the LDS address space of `row` */
			asm volatile("" : "+v"(tie) : "v"(ar), "v"(ai));         /* fetch of chunk c+1 may not pass chunk c-1's sum */
			if (c + 1 < NCH - 1) ld_chunk<WIDE_LOADS>(row + tie, (c + 1) * CH, h1);
			fir_chunk<NW, W, CH>(win, c, h0, ar, ai);
#pragma unroll
			for (int j = 0; j < CH / 2; j++) h0[j] = h1[j];
		}
		if (!skip_last) {
			ld_chunk<WIDE_LOADS>(row, (NCH - 1) * CH, h0);
			fir_chunk<NW, W, CH>(win, NCH - 1, h0, ar, ai);
		}
	} else {
		/* coefficients two chunks ahead of the arithmetic (three rotating buffers: 8 more VGPRs): the LDS latency
		 * under load is about one chunk of arithmetic */
		float2 h[3][CH / 2];
		if (!skip_first) ld_chunk<WIDE_LOADS>(row, 0, h[0]);
		ld_chunk<WIDE_LOADS>(row, CH, h[1]);
#pragma unroll
		for (int c = 0; c < NCH; c++) {
			const bool need_c = c == 0 ? !skip_first : (c == NCH - 1 ? !skip_last : true);
			if (c + 2 < NCH) {
				int tie = 0;
				asm volatile("" : "+v"(tie) : "v"(ar), "v"(ai));     /* fetch of chunk c+2 may not pass chunk c-1's sum */
				const bool need_n = (c + 2 == NCH - 1) ? !skip_last : true;
				if (need_n) ld_chunk<WIDE_LOADS>(row + tie, (c + 2) * CH, h[(c + 2) % 3]);
			}
			if (need_c) fir_chunk<NW, W, CH>(win, c, h[c % 3], ar, ai);
		}
	}
	out_re = ar;
	out_im = ai;
}

/* K blind symbol-clock steps (timing.c:34): the same K rounded float additions, without the loop
 * bookkeeping (3 scalar issue slots per step in a kernel that is issue bound). */
template <int K>
__device__ __forceinline__ float
blind_steps(float p, float f)
{
#pragma unroll
	for (int k = 0; k < K; k++) p = p + f;
	return p;
}

/* ---- the kernel ------------------------------------------------------------------ */

template <int FMT, int OQPSK, bool PACKED, typename G>
__global__ void __launch_bounds__(G::BLOCK, (PACKED && G::NW <= MDEMOD_RW_3WAVES_NW) ? 3 : 2)
demod_kernel_rw(const DemodLaunch L)
{
	typedef Fmt<FMT> F;
	typedef Win<FMT, PACKED> W;
	typedef typename F::sample_t sample_t;
	constexpr int kTaps = G::KT, kBack = G::KB;
	constexpr int NW = G::NW;                    /* window slots                       */
	constexpr int SLIDE = G::SLIDE;              /* slots per slide (whole granules)   */
	constexpr int SG = SLIDE / 4;                /* granules per slide                 */
	constexpr int AMAX = NW - kTaps;             /* alignments 0..AMAX                 */
	constexpr int NST = SG * G::MAXSL;           /* granules staged ahead of the window */
	/* The float window of the std geometry leaves ~20 VGPRs: AGC and NCO state stay in registers there
	 * (5 LDS reads + 5 writes per symbol less, and no exposed LDS latency right after the FIR). */
	constexpr bool REGSTATE = MDEMOD_RW_REGSTATE && !PACKED && G::KT <= 65 && G::NW <= 80;
	/* wave priority (see the main loop): 1 = raised for the scalar stage of a firing; 2 = and for the symbol clock's add
	 * chain that follows it (std QPSK: +2 % more; OQPSK and the compact geometries: -0.3 %, so they stay at 1) */
	constexpr int PRIO = !MDEMOD_RW_SETPRIO ? 0 : (!OQPSK && !G::COMPACT) ? 2 : 1;

	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	float *ctab = reinterpret_cast<float *>(lds);
	float *lut = ctab + L.ctab_floats;
	/* per-lane state slots: field f of this lane is sl[f * 64] */
	enum { S_GAIN, S_BIAS_RE, S_BIAS_IM, S_PHASE, S_FREQ, S_ERR, S_FLAGS, S_TPREV, S_INPHASE, S_EVCALL, S_FIRSTLOCK, S_LASTV, S_COUNT };
	static_assert(S_COUNT == MDEMOD_RW_STATE_SLOTS, "host LDS sizing");
	float *sl = lut + 32 + (threadIdx.x >> 6) * (S_COUNT * 64) + (threadIdx.x & 63);
	int *sli = reinterpret_cast<int *>(sl);
	/* soft-symbol staging: 4 groups of 8 symbols per lane, [group][thread] x 16 B (conflict free), written out as one 64-byte run */
	uint4 *stage = reinterpret_cast<uint4 *>(lut + 32 + (G::BLOCK / 64) * (S_COUNT * 64)) + threadIdx.x;

	const DemodConsts &C = L.c;
	const uint32_t stream = blockIdx.x * blockDim.x + threadIdx.x;
	const bool valid = stream < L.n_streams;

	for (uint32_t i = threadIdx.x; i < L.ctab_floats; i += blockDim.x) ctab[i] = L.ctab[i];
	if (threadIdx.x < 32) lut[threadIdx.x] = L.tanh_lut[threadIdx.x];

	/* ---- per-stream geometry ---- */
	int n = 0;
	const sample_t *src = nullptr;
	if (valid) {
		n = (int)(L.n_samples_arr ? L.n_samples_arr[stream] : L.n_samples);
		const uint64_t off = L.iq_offset ? L.iq_offset[stream] : (uint64_t)stream * L.iq_stride;
		src = reinterpret_cast<const sample_t *>(L.iq) + off;
	}
	const int v_end = kBack + n;                 /* virtual stream = 64 history samples ++ block */
	const int interp = C.interp;

	/* ---- state: symbol-clock variables in registers, everything else in the LDS slots ---- */
	float t_phase = 0.0f, t_freq = C.t_center;
	int dual_state = 1;
	float r_gain = 1.0f, r_bias_re = 0.0f, r_bias_im = 0.0f, r_phase = 0.0f, r_freq = 0.0f;   /* REGSTATE only */
	{
		float gain = 1.0f, bias_re = 0.0f, bias_im = 0.0f, phase = 0.0f, freq = 0.0f, err = 1000.0f, t_prev = 0.0f, inph = 0.0f;
		int fl = MDEMOD_FLAG_UPDOWN_POS | (1 << MDEMOD_FLAG_DUAL_SHIFT);
		if (valid) {
			gain = L.st.agc_gain[stream]; bias_re = L.st.agc_bias_re[stream]; bias_im = L.st.agc_bias_im[stream];
			phase = L.st.pll_phase[stream]; freq = L.st.pll_freq[stream]; err = L.st.pll_err[stream];
			fl = L.st.flags[stream];
			t_phase = L.st.t_phase[stream]; t_freq = L.st.t_freq[stream]; t_prev = L.st.t_prev[stream];
			inph = L.st.inphase[stream];
		}
		dual_state = (fl >> MDEMOD_FLAG_DUAL_SHIFT) & 3;
		sl[S_GAIN * 64] = gain; sl[S_BIAS_RE * 64] = bias_re; sl[S_BIAS_IM * 64] = bias_im;
		sl[S_PHASE * 64] = phase; sl[S_FREQ * 64] = freq; sl[S_ERR * 64] = err;
		sli[S_FLAGS * 64] = fl & 7;                     /* locked | locked_once | updown>0; bit 3 = overflow */
		sl[S_TPREV * 64] = t_prev; sl[S_INPHASE * 64] = inph;
		sli[S_EVCALL * 64] = 0; sli[S_FIRSTLOCK * 64] = -1; sli[S_LASTV * 64] = -1;
		if (REGSTATE) { r_gain = gain; r_bias_re = bias_re; r_bias_im = bias_im; r_phase = phase; r_freq = freq; }
	}

	/* ---- window: slots 0..63 = history, 64..79 = first four granules of the block ---- */
	typename W::elem_t win[NW];
	{
		/* v2 history layout is [stream][64] float2: one base address + immediate offsets */
		const float2 *hist = reinterpret_cast<const float2 *>(L.st.hist) + (size_t)(valid ? stream : 0) * kBack;
#pragma unroll
		for (int k = 0; k < kBack; k++) {
			float2 h = hist[k];
			if (!valid) h = make_float2(0.0f, 0.0f);
			win[k] = W::from_float(h);
		}
#pragma unroll
		for (int g = 0; g < (NW - kBack) / 4; g++) {
			const Gran<FMT> gr = fetch_granule<FMT>(src, 4 * g, n);
#pragma unroll
			for (int u = 0; u < 4; u++) win[kBack + 4 * g + u] = W::pack(gr.s[u]);
		}
	}
	/* the two granules that enter at the next slide are fetched right after the previous one */
	int g_load = (NW - kBack) / 4;                         /* next block granule to fetch (wave-uniform) */
	Gran<FMT> stg[NST];
#pragma unroll
	for (int i = 0; i < NST; i++) stg[i] = fetch_granule<FMT>(src, 4 * (g_load + i), n);
	g_load += NST;

	__syncthreads();                                       /* coefficient rows + LUT visible */

	int base = 0;                                          /* virtual index of logical slot 0 */
	int v_cur = kBack - 1;
	int isub = 0, fire_sub = 0;
	bool fired = false;
	bool done = !valid || n == 0;
	uint32_t sym_call = 0;
	uint32_t ob0 = 0, ob1 = 0, ob2 = 0, ob3 = 0;           /* 8 buffered soft symbols          */
	const int k_safe = C.step_safe;
	const float f_hi = C.step_fmax;
	const uint32_t magic = C.interp_magic;
	const int steps_need = (k_safe + 4 + C.interp - 1) / C.interp;      /* samples that hold k_safe + 4 steps */

	/* Watchdog: every iteration of a wave emits a firing for some lane, slides, or retires a lane, so a wave needs at
	 * most a few iterations per interpolated step of its longest stream.  A bug must not be able to hang the GPU. */
	int n_wave_max = n;
	for (int o = 32; o > 0; o >>= 1) {
		const int other = __shfl_xor(n_wave_max, o);
		n_wave_max = other > n_wave_max ? other : n_wave_max;
	}
	n_wave_max = __builtin_amdgcn_readfirstlane(n_wave_max);
	const uint64_t guard64 = 4ull * (uint64_t)(n_wave_max + kBack) * (uint64_t)interp + 4096ull;
	uint32_t guard = guard64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)guard64;
	do {
		/* ---- (1) symbol clock: timing.c:32-57 ---- */
		if (!fired && !done) {
			const float thr = OQPSK ? (float)dual_state * MD_PI_F : MD_TWO_PI_F;
			/* enough input left for k_safe + 4 steps: (v_end - 1 - v_cur) * interp >= k_safe + 4 (the part of the
			 * current sample that is still to be stepped is ignored: conservative) */
			const bool fast = (t_phase < thr - (float)k_safe * f_hi - 1e-3f) && (v_cur + steps_need < v_end);
			if (fast) {
				float p = t_phase;
				/* cannot reach thr: no compare needed */
				if (k_safe == 14) p = blind_steps<14>(p, t_freq);             /* QPSK 72k @ 230 kS/s, -O 5 */
				else if (k_safe == 6) p = blind_steps<6>(p, t_freq);          /* OQPSK 80k @ 230 kS/s */
				else {                                                         /* e.g. 109 steps at 1 MS/s, -O 8 */
					int k = k_safe;
					for (; k >= 16; k -= 16) p = blind_steps<16>(p, t_freq);
					if (k & 8) p = blind_steps<8>(p, t_freq);
					if (k & 4) p = blind_steps<4>(p, t_freq);
					if (k & 2) p = blind_steps<2>(p, t_freq);
					if (k & 1) p = p + t_freq;
				}
				/* four checked steps.  The increment is positive, so "reached thr" is monotone:
				 * the first hit is after (number of misses) + 1 steps. */
				const float p1 = p + t_freq, p2 = p1 + t_freq, p3 = p2 + t_freq, p4 = p3 + t_freq;
				const bool c1 = p1 >= thr, c2 = p2 >= thr, c3 = p3 >= thr, c4 = p4 >= thr;
				const int m = k_safe + 1 + (c1 ? 0 : 1) + (c2 ? 0 : 1) + (c3 ? 0 : 1);
				float ph = c3 ? p3 : p4;
				ph = c2 ? p2 : ph;
				ph = c1 ? p1 : ph;
				t_phase = ph;
				const uint32_t w = (uint32_t)(isub + m);
				const uint32_t q = (interp == 1) ? w : __umulhi(w, magic);     /* floor(w / interp); the magic of 1 does not fit 32 bits */
				const int isub_new = (int)(w - q * (uint32_t)interp);
				v_cur += (int)q + (isub_new > 0 ? 1 : 0) - (isub > 0 ? 1 : 0);   /* samples pushed: ceil(w/interp) - (isub>0) */
				fire_sub = (isub_new == 0) ? interp - 1 : isub_new - 1;
				isub = isub_new;
				fired = c4;
			}
			while (!fired && !done) {                                   /* generic path */
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
		if (PRIO == 2) __builtin_amdgcn_s_setprio(0);

		/* ---- (2) slide the window by 8 slots when nobody needs slots 0..7 any more ---- */
#pragma unroll
		for (int r = 0; r < G::MAXSL; r++) {
			const int a_now = v_cur - kBack - base;
			if (md_all(done || a_now >= SLIDE)) {
#pragma unroll
				for (int k = 0; k < NW - SLIDE; k++) win[k] = win[k + SLIDE];
#pragma unroll
				for (int g = 0; g < SG; g++)
#pragma unroll
					for (int u = 0; u < 4; u++) win[NW - SLIDE + 4 * g + u] = W::pack(stg[g].s[u]);
				base += SLIDE;
#pragma unroll
				for (int i = 0; i + SG < NST; i++) stg[i] = stg[i + SG];
				/* The refill may only be issued once the old staging registers have been consumed (the empty asm
				 * ties the fetch index to the freshly packed slots): otherwise the scheduler hoists the loads above
				 * the packing, has to give them other registers, copies them into the staging registers right away
				 * and waits a full HBM latency on every slide.  One wave-uniform branch separates the common case
				 * (all granules inside the block) from the ragged end. */
				int tie = 0;
#pragma unroll
				for (int k = 0; k < SLIDE; k += 4)
					asm volatile("" : "+v"(tie) : "v"(win[NW - SLIDE + k]), "v"(win[NW - SLIDE + k + 1]),
					                              "v"(win[NW - SLIDE + k + 2]), "v"(win[NW - SLIDE + k + 3]));
				const int m_new = 4 * g_load + tie;
				if (md_all(m_new + 4 * SG - 1 < n)) {
#pragma unroll
					for (int g = 0; g < SG; g++) __builtin_memcpy(&stg[NST - SG + g], src + m_new + 4 * g, sizeof(Gran<FMT>));
				} else {
#pragma unroll
					for (int g = 0; g < SG; g++) stg[NST - SG + g] = fetch_granule<FMT>(src, m_new + 4 * g, n);
				}
				g_load += SG;
			}
		}

		/* ---- (3) process the firing if its 65 samples are inside the window ---- */
		const int a = v_cur - kBack - base;
		if (fired && a <= AMAX) {
			fired = false;
			const int bank = interp - 1 - fire_sub;                     /* filter.c:52 */
			const float *row;
			if (G::COMPACT) {
				const int o = AMAX - a;                                 /* offset into the padded bank */
				row = ctab + __mul24(bank * 2 + (o & 1), C.ctab_row_stride) + (o & ~1);
			} else {
				row = ctab + __mul24(__mul24(a, interp) + bank, C.ctab_row_stride);     /* small numbers: 24-bit multiplies */
			}
			cf32 y;
			/* slots 0..7 carry only zeros for a lane with a >= 8, the last 8 slots only zeros for
			 * a <= AMAX - 8: when the whole wave agrees the chunk is dropped (exact: acc + 0*x == acc). */
			const bool skip_first = md_all(a >= 8);              /* over the lanes active in this branch */
			const bool skip_last = md_all(a <= AMAX - 8);
			fir_window<NW, W, !G::COMPACT, (MDEMOD_RW_PREFETCH == 2 && !PACKED && !G::COMPACT && FMT != 32) ? 2 : 1>(win, row, skip_first, skip_last, y.re, y.im);
			/* Two waves share a SIMD.  The FIR is ~290 independent VALU instructions, the scalar stage a chain of short
			 * dependent ones (AGC -> NCO -> mix -> loops): when the arbiter interleaves them evenly, the chain waits behind
			 * FIR instructions it does not depend on.  Raising the priority of the wave that is in its scalar stage lets the
			 * chain issue as soon as its operands are ready while the other wave's FIR fills every other slot: +3.2 % on
			 * configs[1], +4.6 % on OQPSK, +1 % on the wide geometry (measured; the FIR at raised priority instead: +1 %;
			 * everything but the FIR raised: +1.4 %; level 3 instead of 2: same).  With PRIO == 2 the priority stays up
			 * through the symbol clock (another dependent chain: 18 float adds) and drops before the slide: 204.5 -> 208.5
			 * GS/s on configs[1]. */
			if (PRIO) __builtin_amdgcn_s_setprio(2);

			/* ---- scalar part: state comes from / goes back to the LDS slots ---- */
			if (REGSTATE) {
				y = md_agc(y, r_gain, r_bias_re, r_bias_im);
			} else {
				float gain = sl[S_GAIN * 64], bias_re = sl[S_BIAS_RE * 64], bias_im = sl[S_BIAS_IM * 64];
				y = md_agc(y, gain, bias_re, bias_im);
				sl[S_GAIN * 64] = gain; sl[S_BIAS_RE * 64] = bias_re; sl[S_BIAS_IM * 64] = bias_im;
			}
			int fl = sli[S_FLAGS * 64];
			PllState pll;
			if (REGSTATE) { pll.phase = r_phase; pll.freq = r_freq; }
			else { pll.phase = sl[S_PHASE * 64]; pll.freq = sl[S_FREQ * 64]; }
			pll.err = sl[S_ERR * 64];
			pll.locked = fl & 1; pll.locked_once = (fl >> 1) & 1; pll.updown = (fl & 4) ? 1 : -1;

			const float sn = md_fast_sin(-pll.phase);
			const float cs = md_fast_cos(-pll.phase);
			bool emit = true;
			float out_re, out_im;
			if (OQPSK) {
				float inphase = sl[S_INPHASE * 64];
				if (dual_state == 1) { inphase = y.re * cs - y.im * sn; emit = false; sl[S_INPHASE * 64] = inphase; }   /* demod.c:66-71 */
				out_re = inphase;
				out_im = y.re * sn + y.im * cs;                                          /* demod.c:76    */
				dual_state = (dual_state % 2) + 1;                                       /* timing.c:52   */
			} else {
				out_re = y.re * cs - y.im * sn;
				out_im = y.re * sn + y.im * cs;
			}
			md_nco_advance(pll.phase, pll.freq);

			if (emit) {
				/* The reference's per-sample loop keeps only the LAST symbol fired inside one input sample (demod.c:33-47,
				 * 62-90: `*sample` and `ret` are overwritten).  It happens when a huge timing error term pulls the clock
				 * back over the threshold at once (full-scale input before the AGC has settled): the symbol emitted for this
				 * sample a moment ago is then replaced, index and all.  The sample index of the last emitted symbol lives in
				 * an LDS state word (deriving "no sample consumed since" from the symbol clock instead measured 2 % slower). */
				const bool again = (v_cur == sli[S_LASTV * 64]);
				sli[S_LASTV * 64] = v_cur;
				const bool any_again = md_any(again);
				if (any_again) { if (again) sym_call--; }
				float t_prev = sl[S_TPREV * 64];
				md_timing_update(t_phase, t_freq, t_prev, C.t_alpha, C.t_beta, C.t_center, C.t_maxdev, out_im);
				sl[S_TPREV * 64] = t_prev;
				int first = 0;
				const int changed = md_pll_update(pll, lut, C.pll_alpha, C.pll_beta, C.pll_fmax, out_re, out_im, first);
				if (md_any(changed)) {                      /* rare: one wave-uniform test on the hot path */
				if (first) sli[S_FIRSTLOCK * 64] = (int)sym_call;
				if (changed) {
					const int ev_call = sli[S_EVCALL * 64];
					if (ev_call < MDEMOD_MAX_LOCK_EVENTS) {
						mdemod_lock_event ev;
						ev.symbol = L.st.n_symbols[stream] + sym_call; ev.locked = pll.locked; ev.pad = 0;
						L.st.events[(size_t)stream * MDEMOD_MAX_LOCK_EVENTS + ev_call] = ev;
					}
					sli[S_EVCALL * 64] = ev_call + 1;
				}
				}
				const uint32_t sym = (uint32_t)(md_quantise(out_re) & 0xFF) | ((uint32_t)(md_quantise(out_im) & 0xFF) << 8);
				if (MDEMOD_RW_STAGE == 2) {
					/* symbol k of the stream lives at ring slot k & 31: group (k >> 3) & 3 of 16 bytes, [group][thread] so that the
					 * 16-byte reads of the flush are conflict free.  A replaced symbol (sym_call was decremented above) lands on
					 * its predecessor's slot; if that one had completed a run, the run is simply written again. */
					const uint32_t k = sym_call & 31u;
					reinterpret_cast<uint16_t *>(stage + (k >> 3) * G::BLOCK)[k & 7u] = (uint16_t)sym;
					sym_call++;
					if ((sym_call & 31u) == 0) {
						int8_t *soft_out = L.soft + (size_t)stream * L.soft_stride * 2;
						if (sym_call <= L.soft_cap) {
							const uint4 a = stage[0], b = stage[G::BLOCK], c = stage[2 * G::BLOCK], d = stage[3 * G::BLOCK];
							int8_t *dst = soft_out + 2 * (size_t)(sym_call - 32);       /* memcpy: buffer and pitch need not be 16-byte aligned */
							__builtin_memcpy(dst, &a, 16); __builtin_memcpy(dst + 16, &b, 16); __builtin_memcpy(dst + 32, &c, 16); __builtin_memcpy(dst + 48, &d, 16);
						} else {
							fl |= 8;
							for (uint32_t i = 0; i < 32 && sym_call - 32 + i < L.soft_cap; i++)
								*reinterpret_cast<uint16_t *>(soft_out + 2 * (size_t)(sym_call - 32 + i)) =
								    reinterpret_cast<const uint16_t *>(stage + (i >> 3) * G::BLOCK)[i & 7u];
						}
					}
				} else {
				if (any_again && again) {
					ob3 = (ob3 & 0xFFFFu) | (sym << 16);                    /* replace the newest buffered symbol */
				} else {
					ob0 = __builtin_amdgcn_alignbit(ob1, ob0, 16);
					ob1 = __builtin_amdgcn_alignbit(ob2, ob1, 16);
					ob2 = __builtin_amdgcn_alignbit(ob3, ob2, 16);
					ob3 = (ob3 >> 16) | (sym << 16);
				}
				sym_call++;
				if ((sym_call & 7u) == 0) {
					const uint32_t sb = sym_call - 8;
					int8_t *soft_out = L.soft + (size_t)stream * L.soft_stride * 2;
					if (MDEMOD_RW_STAGE == 1 && sym_call <= L.soft_cap) {
						/* A 16-byte store per lane every 8 symbols is a quarter of a 64-byte request and the line has left the L2 by
						 * the time the lane comes back to it (measured: 2.2 bytes written to HBM per byte of soft symbols).  Three
						 * groups wait in LDS, the fourth goes out with them: four back-to-back stores, one full 64-byte run. */
						uint4 v; v.x = ob0; v.y = ob1; v.z = ob2; v.w = ob3;
						const uint32_t g = (sb >> 3) & 3u;
						if (g != 3u) {
							stage[g * G::BLOCK] = v;
						} else {
							const uint4 a = stage[0], b = stage[G::BLOCK], c = stage[2 * G::BLOCK];
							uint4 *dst = reinterpret_cast<uint4 *>(soft_out + 2 * (size_t)(sb - 24));
							dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = v;
						}
					} else if (sym_call <= L.soft_cap) {
						uint4 v; v.x = ob0; v.y = ob1; v.z = ob2; v.w = ob3;
						__builtin_memcpy(soft_out + 2 * (size_t)sb, &v, 16);
					} else {
						fl |= 8;
						const uint32_t w4[4] = { ob0, ob1, ob2, ob3 };
						for (uint32_t i = 0; i < 8 && sb + i < L.soft_cap; i++)
							*reinterpret_cast<uint16_t *>(soft_out + 2 * (size_t)(sb + i)) =
							    (uint16_t)(w4[i >> 1] >> ((i & 1) * 16));
					}
				}
				}
			}
			if (REGSTATE) { r_phase = pll.phase; r_freq = pll.freq; }
			else { sl[S_PHASE * 64] = pll.phase; sl[S_FREQ * 64] = pll.freq; }
			sl[S_ERR * 64] = pll.err;
			sli[S_FLAGS * 64] = (fl & 8) | (pll.locked ? 1 : 0) | (pll.locked_once ? 2 : 0) | (pll.updown > 0 ? 4 : 0);
			if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
		}
	} while (--guard);                                               /* the watchdog is the loop's latch */

	if (guard == 0) sli[S_FLAGS * 64] |= 8;                          /* watchdog fired: reported as overflow */

	/* Epilogue addresses are recomputed from an opaque copy of the stream index: otherwise the
	 * compiler keeps ~30 VGPRs of prologue addresses alive across the main loop. */
	uint32_t stream_e = stream;
	asm volatile("" : "+v"(stream_e));
	int8_t *soft_e = L.soft + (size_t)stream_e * L.soft_stride * 2;
	int overflow = (sli[S_FLAGS * 64] >> 3) & 1;

	/* ---- flush the staged groups, then the partial group of soft symbols ---- */
	if (valid && MDEMOD_RW_STAGE == 2) {
		const uint32_t r = sym_call & 31u, sb = sym_call - r;              /* symbols still in the ring: complete groups as 16 bytes, the rest singly */
		for (uint32_t g = 0; g < (r >> 3); g++) {
			if (sb + 8 * g + 8 <= L.soft_cap) { const uint4 qv = stage[g * G::BLOCK]; __builtin_memcpy(soft_e + 2 * (size_t)(sb + 8 * g), &qv, 16); }
			else for (uint32_t i = 0; i < 8; i++) {
				if (sb + 8 * g + i < L.soft_cap) *reinterpret_cast<uint16_t *>(soft_e + 2 * (size_t)(sb + 8 * g + i)) = reinterpret_cast<const uint16_t *>(stage + g * G::BLOCK)[i];
				else overflow = 1;
			}
		}
		for (uint32_t i = r & ~7u; i < r; i++) {
			if (sb + i < L.soft_cap) *reinterpret_cast<uint16_t *>(soft_e + 2 * (size_t)(sb + i)) = reinterpret_cast<const uint16_t *>(stage + (i >> 3) * G::BLOCK)[i & 7u];
			else overflow = 1;
		}
	}
	if (valid && MDEMOD_RW_STAGE == 1) {
		const uint32_t full = sym_call >> 3;                               /* complete groups of 8 so far */
		const uint32_t pending = full & 3u;                                /* ... of which this many still sit in LDS */
		for (uint32_t g = 0; g < pending; g++) {
			const uint32_t sb = (full - pending + g) * 8;
			if (sb + 8 <= L.soft_cap) *reinterpret_cast<uint4 *>(soft_e + 2 * (size_t)sb) = stage[g * G::BLOCK];
			else {
				const uint4 v = stage[g * G::BLOCK];
				const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
				for (uint32_t i = 0; i < 8; i++) {
					if (sb + i < L.soft_cap) *reinterpret_cast<uint16_t *>(soft_e + 2 * (size_t)(sb + i)) = (uint16_t)(w4[i >> 1] >> ((i & 1) * 16));
					else overflow = 1;
				}
			}
		}
	}
	if (valid && MDEMOD_RW_STAGE != 2) {
		const uint32_t r = sym_call & 7u;
		const uint32_t sb = sym_call - r;
		const uint32_t w4[4] = { ob0, ob1, ob2, ob3 };
		for (uint32_t i = 0; i < r; i++) {
			const uint32_t hw = 8 - r + i;
			if (sb + i < L.soft_cap)
				*reinterpret_cast<uint16_t *>(soft_e + 2 * (size_t)(sb + i)) = (uint16_t)(w4[hw >> 1] >> ((hw & 1) * 16));
			else
				overflow = 1;
		}
	}

	/* ---- store state ---- */
	if (REGSTATE) {
		sl[S_GAIN * 64] = r_gain; sl[S_BIAS_RE * 64] = r_bias_re; sl[S_BIAS_IM * 64] = r_bias_im;
		sl[S_PHASE * 64] = r_phase; sl[S_FREQ * 64] = r_freq;
	}
	if (valid) {
		const int fl = sli[S_FLAGS * 64];
		L.st.agc_gain[stream_e] = sl[S_GAIN * 64]; L.st.agc_bias_re[stream_e] = sl[S_BIAS_RE * 64]; L.st.agc_bias_im[stream_e] = sl[S_BIAS_IM * 64];
		L.st.pll_phase[stream_e] = sl[S_PHASE * 64]; L.st.pll_freq[stream_e] = sl[S_FREQ * 64]; L.st.pll_err[stream_e] = sl[S_ERR * 64];
		L.st.flags[stream_e] = (fl & 7) | (dual_state << MDEMOD_FLAG_DUAL_SHIFT);
		L.st.t_phase[stream_e] = t_phase; L.st.t_freq[stream_e] = t_freq; L.st.t_prev[stream_e] = sl[S_TPREV * 64];
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
		float2 *hist = reinterpret_cast<float2 *>(L.st.hist) + (size_t)stream_e * kBack;   /* [stream][64] */
		for (int k = 0; k < kBack; k++) {
			const int idx = n + k;
			float2 h;
			if (idx < kBack) h = hist[idx];
			else { const cf32 s = F::decode(src[idx - kBack]); h = make_float2(s.re, s.im); }
			hist[k] = h;
		}
	}
}

template <int FMT, int OQPSK, bool PACKED, typename G>
hipError_t
launch_rw(const DemodLaunch &L, size_t lds_bytes, hipStream_t stream)
{
	const int block = G::BLOCK;
	const uint32_t blocks = (L.n_streams + block - 1) / block;
	auto kfn = demod_kernel_rw<FMT, OQPSK, PACKED, G>;
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(kfn, dim3(blocks), dim3(block), lds_bytes, stream, L);
	return hipGetLastError();
}

#if MDEMOD_RW_PART != 2
template <int FMT>
hipError_t
launch_rw_mode(const DemodLaunch &L, bool packed, size_t lds_bytes, hipStream_t stream)
{
	if constexpr (FMT != 32) {              /* float input has no packed form */
		if (packed)
			return L.c.oqpsk ? launch_rw<FMT, 1, true, GeoStd>(L, lds_bytes, stream) : launch_rw<FMT, 0, true, GeoStd>(L, lds_bytes, stream);
	}
	return L.c.oqpsk ? launch_rw<FMT, 1, false, GeoStd>(L, lds_bytes, stream) : launch_rw<FMT, 0, false, GeoStd>(L, lds_bytes, stream);
}
#endif

#if MDEMOD_RW_PART != 1
template <int FMT>
hipError_t
launch_rw_wide(const DemodLaunch &L, int mid, size_t lds_bytes, hipStream_t stream)
{
	if constexpr (FMT == 32) {
		/* float input has no packed form: its window is converted-float pairs (2 VGPRs per slot), which fits the 96 slots of the
		 * mid geometry (65 taps at up to 15 samples per firing) and nothing larger */
		if (mid != 1) return hipErrorInvalidValue;
		return L.c.oqpsk ? launch_rw<32, 1, false, GeoMid>(L, lds_bytes, stream) : launch_rw<32, 0, false, GeoMid>(L, lds_bytes, stream);
	} else {
	if (mid == 1) return L.c.oqpsk ? launch_rw<FMT, 1, true, GeoMid>(L, lds_bytes, stream) : launch_rw<FMT, 0, true, GeoMid>(L, lds_bytes, stream);
	if (mid == 2) return L.c.oqpsk ? launch_rw<FMT, 1, true, GeoFar>(L, lds_bytes, stream) : launch_rw<FMT, 0, true, GeoFar>(L, lds_bytes, stream);
	return L.c.oqpsk ? launch_rw<FMT, 1, true, GeoWide>(L, lds_bytes, stream) : launch_rw<FMT, 0, true, GeoWide>(L, lds_bytes, stream);
	}
}
#endif

} /* namespace */

/* The file is compiled twice (build.py): part 1 = std geometry with the max-ILP machine scheduler, part 2 = wide geometry
 * with the default one (max-ILP makes the u8 wide kernels spill inside their main loop). */
#if MDEMOD_RW_PART != 2
hipError_t
mdemod_launch_demod_rw_std(const DemodLaunch &L, int fmt, int packed, size_t lds_bytes, hipStream_t stream)
{
	switch (fmt) {
	case 16: return launch_rw_mode<16>(L, packed != 0, lds_bytes, stream);
	case 8:  return launch_rw_mode<8>(L, packed != 0, lds_bytes, stream);
	case 32: return launch_rw_mode<32>(L, false, lds_bytes, stream);
	default: return hipErrorInvalidValue;
	}
}
#endif

#if MDEMOD_RW_PART != 1
hipError_t
mdemod_launch_demod_rw_wide(const DemodLaunch &L, int fmt, int mid, size_t lds_bytes, hipStream_t stream)
{
	switch (fmt) {
	case 16: return launch_rw_wide<16>(L, mid, lds_bytes, stream);
	case 8:  return launch_rw_wide<8>(L, mid, lds_bytes, stream);
	case 32: return launch_rw_wide<32>(L, mid, lds_bytes, stream);
	default: return hipErrorInvalidValue;
	}
}
#endif
