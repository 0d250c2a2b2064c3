/*
 * demod_device.h — scalar stages of the demodulator as gfx950 device functions.
 *
 * Every function reproduces, operation by operation and rounding by rounding,
 * what the reference computes (reference file:line cited at each function).
 * The file must be compiled with -ffp-contract=off: the reference's output is
 * only defined without FMA contraction.  Temporaries are typed
 * explicitly wherever the reference's C expression mixes float and double.
 */
#ifndef MDEMOD_DEVICE_H
#define MDEMOD_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#define MD_TWO_PI_D 6.283185307179586476925286766559   /* 2*M_PI  (pll.c:61,113; timing.c:80) */
#define MD_HALF_PI_D 1.5707963267948966192313216916398 /* M_PI/2  (sincos.c:39)                */
#define MD_TWO_PI_F 6.28318548202514648437500f         /* 2*(float)M_PI (timing.c:37)          */
#define MD_PI_F     3.14159274101257324218750f         /* (float)M_PI   (timing.c:50)          */

struct cf32 { float re, im; };

/* Wave votes taken on the lane mask itself.  HIP's __all / __any go through an integer per lane (v_cndmask_b32 0/1 + v_cmp_ne_u32:
 * two VALU instructions of 4.4 SIMD-cycles each per vote, and the loop body votes nine times per firing); the ballot of a
 * comparison is the comparison's own mask.  Same semantics: over the active lanes. */
__device__ __forceinline__ bool md_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ bool md_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }

/* MAX(-b, MIN(b, x)) of the reference's macros (utils.h: `(a) < (b) ? (a) : (b)`) for a bound b > 0, as ONE v_med3_f32 instead of
 * two comparisons and two selects (each of the four a 4.4-cycle VALU instruction on gfx950: tools/ubench/valu_mix.hip).  Equal for
 * every finite and infinite x including +-0 (the median of {-b, x, b} is x itself whenever -b <= x <= b, else the bound the
 * macros pick).  A NaN would come out as -b where the macros keep it - but no NaN reaches these points in a run the reference
 * defines: a non-finite sample sends the reference's own tanh look-up out of bounds one symbol later (pll.c:154-159, see
 * md_tanh_lut), and the coefficient tables with a NaN tap are refused. */
__device__ __forceinline__ float md_clamp_sym(float x, float b) { return __builtin_amdgcn_fmed3f(x, -b, b); }

/* sincos.c:24: x = fx * 0x10000 / (2*M_PI) narrowed to int16.  float*int -> float, the
 * division is double, and the double->int16 narrowing is what x86 compilers emit: cvttsd2si
 * to int32, keep the low 16 bits.  Reference form, with the real division. */
__device__ __forceinline__ int32_t
md_turn_code_div(float fx)
{
	return __double2int_rz((double)(fx * 65536.0f) / MD_TWO_PI_D);
}

/* The same integer with ONE double multiplication (a correctly rounded f64 divide is ~35 instructions, and there are two
 * per symbol).  The quotient RN(x / 2pi) and the product RN(x * RN(1/2pi)) are two roundings of almost the same real
 * number, so their truncations differ only if an integer lies between them - and for the 2.2e9 floats with |fx| < 16
 * none does: tools/proofs/verify_turncode_mul.cpp enumerates them all (0 mismatches, also for the constant 1..3 ulp
 * either side: there is margin), the CPU test-suite runs it, and the device self-test mdemod_selftest_turncode()
 * repeats the enumeration on the GPU against the real division.  Scaling by 65536 is exact, so it is folded into the
 * constant.  (Until round 3 this was n0 = trunc(|x| * RN(1/2pi)) plus an exact fma residual and one correction: 7 f64-rate
 * instructions instead of 3; the enumeration shows the correction never fires.)  The PLL produces |fx| < 8.9. */
#define MD_TURNS_PER_RAD_X65536 (65536.0 * (1.0 / MD_TWO_PI_D))
template <bool CHECKED = true>
__device__ __forceinline__ int32_t
md_turn_code(float fx)
{
	int32_t n = __double2int_rz((double)fx * MD_TURNS_PER_RAD_X65536);
	/* outside the proven range: the real division (one wave-uniform test instead of a divergent branch per call; never
	 * taken in practice).  CHECKED = false drops the test: for kernels that only run when the host has established
	 * |fx| < 16 (pll_fmax < 6: the phase leaves pll.c:113's fmod inside (-2pi, 2pi) and pll.c:60 adds at most fmax;
	 * mdemod_set_state refuses anything else; a NaN phase gives 0 on both paths, as cvttsd2si's 0x80000000 does in the
	 * reference). */
	if (CHECKED) {
		const bool out_of_range = !(fabsf(fx) < 16.0f);
		if (__builtin_expect(md_any(out_of_range), 0)) {
			if (out_of_range) n = md_turn_code_div(fx);
		}
	}
	return n;
}

/* dsp/sincos.c:13-34 — Q14 parabola on a 16-bit turn code. */
__device__ __forceinline__ float
md_sin_from_code(int32_t wide)
{
	const int32_t sign = (int32_t)(int16_t)(wide & 0xFFFF);
	int32_t x = (wide & 0x7FFF) - 16384;                    /* sincos.c:26-27 */
	const int32_t x2 = (x * x) >> 14;                       /* sincos.c:29    */
	int32_t y = 19900 - ((x2 * 3516) >> 14);                /* sincos.c:31    */
	y = 16384 - ((x2 * y) >> 14);                           /* sincos.c:32    */
	return (float)(sign < 0 ? -y : y) * (1.0f / 16384.0f);  /* sincos.c:34 (exact: power of two) */
}

/* The same value from a table in LDS.  The parabola depends on the turn code only through |x| = |(code & 0x7FFF) - 16384|
 * (sincos.c:26-32 squares x first) and bit 15 (the sign, sincos.c:34): 16 385 entries replace the integer arithmetic above
 * (md_sin_lut_fill writes them with that very arithmetic; tests sweep all 65 536 codes through both).  T = float: the entry is
 * y / 16384 (exact); the negation is 0 - y, not a sign flip: sincos.c:34 converts the INTEGER -y, so a zero stays +0.
 * T = int16_t: the entry is y, negated as an integer and converted like the reference does. */
template <typename T>
__device__ __forceinline__ void
md_sin_lut_fill(T *tab, int tid, int nthreads)
{
	for (int i = tid; i <= 16384; i += nthreads) {
		const int32_t x2 = (i * i) >> 14;
		int32_t y = 19900 - ((x2 * 3516) >> 14);
		y = 16384 - ((x2 * y) >> 14);
		tab[i] = sizeof(T) == 4 ? (T)((float)y * (1.0f / 16384.0f)) : (T)y;
	}
}

template <typename T>
__device__ __forceinline__ float
md_sin_from_code_lut(const T *tab, int32_t wide)
{
	/* |x| in one instruction (hipcc expands __usad into min, max and a subtraction) */
	uint32_t ax;
	asm("v_sad_u32 %0, %1, %2, 0" : "=v"(ax) : "v"(wide & 0x7FFF), "s"(16384));
	if (sizeof(T) == 4) {
		/* bit 15 of the code becomes the float's sign bit; adding +0 turns the -0 that gives for y == 0 back into the +0 of
		 * sincos.c:34's (float)(-0) and leaves every other value alone */
		const uint32_t yb = __float_as_uint((float)tab[ax]);
		return __uint_as_float(yb | (((uint32_t)wide << 16) & 0x80000000u)) + 0.0f;
	} else {
		const int32_t y = (int32_t)tab[ax];
		return (float)((wide & 0x8000) ? -y : y) * (1.0f / 16384.0f);
	}
}

template <bool CHECKED = true>
__device__ __forceinline__ float
md_fast_sin(float fx)
{
	return md_sin_from_code(md_turn_code<CHECKED>(fx));
}

/* dsp/sincos.c:37-40 */
template <bool CHECKED = true>
__device__ __forceinline__ float
md_fast_cos(float fx)
{
	return md_fast_sin<CHECKED>((float)((double)fx + MD_HALF_PI_D));
}

/* Correctly rounded sqrt of a double that is the sum of two squared floats: 0, or in [2^-298, 2^257], or inf/NaN.  This is
 * the device library's own sequence (v_rsq_f64 + two Goldschmidt/Newton steps with exact fma residuals) without the
 * range scaling it needs for arguments below 2^-767 (two v_ldexp_f64, a compare and two selects less); 0 and inf keep
 * their special-case select.  mdemod_selftest_hypot compares it with the host's sqrt on the device. */
__device__ __forceinline__ double
md_sqrt_sumsq(double s)
{
	const double y = __builtin_amdgcn_rsq(s);
	double g = s * y;
	double h = 0.5 * y;
	const double r = __builtin_fma(-h, g, 0.5);
	g = __builtin_fma(g, r, g);
	h = __builtin_fma(h, r, h);
	double d = __builtin_fma(-g, g, s);
	g = __builtin_fma(d, h, g);
	d = __builtin_fma(-g, g, s);
	g = __builtin_fma(d, h, g);
	return __builtin_amdgcn_class(s, 0x260) ? s : g;       /* -0, +0, +inf: sqrt(s) == s */
}

/* cabsf as glibc 2.35 computes it: one double sqrt of the exact double sum of squares, narrowed to float (agc.c:21) -
 * the reference form: a correctly rounded f64 square root (19 instructions). */
__device__ __forceinline__ float
md_cabsf_exact(float re, float im)
{
	const double s = (double)re * (double)re + (double)im * (double)im;
	return (float)md_sqrt_sumsq(s);
}

/* What the kernels call unless they ask for the short form (below). */
__device__ __forceinline__ float
md_cabsf(float re, float im)
{
	return md_cabsf_exact(re, im);
}

/* The same float with a shorter square root almost always (round 4; taken by the configs[1] instance of the std kernel, where it
 * measured +1.9 %: on configs[2] / [3] and in the latency kernel it measured -0.6 % / -0.6 % / -3 %, the exec-mask detour of the
 * fallback costing what the eight instructions less gain).  Only the FLOAT is wanted, so a square root good to 2^-47 is
 * enough unless it lands next to a float rounding boundary: g = s * rsq(s) (v_rsq_f64: 2^-24 or better), one Newton step with the
 * exact residual d = s - g^2 (fma) leaves a relative error of 1.5 eps^2 < 2^-46.  `near`: the 29 mantissa bits the float drops are
 * within 2^12 double-ulps (2^-40 of the value: 64 times the error bound) of the half-way pattern 1000...0 - there, and for every s
 * outside [2^-200, 2^200] (zero, denormal results, inf, NaN: one unsigned comparison of the exponent field), the lane takes the
 * exact path in a divergent branch: 2^-16 of the lane-firings.  Everywhere else RN24 of the approximation IS RN24(RN53(sqrt(s))):
 * neither rounding has a boundary between the approximation and the root.  mdemod_selftest_cabsf runs 2^32 pairs through both on
 * the device (0 differences; it also counts the fallbacks). */
__device__ __forceinline__ float
md_cabsf_short(float re, float im, uint32_t *fallbacks = nullptr)
{
	const double s = (double)re * (double)re + (double)im * (double)im;
	const double y = __builtin_amdgcn_rsq(s);
	const double g = s * y, h = 0.5 * y;
	const double d = __builtin_fma(-g, g, s);
	const double g1 = __builtin_fma(d, h, g);
	const uint64_t gb = __builtin_bit_cast(uint64_t, g1), sb = __builtin_bit_cast(uint64_t, s);
	const uint32_t drop = (uint32_t)gb & 0x1FFFFFFFu;                                   /* what the narrowing to float rounds away */
	const bool near = (uint32_t)(drop - (0x10000000u - 0x1000u)) < 0x2000u;
	const bool odd = (uint32_t)((uint32_t)(sb >> 32) - ((1023u - 200u) << 20)) >= (400u << 20);     /* s outside [2^-200, 2^200), or not a positive number */
	float r = (float)g1;
	if (__builtin_expect(near || odd, 0)) {
		r = (float)md_sqrt_sumsq(s);
		if (fallbacks) *fallbacks += 1;
	}
	return r;
}

/* dsp/agc.c:13-20: bias tracking and scaling (what the rest of the symbol needs) */
__device__ __forceinline__ cf32
md_agc_apply(cf32 x, float gain, float &bias_re, float &bias_im)
{
	const float keep = 1.0f - 0.001f;
	bias_re = bias_re * keep + 0.001f * x.re;
	bias_im = bias_im * keep + 0.001f * x.im;
	x.re = x.re - bias_re;
	x.im = x.im - bias_im;
	x.re = x.re * gain;
	x.im = x.im * gain;
	return x;
}

/* dsp/agc.c:21-24: the gain for the NEXT symbol from the magnitude of this one */
template <bool SHORT = false>
__device__ __forceinline__ void
md_agc_gain(cf32 scaled, float &gain)
{
	const float mag = SHORT ? md_cabsf_short(scaled.re, scaled.im) : md_cabsf(scaled.re, scaled.im);
	gain = gain + 0.0001f * (190.0f - mag);
	gain = (0.0f > gain) ? 0.0f : gain;
}

/* dsp/agc.c:13-25 */
template <bool SHORT = false>
__device__ __forceinline__ cf32
md_agc(cf32 x, float &gain, float &bias_re, float &bias_im)
{
	x = md_agc_apply(x, gain, bias_re, bias_im);
	md_agc_gain<SHORT>(x, gain);
	return x;
}

/* 2*M_PI = MD_TWO_PI_F + MD_TWO_PI_LO with MD_TWO_PI_F the float above it.  For a float x with 2pi <= |x| < 4pi,
 * (float)((double)x -+ 2*M_PI) == (x -+ MD_TWO_PI_F) -+ MD_TWO_PI_LO in FLOAT arithmetic (the first step is exact: Sterbenz):
 * tools/proofs/verify_wrap_f32.cpp enumerates all 16 777 216 such floats, in the CPU test-suite. */
#define MD_TWO_PI_LO (-0x1.777a5cp-23f)                /* (float)(2*M_PI - (double)MD_TWO_PI_F) = -1.74845553e-07 */

/* NCO phase advance: pll.c:60-61.  F32: the wrap in float arithmetic (needs phase + freq < 4pi, i.e. fmax < 2pi: the
 * v3 kernel, whose host side checks it); otherwise the reference's double subtraction. */
template <bool F32 = false>
__device__ __forceinline__ void
md_nco_advance(float &phase, float freq)
{
	phase = phase + freq;
	/* (double)phase >= 2*pi  <=>  phase >= 6.2831855f: 2*pi lies strictly between the floats 6.2831850 and 6.2831855 */
	if (F32) {
		const float w = (phase - MD_TWO_PI_F) - MD_TWO_PI_LO;
		phase = (phase >= MD_TWO_PI_F) ? w : phase;
	} else if (phase >= MD_TWO_PI_F)
		phase = (float)((double)phase - MD_TWO_PI_D);
}

/* (float)fmod(x, 2*pi) with the dividend's sign (pll.c:113) for a float x.  |x| < 2*pi needs no work; one period off
 * is exact in float arithmetic (MD_TWO_PI_LO above) and happens to some lane of a wave on most firings (a
 * carrier offset walks the phase round the circle), so it is computed branch-free; anything larger takes libm's exact
 * fmod behind a wave-uniform test that is practically never taken (|alpha * e| would have to exceed 2*pi). */
#define MD_FOUR_PI_F 12.56637096405029296875f          /* the float just above 4*pi: |x| < this  <=>  (double)|x| < 4*pi */
__device__ __forceinline__ float
md_wrap_2pi(float xf)
{
	/* |x| < 2*pi (double)  <=>  |xf| < 6.2831855f for a float argument (see md_nco_advance) */
	const bool wraps = !(fabsf(xf) < MD_TWO_PI_F);
	/* one period, in float arithmetic (exact for 2pi <= |x| < 4pi, see MD_TWO_PI_LO) */
	const float once = (xf < 0.0f) ? (xf + MD_TWO_PI_F) + MD_TWO_PI_LO : (xf - MD_TWO_PI_F) - MD_TWO_PI_LO;
	float r = wraps ? once : xf;
	const bool far = !(fabsf(xf) < MD_FOUR_PI_F);              /* also NaN and inf: fmod's business */
	if (__builtin_expect(md_any(far), 0)) {
		if (far) r = (float)fmod((double)xf, MD_TWO_PI_D);
	}
	return r;
}

/* pll.c:154-159 */
/* Branch free: v > 15 reads lut[31] = (float)tanh(15) and v < -16 reads lut[0] = (float)tanh(-16), which ARE
 * 1.0f and -1.0f (|tanh| differs from 1 by 2e-13, far below half a float ulp); mdemod_create checks it. */
__device__ __forceinline__ float
md_tanh_lut(const float *lut, float v)
{
	const float c = __builtin_amdgcn_fmed3f(v, -16.0f, 15.0f);
	return lut[(int)c + 16];
}

/* The same for a value every lane of the wave shares (latency kernel): a locked constellation sits around +-134, so the clamp
 * decides almost every look-up (lut[31] = 1, lut[0] = -1, checked by mdemod_create) and the LDS round trip is only paid for
 * the rare small value. */
__device__ __forceinline__ float
md_tanh_lut_uniform(const float *lut, float v)
{
	float t = (v >= 15.0f) ? 1.0f : -1.0f;
	/* (the look-up is the RARE way: laid out of line, so that the common way is a branch not taken - a lone wave waits ~90 cycles for
	 * its instruction buffer after every branch it takes: r05, SQ_WAIT_ANY of the latency kernel) */
	if (__builtin_expect(md_any(v > -16.0f && v < 15.0f), 0)) t = md_tanh_lut(lut, v);
	return t;
}

/* ... and for the two rails of one symbol behind ONE test (pll.c:143-151 looks both up) */
__device__ __forceinline__ void
md_tanh_lut_uniform2(const float *lut, float a, float b, float &ta, float &tb)
{
	ta = (a >= 15.0f) ? 1.0f : -1.0f;
	tb = (b >= 15.0f) ? 1.0f : -1.0f;
	if (__builtin_expect(md_any((a > -16.0f && a < 15.0f) || (b > -16.0f && b < 15.0f)), 0)) {
		ta = md_tanh_lut(lut, a);
		tb = md_tanh_lut(lut, b);
	}
}

struct PllState {
	float phase, freq, err;
	int   locked, locked_once, updown;
};

/* pll.c:100-130,143-151.  Returns 1 when `locked` changed. */
template <bool UNIFORM = false>
__device__ __forceinline__ int
md_pll_update(PllState &p, const float *lut, float alpha, float beta, float fmax,
              float i, float q, int &just_locked_first)
{
	const float e = UNIFORM ? md_tanh_lut_uniform(lut, i) * q - md_tanh_lut_uniform(lut, q) * i
	                        : md_tanh_lut(lut, i) * q - md_tanh_lut(lut, q) * i;

	const float ph = p.phase + alpha * e;
	p.phase = md_wrap_2pi(ph);
	p.freq = p.freq + beta * e;

	const float decayed = p.err * (1.0f - 0.001f);
	p.err = (float)((double)decayed + fabs((double)e) * (double)0.001f);

	/* pll.c:117-123 as selects (the two conditions exclude each other) */
	const bool lock_now = (p.err < 85.0f) && !p.locked;
	const bool unlock_now = (p.err > 105.0f) && p.locked;
	just_locked_first = (lock_now && !p.locked_once) ? 1 : 0;
	p.locked = lock_now ? 1 : (unlock_now ? 0 : p.locked);
	p.locked_once = lock_now ? 1 : p.locked_once;
	const int changed = (lock_now || unlock_now) ? 1 : 0;

	/* pll.c:125: freq += 0.000001 * updown while unlocked (double; +-1e-6 exactly) */
	if (md_any(!p.locked)) {
		const float swept = (float)((double)p.freq + (p.updown > 0 ? 0.000001 : -0.000001));
		p.freq = p.locked ? p.freq : swept;
	}
	p.updown = (p.freq >= fmax) ? -1 : ((p.freq <= -fmax) ? 1 : p.updown);
	p.freq = (fmax < p.freq) ? fmax : p.freq;        /* MIN(fmax, freq)  */
	p.freq = (-fmax > p.freq) ? -fmax : p.freq;      /* MAX(-fmax, .)    */
	return changed;
}

/* pll.c:100-130 with the three flags kept PACKED in one word (bit 0 locked, bit 1 locked_once, bit 2 updown > 0: the layout of
 * the state array), the lock detector as bit arithmetic: v_and / v_or / v_xor / v_add are 2.6-cycle instructions on gfx950, the
 * compare + select pairs md_pll_update's int fields compile to 4.4 each (and there were nine of each per firing).  Same values,
 * same order of the float and double operations.  Returns nonzero when `locked` changed; first != 0: the first lock ever. */
struct PllWord { float phase, freq, err; };
template <bool UNIFORM = false>
__device__ __forceinline__ uint32_t
md_pll_update_packed(PllWord &p, uint32_t &fl, const float *lut, float alpha, float beta, float fmax, float i, float q, uint32_t &first)
{
	float ti, tq;
	if (UNIFORM) md_tanh_lut_uniform2(lut, i, q, ti, tq);
	else { ti = md_tanh_lut(lut, i); tq = md_tanh_lut(lut, q); }
	const float e = ti * q - tq * i;                                        /* pll.c:143-151 */

	const float ph = p.phase + alpha * e;
	p.phase = md_wrap_2pi(ph);                                               /* pll.c:113 */
	p.freq = p.freq + beta * e;

	const float decayed = p.err * (1.0f - 0.001f);
	p.err = (float)((double)decayed + fabs((double)e) * (double)0.001f);    /* pll.c:117 */

	/* pll.c:117-123: lock below 85, unlock above 105 (the two exclude each other) */
	const uint32_t below = (p.err < 85.0f) ? 1u : 0u, above = (p.err > 105.0f) ? 1u : 0u;
	const uint32_t lock_now = below & (fl ^ 1u);                            /* bit 0: locks on this symbol            */
	const uint32_t unlock_now = above & fl;                                 /* bit 0: unlocks on this symbol          */
	const uint32_t lock2 = lock_now + lock_now;                             /* the same in bit 1                      */
	first = lock2 ^ (lock2 & fl);                                           /* bit 1: locks and never had before      */
	fl = fl | lock_now | lock2;                                             /* locked = locked_once = 1               */
	fl = fl ^ (fl & unlock_now);                                            /* locked = 0                             */

	/* pll.c:125: freq += 0.000001 * updown while unlocked (double; +-1e-6 exactly) */
	if (!(fl & 1u)) p.freq = (float)((double)p.freq + ((fl & 4u) ? 0.000001 : -0.000001));
	/* pll.c:126-128: turn round at +-fmax, then clamp */
	fl = (p.freq >= fmax) ? (fl & ~4u) : ((p.freq <= -fmax) ? (fl | 4u) : fl);
	p.freq = md_clamp_sym(p.freq, fmax);                                    /* MAX(-fmax, MIN(fmax, freq)) */
	return lock_now | unlock_now;
}

/* timing.c:60-87,90-95 */
__device__ __forceinline__ void
md_timing_update(float &t_phase, float &t_freq, float &t_prev,
                 float alpha, float beta, float center, float maxdev, float q)
{
	const float sp = (t_prev < 0.0f) ? -1.0f : 1.0f;
	const float sq = (q < 0.0f) ? -1.0f : 1.0f;
	const float e = sp * q - sq * t_prev;
	t_prev = q;

	float fd = t_freq - center;
	t_phase = (float)((double)t_phase - (MD_TWO_PI_D + (double)(alpha * e)));
	fd = fd - beta * e;
	fd = md_clamp_sym(fd, maxdev);                 /* timing.c:84: MAX(-maxdev, MIN(maxdev, fd)) */
	t_freq = center + fd;
}

/* main.c:305-306 */
__device__ __forceinline__ int
md_quantise(float v)
{
	const float h = md_clamp_sym(v * 0.5f, 127.0f);      /* v/2 is exact; main.c:305: MAX(-127, MIN(127, v/2)) */
	return (int)h;                                       /* truncation toward zero */
}

#endif
