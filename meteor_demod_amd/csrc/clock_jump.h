/*
 * clock_jump.h — the symbol clock's blind steps of configs[3] (109 of the 111 interpolated steps of a symbol at 1 MS/s, -O 8) with
 * most of them in closed form.  Shared by the kernel body (rotwin_body.h) and the host proof (tools/proofs/verify_clock_jump.cpp),
 * so that what is proven is what runs.  Compile with -ffp-contract=off.
 *
 * timing.c:32-38 steps the phase accumulator p by the clock word f once per interpolated step: p = RN(p + f), a ROUNDED float
 * addition, which is why k steps are not p + k*f.  But inside one binade [2^b, 2^(b+1)) p is a multiple of u = ulp(p) = 2^(b-23),
 * f = (F + r) u with F an integer and 0 <= r < 1, and RN(p + f) = p + (F + [r > 1/2]) u: the same multiple of u every time - all
 * such sums are exact, so k steps ARE p + k * fr with fr = RN_u(f), as long as the result stays below 2^(b+1).  The tie r = 1/2
 * rounds to even: after one real addition inside the binade p / u is even and stays even, and the increment is F rounded to even -
 * which is what fr = (f + 2^b) - 2^b gives in either case (2^b / u = 2^23 is even).  So: a step count k that provably does not
 * leave the binade, two exact float operations (k * fr, p + that), and real additions across the binade boundaries - the
 * first one inside the new binade being the one that makes p / u even.
 *
 * Schedule for 109 blind steps (the launcher checks step_safe == 109, i.e. f_hi in (0.056146, 0.056656]):
 *   p0 in (-0.25, 0.30)  -- the caller checks it per lane; other lanes take the generic loop
 *   24 real additions    -> p in (1.09, 1.66), the last one from >= 1.0
 *   jump in [1, 2)       k = floor((2 - p) * inv), inv = (1 - 2^-12) / f_hi: strictly conservative (never reaches 2), at most one short
 *   3 real additions     the first or second crosses 2, the third starts inside [2, 4)
 *   jump in [2, 4)       3 real additions
 *   jump in [4, thr - 0.001)
 * and the caller's four CHECKED additions find the firing (at most three are needed).  A lane that is not where the schedule
 * expects it (never seen; the conditions are checked all the same) jumps by zero steps and is finished by the caller's generic loop.
 */
#ifndef MDEMOD_CLOCK_JUMP_H
#define MDEMOD_CLOCK_JUMP_H

#if defined(__HIPCC__)
#define CJ_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define CJ_HD static inline
#endif

#define CJ109_P_LO  (-0.25f)
#define CJ109_P_HI  0.30f
#define CJ109_R0    24
#define CJ109_RB    3
#define CJ109_MAX_STEPS 121          /* more steps than any lane takes before the checked ones: (thr + 0.25) / f_lo = 116.4 */

/* k steps inside [B, T) in closed form; prev = the input of the last real addition (must have been inside the binade) */
CJ_HD void
cj_jump(float &p, float prev, float f, float B, float T, float inv, float &count)
{
	const float fr = (f + B) - B;                        /* f rounded to a multiple of ulp(B), ties to even */
	float k = floorf((T - p) * inv);
	k = (prev >= B && p < T) ? k : 0.0f;
	p = p + k * fr;                                      /* both exact: multiples of ulp(B) below 2^24 ulp(B) */
	count = count + k;
}

/* from p0 through the 109 + x blind steps; returns the steps taken, p is below thr afterwards */
CJ_HD int
clock_jump_109(float &p, float f, float thr, float inv)
{
	float prev = p, count = 0.0f;
#pragma unroll
	for (int i = 0; i < CJ109_R0; i++) { prev = p; p = p + f; }
	cj_jump(p, prev, f, 1.0f, 2.0f, inv, count);
#pragma unroll
	for (int i = 0; i < CJ109_RB; i++) { prev = p; p = p + f; }
	cj_jump(p, prev, f, 2.0f, 4.0f, inv, count);
#pragma unroll
	for (int i = 0; i < CJ109_RB; i++) { prev = p; p = p + f; }
	cj_jump(p, prev, f, 4.0f, thr - 0.001f, inv, count);
	return CJ109_R0 + 2 * CJ109_RB + (int)count;
}

#endif
