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
#include <hip/hip_runtime.h>
#define CJ_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define CJ_HD static inline
#endif

#ifndef CJ_WORTH
#define CJ_WORTH 0.6                 /* a schedule is taken when its instructions are under this share of the steps it replaces */
#endif
#define CJ_MIN_STEPS 16.0            /* shortest run (thr - S, in steps) cj_schedule looks at: nothing shorter pays for a jump (ra + 14 nb against CJ_WORTH x steps) */
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

/* ---- any rate: the schedule as numbers (round 4) ----------------------------------------------------------------------------
 * The same argument for any clock word: real additions from the start of a firing's run (S = 0, or pi for the second rail of an
 * OQPSK symbol) up to the first binade that holds enough steps to pay for a jump, then jump / three real additions / jump ... up
 * to the binade of the threshold.  The numbers are wave-uniform and come from the host (cj_schedule below, also what the proof
 * uses): ra real additions first (the last of them starts inside the first jump binade for every lane whose run starts inside
 * the schedule's window (lo, hi) - lanes a little below it step up to it first), then nb binades from 2^b0. */
typedef struct {
	int   ra;        /* real additions before the first jump */
	int   nb;        /* binades jumped through (0: this run does not use jumps) */
	float B0;        /* 2^b0: the first of them */
	float lo, hi;    /* the common schedule serves starts in (lo, hi) */
	float floor;     /* a lane that starts in (floor, lo] steps up to lo first; outside (floor, hi): not this way */
	int   max_steps; /* no lane takes more steps than this before the checked ones */
	int   need;      /* input samples that hold max_steps + 4 steps (filled in by the host: -O is its business) */
	int   up_max;    /* no lane in (floor, lo] needs more real additions than this to get above lo (the bound of that loop) */
} cj_sched;

/* host: the schedule of a run that starts around S and ends at thr, for clock words up to f_hi (and down to f_hi (1 - 5e-4)) */
static inline cj_sched
cj_schedule(double S, double thr, double f_hi)
{
	cj_sched J = { 0, 0, 1.0f, 0.0f, 0.0f, 0.0f, 0, 0, 0 };
	/* No schedule where none can pay: a jump costs about 14 instructions, so a run of fewer than CJ_MIN_STEPS steps never takes one
	 * (the "worth it" test at the end says the same, later).  This is also what keeps the searches below finite: with several firings
	 * per interpolated step (symrate >= 2 fs O for OQPSK, 4 fs O for QPSK - accepted by demod_host.cpp and served by the stepping loop)
	 * thr - 0.25 f_hi is not positive and has no binade at all (round 4 looped forever there). */
	if (!(f_hi > 0.0) || !(thr > S) || !(thr - S >= CJ_MIN_STEPS * f_hi) || !(thr < 64.0)) return J;
	const double f_lo = f_hi * (1.0 - 6e-4), w = 3.0 * f_hi, top = thr - 0.25 * f_hi;   /* top >= 15.75 f_hi > 0 */
	int b_last = 0;                                      /* binade of thr - a bit: 2^b_last < top <= 2^(b_last + 1), searched in [-64, 6] */
	while (b_last < 6 && ldexp(1.0, b_last + 1) < top) b_last++;
	while (b_last > -64 && ldexp(1.0, b_last) >= top) b_last--;
	if (!(ldexp(1.0, b_last) < top)) return J;
	/* the first binade to jump through: going down from the threshold's, as long as the next lower one holds at least 14 steps and
	   either lies above the whole start window (the real additions climb into it) or contains it (second OQPSK rail: S = pi in [2, 4)) */
	int b0 = b_last, inside = 0;
	while (b0 > -20 && !inside) {
		const double lower = ldexp(1.0, b0 - 1);
		if (lower < 14.0 * f_hi) break;
		if (lower > S + w) { b0--; continue; }
		if (lower <= S - w && 2.0 * lower > S + w + 2.0 * f_hi) { b0--; inside = 1; }
		break;
	}
	/* the start window: a run starts in [S - alpha e, S + f - alpha e) (timing.c:79: the loop's correction of the phase).  (lo, hi) is
	   what the common schedule serves: three steps below S, and above it whatever room the first binade leaves (at 6 MS/s and up the
	   corrections reach 0.08 rad, five steps).  Lanes between `floor` and lo first step up to lo on their own (real additions, as
	   many as each needs), lanes outside (floor, hi) take the caller's stepping loop. */
	const double B = ldexp(1.0, b0);
	double lo = S - w, hi = S + f_hi + w + 0.25, floor_ = lo - 0.25;
	int ok = 0;
	if (inside) {                                        /* every lane is in the binade already: one real addition for the parity */
		if (lo < B) lo = B;
		if (floor_ < lo) floor_ = lo;
		if (hi > 2.0 * B - 2.0 * f_hi) hi = 2.0 * B - 2.0 * f_hi;
		J.ra = 1;
		ok = lo <= S - w && hi >= S + f_hi + w;
	} else {
		/* ra - 1 additions take the lowest start to 2^b0 at least, and the highest start stays below 2^(b0+1) after ra */
		J.ra = (int)ceil((B - lo) / f_lo) + 1;
		const double room = 2.0 * B - J.ra * f_hi * (1.0 + 1e-6) - 1e-6;
		if (hi > room) hi = room;
		ok = hi >= S + f_hi + w;
	}
	if (!ok) return J;
	if (B < 14.0 * f_hi || (thr - 0.25 * f_hi) - ldexp(1.0, b_last) < 0) return J;
	/* A jump is short of the binade's end by design: inv carries 2^-12, and it is sized for f_hi while the lane's word may be 6e-4
	 * below - together up to 8.5e-4 of the binade's steps.  The three real additions after a jump must cross into the next binade
	 * (at most one step short, plus the floor), the caller's four checked additions must find the firing after the last one (at
	 * most two short).  Past about 1 100 steps in a binade - clock words under 1.8e-3, fs x O over 2.5e8 - that no longer holds:
	 * lanes would fall back to the stepping loop one by one (correct, and slower than no schedule).  Found by tests/sanitize/
	 * fuzz_derive.cpp in round 5; the listed rates of tools/proofs/verify_clock_jump.cpp end at 1.6e8. */
	{
		const double slack = 1.0 / 4096.0 + (1.0 - f_lo / f_hi) + 1e-6;
		for (int b = b0; b <= b_last; b++) {
			const double lo_b = ldexp(1.0, b), hi_b = b == b_last ? thr : 2.0 * lo_b;
			if ((hi_b - lo_b) / f_lo * slack > (b == b_last ? 1.8 : 0.9)) return J;
		}
	}
	J.nb = b_last - b0 + 1;
	J.B0 = (float)B;
	J.lo = (float)lo; J.hi = (float)hi; J.floor = (float)floor_;
	if ((double)J.lo < lo) J.lo = nextafterf(J.lo, 1e30f);
	if ((double)J.hi > hi) J.hi = nextafterf(J.hi, -1e30f);
	if ((double)J.floor < floor_) J.floor = nextafterf(J.floor, 1e30f);
	J.max_steps = (int)ceil((thr - floor_) / f_lo) + 3;
	J.up_max = (int)ceil((lo - floor_) / f_lo) + 2;
	/* worth it?  a jump is about 14 instructions (with its three real additions), a step one */
	if (J.nb < 1 || J.nb > 8 || J.ra + 14 * J.nb > CJ_WORTH * (thr - S) / f_hi) J.nb = 0;
	return J;
}

/* the run of one firing: from p (inside (J.floor, J.hi)) to within three steps below thr; returns the steps taken */
CJ_HD int
clock_jump_run(float &p, float f, float thr, float f_hi, float inv, const cj_sched &J)
{
	float count = 0.0f;
	int early = 0;
	/* a lane the loop's correction set back (rare below 3 MS/s).  Bounded by the host's count: a clock word outside the loop's range
	 * (timing.c:80-86; mdemod_set_state and mdemod_set_clock_seeds keep such words out, this is the second fence) must not spin a wave -
	 * it leaves p wherever it is and the caller's checked additions / stepping loop take over.  NaN ends it by itself. */
	for (; early < J.up_max && p <= J.lo; early++) p = p + f;
	int k = J.ra - 1;
	for (; k >= 8; k -= 8) {
#pragma unroll
		for (int i = 0; i < 8; i++) p = p + f;
	}
	if (k & 4) { p = p + f; p = p + f; p = p + f; p = p + f; }
	if (k & 2) { p = p + f; p = p + f; }
	if (k & 1) p = p + f;
	float prev = p;
	p = p + f;
	float B = J.B0;
	for (int b = 0; b < J.nb; b++) {
		const bool last = b == J.nb - 1;
		cj_jump(p, prev, f, B, last ? thr - 0.125f * f_hi : 2.0f * B, inv, count);
		if (!last) {
#pragma unroll
			for (int i = 0; i < CJ109_RB; i++) { prev = p; p = p + f; }
		}
		B = 2.0f * B;
	}
	return early + J.ra + CJ109_RB * (J.nb - 1) + (int)count;
}

#endif
