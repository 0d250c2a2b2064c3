/*
 * demod_host.cpp — init-time arithmetic of the demodulator (host, libm).
 *
 * Compile with -ffp-contract=off.  Each block cites the reference lines whose
 * mixed float/double evaluation it reproduces; values are pinned by the
 * known-answer fixtures in tests/golden (RRC tables, loop constants, tanh LUT).
 */
#include "demod_host.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#pragma clang fp contract(off)

namespace {

const double kPi = 3.14159265358979323846;   /* M_PI */

/* Second-order loop gains from damping and bandwidth: pll.c:133-140 (damp
 * 1/sqrt2) and timing.c:98-105 (damp 1) evaluate the same float expression. */
void
loop_gains(float damp, float bw, float &alpha, float &beta)
{
	const float two_d_bw = (2.0f * damp) * bw;
	const float denom = (1.0f + two_d_bw) + bw * bw;
	alpha = ((4.0f * damp) * bw) / denom;
	beta = ((4.0f * bw) * bw) / denom;
}

/* One prototype tap: filter.c:71-94.  `n_taps` is the prototype length
 * (taps*interp), `osf` the oversampling of the prototype (osf*interp). */
float
rrc_tap(int stage, unsigned n_taps, float osf, float alpha)
{
	const float norm = static_cast<float>(2.0 / 5.0);
	const int centre = static_cast<int>((n_taps - 1) / 2);

	if (stage == centre) {
		/* filter.c:82-84: (1-a) and 4a are float, the division by pi and the sum are double */
		const double v = static_cast<double>(1.0f - alpha) + static_cast<double>(4.0f * alpha) / kPi;
		return static_cast<float>(static_cast<double>(norm) * v);
	}

	const float t = static_cast<float>(std::abs(centre - stage)) / osf;
	const double pi_t = kPi * static_cast<double>(t);
	const float fat = (4.0f * alpha) * t;                                  /* 4*alpha*t, float */
	const float num = sinf(static_cast<float>(pi_t * static_cast<double>(1.0f - alpha))) +
	                  fat * cosf(static_cast<float>(pi_t * static_cast<double>(1.0f + alpha)));
	const float den = static_cast<float>(pi_t * static_cast<double>(1.0f - fat * fat));

	/* filter.c:90-91: Blackman window, cosf of a double argument narrowed to float, sum in double */
	const double span = static_cast<double>(n_taps - 1);
	const float c1 = cosf(static_cast<float>((2 * kPi) * static_cast<double>(stage) / span));
	const float c2 = cosf(static_cast<float>((4 * kPi) * static_cast<double>(stage) / span));
	const double win = (0.42 - 0.5 * static_cast<double>(c1)) + 0.08 * static_cast<double>(c2);
	const float shaped = static_cast<float>(static_cast<double>(num) * win);

	return (shaped / den) * norm;
}

} /* namespace */

/* ---- mdemod_last_error(): the text of the failure, handed back instead of printed (include/meteor_demod_amd.h, ABI 5) ---------- */
namespace {
thread_local char tl_error[512];
thread_local int  tl_depth;
}

void
mdm_note_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(tl_error, sizeof tl_error, fmt, ap);
	va_end(ap);
	if (getenv("MDEMOD_DEBUG")) fprintf(stderr, "meteor_demod_amd: %s\n", tl_error);
}

mdm_api_scope::mdm_api_scope() { if (tl_depth++ == 0) tl_error[0] = 0; }
mdm_api_scope::~mdm_api_scope() { --tl_depth; }

extern "C" const char *
mdemod_last_error(void)
{
	return tl_error;
}

int
mdemod_host_derive(const mdemod_params &p, HostTables &out, int generation)
{
#define REFUSE(...) do { mdm_note_error(__VA_ARGS__); return MDEMOD_ERR_PARAM; } while (0)
	if (p.interp_factor < 1 || p.interp_factor > 64) REFUSE("-O %d: the interpolation factor must be 1..64", p.interp_factor);
	if (p.rrc_order < 1 || p.rrc_order > 256) REFUSE("-f %d: the RRC order must be 1..256", p.rrc_order);
	if (p.samplerate <= 0 || p.symrate <= 0) REFUSE("sample rate %d and symbol rate %d must be positive", p.samplerate, p.symrate);
	/* Fewer than one input sample per symbol is fine: the reference then fires more than once inside its per-sample loop and keeps
	 * only the LAST symbol of each sample (demod.c:33-47, 62-90: `ret` and `*sample` are overwritten), and so do the kernels
	 * (goldens one_per_symbol, sub_sample, sub_sample_oqpsk).  Below a quarter of a sample per firing nothing has been tested. */
	if (static_cast<double>(p.samplerate) * (p.oqpsk ? 2.0 : 1.0) < static_cast<double>(p.symrate) * 0.25)
		REFUSE("%d samples/s for %d symbols/s: fewer than a quarter of an input sample per firing is untested ground", p.samplerate, p.symrate);
	if (p.bps != 8 && p.bps != 16 && p.bps != 32) REFUSE("%d bits per sample: 8, 16 or 32 expected", p.bps);
	/* demod.c:12-13 multiply in int: `multiplier * symrate` and `samplerate * interp_factor`.  Where those overflow the reference is
	 * undefined; nothing to reproduce, refused. */
	if (static_cast<int64_t>(p.samplerate) * p.interp_factor > INT32_MAX || static_cast<int64_t>(p.symrate) * 2 > INT32_MAX)
		REFUSE("sample rate %d x -O %d (or twice the symbol rate %d) overflows the reference's int arithmetic (demod.c:12-13)", p.samplerate, p.interp_factor, p.symrate);

	DemodConsts &c = out.c;
	c.interp = p.interp_factor;
	c.taps = 2 * p.rrc_order + 1;
	c.oqpsk = p.oqpsk ? 1 : 0;
	c.sin_lut = 0;                                    /* (decided by plan_context once the kernel instance and its LDS are known) */

	/* demod.c:10-14 */
	const int mult = p.oqpsk ? 1 : 2;
	const float pll_bw = static_cast<float>((2 * kPi) * static_cast<double>(p.pll_bw) /
	                                        static_cast<double>(mult * p.symrate));
	const float sym_freq = static_cast<float>((2 * kPi) * static_cast<double>(p.symrate) /
	                                          static_cast<double>(p.samplerate * p.interp_factor));
	const float sym_bw = p.sym_bw / static_cast<float>(p.interp_factor);
	out.osf = static_cast<float>(p.samplerate) / static_cast<float>(p.symrate);

	/* pll.c:29-44 */
	float fmax = p.freq_max;
	if (fmax < 0) fmax = 0.3f;
	else fmax = (1.0f < fmax) ? 1.0f : fmax;
	c.pll_fmax = p.oqpsk ? fmax / 2.0f : fmax;
	loop_gains(0.7071067811865475f, pll_bw, c.pll_alpha, c.pll_beta);
	for (int i = 0; i < 32; i++) out.tanh_lut[i] = static_cast<float>(tanh(static_cast<double>(i - 16)));
	/* the kernels clamp the LUT argument instead of branching on v > 15 / v < -16 (pll.c:156-157) */
	if (out.tanh_lut[31] != 1.0f || out.tanh_lut[0] != -1.0f) REFUSE("this host's tanh() does not saturate to +-1 at +-15: the kernels' clamped table lookup (pll.c:156-157) would differ");

	/* timing.c:19-27 */
	c.t_center = sym_freq;
	c.t_maxdev = sym_freq / static_cast<float>(1 << 12);
	loop_gains(1.0f, sym_bw, c.t_alpha, c.t_beta);

	/* filter.c:10-28 */
	const unsigned taps = static_cast<unsigned>(c.taps), banks = static_cast<unsigned>(c.interp);
	const float rrc_alpha = static_cast<float>(0.6);                       /* demod.h:8 */
	out.rrc.resize(static_cast<size_t>(taps) * banks);
	for (unsigned j = 0; j < banks; j++)
		for (unsigned i = 0; i < taps; i++)
			out.rrc[j * taps + i] = rrc_tap(static_cast<int>(i * banks + j), taps * banks,
			                                out.osf * static_cast<float>(banks), rrc_alpha);
	/* filter.c:86-93 has no guard for t = 1/(4 alpha): where samples-per-symbol x -O / 2.4 lands on a tap (230.4 kS/s at 80 000
	 * symbols/s with -O 5; 1.08 MS/s at 72 000 with -O 4) the tap is x/0 or 0/0.  The reference then filters inf / NaN into its
	 * loops and indexes its tanh table with (int)NaN (pll.c:154-159): nothing to reproduce.  Refused, with the way out. */
	for (float tap : out.rrc)
		if (!std::isfinite(tap)) {
			REFUSE("the RRC filter for %d samples/s, %d symbols/s, -O %d has a tap that is not finite (filter.c:86-93 divides by "
			       "zero there; the reference's output is undefined): choose another -O", p.samplerate, p.symrate, p.interp_factor);
		}

	/* ---- symbol-clock fast path constants ---- */
	{
		const double f_hi = (static_cast<double>(c.t_center) + static_cast<double>(c.t_maxdev)) * (1.0 + 1e-6);
		c.step_fmax = nextafterf(static_cast<float>(f_hi), 1e30f);
		/* Blind steps: a lane whose phase is below thr - k*f_hi cannot fire within k steps.  After a
		 * symbol the phase restarts in [-alpha*e, f - alpha*e), so k is sized for a start below
		 * f_hi + 0.05; the kernel checks the bound per lane and falls back to its generic loop. */
		const double thr_min = p.oqpsk ? kPi : 2 * kPi;          /* first threshold a fresh symbol meets */
		int ks = static_cast<int>(floor((thr_min - f_hi - 0.05 - 1e-3) / f_hi));
		c.step_safe = ks < 0 ? 0 : ks;
		c.step_check = 4;                                         /* fixed in the kernel */
		c.interp_magic = static_cast<uint32_t>((1ull << 32) / static_cast<uint64_t>(c.interp)) + 1u;
		c.step_inv = static_cast<float>((1.0 - 1.0 / 4096.0) / static_cast<double>(c.step_fmax));
		/* long runs (a high sample rate times -O): most of the steps in closed form, binade by binade (clock_jump.h; the v3 body only) */
		const double pi_f = static_cast<double>(static_cast<float>(kPi)), two_pi_f = 2.0 * pi_f;
		const cj_sched none = { 0, 0, 1.0f, 0.0f, 0.0f, 0.0f, 0, 0, 0 };
		c.jump[0] = c.jump[1] = none;
		if (!(p.reserved & MDEMOD_FLAG_NO_CLOCK_JUMP)) {
			c.jump[0] = cj_schedule(0.0, p.oqpsk ? pi_f : two_pi_f, static_cast<double>(c.step_fmax));
			if (p.oqpsk) c.jump[1] = cj_schedule(pi_f, two_pi_f, static_cast<double>(c.step_fmax));
		}
		for (cj_sched &j : c.jump) j.need = (j.max_steps + 4 + c.interp - 1) / c.interp;
	}

	/* ---- kernel selection + geometry ---- */
	const bool v3 = generation != 0;                  /* 0: the v1 ring kernel whatever the configuration (tests) */
	const double per_firing = static_cast<double>(out.osf) / (p.oqpsk ? 2.0 : 1.0);   /* samples consumed per firing */
	const bool std_ok = c.taps <= 65 && per_firing <= 3.6;
	/* wide: packed window only (s16 / u8), up to 129 taps, up to 15 samples per firing */
	const bool wide_ok = !std_ok && per_firing <= 15.0 && (p.bps != 32 ? c.taps <= 129 : c.taps <= 65);   /* float input: only the 96-slot mid window fits as float pairs */
	/* mid: the short filter at a high sample rate (e.g. the default -f 32 at 1.024 MS/s): same lane spread as wide, but
	 * a 96-slot window instead of 160 */
	const bool mid_ok = wide_ok && c.taps <= 65;
	/* far: the short filter at 2-3 MS/s-class rates: up to 30 (v3: 46) samples per firing, two 16-slot slides per iteration */
	const bool far_ok = !std_ok && !wide_ok && c.taps <= 65 && per_firing <= 46.0 && p.bps != 32;   /* (47 alignments, as many loop iterations per firing as it takes) */
	/* hybrid: float input outside the std geometry, up to 129 taps at up to 15 samples per firing or up to 65 taps at up to 30 (a float window of 160 slots is 320
	 * registers: the older 80 slots in VGPRs, the newer ones in AccVGPRs, one wave per SIMD: demod_kernel_rot.hip, WinH; with up to 65
	 * taps - rw_mid as well - the window has 96 slots) */
	const bool hyb_far_ok = v3 && !std_ok && p.bps == 32 && per_firing > 30.0 && per_firing <= 54.0 && c.taps <= 65;   /* 120 slots, 55 alignments */
	const bool hyb_ok = v3 && !std_ok && p.bps == 32 && ((per_firing <= 30.0 && c.taps <= 129) || hyb_far_ok);
	/* the long filter at 15..30 samples per firing (s16 / u8): the wide window has the 31 alignments for it and slides once per
	 * loop iteration, so such a firing takes two iterations */
	const bool wide_far_ok = v3 && !std_ok && !wide_ok && !far_ok && per_firing <= 30.0 && c.taps <= 129 && p.bps != 32;
	/* gather: rates none of the windows reaches (more than 46 samples per firing, 30 with the long filter): no window,
	 * every firing loads its own taps (demod_kernel_gat.hip) */
	const bool gather_ok = v3 && c.taps <= (p.bps == 32 ? 65 : 129) && !std_ok && !wide_ok && !far_ok && !wide_far_ok && !hyb_ok;
	const bool allow_rw = v3;
	out.rw_gather = gather_ok;
	out.rw_hyb = hyb_ok;
	out.rw_std_compact = false;
	out.rw_compact4 = false;          /* (assigned again below for the packed windows; the std and gather returns leave before that) */
	out.use_rw = allow_rw && (std_ok || wide_ok || far_ok || hyb_ok || wide_far_ok || gather_ok);
	out.rw_mid = out.use_rw && !std_ok && (mid_ok || (hyb_ok && !hyb_far_ok && c.taps <= 65));
	out.rw_far = out.use_rw && (far_ok || hyb_far_ok);
	out.rw_wide = out.use_rw && !std_ok && !mid_ok && !far_ok && !hyb_ok && !gather_ok;
	c.chunk_granules = 2;
	if (out.use_rw && !out.rw_wide && !out.rw_mid && !out.rw_far && !out.rw_hyb && !out.rw_gather) {
		/* std: 80-slot register window, filter embedded as 65 taps (leading zeros), 16 alignments */
		const int kTaps = 65, NW = 80, AL = NW - kTaps + 1;
		c.hpad = kTaps - 1;
		c.win_granules = NW / 4;
		c.ring_granules = 0;
		c.ctab_row_floats = NW;
		c.ctab_row_stride = NW + 4;                       /* 21 x 16 B: rows 16-byte aligned for ds_read_b128, odd => consecutive rows on distinct 16-byte slots */
		const int lead = kTaps - c.taps;
		/* sixteen rows per bank are 5.4 KB: past -O 18 (100 KB) the v3 kernel takes the compact4 layout of the other geometries instead
		 * (four shifted copies of the padded taps per bank, 1.6 KB: demod_kernel_rot.hip, WinF<.., COMPACT>) - two more instructions
		 * per firing for the row address, and -O 29 .. 80 stay off the v1 ring kernel */
#ifndef MDEMOD_STD_COMPACT_ABOVE
#define MDEMOD_STD_COMPACT_ABOVE (100 * 1024)
#endif
		out.rw_std_compact = v3 && static_cast<size_t>(AL) * banks * c.ctab_row_stride * sizeof(float) > MDEMOD_STD_COMPACT_ABOVE;
		if (out.rw_std_compact) {
			const int AMAX = NW - kTaps, LP = kTaps + 2 * AMAX;
			c.ctab_row_floats = LP;
			c.ctab_row_stride = (LP + 3) / 4 * 4;
			if ((c.ctab_row_stride / 4) % 2 == 0) c.ctab_row_stride += 4;
			out.ctab.assign(static_cast<size_t>(4) * banks * c.ctab_row_stride, 0.0f);
			for (unsigned b = 0; b < banks; b++) {
				std::vector<float> P(static_cast<size_t>(LP) + 4, 0.0f);
				for (int k = 0; k < c.taps; k++) P[AMAX + lead + k] = out.rrc[b * taps + k];
				for (int k = 0; k < 4; k++) {
					float *row = &out.ctab[(static_cast<size_t>(b) * 4 + k) * c.ctab_row_stride];
					for (int i = 0; i < LP; i++) row[i] = P[i + k];
				}
			}
			return MDEMOD_OK;
		}
		out.ctab.assign(static_cast<size_t>(AL) * banks * c.ctab_row_stride, 0.0f);
		for (int a = 0; a < AL; a++)
			for (unsigned b = 0; b < banks; b++) {
				float *row = &out.ctab[(static_cast<size_t>(a) * banks + b) * c.ctab_row_stride];
				for (int k = 0; k < c.taps; k++) row[a + lead + k] = out.rrc[b * taps + k];
			}
		return MDEMOD_OK;
	}
	if (out.rw_gather) {
		/* compact4 layout with as many alignments as samples in a 16-byte load: the samples between such a step and the oldest tap */
		const int kTaps = c.taps <= 65 ? 65 : 129, AMAX = (p.bps == 8 ? 8 : (p.bps == 16 ? 4 : 2)) - 1, LP = kTaps + 2 * AMAX;
		c.hpad = kTaps - 1;
		c.win_granules = (kTaps + 3) / 4;
		c.ring_granules = 0;
		c.ctab_row_floats = LP;
		c.ctab_row_stride = (LP + 3) / 4 * 4;
		if ((c.ctab_row_stride / 4) % 2 == 0) c.ctab_row_stride += 4;
		out.ctab.assign(static_cast<size_t>(4) * banks * c.ctab_row_stride, 0.0f);
		const int lead = kTaps - c.taps;
		for (unsigned b = 0; b < banks; b++) {
			std::vector<float> P(static_cast<size_t>(LP) + 4, 0.0f);
			for (int k = 0; k < c.taps; k++) P[AMAX + lead + k] = out.rrc[b * taps + k];
			for (int k = 0; k < 4; k++) {
				float *row = &out.ctab[(static_cast<size_t>(b) * 4 + k) * c.ctab_row_stride];
				for (int i = 0; i < LP; i++) row[i] = P[i + k];
			}
		}
		return MDEMOD_OK;
	}
	out.rw_compact4 = v3 && (out.rw_wide || out.rw_mid || out.rw_far) && p.bps != 32;
	if (out.rw_compact4 || out.rw_hyb) {
		/* v3 packed rotating window: per bank the padded sequence P = AMAX zeros ++ taps ++ AMAX zeros, stored FOUR times:
		 * copy (bank, k)[i] = P[i + k].  A lane at alignment a reads P[(AMAX - a) + s] for slot s, i.e. copy ((AMAX - a) & 3) at the
		 * 16-byte aligned index ((AMAX - a) & ~3) + s: every group of four taps is one ds_read_b128.  The FIR's prefetch runs up
		 * to six groups (24 floats) past the last tap it uses: rows carry that much padding. */
		const int kTaps = (out.rw_wide || (out.rw_hyb && !out.rw_mid && !out.rw_far)) ? 129 : 65;
		const int NW = out.rw_mid ? MDEMOD_RW_MID_NW : (out.rw_far ? (out.rw_hyb ? 120 : MDEMOD_RW_FAR_NW) : MDEMOD_RW_WIDE_NW), AMAX = NW - kTaps;
		const int LP = kTaps + 2 * AMAX;
		c.hpad = kTaps - 1;
		c.win_granules = NW / 4;
		c.ring_granules = 0;
		c.ctab_row_floats = LP;
		/* (the hybrid window reads exactly the NW slots of its alignment, no prefetch past them) */
		c.ctab_row_stride = (LP + 3 + (out.rw_hyb ? 0 : 24 + 3)) / 4 * 4; /* whole 16-byte groups ...            */
		if ((c.ctab_row_stride / 4) % 2 == 0) c.ctab_row_stride += 4;    /* ... an odd number of them           */
		out.ctab.assign(static_cast<size_t>(4) * banks * c.ctab_row_stride + (out.rw_hyb ? 0 : 32), 0.0f);
		const int lead = kTaps - c.taps;
		for (unsigned b = 0; b < banks; b++) {
			std::vector<float> P(static_cast<size_t>(LP) + 4, 0.0f);
			for (int k = 0; k < c.taps; k++) P[AMAX + lead + k] = out.rrc[b * taps + k];
			for (int k = 0; k < 4; k++) {
				float *row = &out.ctab[(static_cast<size_t>(b) * 4 + k) * c.ctab_row_stride];
				for (int i = 0; i < LP; i++) row[i] = P[i + k];
			}
		}
		return MDEMOD_OK;
	}
	/* v1: LDS ring */
	c.hpad = ((c.taps - 1 + 7) / 8) * 8;
	if (c.hpad < 8) c.hpad = 8;
	c.win_granules = (c.taps + 6) / 4;                 /* ceil((3 + taps) / 4): any start alignment */
	c.ring_granules = c.hpad / 4 + 8;
	c.ctab_row_floats = 4 * c.win_granules;
	const int stride_granules = (c.win_granules & 1) ? c.win_granules : c.win_granules + 1;   /* odd: distinct bank slots */
	c.ctab_row_stride = 4 * stride_granules;

	/* aligned coefficient rows: row (a, bank), slot s holds tap s-a or 0 */
	out.ctab.assign(static_cast<size_t>(4) * banks * c.ctab_row_stride, 0.0f);
	for (int a = 0; a < 4; a++)
		for (unsigned b = 0; b < banks; b++) {
			float *row = &out.ctab[(static_cast<size_t>(a) * banks + b) * c.ctab_row_stride];
			for (int k = 0; k < c.taps; k++) row[a + k] = out.rrc[b * taps + k];
		}
	return MDEMOD_OK;
}
