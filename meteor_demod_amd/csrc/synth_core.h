/*
 * synth_core.h — deterministic synthetic LRPT IQ generator (test / bench input).
 *
 * Not part of the reference boundary: the reference ships no signal source.
 * This produces the synthetic recordings SURVEY §8(d) specifies (random QPSK /
 * OQPSK symbols, RRC alpha = 0.6 shaping, resampling to fs with a clock error,
 * carrier offset, AWGN, DC offset, quantisation to u8 / s16 / f32).
 *
 * Every sample is a pure function of (stream params, sample index) built from
 * integer hashing, table look-ups and IEEE double +,-,* only, so the SAME
 * inline function compiled by hipcc for gfx950 and for the host produces
 * bit-identical samples (compile with -ffp-contract=off).  That is what lets a
 * 100+ GB device-resident buffer be spot-checked tile by tile on the CPU.
 */
#ifndef MDEMOD_SYNTH_CORE_H
#define MDEMOD_SYNTH_CORE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define SYNTH_HD __host__ __device__ inline
#else
#define SYNTH_HD static inline
#endif

#define SYNTH_SPAN      6                 /* pulse half-length in symbols          */
#define SYNTH_OS        256               /* pulse table points per symbol         */
#define SYNTH_PULSE_LEN (2 * SYNTH_SPAN * SYNTH_OS + 2)
#define SYNTH_TRIG_LEN  1024              /* two-level carrier table, 10 + 10 bits */

/* Tables shared by all streams; filled on the host by synth_tables_init(). */
typedef struct {
	double pulse[SYNTH_PULSE_LEN];        /* RRC(alpha) at tau = -SPAN + i/OS      */
	double cos_hi[SYNTH_TRIG_LEN], sin_hi[SYNTH_TRIG_LEN];   /* angle = i / 2^10 turn */
	double cos_lo[SYNTH_TRIG_LEN], sin_lo[SYNTH_TRIG_LEN];   /* angle = i / 2^20 turn */
} synth_tables;

/* One stream (recording).  All time bases are fixed point so that sample n is
 * addressable without accumulation error. */
typedef struct {
	uint64_t seed;
	uint64_t sym_step;      /* symbols per input sample, 32.32 fixed point       */
	uint64_t sym_phase0;    /* symbol-clock offset at sample 0, 32.32            */
	uint32_t car_step;      /* carrier turns per sample, units of 2^-32 turn     */
	uint32_t car_phase0;    /* carrier phase at sample 0, same units             */
	double   amp;           /* per-component scale: complex RMS = amp*sqrt(2)... see synth.py */
	double   noise_scale;   /* multiplies the zero-mean integer noise sum        */
	double   dc_i, dc_q;
	int32_t  oqpsk;         /* Q rail delayed by half a symbol                   */
	int32_t  fmt;           /* 8, 16, 32 — the reference's --bps values          */
	int64_t  car_ramp48;    /* Doppler ramp: change of car_step per sample, units of 2^-48 turn per sample^2 */
	int64_t  clk_ramp64;    /* Doppler on the symbol clock: change of sym_step per sample, units of 2^-64 symbol per sample^2 */
} synth_stream;

/* (a * b) >> 32 for a < 2^63 and a signed b, exact through the 128-bit product */
SYNTH_HD uint64_t
synth_mul_shr32(uint64_t a, int64_t b)
{
	const uint64_t m = b < 0 ? (uint64_t)(-b) : (uint64_t)b;
#if defined(__HIP_DEVICE_COMPILE__)
	const uint64_t hi = __umul64hi(a, m), lo = a * m;
#else
	const unsigned __int128 pr = (unsigned __int128)a * m;
	const uint64_t hi = (uint64_t)(pr >> 64), lo = (uint64_t)pr;
#endif
	const uint64_t r = (hi << 32) | (lo >> 32);
	return b < 0 ? (uint64_t)0 - r : r;
}

SYNTH_HD uint64_t
synth_mix64(uint64_t z)
{
	z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
	z ^= z >> 27; z *= 0x94D049BB133111EBull;
	z ^= z >> 31;
	return z;
}

/* +-1 symbol rails of symbol j */
SYNTH_HD void
synth_symbol(uint64_t seed, uint64_t j, double *si, double *sq)
{
	const uint64_t h = synth_mix64(seed + (j + 1) * 0x9E3779B97F4A7C15ull);
	*si = (h & 1) ? 1.0 : -1.0;
	*sq = (h & 2) ? 1.0 : -1.0;
}

/* Interpolated pulse at tau = (d + frac/2^32) symbols, d integer in (-SPAN, SPAN). */
SYNTH_HD double
synth_pulse(const synth_tables *tb, int d, uint32_t frac)
{
	const int idx = (d + SYNTH_SPAN) * SYNTH_OS + (int)(frac >> 24);
	const double r = (double)(frac & 0xFFFFFFu) * (1.0 / 16777216.0);
	const double a = tb->pulse[idx];
	const double b = tb->pulse[idx + 1];
	return a + r * (b - a);
}

/* One baseband rail at symbol time t (32.32): sum over the 2*SPAN nearest symbols. */
SYNTH_HD double
synth_rail(const synth_tables *tb, uint64_t seed, uint64_t t, int rail)
{
	const uint64_t k = t >> 32;
	const uint32_t frac = (uint32_t)t;
	double acc = 0.0;
	for (int m = -SYNTH_SPAN + 1; m <= SYNTH_SPAN; m++) {
		/* symbol j = k + m sits at tau = frac - m */
		double si, sq;
		synth_symbol(seed, k + (uint64_t)(int64_t)m, &si, &sq);
		const double p = synth_pulse(tb, -m, frac);
		acc = acc + (rail ? sq : si) * p;
	}
	return acc;
}

/* Zero-mean sum of eight 16-bit uniforms (Irwin-Hall, variance 8*(2^32-1)/12). */
SYNTH_HD double
synth_noise(uint64_t seed, uint64_t n, uint64_t salt)
{
	const uint64_t h1 = synth_mix64(seed ^ (salt + n * 0xD1B54A32D192ED03ull));
	const uint64_t h2 = synth_mix64(h1 + 0x9E3779B97F4A7C15ull);
	int64_t s = 0;
	for (int i = 0; i < 4; i++) {
		s += (int64_t)((h1 >> (16 * i)) & 0xFFFF);
		s += (int64_t)((h2 >> (16 * i)) & 0xFFFF);
	}
	return (double)(2 * s - 8 * 65535) * 0.5;
}

/* Sample n of a stream as real-valued (I, Q) before quantisation. */
SYNTH_HD void
synth_sample(const synth_tables *tb, const synth_stream *st, uint64_t n, double *oi, double *oq)
{
	const uint64_t tri = (n & 1) ? n * ((n - 1) >> 1) : (n >> 1) * (n - 1);       /* n(n-1)/2 */
	const uint64_t t = st->sym_phase0 + n * st->sym_step + (st->clk_ramp64 ? synth_mul_shr32(tri, st->clk_ramp64) : 0);
	const double bi = synth_rail(tb, st->seed, t, 0);
	const double bq = synth_rail(tb, st->seed, st->oqpsk ? t - 0x80000000ull : t, 1);

	/* phase = phase0 + n*step + ramp*n(n-1)/2, all modulo one turn: only bits 16..47 of the 2^-48 product matter, so the
	 * 64-bit wrap-around of the multiplication is harmless */
	const uint32_t th = st->car_phase0 + (uint32_t)n * st->car_step + (uint32_t)((tri * (uint64_t)st->car_ramp48) >> 16);
	const uint32_t hi = th >> 22, lo = (th >> 12) & 0x3FFu;
	const double c = tb->cos_hi[hi] * tb->cos_lo[lo] - tb->sin_hi[hi] * tb->sin_lo[lo];
	const double s = tb->sin_hi[hi] * tb->cos_lo[lo] + tb->cos_hi[hi] * tb->sin_lo[lo];

	const double ri = bi * c - bq * s;
	const double rq = bi * s + bq * c;
	*oi = (st->amp * ri + st->dc_i) + st->noise_scale * synth_noise(st->seed, n, 0x1234567ull);
	*oq = (st->amp * rq + st->dc_q) + st->noise_scale * synth_noise(st->seed, n, 0x89ABCDEull);
}

SYNTH_HD int32_t
synth_round_clip(double v, int32_t lo, int32_t hi)
{
	/* round half up via floor-free integer conversion: trunc(v + 0.5) corrected for negatives */
	double w = v + 0.5;
	int64_t q = (int64_t)w;
	if ((double)q > w) q -= 1;          /* floor for negative non-integers */
	if (q < lo) q = lo;
	if (q > hi) q = hi;
	return (int32_t)q;
}

/* Write sample n in the stream's format at dst (which points at that sample's slot). */
SYNTH_HD void
synth_store(const synth_tables *tb, const synth_stream *st, uint64_t n, void *dst)
{
	double i, q;
	synth_sample(tb, st, n, &i, &q);
	if (st->fmt == 8) {
		uint8_t *p = (uint8_t *)dst;
		p[0] = (uint8_t)(synth_round_clip(i, -128, 127) + 128);
		p[1] = (uint8_t)(synth_round_clip(q, -128, 127) + 128);
	} else if (st->fmt == 16) {
		int16_t *p = (int16_t *)dst;
		p[0] = (int16_t)synth_round_clip(i, -32768, 32767);
		p[1] = (int16_t)synth_round_clip(q, -32768, 32767);
	} else {
		float *p = (float *)dst;
		p[0] = (float)i;
		p[1] = (float)q;
	}
}

#endif
