/*
 * lrpt_oracle.c — CPU restatement of the meteor_demod hot path.
 * TEST INFRASTRUCTURE ONLY (see lrpt_oracle.h).  Must be compiled with
 * -ffp-contract=off: the reference's output is only reproducible without FMA
 * contraction.
 *
 * The arithmetic below spells out every rounding step and every float/double
 * promotion of the reference explicitly (temporaries are typed on purpose).
 */
#include "lrpt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static const double TWO_PI_D = 2 * M_PI;               /* "2*M_PI" literals: pll.c:61,113 timing.c:80 */
static const float  TWO_PI_F = 2 * (float)M_PI;        /* timing.c:37 */
static const float  PI_F     = (float)M_PI;            /* timing.c:50 */

/* ------------------------------------------------------------------------- */
/* Init-time                                                                 */
/* ------------------------------------------------------------------------- */

/* dsp/filter.c:71-94 — one tap of the Blackman-windowed RRC prototype. */
float
orc_rrc_coeff(int stage_no, unsigned taps, float osf, float alpha)
{
	const float norm = (float)(2.0 / 5.0);
	const int order = (int)((taps - 1) / 2);

	if (stage_no == order) {                              /* filter.c:82-84 */
		double centre = (double)(1.0f - alpha) + (double)(4.0f * alpha) / M_PI;
		return (float)((double)norm * centre);
	}

	const float t = (float)abs(order - stage_no) / osf;  /* filter.c:86 */
	const float four_at = (4.0f * alpha) * t;

	/* filter.c:87: sinf()/cosf() take the double products narrowed to float */
	const float s_arg = (float)((M_PI * (double)t) * (double)(1.0f - alpha));
	const float c_arg = (float)((M_PI * (double)t) * (double)(1.0f + alpha));
	float coeff = sinf(s_arg) + four_at * cosf(c_arg);

	/* filter.c:88 */
	const float interm = (float)((M_PI * (double)t) * (double)(1.0f - four_at * four_at));

	/* filter.c:90-91 (Blackman 0.42/0.5/0.08, labelled Hamming upstream) */
	const double span = (double)(taps - 1);
	const float w1 = cosf((float)((2 * M_PI) * (double)stage_no / span));
	const float w2 = cosf((float)((4 * M_PI) * (double)stage_no / span));
	const double window = (0.42 - 0.5 * (double)w1) + 0.08 * (double)w2;
	coeff = (float)((double)coeff * window);

	return (coeff / interm) * norm;                       /* filter.c:93 */
}

/* loop gains: dsp/pll.c:133-140 and dsp/timing.c:98-105 (identical formula). */
static void
loop_gains(float damp, float bw, float *alpha, float *beta)
{
	const float denom = (1.0f + (2.0f * damp) * bw) + bw * bw;
	*alpha = ((4.0f * damp) * bw) / denom;
	*beta  = ((4.0f * bw) * bw) / denom;
}

/* demod.c:8-15 -> pll.c:25-44, timing.c:19-27, filter.c:10-28 */
int
orc_consts_init(orc_consts *c, const orc_params *p)
{
	memset(c, 0, sizeof(*c));
	if (p->interp < 1 || p->rrc_order < 0 || p->symrate <= 0 || p->samplerate <= 0)
		return -1;

	const int mult = p->oqpsk ? 1 : 2;                                   /* demod.c:10 */
	const float pll_bw = (float)(TWO_PI_D * (double)p->pll_bw / (double)(mult * p->symrate)); /* demod.c:12 */
	const float sym_freq = (float)(TWO_PI_D * (double)p->symrate / (double)(p->samplerate * p->interp)); /* demod.c:13 */
	const float sym_bw = p->sym_bw / (float)p->interp;                   /* demod.c:13 */
	const float osf = (float)p->samplerate / (float)p->symrate;          /* demod.c:14 */
	const float rrc_alpha = (float)0.6;                                  /* demod.h:8 */

	/* pll.c:29-44 */
	float fmax = p->freq_max;
	if (fmax < 0) fmax = 0.3f;
	else fmax = (1.0f < fmax) ? 1.0f : fmax;
	c->pll_fmax = p->oqpsk ? fmax / 2.0f : fmax;
	for (int i = 0; i < 32; i++) c->tanh_lut[i] = (float)tanh((double)(i - 16));
	loop_gains(0.7071067811865475f, pll_bw, &c->pll_alpha, &c->pll_beta);

	/* timing.c:19-27 */
	c->t_center = sym_freq;
	c->t_maxdev = sym_freq / (float)(1 << 12);
	loop_gains(1.0f, sym_bw, &c->t_alpha, &c->t_beta);

	/* filter.c:10-28 */
	c->interp = p->interp;
	c->taps = 2 * p->rrc_order + 1;
	c->oqpsk = p->oqpsk ? 1 : 0;
	c->osf = osf;
	c->coeffs = (float *)malloc(sizeof(float) * (size_t)c->taps * (size_t)c->interp);
	if (!c->coeffs) return -2;
	const unsigned taps = (unsigned)c->taps, factor = (unsigned)c->interp;
	for (unsigned j = 0; j < factor; j++)
		for (unsigned i = 0; i < taps; i++)
			c->coeffs[j * taps + i] =
			    orc_rrc_coeff((int)(i * factor + j), taps * factor, osf * (float)factor, rrc_alpha);
	return 0;
}

void
orc_consts_free(orc_consts *c)
{
	free(c->coeffs);
	c->coeffs = NULL;
}

int
orc_stream_init(orc_stream *st, const orc_params *p)
{
	memset(st, 0, sizeof(*st));
	int rc = orc_consts_init(&st->c, p);
	if (rc) return rc;
	st->s.hist = (orc_cf *)calloc((size_t)st->c.taps, sizeof(orc_cf));   /* filter.c:16 */
	if (!st->s.hist) { orc_consts_free(&st->c); return -2; }
	st->s.gain = 1.0f;                      /* agc.c:9  */
	st->s.pll_err = 1000.0f;                /* pll.c:36 */
	st->s.updown = 1;                       /* pll.c:112 */
	st->s.t_freq = st->c.t_center;          /* timing.c:21 */
	st->s.dual_state = 1;                   /* timing.c:43 */
	st->s.first_lock_symbol = -1;
	return 0;
}

void
orc_stream_free(orc_stream *st)
{
	free(st->s.hist);
	st->s.hist = NULL;
	orc_consts_free(&st->c);
}

orc_stream *
orc_stream_new(const orc_params *p)
{
	orc_stream *st = (orc_stream *)malloc(sizeof(*st));
	if (!st) return NULL;
	if (orc_stream_init(st, p)) { free(st); return NULL; }
	return st;
}

void
orc_stream_delete(orc_stream *st)
{
	if (!st) return;
	orc_stream_free(st);
	free(st);
}

/* ------------------------------------------------------------------------- */
/* Fixed-point sine: dsp/sincos.c:13-47                                      */
/* ------------------------------------------------------------------------- */

float
orc_fast_sin_code(int16_t code)
{
	const int32_t a = 1 << 14;
	const int32_t b = (int32_t)((2 - 3.14159 / 4) * (1 << 14));   /* 19900 */
	const int32_t cc = b - (1 << 14);                             /* 3516  */

	const int16_t sign = code;
	int16_t x = (int16_t)(code & 0x7FFF);          /* sincos.c:26: clear bit 15 */
	x = (int16_t)(x - (1 << 14));                  /* sincos.c:27 */
	const int32_t x2 = ((int32_t)x * x) >> 14;     /* sincos.c:29 */
	int32_t y = b - (int32_t)(((int64_t)(x2 * cc)) >> 14);   /* sincos.c:31,43-47 */
	y = a - (int32_t)(((int64_t)(x2 * y)) >> 14);            /* sincos.c:32 */
	return (float)(sign < 0 ? -y : y) / (float)(1 << 14);    /* sincos.c:34 */
}

float
orc_fast_sin(float fx)
{
	/* sincos.c:24: float*int stays float, the division is double, the
	 * double->int16 narrowing keeps the low 16 bits of the int32 truncation
	 * (what gcc/clang emit on x86-64). */
	const double xd = (double)(fx * 65536.0f) / TWO_PI_D;
	const int32_t wide = (int32_t)xd;
	const int16_t code = (int16_t)(uint16_t)((uint32_t)wide & 0xFFFFu);
	return orc_fast_sin_code(code);
}

float
orc_fast_cos(float fx)
{
	return orc_fast_sin((float)((double)fx + M_PI / 2));   /* sincos.c:39 */
}

/* ------------------------------------------------------------------------- */
/* Per-symbol stages                                                         */
/* ------------------------------------------------------------------------- */

/* dsp/filter.c:46-65: one polyphase bank, oldest sample first, strictly sequential. */
static orc_cf
fir_eval(const orc_consts *c, const orc_state *s, int phase)
{
	const float *h = c->coeffs + (size_t)(c->interp - phase - 1) * (size_t)c->taps;
	orc_cf acc = { 0.0f, 0.0f };
	int pos = s->hidx;
	for (int k = 0; k < c->taps; k++) {
		const float pr = s->hist[pos].re * h[k];
		const float pi = s->hist[pos].im * h[k];
		acc.re = acc.re + pr;
		acc.im = acc.im + pi;
		if (++pos == c->taps) pos = 0;
	}
	return acc;
}

/* dsp/agc.c:13-25 */
static orc_cf
agc(orc_state *s, orc_cf x)
{
	const float keep = 1.0f - 0.001f;
	s->bias.re = s->bias.re * keep + 0.001f * x.re;
	s->bias.im = s->bias.im * keep + 0.001f * x.im;
	x.re = x.re - s->bias.re;
	x.im = x.im - s->bias.im;
	x.re = x.re * s->gain;
	x.im = x.im * s->gain;
	const float mag = hypotf(x.re, x.im);                 /* cabsf, agc.c:21 */
	s->gain = s->gain + 0.0001f * (190.0f - mag);
	if (0.0f > s->gain) s->gain = 0.0f;                   /* MAX(0, gain), agc.c:22 */
	return x;
}

/* NCO phase advance shared by pll_mix / pll_mix_i / pll_mix_q: pll.c:60-61,76-77,93-94 */
static void
nco_advance(orc_state *s)
{
	s->pll_phase = s->pll_phase + s->pll_freq;
	if ((double)s->pll_phase >= TWO_PI_D)
		s->pll_phase = (float)((double)s->pll_phase - TWO_PI_D);
}

/* pll.c:51-64 (and the I-only / Q-only variants :67-97) */
static orc_cf
nco_mix(orc_state *s, orc_cf x)
{
	const float sn = orc_fast_sin(-s->pll_phase);
	const float cs = orc_fast_cos(-s->pll_phase);
	orc_cf y;
	y.im = x.re * sn + x.im * cs;
	/* pll.c:58: "(a) + I*(b)" — I is the complex constant 0+1i, so the real part
	 * is a + 0*b.  Only the sign of an exact zero can differ from plain `a`. */
	y.re = (x.re * cs - x.im * sn) + 0.0f * y.im;
	nco_advance(s);
	return y;
}

/* timing.c:60-87,90-95 */
static void
timing_update(const orc_consts *c, orc_state *s, float q)
{
	const float sp = (s->t_prev < 0) ? -1.0f : 1.0f;      /* utils.h:27 sgn(0)=+1 */
	const float sq = (q < 0) ? -1.0f : 1.0f;
	const float e = sp * q - sq * s->t_prev;
	s->t_prev = q;

	float fd = s->t_freq - c->t_center;
	s->t_phase = (float)((double)s->t_phase - (TWO_PI_D + (double)(c->t_alpha * e)));
	fd = fd - c->t_beta * e;
	if (fd > c->t_maxdev) fd = c->t_maxdev;               /* MIN(maxdev, fd) */
	if (fd < -c->t_maxdev) fd = -c->t_maxdev;             /* MAX(-maxdev, .) */
	s->t_freq = c->t_center + fd;
}

/* pll.c:154-159 */
static float
tanh_lookup(const orc_consts *c, float v)
{
	if (v > 15) return 1.0f;
	if (v < -16) return -1.0f;
	return c->tanh_lut[(int)v + 16];
}

/* pll.c:100-130,143-151 */
static void
pll_update(const orc_consts *c, orc_state *s, float i, float q)
{
	const float e = tanh_lookup(c, i) * q - tanh_lookup(c, q) * i;

	const float ph = s->pll_phase + c->pll_alpha * e;
	s->pll_phase = (float)fmod((double)ph, TWO_PI_D);
	s->pll_freq = s->pll_freq + c->pll_beta * e;

	const float decayed = s->pll_err * (1.0f - 0.001f);
	s->pll_err = (float)((double)decayed + fabs((double)e) * (double)0.001f);
	if (s->pll_err < 85 && !s->locked) {
		s->locked = 1;
		if (!s->locked_once) s->first_lock_symbol = (int64_t)s->n_symbols;
		s->locked_once = 1;
	} else if (s->pll_err > 105 && s->locked) {
		s->locked = 0;
	}

	if (!s->locked)
		s->pll_freq = (float)((double)s->pll_freq + 0.000001 * (double)s->updown);
	if (s->pll_freq >= c->pll_fmax) s->updown = -1;
	else if (s->pll_freq <= -c->pll_fmax) s->updown = 1;
	if (s->pll_freq > c->pll_fmax) s->pll_freq = c->pll_fmax;
	if (s->pll_freq < -c->pll_fmax) s->pll_freq = -c->pll_fmax;
}

/* ------------------------------------------------------------------------- */
/* Per-sample driver: demod.c:24-48 (QPSK) and demod.c:51-91 (OQPSK)         */
/* ------------------------------------------------------------------------- */

int
orc_push(orc_stream *st, float re, float im, orc_cf *out)
{
	const orc_consts *c = &st->c;
	orc_state *s = &st->s;
	int produced = 0;

	/* filter.c:39-43 */
	s->hist[s->hidx].re = re;
	s->hist[s->hidx].im = im;
	s->hidx = (s->hidx + 1) % c->taps;
	s->n_samples++;

	for (int i = 0; i < c->interp; i++) {
		s->t_phase = s->t_phase + s->t_freq;              /* timing.c:34,47 */

		if (!c->oqpsk) {
			if (!(s->t_phase >= TWO_PI_F)) continue;      /* timing.c:37 */
			orc_cf y = fir_eval(c, s, i);
			y = agc(s, y);
			y = nco_mix(s, y);
			timing_update(c, s, y.im);
			pll_update(c, s, y.re, y.im);
			*out = y;
			produced = 1;
		} else {
			if (!(s->t_phase >= (float)s->dual_state * PI_F)) continue;   /* timing.c:50 */
			const int which = s->dual_state;
			s->dual_state = (s->dual_state % 2) + 1;      /* timing.c:52 */
			orc_cf y = fir_eval(c, s, i);
			y = agc(s, y);
			if (which == 1) {                             /* demod.c:66-71 */
				const float sn = orc_fast_sin(-s->pll_phase);
				const float cs = orc_fast_cos(-s->pll_phase);
				s->inphase = y.re * cs - y.im * sn;
				nco_advance(s);
			} else {                                      /* demod.c:72-83 */
				const float sn = orc_fast_sin(-s->pll_phase);
				const float cs = orc_fast_cos(-s->pll_phase);
				const float quad = y.re * sn + y.im * cs;
				nco_advance(s);
				out->re = s->inphase + 0.0f * quad;           /* demod.c:78: inphase + I*quad */
				out->im = quad;
				timing_update(c, s, quad);
				pll_update(c, s, s->inphase, quad);
				produced = 1;
			}
		}
	}
	/* symbols are counted as the caller sees them: one per input sample that returns one (demod.c:33-47, 62-90 overwrite `*sample`
	 * and `ret` when the clock fires twice inside a sample: the earlier symbol is gone, and pll.c's lock flags are read by
	 * main.c:312 against this count) */
	if (produced) s->n_symbols++;
	return produced;
}

/* main.c:305-306: MAX(-127, MIN(127, v/2)) then float->int8 truncation. */
int8_t
orc_quantise(float v)
{
	float h = v / 2.0f;
	if (!(127.0f < h)) { /* MIN(127, h) keeps h unless 127 < h */ } else h = 127.0f;
	if (-127.0f > h) h = -127.0f;
	return (int8_t)h;
}

/* wavfile.c:58-69 */
static inline void
load_sample(const void *iq, size_t idx, int fmt, float *re, float *im)
{
	switch (fmt) {
	case ORC_FMT_U8: {
		const uint8_t *p = (const uint8_t *)iq + 2 * idx;
		*re = (float)((int)p[0] - 128);
		*im = (float)((int)p[1] - 128);
		break; }
	case ORC_FMT_S16: {
		const int16_t *p = (const int16_t *)iq + 2 * idx;
		*re = (float)p[0];
		*im = (float)p[1];
		break; }
	default: {
		const float *p = (const float *)iq + 2 * idx;
		*re = p[0];
		*im = p[1];
		break; }
	}
}

long
orc_run(orc_stream *st, const void *iq, size_t n, int fmt,
        int8_t *soft, size_t soft_cap,
        orc_trace *trace,
        orc_lock_event *events, size_t events_cap, size_t *n_events)
{
	if (fmt != ORC_FMT_U8 && fmt != ORC_FMT_S16 && fmt != ORC_FMT_F32) return -1;
	size_t produced = 0, nev = 0;

	for (size_t k = 0; k < n; k++) {
		float re, im;
		orc_cf y;
		load_sample(iq, k, fmt, &re, &im);
		const int was_locked = st->s.locked;
		const uint64_t sym_index = st->s.n_symbols;
		if (!orc_push(st, re, im, &y)) continue;
		if (produced >= soft_cap) return -1;
		if (soft) {
			soft[2 * produced]     = orc_quantise(y.re);
			soft[2 * produced + 1] = orc_quantise(y.im);
		}
		if (trace) {
			orc_trace *t = &trace[produced];
			t->sample_index = st->s.n_samples - 1;
			t->re = y.re;
			t->im = y.im;
			t->pll_freq = st->s.pll_freq;
			t->omega = st->s.t_freq;
			t->gain = st->s.gain;
			t->locked = st->s.locked;
		}
		if (st->s.locked != was_locked) {
			if (events && nev < events_cap) {
				events[nev].symbol = sym_index;
				events[nev].locked = st->s.locked;
			}
			nev++;
		}
		produced++;
	}
	if (n_events) *n_events = nev;
	return (long)produced;
}

/* main.c:285-329 with wavfile.c:51-80's whole-buffer reads. */
long
orc_file_model(orc_stream *st, const uint8_t *data, size_t nbytes, int fmt,
               uint8_t *out, size_t out_cap)
{
	enum { FILE_BUF = 32768, RING = 512 };
	if (fmt != ORC_FMT_U8 && fmt != ORC_FMT_S16 && fmt != ORC_FMT_F32) return -1;
	const size_t bytes_per_sample = 2 * (size_t)fmt / 8;
	const size_t n = (nbytes / FILE_BUF) * (FILE_BUF / bytes_per_sample);   /* wavfile.c:55 */

	int8_t ring[2 * RING];
	memset(ring, 0, sizeof(ring));                    /* static storage, main.c:34 */
	unsigned ring_idx = 0;
	size_t written = 0;

	for (size_t k = 0; k < n; k++) {
		float re, im;
		orc_cf y;
		load_sample(data, k, fmt, &re, &im);
		if (!orc_push(st, re, im, &y)) continue;
		ring[ring_idx++] = orc_quantise(y.re);
		ring[ring_idx++] = orc_quantise(y.im);
		if (ring_idx >= 2 * RING) {
			ring_idx = 0;
			if (st->s.locked_once) {                  /* main.c:312 */
				if (written + sizeof(ring) > out_cap) return -1;
				memcpy(out + written, ring, sizeof(ring));
				written += sizeof(ring);
			}
		}
	}
	/* main.c:321: fwrite(ring, ring_idx, 2, f) == 2*ring_idx bytes */
	if (2 * (size_t)ring_idx > sizeof(ring)) return -1;   /* reference reads out of bounds here */
	if (written + 2 * (size_t)ring_idx > out_cap) return -1;
	memcpy(out + written, ring, 2 * (size_t)ring_idx);
	written += 2 * (size_t)ring_idx;
	return (long)written;
}
