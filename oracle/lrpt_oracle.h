/*
 * lrpt_oracle — CPU restatement of the meteor_demod IQ-in -> soft-QPSK-out path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity checker for the HIP path and the
 * "port" CPU baseline of bench.py.  Nothing under meteor_demod_amd/ or host/ may
 * include, link or call it; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * Parity status: PINNED.  oracle/Makefile builds the reference's own sources
 * (from /root/reference, strict -ffp-contract=off flags) into oracle/_ref/ and
 * tests/golden/make_golden.py proves this restatement byte-identical to it on
 * the committed fixtures (soft symbols, per-symbol float traces, RRC tables,
 * fast_sin over every int16 code).  See tests/golden/MANIFEST.json.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference repo root).
 */
#ifndef LRPT_ORACLE_H
#define LRPT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Input sample formats: the reference's --bps values (wavfile.c:58-69). */
enum { ORC_FMT_U8 = 8, ORC_FMT_S16 = 16, ORC_FMT_F32 = 32 };

/* The eight arguments of demod_init (demod.h:29) as given on the command line
 * (main.c:64-78,187) before any derivation. */
typedef struct {
	float pll_bw;        /* -b, default 1 (demod.h:15)                         */
	float sym_bw;        /* fixed SYM_BW 0.00005 (demod.h:14, main.c:187)      */
	int   samplerate;    /* -s / WAV header                                    */
	int   symrate;       /* -r, default 72000                                  */
	int   interp;        /* -O, default 5                                      */
	int   rrc_order;     /* -f, default 32                                     */
	int   oqpsk;         /* -m oqpsk                                           */
	float freq_max;      /* -d after main.c:136 scaling; negative => default   */
} orc_params;

/* Loop constants produced by demod_init -> pll_init/timing_init/filter_init_rrc. */
typedef struct {
	int   interp, taps, oqpsk;
	float pll_alpha, pll_beta, pll_fmax;
	float t_alpha, t_beta, t_center, t_maxdev;
	float osf;
	float tanh_lut[32];
	float *coeffs;           /* interp * taps, bank-major (filter.c:20) */
} orc_consts;

typedef struct { float re, im; } orc_cf;

/* Complete mutable state of one stream. */
typedef struct {
	orc_cf  *hist;           /* taps entries, circular (filter.h:6)          */
	int      hidx;           /* filter.h:10                                   */
	float    gain;           /* agc.c:9                                       */
	orc_cf   bias;           /* agc.c:10                                      */
	float    pll_phase, pll_freq, pll_err;   /* pll.c:16,20                   */
	int      locked, locked_once;            /* pll.c:20                      */
	int      updown;                         /* pll.c:112                     */
	float    t_phase, t_freq, t_prev;        /* timing.c:13-14                */
	int      dual_state;                     /* timing.c:43                   */
	float    inphase;                        /* demod.c:54                    */
	uint64_t n_samples;      /* samples consumed so far                       */
	uint64_t n_symbols;      /* symbols emitted so far                        */
	int64_t  first_lock_symbol; /* index of the symbol whose PLL update set
	                               locked_once, -1 if never                   */
} orc_state;

typedef struct {
	orc_consts c;
	orc_state  s;
} orc_stream;

/* Per-symbol float trace record (what the reference exposes through its getters). */
typedef struct {
	uint64_t sample_index;   /* index of the input sample the symbol fired on */
	float    re, im;         /* demodulated symbol before quantisation        */
	float    pll_freq;       /* pll_get_freq()  pll.c:46                      */
	float    omega;          /* mm_omega()      timing.c:29                   */
	float    gain;           /* agc_get_gain()  agc.c:28                      */
	int32_t  locked;         /* pll_get_locked() pll.c:47                     */
} orc_trace;

/* Lock transition event: symbol index (absolute) and new lock state. */
typedef struct { uint64_t symbol; int32_t locked; } orc_lock_event;

/* --- init-time pieces ---------------------------------------------------- */
float orc_rrc_coeff(int stage_no, unsigned taps, float osf, float alpha);
int   orc_consts_init(orc_consts *c, const orc_params *p);
void  orc_consts_free(orc_consts *c);

int   orc_stream_init(orc_stream *st, const orc_params *p);
void  orc_stream_free(orc_stream *st);

/* --- hot path ------------------------------------------------------------ */
float orc_fast_sin(float fx);
float orc_fast_cos(float fx);
float orc_fast_sin_code(int16_t code);   /* polynomial part only, for the 65536-code KAT */

/* Feed one sample; returns 1 and fills *out when a symbol was produced. */
int   orc_push(orc_stream *st, float re, float im, orc_cf *out);

/* Quantise one soft component exactly as main.c:305-306. */
int8_t orc_quantise(float v);

/*
 * Block API: consume n IQ samples of format fmt from iq, append int8 I,Q pairs
 * to soft (capacity soft_cap SYMBOLS).  trace/events may be NULL.
 * Returns number of symbols produced, or -1 on overflow of soft_cap / bad fmt.
 */
long  orc_run(orc_stream *st, const void *iq, size_t n, int fmt,
              int8_t *soft, size_t soft_cap,
              orc_trace *trace,
              orc_lock_event *events, size_t events_cap, size_t *n_events);

/*
 * File-level model of main.c:285-329 + wavfile.c:51-80: input is consumed in
 * whole 32768-byte reads, 512-symbol chunks are written only once locked_once,
 * final flush writes 2*ring_idx bytes.  `data` points at the first sample byte
 * (after any WAV header).  Output is appended to out (capacity out_cap bytes);
 * returns bytes written or -1 (overflow; or ring_idx > 512 at EOF, where the
 * reference reads out of bounds — undefined, not modelled).
 */
long  orc_file_model(orc_stream *st, const uint8_t *data, size_t nbytes, int fmt,
                     uint8_t *out, size_t out_cap);

/* Convenience for ctypes-based tests: heap-allocated stream. */
orc_stream *orc_stream_new(const orc_params *p);
void        orc_stream_delete(orc_stream *st);

#ifdef __cplusplus
}
#endif
#endif
