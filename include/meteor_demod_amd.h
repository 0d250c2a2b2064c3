/*
 * meteor_demod_amd.h — C-ABI of the MI355X-native LRPT demodulator.
 *
 * Drop-in boundary for ONE path of dbdexter-dev/meteor_demod: IQ samples in ->
 * 8-bit soft QPSK/OQPSK symbols out.  Citations are reference file:line.
 *
 * The reference's seam is a per-sample function pointer
 *     int (*demod)(float complex *sample)            main.c:22-29,72,104,304
 * bound to demod_qpsk (demod.c:24) or demod_oqpsk (demod.c:51) and configured by
 *     void demod_init(float pll_bw, float sym_bw, int samplerate, int symrate,
 *                     int interp_factor, int rrc_order, int oqpsk,
 *                     float freq_max)                 demod.h:29, demod.c:8-15
 * with state in file-static globals.  A GPU needs batches and explicit state, so
 * the same seam is exported here as a context object that demodulates BLOCKS of
 * samples for N independent streams (recordings, or tiles of one recording):
 *
 *   reference                                  this library
 *   ---------                                  ------------
 *   demod_init(...)            demod.h:29      mdemod_create(&params, &ctx)
 *   demod_deinit()             demod.h:34      mdemod_destroy(ctx)
 *   demod_qpsk / demod_oqpsk   demod.h:42,50   mdemod_process_device[_uniform], mdemod_process_host
 *   ring quantise  main.c:305-306              done on device: soft = int8 I,Q pairs
 *   pll_get_freq / pll_get_locked /
 *   pll_did_lock_once          pll.h:20,27,34  mdemod_status.{pll_freq,locked,locked_once}
 *   mm_omega                   timing.h:32     mdemod_status.omega
 *   agc_get_gain               agc.h:18        mdemod_status.gain
 *   lock gating    main.c:312                  mdemod_status.first_lock_symbol + lock events
 *
 * Results are bit-identical, stream by stream, to the reference compiled with
 * -ffp-contract=off (the only compiler-independent build of it).
 * State persists in the context between calls exactly as the reference's
 * globals persist between per-sample calls, so feeding a recording in blocks of
 * any size gives the same bytes as feeding it in one call.
 *
 * What "drop-in" means here, and what it does not:
 *   - The per-sample symbols `int demod_qpsk(float complex *)` / `demod_oqpsk` (demod.h:42,50) and the getter symbols
 *     pll_get_freq / pll_get_locked / pll_did_lock_once / mm_omega / agc_get_gain (pll.h:20,27,34, timing.h:32, agc.h:18) are
 *     NOT exported (SURVEY 8(b) suggests them as a shim over a CPU backend; this library has no CPU backend by construction, and a
 *     symbol returned in place per call would be a kernel launch per sample).  An UNMODIFIED main.c therefore does not link
 *     against this library: the binding is the block call - the patch to main.c's thread_process shown in INTEGRATION.md 1, or
 *     the C host shipped in host/ with the reference's option table.
 *   - Exact mode on ONE stream is one wavefront: about 5 Msamples/s on configs[1] (r05), SLOWER than the reference on one host core
 *     (about 26 Msamples/s on the bench host).  The GPU is meant to be used on batches of streams (mdemod_process_*, thousands
 *     of recordings or tiles at 250 Gsamples/s) or on one long recording through mdemod_demodulate_recording (`--tiled`:
 *     overlapped tiles, agreement with the serial run at the reference's own perturbation floor, not bit-exact).
 *
 * Plain C: pointers and sizes only.  Device pointers are raw HIP device
 * addresses; `hip_stream` is a hipStream_t passed as void* (NULL = default).
 * There is no CPU fallback: every entry fails with MDEMOD_ERR_HIP if no gfx950
 * device / kernel image is available.
 */
#ifndef METEOR_DEMOD_AMD_H
#define METEOR_DEMOD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDEMOD_ABI_VERSION 5    /* 5 (r06): + mdemod_last_error, + mdemod_fanin_peer; the library no longer writes to stderr.  4 (r05): + mdemod_pin_host_buffer / mdemod_unpin_host_buffer; settings with a non-finite RRC tap refused */

/* Error codes (the reference surfaces none: demod_init returns void and drops
 * filter_init_rrc's status, demod.c:14).  Nothing in the library calls exit() or abort(), and no C++ exception leaves it: an
 * allocation or a thread that the host refuses comes back as MDEMOD_ERR_NOMEM from the entry it happened in. */
enum {
	MDEMOD_OK            =  0,
	MDEMOD_ERR_PARAM     = -1,   /* bad argument / unsupported configuration */
	MDEMOD_ERR_NOMEM     = -2,   /* host or device allocation failed         */
	MDEMOD_ERR_HIP       = -3,   /* HIP runtime / launch failure             */
	MDEMOD_ERR_OVERFLOW  = -4,   /* soft-symbol capacity too small           */
	MDEMOD_ERR_RANGE     = -5    /* stream index out of range                */
};
/* What went wrong, in words, when a code alone does not say it: which setting was refused and why, which HIP call failed with
 * which runtime message.  The library itself prints nothing (it may sit behind a caller's own terminal UI, as the reference's
 * demodulator sits behind tui.c); only with MDEMOD_DEBUG in the environment is the same text also written to stderr.
 * The text belongs to the CALLING THREAD's most recent entry that returned < 0 ("" when that entry had nothing to add to its code,
 * and after any entry that succeeded); it stays valid until that thread calls into the library again.  Never NULL.
 * (The reference has no counterpart: demod_init returns void, demod.c:14 drops filter_init_rrc's status.) */
const char *mdemod_last_error(void);

/* Defaults of the reference CLI (demod.h:8-15). */
#define MDEMOD_DEFAULT_SYM_RATE   72000
#define MDEMOD_DEFAULT_RRC_ORDER  32
#define MDEMOD_DEFAULT_INTERP     5
#define MDEMOD_DEFAULT_SYM_BW     0.00005f
#define MDEMOD_DEFAULT_PLL_BW     1.0f

typedef struct mdemod_ctx mdemod_ctx;

/* The eight demod_init arguments (same meaning, same order: demod.h:17-29),
 * plus what the GPU boundary needs in addition. */
typedef struct {
	float    pll_bw;        /* -b  carrier loop bandwidth                (main.c:88)  */
	float    sym_bw;        /* SYM_BW, fixed 5e-5 upstream               (main.c:187) */
	int32_t  samplerate;    /* -s / WAV header                           (main.c:121) */
	int32_t  symrate;       /* -r                                        (main.c:118) */
	int32_t  interp_factor; /* -O                                        (main.c:109) */
	int32_t  rrc_order;     /* -f  (taps = 2*order+1)                    (main.c:97)  */
	int32_t  oqpsk;         /* -m oqpsk                                  (main.c:103) */
	float    freq_max;      /* -d already scaled by 2*pi/symrate (main.c:136);
	                           negative selects the 0.3 default (pll.c:31)            */
	int32_t  bps;           /* input format: 8 (u8 offset-128), 16 (s16), 32 (f32):
	                           wavfile.c:58-69                                        */
	int32_t  device;        /* HIP device ordinal                                     */
	uint32_t n_streams;     /* independent streams held by this context               */
	uint32_t reserved;      /* 0: the library picks the kernel (what a caller passes).  MDEMOD_FLAG_* below pin a variant: the
	                           test-suite runs every golden through all of them.  The library reads one environment variable, MDEMOD_PACK_THREADS
	                           (threads that pack host buffers in mdemod_process_host, default 12; no result depends on it). */
} mdemod_params;

/* mdemod_params.reserved (diagnosis and tests only; every variant produces the same bytes) */
#define MDEMOD_FLAG_KERNEL_MASK 0x3u    /* 0 or 3 = v3 rotating windows where they apply (the default), 1 = v1 LDS ring; 2 = the retired v2: MDEMOD_ERR_PARAM */
#define MDEMOD_FLAG_LAT_OFF     0x4u    /* never the wave-per-stream (latency) kernel, however few the streams */
#define MDEMOD_FLAG_LAT_ON      0x8u    /* always, when the configuration fits it */
#define MDEMOD_FLAG_V2_PACKED   0x10u   /* (retired with the v2 kernel in round 4: ignored) */
#define MDEMOD_FLAG_NO_CLOCK_JUMP 0x20u /* the symbol clock's long runs one rounded addition at a time, not in closed form (csrc/clock_jump.h): A/B only */

/* Value snapshot of one stream after a call (replaces the reference's racy
 * getters polled from the UI thread, main.c:231-237,250-258). */
typedef struct {
	uint64_t n_samples;          /* samples consumed since create/reset           */
	uint64_t n_symbols;          /* symbols emitted since create/reset            */
	int64_t  first_lock_symbol;  /* symbol whose PLL update set locked_once; -1   */
	uint32_t symbols_this_call;  /* symbols written to soft by the last call      */
	uint32_t lock_events_this_call; /* lock transitions in the last call (may
	                                exceed MDEMOD_MAX_LOCK_EVENTS; extra dropped) */
	float    pll_freq;           /* pll_get_freq()      rad/symbol                */
	float    omega;              /* mm_omega()          rad/interpolated sample   */
	float    gain;               /* agc_get_gain()                                */
	int32_t  locked;             /* pll_get_locked()                              */
	int32_t  locked_once;        /* pll_did_lock_once()                           */
	int32_t  overflow;           /* 1 if soft capacity was exceeded (symbols past
	                                the capacity are dropped, state still advances)*/
} mdemod_status;

#define MDEMOD_MAX_LOCK_EVENTS 32
typedef struct {
	uint64_t symbol;             /* absolute symbol index of the transition       */
	int32_t  locked;             /* new lock state                                */
	int32_t  pad;
} mdemod_lock_event;

/* Complete loop state of one stream, for chaining tiles, seeding
 * tiles with converged state, checkpoints and tests.  History is exchanged
 * separately (mdemod_get_history / mdemod_set_history). */
typedef struct {
	float    agc_gain, agc_bias_re, agc_bias_im;      /* agc.c:9-10              */
	float    pll_phase, pll_freq, pll_err;            /* pll.c:16,20             */
	int32_t  pll_locked, pll_locked_once, pll_updown; /* pll.c:20,112            */
	float    t_phase, t_freq, t_prev;                 /* timing.c:13-14          */
	int32_t  t_dual_state;                            /* timing.c:43             */
	float    oqpsk_inphase;                           /* demod.c:54              */
	uint64_t n_samples, n_symbols;
	int64_t  first_lock_symbol;
} mdemod_stream_state;

/* ---- lifecycle ----------------------------------------------------------- */

uint32_t mdemod_abi_version(void);
const char *mdemod_strerror(int code);
/* Optional: brings up the HIP runtime on `device` and loads the library's code objects (0.1-0.2 s in a new process) - e.g. from
 * a second thread while the host reads its input file.  Every other entry does this by itself when it has not happened yet. */
int  mdemod_init_device(int device);
/* GPUs this process sees (hipGetDeviceCount); 0 when there is none or the runtime cannot start. */
int  mdemod_device_count(void);

/* Replaces demod_init (demod.c:8-15).  Derives loop constants, RRC taps and the
 * tanh LUT on the host with the reference's exact mixed float/double
 * expressions, uploads them, allocates per-stream state.
 * MDEMOD_ERR_PARAM: -O outside 1..64, -f outside 1..256, a sample format other than 8 / 16 / 32 bits, fewer than a quarter of an
 * input sample per firing, samplerate x interp or 2 x symrate beyond int (demod.c:12-13 multiplies in int), a carrier range of
 * 6 rad per symbol or more, and settings whose RRC table has a tap that is not finite (filter.c:86-93 divides by zero where
 * samples-per-symbol x -O / 2.4 lands on a tap - 230.4 kS/s OQPSK 80k with -O 5, 1.08 MS/s with -O 4 -; the reference's output is
 * undefined there; another -O is the way out, and mdemod_last_error() says so). */
int  mdemod_create(const mdemod_params *params, mdemod_ctx **out);
/* Replaces demod_deinit (demod.c:18-21). */
void mdemod_destroy(mdemod_ctx *ctx);
/* Put every stream back into the reference's power-on state. */
int  mdemod_reset(mdemod_ctx *ctx, void *hip_stream);

/* Soft-symbol capacity (in SYMBOLS) that is always enough for n input samples: one symbol per
 * sample is the hard bound of the reference's per-sample loop (demod.c:33-47); the nominal
 * n*symrate/samplerate is exceeded while the symbol clock drains a large phase excursion. */
uint64_t mdemod_max_symbols(const mdemod_ctx *ctx, uint64_t n_samples);

/* Row pitch (in SYMBOLS, a multiple of 8) that holds what n input samples produce at the nominal symbol rate with 1 % slack:
 * what a consumer of many streams wants to move or keep (0.63 B per input sample at 72k in 230 kS/s) instead of the hard
 * bound above.  NOT a bound: rows are truncated to it by mdemod_compact_soft (overflow is reported by the process call
 * against the capacity it was given). */
uint64_t mdemod_nominal_pitch(const mdemod_ctx *ctx, uint64_t n_samples);
/* Copy the symbols the LAST process call wrote (per-stream counts kept by the context) from rows of soft_stride_symbols
 * to rows of out_pitch_symbols (both multiples of 8; 16-byte moves): the compaction in front of a PCIe copy or of the
 * RCCL fan-in of soft-symbol buffers (main.c:305-315 writes symbols back to back; the hard-bound pitch is a GPU artefact). */
int  mdemod_compact_soft(mdemod_ctx *ctx, const int8_t *soft_dev, uint64_t soft_stride_symbols,
                         int8_t *out_dev, uint64_t out_pitch_symbols, void *hip_stream);
/* The fan-in of soft-symbol buffers for a host that drives several GPUs from ONE process (host/meteor_demod_amd.c --devices: one worker
 * thread and one context per GPU; the reference has one demodulator thread, main.c:218, and nothing to gather).  The symbols of src's LAST
 * process call, compacted as by mdemod_compact_soft to rows of pitch_symbols, are written by src's GPU straight into memory of dst_device:
 * rows first_row .. first_row + n_streams - 1 of dst_soft_dev, and - dst_counts_dev may be NULL - the per-stream symbol counts to
 * dst_counts_dev[first_row ...].  The compaction kernel's own stores cross xGMI (peer access from src's device to dst_device is switched
 * on at the first call; dst_device may be src's own, then this is mdemod_compact_soft plus the counts): no staging buffer, no second copy,
 * no RCCL - that one serves the one-process-per-GPU layout (meteor_demod_amd.sharding.fanin_soft).  MDEMOD_ERR_HIP when the two devices have
 * no peer access to each other.  Asynchronous on hip_stream, a stream of src's device; the caller orders it against the consumer on dst_device. */
int  mdemod_fanin_peer(mdemod_ctx *src, const int8_t *soft_dev, uint64_t soft_stride_symbols, int dst_device,
                       int8_t *dst_soft_dev, uint64_t pitch_symbols, uint64_t first_row, uint32_t *dst_counts_dev, void *hip_stream);

/* ---- the hot path (replaces the main.c:303-306 loop body) ---------------- */

/*
 * All streams share one layout: stream s reads n_samples IQ samples starting at
 * iq_dev + s*iq_stride_samples (in units of one IQ sample) and writes its int8
 * I,Q pairs at soft_dev + s*soft_stride_symbols*2.  Asynchronous on hip_stream.
 */
int  mdemod_process_device_uniform(mdemod_ctx *ctx,
                                   const void *iq_dev, uint64_t iq_stride_samples,
                                   uint32_t n_samples,
                                   int8_t *soft_dev, uint64_t soft_stride_symbols,
                                   uint32_t soft_cap_symbols,
                                   void *hip_stream);

/*
 * Ragged batch: per-stream sample offsets (from iq_dev, in IQ samples) and
 * counts, both arrays of n_streams entries in DEVICE memory.  count 0 is legal.
 */
int  mdemod_process_device(mdemod_ctx *ctx,
                           const void *iq_dev,
                           const uint64_t *iq_offset_dev, const uint32_t *n_samples_dev,
                           int8_t *soft_dev, uint64_t soft_stride_symbols,
                           uint32_t soft_cap_symbols,
                           void *hip_stream);

/*
 * Host-buffer convenience (PCIe inclusive, synchronous): iq_host[s] points at
 * n_samples[s] IQ samples of stream s; soft_host[s] receives up to soft_cap[s]
 * symbols (2 bytes each); n_symbols[s] is set to the number produced.
 */
int  mdemod_process_host(mdemod_ctx *ctx,
                         const void *const *iq_host, const uint32_t *n_samples,
                         int8_t *const *soft_host, const uint32_t *soft_cap,
                         uint32_t *n_symbols);

/*
 * Optional, for a caller that reads into the same buffer block after block (main.c:303 does: wavfile.c's 32 KiB buffer; the C host
 * here: 4 MiB per file): pins [base, base + bytes) for the HIP runtime (hipHostRegister; ~60 ms per GB, once).  mdemod_process_host
 * then copies a batch that lies inside a pinned range straight from the caller's pages - when every stream's block has the same
 * length and the blocks sit one stride apart (iq_host[s] = iq_host[0] + s * stride: a batch read into one buffer) - instead of
 * staging it through the library's own pinned ring with the CPU.  Any other layout, and anything outside the pinned ranges, takes the
 * staged path; results never depend on it.  The caller keeps the range mapped until mdemod_unpin_host_buffer (with the same base)
 * or mdemod_destroy, which unpins what is left.  MDEMOD_ERR_PARAM for a range that overlaps one pinned already through this call.
 * Memory that is pinned anyway (from hipHostMalloc, or registered by the caller) is accepted and left as it is - when that
 * pinning holds ALL of [base, base + bytes); one that starts at `base` and stops short of the end is MDEMOD_ERR_PARAM (r06: the direct
 * copy would touch pages nobody locked), and such rows simply keep taking the staged path.
 */
int  mdemod_pin_host_buffer(mdemod_ctx *ctx, const void *base, size_t bytes);
int  mdemod_unpin_host_buffer(mdemod_ctx *ctx, const void *base);

/* ---- status / state (synchronise with the last call on hip_stream first) -- */

int  mdemod_get_status(mdemod_ctx *ctx, uint32_t first, uint32_t count,
                       mdemod_status *out, void *hip_stream);
int  mdemod_get_lock_events(mdemod_ctx *ctx, uint32_t stream,
                            mdemod_lock_event *out, uint32_t cap, uint32_t *n,
                            void *hip_stream);
int  mdemod_get_state(mdemod_ctx *ctx, uint32_t stream, mdemod_stream_state *out,
                      void *hip_stream);
/* MDEMOD_ERR_PARAM for a carrier state the reference's loop cannot hold: |pll_phase| + |pll_freq| >= 12.5 (pll.c:113
 * keeps the phase inside (-2pi, 2pi), pll.c:126-128 the frequency word inside +-fmax), or a clock word t_freq further than
 * centre / 4096 from the centre 2 pi symrate / (samplerate * interp_factor) (timing.c:80-86 keeps it there; a state exported
 * by mdemod_get_state always passes). */
int  mdemod_set_state(mdemod_ctx *ctx, uint32_t stream, const mdemod_stream_state *in,
                      void *hip_stream);
/* Filter history: the last mdemod_history_len() input samples, oldest first,
 * as float I,Q pairs (filter.h:6). */
uint32_t mdemod_history_len(const mdemod_ctx *ctx);
int  mdemod_get_history(mdemod_ctx *ctx, uint32_t stream, float *iq_pairs, void *hip_stream);
int  mdemod_set_history(mdemod_ctx *ctx, uint32_t stream, const float *iq_pairs, void *hip_stream);

/* ---- overlapped tiles of ONE recording ------------------------------------ */

/* ONE recording on many lanes (NOTEBOOK.md 3.1).  The reference runs a recording as one serial recurrence (main.c:303-316);
 * here only its head runs serially (the "pilot": from the reference's power-on state until the carrier loop has locked and
 * settled - those symbols ARE the reference's symbols, lock gate included), the rest as tiles, one lane each:
 *   estimates the carrier (4th-power spectrum, de-chirped) and the symbol clock (symbol-rate line) along the whole recording, on
 *             windows side by side: taken on a host thread and a stream of the call's own while the head runs;
 *   acquire   every tile starts (acquire + frame + settle) samples early from the pilot's loop state, the carrier and clock
 *             read off those curves at its position and its gain estimate;
 *   re-seed   after `acquire_samples` the two loop integrators (pll.c:115 freq, timing.c:84 freq) are put back on their
 *             seeds: the acquisition transient kicks them and they need 8-16 k symbols to come back on their own;
 *   frame     a Costas loop locks on one of four rotations.  After `frame_samples` more the rotation of every tile relative
 *             to its predecessor follows from the two NCO phases and the carrier estimate between them (dead reckoning:
 *             theta_b - theta_a - f * steps, rounded to quarter turns); prefix-summed from the pilot and undone IN THE STATE
 *             (mdemod_rotate_carrier), so that every tile settles in the rotation the serial run is in (the timing detector
 *             reads only the Q rail, timing.c:65-66: a tile that settles a quarter turn off tracks the other rail's noise);
 *   settle    `settle_samples` more (not emitted), then the body (emitted);
 *   seams     the last symbols before each tile's body were demodulated by its predecessor too: an int8 correlation gives
 *             the residual rotation (expected 0; otherwise the tile and its successors are repaired: 180 degrees on the
 *             output, odd quarter turns by one more settle + body pass from the saved post-acquisition state) and the
 *             one-symbol duplicate / gap at the seam.
 * Tile 0 is the exact continuation of the pilot.  iq_dev: n_samples IQ samples in the format of params->bps, in device
 * memory; soft_dev: device buffer for soft_cap_symbols int8 pairs.  params->n_streams is ignored.  Synchronous on hip_stream (the
 * estimates run on one more stream and host thread, both gone when the call returns). */
typedef struct {
	uint32_t tile_samples;          /* body samples per tile; 0 = automatic: 8 192 ... 41 072 symbols worth: ~1000 tiles (one per
	                                   wave: latency kernel) while that fits, else as short as keeps them within 131 072 lanes */
	uint32_t acquire_samples;       /* 0xFFFFFFFF = 2 000 symbols worth                                                  */
	uint32_t frame_samples;         /* 0xFFFFFFFF = 1 500 symbols worth                                                  */
	uint32_t settle_samples;        /* 0xFFFFFFFF = 32 000 symbols worth, 24 000 for the tiles that start within 50 000 symbols of the
	                                   hand-over (the serial run itself is still drifting in there: a longer lead agrees with it less) */
	uint32_t pilot_block;           /* pilot granularity in samples               (65536)  */
	uint32_t pilot_margin_symbols;  /* symbols between the first lock and the hand-over; 0xFFFFFFFF = 15 000 (OQPSK: 20 000); 20 000 (30 000) when the tiles take the pilot's clock or carrier word */
	uint64_t max_pilot_samples;     /* give up waiting for lock after this many; 0xFFFFFFFFFFFFFFFF = 1 500 000 symbols worth: the reference's
	                                   sweep (1e-6 rad per symbol, up first, pll.c:125) has been to +fmax and down to -fmax by then */
	uint32_t match_symbols;         /* symbols compared across a seam; at least 32 are used (192) */
	int32_t  repair;                /* 1: tiles whose seam shows an odd residual rotation run settle + body again (1)    */
	uint32_t carrier_seed;          /* 1: every tile from its own 4th-power spectrum (follows Doppler); 0: all tiles from
	                                   the pilot's carrier estimate (dead reckoning then rarely holds: repair does the work) (1) */
	uint32_t clock_seed;            /* 0: every tile's symbol clock from its own spectral line (mdemod_estimate_clock: follows the
	                                   Doppler on the clock, good to a tenth of the loop's own wander); 1: the pilot's omega for all (0) */
	int32_t  debug;                 /* 0; 1: stage timings and a per-seam trace of the odd seams on stderr, 2: every seam, 3: and one line per tile (seeds, end state) (0) */
	int32_t  debug_tile;            /* with debug: also trace the framing of the tiles around this index; -1 = none (-1) */
} mdemod_recording_opts;

typedef struct {
	uint64_t n_symbols;             /* symbols written to soft_dev                           */
	uint64_t pilot_samples;         /* samples [0, pilot_samples) were demodulated serially  */
	uint64_t pilot_symbols;         /* ... and produced this many symbols (bit-exact)        */
	uint64_t exact_symbols;         /* leading symbols that are the reference's own (pilot + its exact continuation, tile 0) */
	int64_t  first_lock_symbol;     /* as mdemod_status, from the pilot                      */
	uint64_t samples_demodulated;   /* kernel work including acquisition, settling, repairs  */
	uint32_t n_tiles;
	uint32_t tile_samples;          /* body samples per tile actually used                   */
	uint32_t weak_seams;            /* seams whose correlation was too weak to trust         */
	uint32_t seam_fixes;            /* one-symbol duplicates / gaps repaired                 */
	int32_t  pilot_locked;          /* 0: the head gave up unlocked; 1: locked; 2: handed over on a FALSE lock of the reference - its PLL reports
	                                   lock, but its carrier word is still more than 100 Hz from the signal's own spectral line after
	                                   max_pilot_samples (the head does not hand over on such a lock while it has patience left: the
	                                   reference's OQPSK loop and any float recording's first seconds produce them).  From there on the
	                                   reference's output is not a demodulation of this signal, the tiles' is: the two cannot agree */
	uint32_t weak_carrier_tiles;    /* carrier_seed=1: carrier windows (~20 000 symbols each, side by side) without a clear spectral line: the curve is interpolated across them */
	double   pilot_seconds;         /* wall time of the serial head                          */
	double   tiles_seconds;         /* wall time of everything after it                      */
	uint32_t frame_misses;          /* seams where dead reckoning put the tile in another rotation than the correlation found */
	uint32_t repaired_tiles;        /* tiles that ran settle + body a second time (odd residual rotation)               */
	uint32_t rotation_jumps;        /* seams that still show an unexpected rotation after the repairs (a cycle slip inside a
	                                   body): the output is rotated from there on, like a serial run after a cycle slip.
	                                   (A re-run tile that sits half a turn off on both its seams is not one: its output is
	                                   turned, which is exact, and continuous with its neighbours.) */
	float    frame_residual_rms;    /* rad: dead-reckoned minus measured NCO phase, after removing the quarter turns (0.785 = limit) */
	uint32_t odd_tiles_kept;        /* tiles left an odd number of quarter turns off because they were too few to be worth a repair
	                                   pass (< 0.5 % of the tiles): output turned (decisions exact), soft values on the other rail's timing */
	uint32_t weak_clock_tiles;      /* clock_seed=0: tiles with no clear symbol-rate line in any window around them: they start from the pilot's omega */
} mdemod_recording_report;

void mdemod_recording_default_opts(mdemod_recording_opts *opts);
/* Stream ordering: synchronous for the host (returns when soft_dev is complete), ordered on the device AFTER everything already
 * queued on hip_stream - a caller may produce iq_dev asynchronously on hip_stream (kernel, cast, non-blocking copy) and call this
 * without synchronising first: the internal streams (serial head, carrier / clock estimators) wait on an event recorded on
 * hip_stream at entry.  Work queued on OTHER streams is the caller's to order. */
int  mdemod_demodulate_recording(const mdemod_params *params, const mdemod_recording_opts *opts,
                                 const void *iq_dev, uint64_t n_samples,
                                 int8_t *soft_dev, uint64_t soft_cap_symbols,
                                 mdemod_recording_report *report, void *hip_stream);

/* Same with host buffers (PCIe inclusive, synchronous): the C CLI's --tiled mode. */
int  mdemod_demodulate_recording_host(const mdemod_params *params, const mdemod_recording_opts *opts,
                                      const void *iq_host, uint64_t n_samples,
                                      int8_t *soft_host, uint64_t soft_cap_symbols,
                                      mdemod_recording_report *report);

/* Name of the kernel variant this context launches (for logs and bench output). */
const char *mdemod_kernel_name(const mdemod_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
