/*
 * mdemod_internal_api.h — entries of libmeteor_demod_amd.so that are NOT part of the drop-in boundary (include/meteor_demod_amd.h):
 * the primitives the recording stitcher (csrc/recording.hip) is built from, and the known-answer / self-test hooks of the
 * test-suite.  They are exported (the tests reach them through ctypes) but a caller of the library does not need them:
 * mdemod_demodulate_recording[_host] is the public face of everything in the first block.
 */
#ifndef MDEMOD_INTERNAL_API_H
#define MDEMOD_INTERNAL_API_H

#include "../../include/meteor_demod_amd.h"

#ifdef __cplusplus
/* The boundary is C: no exception may cross it (a caller written in C has nothing to catch it with, and an uncaught one is abort()).
 * Every int-returning entry is a function-try-block that ends in this: std::bad_alloc from a vector, std::system_error from a thread
 * that could not be started and anything else come back as MDEMOD_ERR_NOMEM - "a resource was not to be had". */
#include <new>
/* The text behind mdemod_last_error() (ABI 5): thread-local, written by the entry that fails, never printed unless the environment
 * has MDEMOD_DEBUG (a library behind someone else's TUI has no business on stderr).  MDEMOD_API_ENTER, first statement of every
 * int-returning entry, empties it when the OUTERMOST entry on this thread begins, so a text never outlives the call it belongs to
 * (entries call each other: mdemod_create -> mdemod_reset / mdemod_destroy). */
void mdm_note_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
struct mdm_api_scope { mdm_api_scope(); ~mdm_api_scope(); };
#define MDEMOD_API_ENTER mdm_api_scope api_scope_;
#define MDEMOD_API_CATCH catch (...) { mdm_note_error("a C++ exception reached the boundary (allocation failed, or a thread could not be started)"); return MDEMOD_ERR_NOMEM; }
extern "C" {
#endif

/* ---- what the stitcher needs besides the ragged process call -----------------
 * The reference demodulates a recording as one serial recurrence (main.c:303).
 * To run tiles of it in parallel each tile is an independent stream that starts
 * early from a seed state. */

/* Every stream := *seed (loop state and counters); filter history zeroed. */
int  mdemod_set_state_all(mdemod_ctx *ctx, const mdemod_stream_state *seed, void *hip_stream);
/* pll phase of stream s += quarter_turns_dev[s] * pi/2 (device array, n_streams
 * entries; wrapped like pll.c:113): moves a stream that locked k*90 degrees away
 * from its predecessor onto the predecessor's constellation rotation.  In OQPSK
 * mode an odd number of quarter turns also moves the symbol clock by half a symbol
 * (t_phase +- pi, dual state toggled: the rails swap and their firings are half a
 * symbol apart). */
int  mdemod_rotate_carrier(mdemod_ctx *ctx, const int32_t *quarter_turns_dev, void *hip_stream);

/* Per-stream carrier seeds (device arrays of n_streams entries): pll frequency in rad/symbol and sweep direction
 * (+1 / -1, pll.c:112) of stream s := freq_dev[s], updown_dev[s]; everything else is left as it is.  Used with
 * mdemod_set_state_all when the carrier moves along the recording (Doppler) and every tile needs its local estimate. */
int  mdemod_set_carrier_seeds(mdemod_ctx *ctx, const float *freq_dev, const int32_t *updown_dev, void *hip_stream);
/* Per-stream AGC gain seeds (agc.c:9, device array of n_streams entries, negative values become 0 like agc.c:23):
 * for tiles of a recording whose amplitude changes faster than the reference's AGC follows at that gain. */
int  mdemod_set_gain_seeds(mdemod_ctx *ctx, const float *gain_dev, void *hip_stream);

/* Per-stream symbol-clock frequency seeds (timing.c:14 `freq`, rad per interpolated sample; device array of n_streams entries).
 * Words outside centre +- centre / 4096 - what timing.c:80-86 can hold, and what the kernels' symbol clock counts on - are clamped to
 * that range on the device as they are copied in (mdemod_set_state refuses such a word; a device array cannot be refused). */
int  mdemod_set_clock_seeds(mdemod_ctx *ctx, const float *t_freq_dev, void *hip_stream);
/* mdemod_get_state for `count` streams from `first` in one round trip (tiles of a recording: thousands of streams). */
int  mdemod_get_states(mdemod_ctx *ctx, uint32_t first, uint32_t count, mdemod_stream_state *out, void *hip_stream);
/* dst := src for every stream (loop state, counters, filter history, lock events): a checkpoint of a whole bank.  Both
 * contexts must have been created from the same parameters (same n_streams, format, device).  Asynchronous on hip_stream. */
int  mdemod_copy_state(mdemod_ctx *dst, mdemod_ctx *src, void *hip_stream);

/* Feed-forward carrier estimate of n_windows windows of one recording (device arrays; what the recording entry below seeds
 * its tiles with): the 4th power of the samples has a spectral
 * line at 4x the carrier offset whatever the data (QPSK and RRC-shaped OQPSK).  Window w covers window samples from
 * starts_dev[w] (reads past the end of the recording repeat its last sample); the window length actually used is
 * mdemod_carrier_window_samples(): window_samples rounded down to a power of two in [4096, 2^18].  freq_dev[w]: carrier in
 * rad per NCO step (per symbol; per half symbol for OQPSK: pll.c:77,93) at the MIDDLE of the window, as pll_get_freq()
 * would report it; quality_dev[w]: line / mean of the searched band (+-0.33 rad/symbol): noise alone gives 3-4, a 12 dB
 * signal 40-50.  One kernel (z^4, boxcar decimation, FFT in LDS, peak search), asynchronous on hip_stream; only
 * samplerate, symrate, oqpsk and bps of params are used. */
uint32_t mdemod_carrier_window_samples(const mdemod_params *params, uint32_t window_samples);
int  mdemod_estimate_carrier(const mdemod_params *params, const void *iq_dev, uint64_t n_samples,
                             const uint64_t *starts_dev, uint32_t n_windows, uint32_t window_samples,
                             float *freq_dev, float *quality_dev, void *hip_stream);
/* The same with a de-chirp: chirp_dev[w] (may be NULL) = carrier slope of window w in rad per NCO step per SAMPLE; the line of a
 * carrier that moves by more than a bin inside the window (Doppler: up to 40 Hz/s) is smeared without it. */
int  mdemod_estimate_carrier_chirp(const mdemod_params *params, const void *iq_dev, uint64_t n_samples,
                                   const uint64_t *starts_dev, const float *chirp_dev, uint32_t n_windows, uint32_t window_samples,
                                   float *freq_dev, float *quality_dev, void *hip_stream);
/* Feed-forward symbol-clock estimate of the same windows: the symbol-rate line of |z|^2 (QPSK) or the two lines of z^2 at twice
 * the carrier +- the symbol rate (OQPSK: carrier_dev[w] = that window's carrier in rad per NCO step, e.g. from
 * mdemod_estimate_carrier; chirp_dev as above; both may be NULL, and are not read for QPSK).  t_freq_dev[w]: the symbol clock in
 * rad per interpolated step, as mm_omega() / timing.c:14 would hold it when locked (2 pi * symrate / samplerate / interp for a
 * perfect clock), searched within the reference's own +-1/4096 of the nominal rate; quality_dev[w]: line / mean of +-128 bins
 * (noise alone 2-4, a 12 dB signal 20-30).  window_samples is rounded down to a power of two in [4096, 2^18].  At 12 dB and
 * 2^18 samples the estimate is good to 1e-7 of the rate (the reference's own loop wanders by 3e-6 around it).  Asynchronous on
 * hip_stream; samplerate, symrate, interp_factor, oqpsk and bps of params are used. */
int  mdemod_estimate_clock(const mdemod_params *params, const void *iq_dev, uint64_t n_samples,
                           const uint64_t *starts_dev, const float *carrier_dev, const float *chirp_dev,
                           uint32_t n_windows, uint32_t window_samples, float *t_freq_dev, float *quality_dev, void *hip_stream);

/* ---- init-time tables, exposed for known-answer tests -------------------- */

/* Host-only: derive the init-time tables for `params` without touching a device
 * (what demod_init computes through pll_init / timing_init / filter_init_rrc).
 * rrc_out (may be NULL) receives interp*taps floats, consts_out the 8 loop
 * constants (order below), lut_out the 32-entry tanh LUT.  Returns the number
 * of RRC floats, or <0. */
int  mdemod_derive_tables(const mdemod_params *params, float *rrc_out, uint32_t rrc_cap,
                          float consts_out[8], float lut_out[32]);
/* What mdemod_create would pick for `params` - host only, no device: the kernel's name as mdemod_kernel_name gives it (a
 * v1 kernel that leaves its table in global memory is marked), the dynamic LDS bytes and the threads per block. */
int  mdemod_plan_kernel(const mdemod_params *params, char *name, uint32_t name_cap, uint32_t *lds_bytes, uint32_t *block_threads);

/* RRC polyphase table as filter_init_rrc lays it out (filter.c:18-22):
 * interp*taps floats, bank-major.  Returns number of floats, or <0. */
int  mdemod_get_rrc_table(const mdemod_ctx *ctx, float *out, uint32_t cap);
/* Derived loop constants in the order:
 * pll_alpha, pll_beta, pll_fmax, t_alpha, t_beta, t_center, t_maxdev, osf */
int  mdemod_get_loop_constants(const mdemod_ctx *ctx, float out[8]);
/* tanh LUT (pll.c:40-42), 32 floats. */
int  mdemod_get_tanh_lut(const mdemod_ctx *ctx, float out[32]);

/* Device self-test of the scalar primitives (fixed-point sine, hypot, wrap):
 * evaluates them on `n` inputs on the GPU.  x: n floats in; sin_out/cos_out: n
 * floats each; for hypot: pairs (x[2i], x[2i+1]) -> n/2 results in sin_out. */
int  mdemod_selftest_sincos(mdemod_ctx *ctx, const float *x, uint32_t n,
                            float *sin_out, float *cos_out);
int  mdemod_selftest_hypot(mdemod_ctx *ctx, const float *xy, uint32_t n_pairs, float *out);
/* Exhaustive on-device check of the division-free turn code of fast_sin against the real
 * double division over every float with |x| < 16 (see csrc/demod_device.h). */
int  mdemod_selftest_turncode(mdemod_ctx *ctx, uint64_t *n_checked, uint64_t *n_mismatch);
int  mdemod_selftest_cabsf(mdemod_ctx *ctx, uint64_t pairs, uint64_t *n_mismatch, uint64_t *n_fallback);   /* the short cabsf against the correctly rounded one on `pairs` pseudo-random pairs */
int  mdemod_selftest_sinlut(mdemod_ctx *ctx, uint64_t *n_checked, uint64_t *n_mismatch);   /* the sine table in LDS against sincos.c's integer arithmetic: every turn code */

#ifdef __cplusplus
}
#endif
#endif
