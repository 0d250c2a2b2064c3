/*
 * demod_internal.h — structures shared by the host side (demod_api.cpp) and the
 * gfx950 kernels (demod_kernel.hip).  Not part of the public C-ABI.
 */
#ifndef MDEMOD_INTERNAL_H
#define MDEMOD_INTERNAL_H

#include <stdint.h>
#include "../../include/meteor_demod_amd.h"
#include "mdemod_internal_api.h"
#include "clock_jump.h"

#define MDEMOD_WAVE            64
#define MDEMOD_GRANULE_SAMPLES 4      /* ring granule = 4 consecutive IQ samples */
#define MDEMOD_RW_STATE_SLOTS   12     /* per-lane LDS state words of the register-window kernels */
#define MDEMOD_RW_WIDE_BLOCK    512    /* threads per block of the wide geometry */
#ifndef MDEMOD_RW_WIDE_NW
#define MDEMOD_RW_WIDE_NW       160    /* window slots of the wide geometry (129 taps + 32 alignments) */
#define MDEMOD_RW_WIDE_SLIDE    16     /* slots per slide of the wide geometry      */
#define MDEMOD_RW_WIDE_MAXSL    1      /* slides per loop iteration                 */
#endif
#define MDEMOD_SIN_LUT_BYTES    65552  /* fast_sin's parabola as 16 385 floats in LDS, rounded up to 16 bytes (rotwin_body.h, LUT instances) */
#define MDEMOD_RW_LUT_BLOCK     512    /* threads per block of the std kernel's instances with the sine table: one table per CU */
#define MDEMOD_RW_MID_NW        96     /* window slots of the mid geometry (65 taps + 32 alignments)      */
#define MDEMOD_RW_FAR_NW        112    /* window slots of the far geometry (65 taps + 48 alignments)      */
#ifndef MDEMOD_RW_BLOCK
#define MDEMOD_RW_BLOCK         256    /* threads per block of the other register-window kernels */
#endif

/* Loop constants + geometry, passed to the kernel by value. */
struct DemodConsts {
	int32_t interp;          /* polyphase banks (-O)                              */
	int32_t taps;            /* 2*order+1                                         */
	int32_t oqpsk;           /* -m oqpsk                                          */
	int32_t hpad;            /* history samples kept per stream (>= taps-1, %8==0)*/
	int32_t win_granules;    /* granules covering one FIR window at any alignment */
	int32_t ring_granules;   /* LDS ring capacity per lane, in granules           */
	int32_t chunk_granules;  /* granules fetched per refill                       */
	int32_t ctab_row_floats; /* padded row length of the aligned coefficient table*/
	int32_t ctab_row_stride; /* floats between rows (bank-conflict-free stride)   */
	float   pll_alpha, pll_beta, pll_fmax;
	float   t_alpha, t_beta, t_center, t_maxdev;
	/* symbol-clock fast path: see demod_host.cpp */
	int32_t  step_safe;      /* blind steps that provably cannot fire            */
	int32_t  step_check;     /* predicated checked steps after them              */
	float    step_fmax;      /* upper bound of the per-step phase increment      */
	uint32_t interp_magic;   /* floor(2^32/interp)+1: x/interp == mulhi(x, magic) */
	float    step_inv;       /* (1 - 2^-12) / step_fmax: steps that fit a phase distance, strictly conservative (clock_jump.h) */
	int32_t  sin_lut;        /* host only: this context launches the kernel instance with the sine table in LDS */
	cj_sched jump[2];        /* the symbol clock's runs in closed form (clock_jump.h): [0] from 0, [1] the second rail of an OQPSK symbol; nb == 0: not used */
};

/* Per-stream state, structure-of-arrays in HBM so that lane s of a wave touches
 * element s of each array (fully coalesced state load/store). */
struct DemodStateSoA {
	float    *agc_gain, *agc_bias_re, *agc_bias_im;
	float    *pll_phase, *pll_freq, *pll_err;
	float    *t_phase, *t_freq, *t_prev;
	float    *inphase;
	int32_t  *flags;             /* bit0 locked, bit1 locked_once, bit2 updown>0, bits 4-5 dual_state */
	uint64_t *n_samples, *n_symbols;
	int64_t  *first_lock;
	uint32_t *sym_this_call, *ev_this_call;
	int32_t  *overflow;
	void     *hist;              /* [hpad][n_streams] raw samples (format of the ctx) */
	mdemod_lock_event *events;   /* [n_streams][MDEMOD_MAX_LOCK_EVENTS]               */
};

#define MDEMOD_FLAG_LOCKED       1
#define MDEMOD_FLAG_LOCKED_ONCE  2
#define MDEMOD_FLAG_UPDOWN_POS   4
#define MDEMOD_FLAG_DUAL_SHIFT   4

/* Launch arguments of the demod kernel. */
struct DemodLaunch {
	DemodConsts   c;
	DemodStateSoA st;
	const void   *iq;                /* base of all streams                          */
	const uint64_t *iq_offset;       /* per-stream offsets (samples) or NULL         */
	const uint32_t *n_samples_arr;   /* per-stream counts or NULL                    */
	uint64_t      iq_stride;         /* uniform layout: stream s starts at s*stride  */
	uint32_t      n_samples;         /* uniform count                                */
	int8_t       *soft;
	uint64_t      soft_stride;       /* symbols                                      */
	uint32_t      soft_cap;          /* symbols                                      */
	uint32_t      n_streams;
	const float  *ctab;              /* aligned, zero-padded coefficient table in HBM*/
	uint32_t      ctab_floats;
	const float  *tanh_lut;          /* 32 floats                                    */
};


#ifdef __HIPCC__
#include <hip/hip_runtime.h>
hipError_t mdemod_launch_demod(const DemodLaunch &L, int fmt, int block, int global_table, size_t lds_bytes, hipStream_t stream);
hipError_t mdemod_launch_demod_rot(const DemodLaunch &L, int fmt, int compact, size_t lds_bytes, hipStream_t stream);   /* v3: rotating register window (std geometry); compact: compact4 coefficient table */
hipError_t mdemod_launch_demod_gat(const DemodLaunch &L, int fmt, int long_filter, size_t lds_bytes, hipStream_t stream);   /* v3: gather geometry (no window: every firing loads its taps); long_filter: 129 embedded taps instead of 65 (not for float input) */
hipError_t mdemod_launch_demod_roth(const DemodLaunch &L, int geom, int many_slides, size_t lds_bytes, hipStream_t stream);   /* v3: hybrid window (float input: VGPRs + AccVGPRs, one wave per SIMD); geom 0: 160 slots (<= 129 taps), 1: 96 slots (<= 65 taps; many_slides: the instance for more than 20 samples per firing), 2: 120 slots (<= 65 taps, up to 54 samples per firing) */
hipError_t mdemod_launch_demod_rotp(const DemodLaunch &L, int fmt, int geom /* 0 wide, 1 mid, 2 far */, size_t lds_bytes, hipStream_t stream);   /* v3: rotating packed window */
hipError_t mdemod_launch_demod_lat(const DemodLaunch &L, int fmt, const float *rrc_dev, int ring_size, int span, int float_history, size_t lds_bytes, hipStream_t stream);
bool mdemod_lat_geometry(const DemodConsts &c, double samples_per_firing, int *ring_size, int *span, size_t *lds_bytes);
hipError_t mdemod_launch_warm(hipStream_t stream);   /* empty kernel: loads the code objects */
hipError_t mdemod_launch_reset(const DemodStateSoA &st, const DemodConsts &c, int fmt, int float_history, uint32_t n_streams, hipStream_t stream);
hipError_t mdemod_launch_seed(const DemodStateSoA &st, const DemodConsts &c, const mdemod_stream_state &v, int32_t flags, int fmt,
                              int float_history, uint32_t n_streams, hipStream_t stream);
hipError_t mdemod_launch_rotate(const DemodStateSoA &st, const int32_t *quarter_turns_dev, uint32_t n_streams, int oqpsk, hipStream_t stream);
hipError_t mdemod_launch_gain_seeds(const DemodStateSoA &st, const float *gain_dev, uint32_t n_streams, hipStream_t stream);
hipError_t mdemod_launch_clock_seeds(const DemodStateSoA &st, const float *t_freq_dev, float lo, float hi, uint32_t n_streams, hipStream_t stream);
hipError_t mdemod_launch_carrier_seeds(const DemodStateSoA &st, const float *freq_dev, const int32_t *updown_dev, uint32_t n_streams, hipStream_t stream);
hipError_t mdemod_launch_fill_uniform_rows(uint64_t *off_dev, uint32_t *cnt_dev, uint64_t pitch_samples, uint32_t count, uint32_t n_streams, hipStream_t stream);
hipError_t mdemod_launch_copy_events(const DemodStateSoA &st, mdemod_lock_event *dst, uint32_t n_streams, uint32_t *sym_out, uint32_t *ev_out, hipStream_t stream);
hipError_t mdemod_launch_compact_rows(const int8_t *src, uint64_t src_pitch_sym, int8_t *dst, uint64_t dst_pitch_sym,
                                      const uint32_t *counts_dev, uint32_t n_streams, hipStream_t stream);
/* symbols n samples produce at the nominal symbol rate, with slack: NOT a bound (see mdemod_max_symbols) */
uint64_t mdemod_nominal_symbols(const mdemod_ctx *ctx, uint64_t n_samples);
/* host_pipe.cpp: the pipelined host-buffer path behind mdemod_process_host */
int  mdemod_hostpipe_run(mdemod_ctx *ctx, void **pipe_slot, const DemodStateSoA &st, uint32_t n_streams, size_t sample_bytes,
                         const void *const *iq_host, const uint32_t *n_samples,
                         int8_t *const *soft_host, const uint32_t *soft_cap, uint32_t *n_symbols);
void mdemod_hostpipe_free(void *pipe);
int  mdemod_hostpipe_pin(void **pipe_slot, const void *base, size_t bytes);
int  mdemod_hostpipe_unpin(void *pipe, const void *base);
hipError_t mdemod_launch_selftest_sincos(const float *x, uint32_t n, float *s, float *c, hipStream_t stream);
hipError_t mdemod_launch_selftest_turncode(unsigned long long *mismatch_dev, hipStream_t stream);
hipError_t mdemod_launch_selftest_sinlut(unsigned long long *mismatch_dev, hipStream_t stream);
hipError_t mdemod_launch_selftest_cabsf(uint64_t pairs, unsigned long long *out_dev, hipStream_t stream);
hipError_t mdemod_launch_selftest_hypot(const float *xy, uint32_t n, float *out, hipStream_t stream);
#endif

#endif
