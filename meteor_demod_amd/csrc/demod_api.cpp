/*
 * demod_api.cpp — implementation of the C-ABI declared in
 * include/meteor_demod_amd.h: context lifecycle, device buffers, launches.
 * There is no CPU fallback anywhere in this file: without a HIP device every
 * entry that needs one returns MDEMOD_ERR_HIP.
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "demod_host.h"
#include "demod_internal.h"

struct mdemod_ctx {
	mdemod_params params;
	HostTables    tab;
	int           block_threads;
	size_t        lds_bytes;
	size_t        sample_bytes;

	/* device */
	DemodStateSoA st;
	float        *d_ctab;
	float        *d_lut;
	float        *d_rrc;       /* plain polyphase table [bank][taps] (filter.c:18-22) for the latency kernel */
	int           hyb_block;   /* hybrid window (tab.rw_hyb): threads per block */
	bool          v1_global_table;   /* v1 ring kernel with its coefficient table left in global memory (it does not fit the LDS) */
	bool          use_rot;     /* std geometry: the v3 rotating-window kernel of demod_kernel_rot.hip */
	bool          lat_ok;      /* the latency kernel (one stream per wave) fits this configuration */
	int           lat_ring, lat_span;
	size_t        lat_lds;
	std::vector<void *> allocs;

	void   *pipe;      /* host_pipe.cpp: pinned staging, streams, events of mdemod_process_host (grow only) */
};

namespace {

#define HIP_TRY(expr)                                                                   \
	do {                                                                                \
		hipError_t e_ = (expr);                                                         \
		if (e_ != hipSuccess) {                                                         \
			mdm_note_error("%s failed: %s (%s:%d)", #expr,                           \
			        hipGetErrorString(e_), __FILE__, __LINE__);                         \
			(void)hipGetLastError();   /* reported: it must not be read again as the status of somebody's next launch */ \
			return e_ == hipErrorOutOfMemory ? MDEMOD_ERR_NOMEM : MDEMOD_ERR_HIP;       \
		}                                                                               \
	} while (0)

template <typename T>
int
dev_alloc(mdemod_ctx *ctx, T **ptr, size_t count)
{
	void *p = nullptr;
	HIP_TRY(hipMalloc(&p, count * sizeof(T) + 16));
	ctx->allocs.push_back(p);
	*ptr = static_cast<T *>(p);
	return MDEMOD_OK;
}

template <typename T>
int
fetch(const T *dev, uint32_t first, uint32_t count, std::vector<T> &host, hipStream_t st)
{
	host.resize(count);
	HIP_TRY(hipMemcpyAsync(host.data(), dev + first, sizeof(T) * count, hipMemcpyDeviceToHost, st));
	return MDEMOD_OK;
}

/* the wave-per-stream kernel for few streams, the lane-per-stream kernels for many.  A wave runs its one stream 2-3 times faster than
 * a lane does, until every SIMD has a wave or two and the vector unit - 63 of 64 lanes of it idle - is the bound: the wave kernel
 * levels off at a TOTAL rate (10-11 GS/s at configs[1], 5.5-6 at configs[2], 15-17 at configs[3], 25-28 at 2 MS/s) while the lane
 * kernels keep their per-stream rate up to 131 072 streams.  Measured r05 (tools/lat_bench.py, MS/s per stream, wave vs lane):
 * configs[1] 3.8 vs 2.1 at 2 048 streams, 2.5 vs 2.1 at 4 096, 1.3 vs 2.1 at 8 192; configs[2] 1.34 vs 1.06 at 4 096, 0.72 vs 1.06
 * at 8 192; 97 taps of float input at 230 kS/s 2.15 vs 1.41 at 4 096; configs[3] 5.9 vs 4.0 at 2 048, 3.7 vs 4.0 at 4 096 (float
 * input 5.8 vs 4.7, 3.6 vs 4.7); 1.024 MS/s 7.9 vs 7.1 at 1 024, 4.1 vs 7.0 at 2 048 (float input 4.1 vs 6.0); 2.048 MS/s 17.9 vs
 * 9.1 at 1 024, 9.7 vs 10.2 at 2 048.  MDEMOD_FLAG_LAT_ON / _OFF pin the choice.  (r04: the farm's symbol clock takes the closed-form
 * schedule of clock_jump.h at high sample rates as the v3 kernels do: one stream at 1.8 MS/s 9.3 -> 18.7 MS/s.) */
bool
wants_latency_kernel(const mdemod_ctx *ctx)
{
	if (!ctx->lat_ok || (ctx->params.reserved & MDEMOD_FLAG_LAT_OFF)) return false;
	if (ctx->params.reserved & MDEMOD_FLAG_LAT_ON) return true;
	const double per_firing = static_cast<double>(ctx->tab.osf) / (ctx->params.oqpsk ? 2.0 : 1.0);
	/* from about 32 samples per firing (3.2 MS/s QPSK, 6 MS/s OQPSK) the farm's batches hold too few firings: one lane of a v3 kernel
	 * is faster even for ONE stream (r04, tools/one_stream_rates.py: 3.2 MS/s 10.4 vs 9.6 MS/s, OQPSK 80k at 6 MS/s 9.7 vs 8.0) */
	if (ctx->tab.use_rw && per_firing > 32.0) return false;
	/* where the two meet (the table above): the mid and far windows' lanes are quick, the wide one's less so, and at a few samples
	 * per firing a lane spends most of its time in the per-symbol arithmetic the wave does no faster */
	const uint32_t most = (ctx->tab.use_rw && (ctx->tab.rw_mid || ctx->tab.rw_far)) ? 1024u : (per_firing >= 8.0 ? 2048u : 4096u);
	return ctx->params.n_streams <= most;
}

int
select_device(const mdemod_ctx *ctx)
{
	/* Every device entry begins here.  The launch wrappers report hipGetLastError() after their launch: an error some EARLIER call of
	 * this thread left behind (the caller's own, another library's, a refused hipSetDevice) would come back as the status of a launch
	 * that went through (r06: a context made right after mdemod_create had refused a device that does not exist failed in
	 * mdemod_launch_reset with that device's error).  A launch's status is the launch's: what is pending is dropped first. */
	(void)hipGetLastError();
	HIP_TRY(hipSetDevice(ctx->params.device));
	return MDEMOD_OK;
}

int
launch(mdemod_ctx *ctx, DemodLaunch &L, hipStream_t stream)
{
	L.c = ctx->tab.c;
	L.st = ctx->st;
	L.n_streams = ctx->params.n_streams;
	L.ctab = ctx->d_ctab;
	L.ctab_floats = static_cast<uint32_t>(ctx->tab.ctab.size());
	L.tanh_lut = ctx->d_lut;
	/* Few streams: one stream per WAVE (demod_kernel_lat.hip) instead of one per lane.  A lane runs ~0.8 M symbols/s whatever
	 * the batch, a wave 1.5 times that, and below a few thousand streams most of the GPU idles either way. */
	if (wants_latency_kernel(ctx)) {
		HIP_TRY(mdemod_launch_demod_lat(L, ctx->params.bps, ctx->d_rrc, ctx->lat_ring, ctx->lat_span, ctx->tab.use_rw ? 1 : 0, ctx->lat_lds, stream));
		return MDEMOD_OK;
	}
	if (ctx->tab.use_rw && ctx->tab.rw_gather)
		HIP_TRY(mdemod_launch_demod_gat(L, ctx->params.bps, ctx->tab.c.taps > 65 ? 1 : 0, ctx->lds_bytes, stream));
	else if (ctx->tab.use_rw && ctx->tab.rw_hyb)
		HIP_TRY(mdemod_launch_demod_roth(L, ctx->tab.rw_mid ? 1 : (ctx->tab.rw_far ? 2 : 0),
		                                 static_cast<double>(ctx->tab.osf) / (ctx->params.oqpsk ? 2.0 : 1.0) > 20.0 ? 1 : 0, ctx->lds_bytes, stream));
	else if (ctx->tab.use_rw && ctx->use_rot)
		HIP_TRY(mdemod_launch_demod_rot(L, ctx->params.bps, ctx->tab.rw_std_compact ? 1 : 0, ctx->lds_bytes, stream));
	else if (ctx->tab.use_rw && ctx->tab.rw_compact4)
		HIP_TRY(mdemod_launch_demod_rotp(L, ctx->params.bps, ctx->tab.rw_mid ? 1 : (ctx->tab.rw_far ? 2 : 0), ctx->lds_bytes, stream));
	else if (ctx->tab.use_rw)
		return MDEMOD_ERR_PARAM;                      /* (cannot happen: every register-window plan is one of the v3 kernels above) */
	else
		HIP_TRY(mdemod_launch_demod(L, ctx->params.bps, ctx->block_threads, ctx->v1_global_table ? 1 : 0, ctx->lds_bytes, stream));
	return MDEMOD_OK;
}

} /* namespace */

uint64_t
mdemod_nominal_symbols(const mdemod_ctx *ctx, uint64_t n_samples)
{
	if (!ctx) return 0;
	const double steps = static_cast<double>(n_samples) * ctx->tab.c.interp;
	const double per_step = static_cast<double>(ctx->tab.c.t_center) * (1.0 + 1.0 / 4096.0) / 6.283185307179586;
	return ((static_cast<uint64_t>(steps * per_step * 1.01) + 16 + 7) / 8) * 8;
}

/* Everything mdemod_create decides without a device: the kernel generation, its tables and geometry, the LDS it needs, whether the
 * latency kernel fits.  (mdemod_plan_kernel exposes it to the CPU tests.) */
static int
plan_context(mdemod_ctx *ctx)
{
	const mdemod_params *params = &ctx->params;
	/* MDEMOD_FLAG_KERNEL_MASK pins a generation (the tests run v1 and v3 on the same inputs); otherwise v3 wherever one of its
	 * windows fits, v1 for the rest.  (v2, the moving register window of round 2, was retired in round 4: every geometry it had is
	 * served by a v3 kernel, and the one case that used to route to it - a carrier range of 6 rad per symbol or more, for which the
	 * v3 kernels' float wraps and unchecked turn code do not hold - cannot occur: pll.c:29-33 clamps fmax to 1 rad per symbol,
	 * and so does demod_host.cpp.) */
	const uint32_t kforce = params->reserved & MDEMOD_FLAG_KERNEL_MASK;
	if (kforce == 2) { mdm_note_error("kernel generation 2 (round 2's moving register window) was retired: 0 / 3 = the rotating windows, 1 = the LDS ring"); return MDEMOD_ERR_PARAM; }
	const int generation = kforce == 1 ? 0 : 2;
	int rc = mdemod_host_derive(*params, ctx->tab, generation);
	if (rc) return rc;
	/* the v3 and latency kernels wrap the NCO phase in float arithmetic and skip the range test of its turn code
	 * (demod_device.h): both need phase + freq < 4 pi, i.e. fmax < 2 pi rad per symbol */
	if (!(ctx->tab.c.pll_fmax < 6.0f)) { mdm_note_error("a carrier range of %g rad per symbol: 6 or more cannot be held (pll.c:113 wraps the phase once per symbol)", static_cast<double>(ctx->tab.c.pll_fmax)); return MDEMOD_ERR_PARAM; }
	ctx->sample_bytes = 2 * static_cast<size_t>(params->bps) / 8;
	ctx->use_rot = generation == 2 && ctx->tab.use_rw && !ctx->tab.rw_wide && !ctx->tab.rw_mid && !ctx->tab.rw_far && !ctx->tab.rw_hyb && !ctx->tab.rw_gather;
	ctx->hyb_block = 0;

	/* tunables (experiments only; defaults are the measured best) */
	DemodConsts &c = ctx->tab.c;
	if (!ctx->tab.use_rw) {
		c.ring_granules = c.hpad / 4 + 8;            /* v1: eight granules of slack behind the history (measured best, r01) */
	}
	ctx->block_threads = 64 * 3;                     /* v1 kernel: three waves per block (__launch_bounds__(256); measured best, r01) */

	if (ctx->tab.rw_hyb) ctx->hyb_block = MDEMOD_RW_BLOCK;
	/* The kernel instances compiled for the BASELINE settings (their symbol clock's blind steps are template parameters: 14 / 6 on
	 * the std window, 109 on the wide packed one) also keep fast_sin's parabola as a table in LDS (rotwin_body.h, LUT): 64 KB, one
	 * 512-thread block per CU.  Only when table, coefficient rows, state slots and output rings fit the 160 KB together. */
	c.sin_lut = ((ctx->use_rot && !ctx->tab.rw_std_compact && c.step_safe == (c.oqpsk ? 6 : 14)) ||
	             (ctx->tab.rw_compact4 && ctx->tab.rw_wide && params->bps == 16 && !c.oqpsk && c.step_safe == 109)) ? 1 : 0;
	auto rw_block = [&]() { return ctx->tab.rw_wide ? MDEMOD_RW_WIDE_BLOCK : (c.sin_lut ? MDEMOD_RW_LUT_BLOCK : MDEMOD_RW_BLOCK); };
	auto lds_need = [&](int threads) {
		return (ctx->tab.ctab.size() + 32) * sizeof(float) +
		       (ctx->tab.use_rw ? static_cast<size_t>(rw_block() / 64) * MDEMOD_RW_STATE_SLOTS * 64 * sizeof(float)
		                          + static_cast<size_t>(rw_block()) * 64     /* soft-symbol staging: a 32-symbol ring (4 x 16 B) per thread */
		                          + (c.sin_lut ? static_cast<size_t>(MDEMOD_SIN_LUT_BYTES) : 0)
		                        : static_cast<size_t>(threads / 64) * c.ring_granules * 64 * 4 * ctx->sample_bytes);
	};
	if (c.sin_lut && lds_need(ctx->block_threads) > 160 * 1024) c.sin_lut = 0;      /* (e.g. -O 7 at a rate that has 14 blind steps: 38 KB of rows) */
	if (ctx->tab.use_rw && lds_need(ctx->block_threads) > 160 * 1024) {
		/* The per-alignment coefficient rows of the std geometry grow with -O (16 alignments x interp x 84 floats: past the
		 * 160 KB of LDS from -O 29 on); the v1 ring kernel keeps 4 alignments and still fits: fall back to it. */
		rc = mdemod_host_derive(*params, ctx->tab, 0);
		ctx->use_rot = false;
		if (rc) return rc;
		c.ring_granules = c.hpad / 4 + 8;
		c.sin_lut = 0;
	}
	while (ctx->block_threads > 64 && lds_need(ctx->block_threads) > 160 * 1024) ctx->block_threads -= 64;
	ctx->lds_bytes = lds_need(ctx->block_threads);
	ctx->v1_global_table = false;
	if (ctx->lds_bytes > 160 * 1024) {
		/* not even one wave's ring fits next to the table (-O 64 with 129 taps, ...): the v1 kernel reads its coefficients from
		   global memory then - slow, but the reference takes such a configuration and so does this */
		if (ctx->tab.use_rw) return MDEMOD_ERR_PARAM;
		ctx->v1_global_table = true;
		ctx->block_threads = 64 * 3;
		auto ring_need = [&](int threads) { return 32 * sizeof(float) + static_cast<size_t>(threads / 64) * c.ring_granules * 64 * 4 * ctx->sample_bytes; };
		while (ctx->block_threads > 64 && ring_need(ctx->block_threads) > 160 * 1024) ctx->block_threads -= 64;
		ctx->lds_bytes = ring_need(ctx->block_threads);
		if (ctx->lds_bytes > 160 * 1024) return MDEMOD_ERR_PARAM;
	}

	/* like the v3 kernels, the latency kernel takes the NCO's short cuts that need fmax < 2pi (demod_device.h) */
	ctx->lat_ok = c.pll_fmax < 6.0f &&
	              mdemod_lat_geometry(c, static_cast<double>(ctx->tab.osf) / (params->oqpsk ? 2.0 : 1.0), &ctx->lat_ring, &ctx->lat_span, &ctx->lat_lds);
	return MDEMOD_OK;
}

extern "C" {

uint32_t
mdemod_abi_version(void)
{
	return MDEMOD_ABI_VERSION;
}

int
mdemod_init_device(int device)
try { MDEMOD_API_ENTER
	(void)hipGetLastError();                                               /* (see select_device) */
	HIP_TRY(hipSetDevice(device));
	HIP_TRY(hipFree(nullptr));                                             /* forces the context */
	/* ... and the code objects (loaded at the first launch of a process), on a stream of its own */
	hipStream_t s = nullptr;
	HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	const bool ok = mdemod_launch_warm(s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
	(void)hipStreamDestroy(s);
	return ok ? MDEMOD_OK : MDEMOD_ERR_HIP;
} MDEMOD_API_CATCH

int
mdemod_device_count(void)
try { MDEMOD_API_ENTER
	int n = 0;
	return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
} MDEMOD_API_CATCH

const char *
mdemod_strerror(int code)
{
	switch (code) {
	case MDEMOD_OK: return "ok";
	case MDEMOD_ERR_PARAM: return "bad parameter or unsupported configuration";
	case MDEMOD_ERR_NOMEM: return "out of memory";
	case MDEMOD_ERR_HIP: return "HIP runtime failure (no gfx950 device, or launch error)";
	case MDEMOD_ERR_OVERFLOW: return "soft-symbol capacity too small";
	case MDEMOD_ERR_RANGE: return "stream index out of range";
	default: return "unknown error";
	}
}

int
mdemod_derive_tables(const mdemod_params *params, float *rrc_out, uint32_t rrc_cap,
                     float consts_out[8], float lut_out[32])
try { MDEMOD_API_ENTER
	if (!params) return MDEMOD_ERR_PARAM;
	HostTables t;
	int rc = mdemod_host_derive(*params, t);
	if (rc) return rc;
	if (rrc_out) {
		if (rrc_cap < t.rrc.size()) return MDEMOD_ERR_PARAM;
		memcpy(rrc_out, t.rrc.data(), t.rrc.size() * sizeof(float));
	}
	if (consts_out) {
		const float v[8] = { t.c.pll_alpha, t.c.pll_beta, t.c.pll_fmax, t.c.t_alpha,
		                     t.c.t_beta, t.c.t_center, t.c.t_maxdev, t.osf };
		memcpy(consts_out, v, sizeof(v));
	}
	if (lut_out) memcpy(lut_out, t.tanh_lut, sizeof(t.tanh_lut));
	return static_cast<int>(t.rrc.size());
} MDEMOD_API_CATCH

int
mdemod_create(const mdemod_params *params, mdemod_ctx **out)
{
	MDEMOD_API_ENTER
	if (!params || !out || params->n_streams == 0) { mdm_note_error("mdemod_create: params, out and at least one stream are needed"); return MDEMOD_ERR_PARAM; }
	*out = nullptr;
	mdemod_ctx *ctx = new (std::nothrow) mdemod_ctx();
	if (!ctx) { mdm_note_error("mdemod_create: no memory for a context"); return MDEMOD_ERR_NOMEM; }
	ctx->params = *params;
	ctx->pipe = nullptr;
	try {
	int rc = plan_context(ctx);
	if (rc) { delete ctx; return rc; }
	DemodConsts &c = ctx->tab.c;

#define CREATE_TRY(expr) do { rc = (expr); if (rc) { mdemod_destroy(ctx); return rc; } } while (0)
	{
		(void)hipGetLastError();                       /* (see select_device) */
		hipError_t e = hipSetDevice(params->device);
		if (e != hipSuccess) {
			mdm_note_error("no usable HIP device %d: %s", params->device, hipGetErrorString(e));
			(void)hipGetLastError();
			delete ctx;
			return MDEMOD_ERR_HIP;
		}
	}
	const size_t n = params->n_streams;
	DemodStateSoA &s = ctx->st;
	CREATE_TRY(dev_alloc(ctx, &s.agc_gain, n));   CREATE_TRY(dev_alloc(ctx, &s.agc_bias_re, n));
	CREATE_TRY(dev_alloc(ctx, &s.agc_bias_im, n)); CREATE_TRY(dev_alloc(ctx, &s.pll_phase, n));
	CREATE_TRY(dev_alloc(ctx, &s.pll_freq, n));   CREATE_TRY(dev_alloc(ctx, &s.pll_err, n));
	CREATE_TRY(dev_alloc(ctx, &s.t_phase, n));    CREATE_TRY(dev_alloc(ctx, &s.t_freq, n));
	CREATE_TRY(dev_alloc(ctx, &s.t_prev, n));     CREATE_TRY(dev_alloc(ctx, &s.inphase, n));
	CREATE_TRY(dev_alloc(ctx, &s.flags, n));
	CREATE_TRY(dev_alloc(ctx, &s.n_samples, n));  CREATE_TRY(dev_alloc(ctx, &s.n_symbols, n));
	CREATE_TRY(dev_alloc(ctx, &s.first_lock, n));
	CREATE_TRY(dev_alloc(ctx, &s.sym_this_call, n)); CREATE_TRY(dev_alloc(ctx, &s.ev_this_call, n));
	CREATE_TRY(dev_alloc(ctx, &s.overflow, n));
	{
		unsigned char *h = nullptr;
		CREATE_TRY(dev_alloc(ctx, &h, static_cast<size_t>(c.hpad) * n * (ctx->tab.use_rw ? 8 : ctx->sample_bytes)));
		s.hist = h;
	}
	CREATE_TRY(dev_alloc(ctx, &s.events, n * MDEMOD_MAX_LOCK_EVENTS));
	CREATE_TRY(dev_alloc(ctx, &ctx->d_ctab, ctx->tab.ctab.size()));
	CREATE_TRY(dev_alloc(ctx, &ctx->d_lut, 32));
	CREATE_TRY(dev_alloc(ctx, &ctx->d_rrc, ctx->tab.rrc.size()));
	{
		/* tables and the power-on state go in on a stream of their own, and only that stream is waited for: a context made while
		   other contexts run (a second host thread, a recording's tile bank next to its serial head) must not wait for their
		   kernels, which hipDeviceSynchronize and the null stream's copies did */
		hipStream_t s0 = nullptr;
		hipError_t e = hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
		if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_ctab, ctx->tab.ctab.data(), ctx->tab.ctab.size() * sizeof(float), hipMemcpyHostToDevice, s0);
		if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_rrc, ctx->tab.rrc.data(), ctx->tab.rrc.size() * sizeof(float), hipMemcpyHostToDevice, s0);
		if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_lut, ctx->tab.tanh_lut, sizeof(ctx->tab.tanh_lut), hipMemcpyHostToDevice, s0);
		int rc_reset = MDEMOD_OK;
		if (e == hipSuccess) rc_reset = mdemod_reset(ctx, s0);
		if (e == hipSuccess) e = hipStreamSynchronize(s0);
		if (s0) (void)hipStreamDestroy(s0);
		if (e != hipSuccess || rc_reset != MDEMOD_OK) { mdemod_destroy(ctx); return rc_reset != MDEMOD_OK ? rc_reset : MDEMOD_ERR_HIP; }
	}
#undef CREATE_TRY
	} catch (...) {                                    /* (see MDEMOD_API_CATCH; what the context holds so far is given back) */
		mdm_note_error("mdemod_create: a C++ exception reached the boundary (allocation failed)");
		mdemod_destroy(ctx);
		return MDEMOD_ERR_NOMEM;
	}
	*out = ctx;
	return MDEMOD_OK;
}

void
mdemod_destroy(mdemod_ctx *ctx)
{
	if (!ctx) return;
	(void)hipSetDevice(ctx->params.device);
	for (void *p : ctx->allocs) (void)hipFree(p);
	mdemod_hostpipe_free(ctx->pipe);
	delete ctx;
}

int
mdemod_reset(mdemod_ctx *ctx, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!ctx) return MDEMOD_ERR_PARAM;
	int rc = select_device(ctx);
	if (rc) return rc;
	HIP_TRY(hipMemsetAsync(ctx->st.events, 0, sizeof(mdemod_lock_event) * MDEMOD_MAX_LOCK_EVENTS * ctx->params.n_streams,
	                       static_cast<hipStream_t>(hip_stream)));
	HIP_TRY(mdemod_launch_reset(ctx->st, ctx->tab.c, ctx->params.bps, ctx->tab.use_rw ? 1 : 0, ctx->params.n_streams,
	                            static_cast<hipStream_t>(hip_stream)));
	return MDEMOD_OK;
} MDEMOD_API_CATCH

uint64_t
mdemod_max_symbols(const mdemod_ctx *ctx, uint64_t n_samples)
{
	if (!ctx) return 0;
	/* The reference emits at most one symbol per input sample (demod.c:33-47: its per-sample loop keeps the last firing),
	 * and so do the kernels.  A bound from the nominal symbol rate (samples * symrate / samplerate) is NOT safe: while the
	 * symbol clock drains a large phase excursion (a full-scale burst after silence) it fires on every sample. */
	return n_samples + 8;
}

int
mdemod_process_device_uniform(mdemod_ctx *ctx, const void *iq_dev, uint64_t iq_stride_samples,
                              uint32_t n_samples, int8_t *soft_dev, uint64_t soft_stride_symbols,
                              uint32_t soft_cap_symbols, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!ctx || (!iq_dev && n_samples) || !soft_dev) return MDEMOD_ERR_PARAM;
	if (n_samples > 0x3FFFFF00u) return MDEMOD_ERR_PARAM;
	if (soft_cap_symbols > soft_stride_symbols) return MDEMOD_ERR_PARAM;
	int rc = select_device(ctx);
	if (rc) return rc;
	DemodLaunch L;
	memset(&L, 0, sizeof(L));
	L.iq = iq_dev; L.iq_stride = iq_stride_samples; L.n_samples = n_samples;
	L.soft = soft_dev; L.soft_stride = soft_stride_symbols; L.soft_cap = soft_cap_symbols;
	return launch(ctx, L, static_cast<hipStream_t>(hip_stream));
} MDEMOD_API_CATCH

int
mdemod_process_device(mdemod_ctx *ctx, const void *iq_dev, const uint64_t *iq_offset_dev,
                      const uint32_t *n_samples_dev, int8_t *soft_dev, uint64_t soft_stride_symbols,
                      uint32_t soft_cap_symbols, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!ctx || !iq_dev || !iq_offset_dev || !n_samples_dev || !soft_dev) return MDEMOD_ERR_PARAM;
	if (soft_cap_symbols > soft_stride_symbols) return MDEMOD_ERR_PARAM;
	int rc = select_device(ctx);
	if (rc) return rc;
	DemodLaunch L;
	memset(&L, 0, sizeof(L));
	L.iq = iq_dev; L.iq_offset = iq_offset_dev; L.n_samples_arr = n_samples_dev;
	L.soft = soft_dev; L.soft_stride = soft_stride_symbols; L.soft_cap = soft_cap_symbols;
	return launch(ctx, L, static_cast<hipStream_t>(hip_stream));
} MDEMOD_API_CATCH

int
mdemod_process_host(mdemod_ctx *ctx, const void *const *iq_host, const uint32_t *n_samples,
                    int8_t *const *soft_host, const uint32_t *soft_cap, uint32_t *n_symbols)
try { MDEMOD_API_ENTER
	if (!ctx || !iq_host || !n_samples || !soft_host || !soft_cap) return MDEMOD_ERR_PARAM;
	int rc = select_device(ctx);
	if (rc) return rc;
	/* other work of this context (e.g. an asynchronous reset on the caller's stream) must be through before the
	 * pipeline's own streams touch the state */
	HIP_TRY(hipDeviceSynchronize());
	return mdemod_hostpipe_run(ctx, &ctx->pipe, ctx->st, ctx->params.n_streams, ctx->sample_bytes,
	                           iq_host, n_samples, soft_host, soft_cap, n_symbols);
} MDEMOD_API_CATCH

int
mdemod_pin_host_buffer(mdemod_ctx *ctx, const void *base, size_t bytes)
try { MDEMOD_API_ENTER
	if (!ctx || !base || !bytes) return MDEMOD_ERR_PARAM;
	int rc = select_device(ctx);
	if (rc) return rc;
	return mdemod_hostpipe_pin(&ctx->pipe, base, bytes);
} MDEMOD_API_CATCH

int
mdemod_unpin_host_buffer(mdemod_ctx *ctx, const void *base)
try { MDEMOD_API_ENTER
	if (!ctx || !base) return MDEMOD_ERR_PARAM;
	int rc = select_device(ctx);
	if (rc) return rc;
	return mdemod_hostpipe_unpin(ctx->pipe, base);
} MDEMOD_API_CATCH

/* ---- status / state ---------------------------------------------------------- */


int
mdemod_get_status(mdemod_ctx *ctx, uint32_t first, uint32_t count, mdemod_status *out, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!ctx || !out) return MDEMOD_ERR_PARAM;
	if (static_cast<uint64_t>(first) + count > ctx->params.n_streams) return MDEMOD_ERR_RANGE;
	if (!count) return MDEMOD_OK;
	int rc = select_device(ctx);
	if (rc) return rc;
	hipStream_t st = static_cast<hipStream_t>(hip_stream);
	const DemodStateSoA &s = ctx->st;
	std::vector<uint64_t> nsamp, nsym; std::vector<int64_t> fl;
	std::vector<uint32_t> symc, evc; std::vector<float> pf, om, gn; std::vector<int32_t> flags, ovf;
	if ((rc = fetch(s.n_samples, first, count, nsamp, st))) return rc;
	if ((rc = fetch(s.n_symbols, first, count, nsym, st))) return rc;
	if ((rc = fetch(s.first_lock, first, count, fl, st))) return rc;
	if ((rc = fetch(s.sym_this_call, first, count, symc, st))) return rc;
	if ((rc = fetch(s.ev_this_call, first, count, evc, st))) return rc;
	if ((rc = fetch(s.pll_freq, first, count, pf, st))) return rc;
	if ((rc = fetch(s.t_freq, first, count, om, st))) return rc;
	if ((rc = fetch(s.agc_gain, first, count, gn, st))) return rc;
	if ((rc = fetch(s.flags, first, count, flags, st))) return rc;
	if ((rc = fetch(s.overflow, first, count, ovf, st))) return rc;
	HIP_TRY(hipStreamSynchronize(st));
	for (uint32_t i = 0; i < count; i++) {
		mdemod_status &o = out[i];
		o.n_samples = nsamp[i]; o.n_symbols = nsym[i]; o.first_lock_symbol = fl[i];
		o.symbols_this_call = symc[i]; o.lock_events_this_call = evc[i];
		o.pll_freq = pf[i]; o.omega = om[i]; o.gain = gn[i];
		o.locked = (flags[i] & MDEMOD_FLAG_LOCKED) ? 1 : 0;
		o.locked_once = (flags[i] & MDEMOD_FLAG_LOCKED_ONCE) ? 1 : 0;
		o.overflow = ovf[i];
	}
	return MDEMOD_OK;
} MDEMOD_API_CATCH

int
mdemod_get_lock_events(mdemod_ctx *ctx, uint32_t stream, mdemod_lock_event *out, uint32_t cap,
                       uint32_t *n, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!ctx || !n) return MDEMOD_ERR_PARAM;
	if (stream >= ctx->params.n_streams) return MDEMOD_ERR_RANGE;
	int rc = select_device(ctx);
	if (rc) return rc;
	hipStream_t st = static_cast<hipStream_t>(hip_stream);
	uint32_t cnt = 0;
	HIP_TRY(hipMemcpyAsync(&cnt, ctx->st.ev_this_call + stream, sizeof(cnt), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	*n = cnt;
	uint32_t m = cnt < MDEMOD_MAX_LOCK_EVENTS ? cnt : MDEMOD_MAX_LOCK_EVENTS;
	if (m > cap) m = cap;
	if (m && out) {
		HIP_TRY(hipMemcpyAsync(out, ctx->st.events + static_cast<size_t>(stream) * MDEMOD_MAX_LOCK_EVENTS,
		                       sizeof(mdemod_lock_event) * m, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
	}
	return MDEMOD_OK;
} MDEMOD_API_CATCH

#define ONE(field, hostvar, dir)                                                                   \
	HIP_TRY(hipMemcpyAsync(dir ? static_cast<void *>(ctx->st.field + stream) : static_cast<void *>(&(hostvar)), \
	                       dir ? static_cast<const void *>(&(hostvar)) : static_cast<const void *>(ctx->st.field + stream), \
	                       sizeof(hostvar), dir ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, st))

int
mdemod_get_state(mdemod_ctx *ctx, uint32_t stream, mdemod_stream_state *out, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!ctx || !out) return MDEMOD_ERR_PARAM;
	if (stream >= ctx->params.n_streams) return MDEMOD_ERR_RANGE;
	int rc = select_device(ctx);
	if (rc) return rc;
	hipStream_t st = static_cast<hipStream_t>(hip_stream);
	int32_t flags = 0;
	ONE(agc_gain, out->agc_gain, 0); ONE(agc_bias_re, out->agc_bias_re, 0); ONE(agc_bias_im, out->agc_bias_im, 0);
	ONE(pll_phase, out->pll_phase, 0); ONE(pll_freq, out->pll_freq, 0); ONE(pll_err, out->pll_err, 0);
	ONE(t_phase, out->t_phase, 0); ONE(t_freq, out->t_freq, 0); ONE(t_prev, out->t_prev, 0);
	ONE(inphase, out->oqpsk_inphase, 0); ONE(flags, flags, 0);
	ONE(n_samples, out->n_samples, 0); ONE(n_symbols, out->n_symbols, 0); ONE(first_lock, out->first_lock_symbol, 0);
	HIP_TRY(hipStreamSynchronize(st));
	out->pll_locked = (flags & MDEMOD_FLAG_LOCKED) ? 1 : 0;
	out->pll_locked_once = (flags & MDEMOD_FLAG_LOCKED_ONCE) ? 1 : 0;
	out->pll_updown = (flags & MDEMOD_FLAG_UPDOWN_POS) ? 1 : -1;
	out->t_dual_state = (flags >> MDEMOD_FLAG_DUAL_SHIFT) & 3;
	return MDEMOD_OK;
} MDEMOD_API_CATCH

/* A carrier phase / frequency word the reference's loop can hold: pll.c:113 leaves the phase inside (-2pi, 2pi), pll.c:60 adds
 * the frequency word, pll.c:126-128 clamp that to +-fmax.  The kernels' NCO relies on |phase| + |freq| < 4pi (demod_device.h:
 * md_turn_code, md_nco_advance); NaN (what NaN input makes of them) goes through every path as in the reference. */
static bool
carrier_in_domain(const mdemod_stream_state &v)
{
	const float a = fabsf(v.pll_phase) + fabsf(v.pll_freq);
	return !(a >= 12.5f);
}

/* A clock word the reference's loop can hold: timing.c:80-86 keeps it within center / 4096 of the centre, and the kernels' symbol
 * clock counts on it (steps that provably cannot fire: step_fmax, clock_jump.h).  NaN is refused like any other word outside (r06: a
 * NaN clock never fires - every kernel would spin to its watchdog and report an overflow after a full-length run). */
static bool
clock_in_domain(const mdemod_ctx *ctx, const mdemod_stream_state &v)
{
	const DemodConsts &c = ctx->tab.c;
	/* (centre + fd is a rounded float sum: the bound carries the same 1e-6 of slack as step_fmax, the one the kernels use) */
	const double lo = (static_cast<double>(c.t_center) - static_cast<double>(c.t_maxdev)) * (1.0 - 1e-6);
	return v.t_freq <= c.step_fmax && static_cast<double>(v.t_freq) >= lo;          /* (positive comparisons: NaN is outside) */
}

static void
note_domain(const mdemod_ctx *ctx, const mdemod_stream_state &v)
{
	const DemodConsts &c = ctx->tab.c;
	if (!carrier_in_domain(v))
		mdm_note_error("carrier state pll_phase %g, pll_freq %g: |phase| + |freq| must stay below 12.5 (pll.c:113 wraps once per symbol)",
		                  static_cast<double>(v.pll_phase), static_cast<double>(v.pll_freq));
	else
		mdm_note_error("symbol-clock word t_freq %.9g is outside what the reference's loop can hold, %.9g +- %.9g (timing.c:80-86)",
		                  static_cast<double>(v.t_freq), static_cast<double>(c.t_center), static_cast<double>(c.t_maxdev));
}

int
mdemod_set_state(mdemod_ctx *ctx, uint32_t stream, const mdemod_stream_state *in, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!ctx || !in) return MDEMOD_ERR_PARAM;
	if (stream >= ctx->params.n_streams) return MDEMOD_ERR_RANGE;
	if (in->t_dual_state != 1 && in->t_dual_state != 2) return MDEMOD_ERR_PARAM;
	if (!carrier_in_domain(*in) || !clock_in_domain(ctx, *in)) { note_domain(ctx, *in); return MDEMOD_ERR_PARAM; }
	int rc = select_device(ctx);
	if (rc) return rc;
	hipStream_t st = static_cast<hipStream_t>(hip_stream);
	mdemod_stream_state v = *in;
	int32_t flags = (v.pll_locked ? MDEMOD_FLAG_LOCKED : 0) | (v.pll_locked_once ? MDEMOD_FLAG_LOCKED_ONCE : 0) |
	                (v.pll_updown > 0 ? MDEMOD_FLAG_UPDOWN_POS : 0) | (v.t_dual_state << MDEMOD_FLAG_DUAL_SHIFT);
	ONE(agc_gain, v.agc_gain, 1); ONE(agc_bias_re, v.agc_bias_re, 1); ONE(agc_bias_im, v.agc_bias_im, 1);
	ONE(pll_phase, v.pll_phase, 1); ONE(pll_freq, v.pll_freq, 1); ONE(pll_err, v.pll_err, 1);
	ONE(t_phase, v.t_phase, 1); ONE(t_freq, v.t_freq, 1); ONE(t_prev, v.t_prev, 1);
	ONE(inphase, v.oqpsk_inphase, 1); ONE(flags, flags, 1);
	ONE(n_samples, v.n_samples, 1); ONE(n_symbols, v.n_symbols, 1); ONE(first_lock, v.first_lock_symbol, 1);
	HIP_TRY(hipStreamSynchronize(st));
	return MDEMOD_OK;
} MDEMOD_API_CATCH
#undef ONE

int
mdemod_set_state_all(mdemod_ctx *ctx, const mdemod_stream_state *seed, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!ctx || !seed) return MDEMOD_ERR_PARAM;
	if (seed->t_dual_state != 1 && seed->t_dual_state != 2) return MDEMOD_ERR_PARAM;
	if (!carrier_in_domain(*seed) || !clock_in_domain(ctx, *seed)) { note_domain(ctx, *seed); return MDEMOD_ERR_PARAM; }
	int rc = select_device(ctx);
	if (rc) return rc;
	const int32_t flags = (seed->pll_locked ? MDEMOD_FLAG_LOCKED : 0) | (seed->pll_locked_once ? MDEMOD_FLAG_LOCKED_ONCE : 0) |
	                      (seed->pll_updown > 0 ? MDEMOD_FLAG_UPDOWN_POS : 0) | (seed->t_dual_state << MDEMOD_FLAG_DUAL_SHIFT);
	HIP_TRY(mdemod_launch_seed(ctx->st, ctx->tab.c, *seed, flags, ctx->params.bps, ctx->tab.use_rw ? 1 : 0,
	                           ctx->params.n_streams, static_cast<hipStream_t>(hip_stream)));
	return MDEMOD_OK;
} MDEMOD_API_CATCH

int
mdemod_rotate_carrier(mdemod_ctx *ctx, const int32_t *quarter_turns_dev, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!ctx || !quarter_turns_dev) return MDEMOD_ERR_PARAM;
	int rc = select_device(ctx);
	if (rc) return rc;
	HIP_TRY(mdemod_launch_rotate(ctx->st, quarter_turns_dev, ctx->params.n_streams, ctx->params.oqpsk ? 1 : 0, static_cast<hipStream_t>(hip_stream)));
	return MDEMOD_OK;
} MDEMOD_API_CATCH

int
mdemod_set_carrier_seeds(mdemod_ctx *ctx, const float *freq_dev, const int32_t *updown_dev, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!ctx || !freq_dev || !updown_dev) return MDEMOD_ERR_PARAM;
	int rc = select_device(ctx);
	if (rc) return rc;
	HIP_TRY(mdemod_launch_carrier_seeds(ctx->st, freq_dev, updown_dev, ctx->params.n_streams, static_cast<hipStream_t>(hip_stream)));
	return MDEMOD_OK;
} MDEMOD_API_CATCH

int
mdemod_set_gain_seeds(mdemod_ctx *ctx, const float *gain_dev, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!ctx || !gain_dev) return MDEMOD_ERR_PARAM;
	int rc = select_device(ctx);
	if (rc) return rc;
	HIP_TRY(mdemod_launch_gain_seeds(ctx->st, gain_dev, ctx->params.n_streams, static_cast<hipStream_t>(hip_stream)));
	return MDEMOD_OK;
} MDEMOD_API_CATCH

int
mdemod_set_clock_seeds(mdemod_ctx *ctx, const float *t_freq_dev, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!ctx || !t_freq_dev) return MDEMOD_ERR_PARAM;
	int rc = select_device(ctx);
	if (rc) return rc;
	/* clamped on the way in to the words the reference's loop can hold (clock_in_domain above: what mdemod_set_state refuses) */
	const DemodConsts &c = ctx->tab.c;
	float lo = static_cast<float>((static_cast<double>(c.t_center) - static_cast<double>(c.t_maxdev)) * (1.0 - 1e-6));
	if (static_cast<double>(lo) < (static_cast<double>(c.t_center) - static_cast<double>(c.t_maxdev)) * (1.0 - 1e-6)) lo = nextafterf(lo, 1e30f);
	HIP_TRY(mdemod_launch_clock_seeds(ctx->st, t_freq_dev, lo, c.step_fmax, ctx->params.n_streams, static_cast<hipStream_t>(hip_stream)));
	return MDEMOD_OK;
} MDEMOD_API_CATCH

int
mdemod_get_states(mdemod_ctx *ctx, uint32_t first, uint32_t count, mdemod_stream_state *out, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!ctx || !out) return MDEMOD_ERR_PARAM;
	if (static_cast<uint64_t>(first) + count > ctx->params.n_streams) return MDEMOD_ERR_RANGE;
	if (!count) return MDEMOD_OK;
	int rc = select_device(ctx);
	if (rc) return rc;
	hipStream_t st = static_cast<hipStream_t>(hip_stream);
	const DemodStateSoA &s = ctx->st;
	std::vector<float> f[10]; std::vector<int32_t> flags; std::vector<uint64_t> nsamp, nsym; std::vector<int64_t> fl;
	float *const src[10] = { s.agc_gain, s.agc_bias_re, s.agc_bias_im, s.pll_phase, s.pll_freq, s.pll_err, s.t_phase, s.t_freq, s.t_prev, s.inphase };
	for (int k = 0; k < 10; k++) if ((rc = fetch(src[k], first, count, f[k], st))) return rc;
	if ((rc = fetch(s.flags, first, count, flags, st))) return rc;
	if ((rc = fetch(s.n_samples, first, count, nsamp, st))) return rc;
	if ((rc = fetch(s.n_symbols, first, count, nsym, st))) return rc;
	if ((rc = fetch(s.first_lock, first, count, fl, st))) return rc;
	HIP_TRY(hipStreamSynchronize(st));
	for (uint32_t i = 0; i < count; i++) {
		mdemod_stream_state &o = out[i];
		o.agc_gain = f[0][i]; o.agc_bias_re = f[1][i]; o.agc_bias_im = f[2][i];
		o.pll_phase = f[3][i]; o.pll_freq = f[4][i]; o.pll_err = f[5][i];
		o.t_phase = f[6][i]; o.t_freq = f[7][i]; o.t_prev = f[8][i]; o.oqpsk_inphase = f[9][i];
		o.pll_locked = (flags[i] & MDEMOD_FLAG_LOCKED) ? 1 : 0;
		o.pll_locked_once = (flags[i] & MDEMOD_FLAG_LOCKED_ONCE) ? 1 : 0;
		o.pll_updown = (flags[i] & MDEMOD_FLAG_UPDOWN_POS) ? 1 : -1;
		o.t_dual_state = (flags[i] >> MDEMOD_FLAG_DUAL_SHIFT) & 3;
		o.n_samples = nsamp[i]; o.n_symbols = nsym[i]; o.first_lock_symbol = fl[i];
	}
	return MDEMOD_OK;
} MDEMOD_API_CATCH

int
mdemod_copy_state(mdemod_ctx *dst, mdemod_ctx *src, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!dst || !src) return MDEMOD_ERR_PARAM;
	const mdemod_params &a = dst->params, &b = src->params;
	/* the same layout (streams, sample format, history) - and the same meaning: a QPSK bank's state words are not an OQPSK bank's (the
	 * rail of timing.c:43 lives in the flags), nor one filter's history another's */
	if (a.n_streams != b.n_streams || a.bps != b.bps || a.device != b.device || dst->tab.use_rw != src->tab.use_rw ||
	    dst->tab.c.hpad != src->tab.c.hpad || (a.oqpsk != 0) != (b.oqpsk != 0) || a.rrc_order != b.rrc_order || a.interp_factor != b.interp_factor) {
		mdm_note_error("mdemod_copy_state: the two contexts differ in streams (%u / %u), sample format, device, modulation, -f or -O: a bank's state only fits a bank made from the same parameters",
		               a.n_streams, b.n_streams);
		return MDEMOD_ERR_PARAM;
	}
	int rc = select_device(dst);
	if (rc) return rc;
	hipStream_t st = static_cast<hipStream_t>(hip_stream);
	const size_t n = a.n_streams;
	const DemodStateSoA &d = dst->st, &s = src->st;
#define CP(field) HIP_TRY(hipMemcpyAsync(d.field, s.field, sizeof(*d.field) * n, hipMemcpyDeviceToDevice, st))
	CP(agc_gain); CP(agc_bias_re); CP(agc_bias_im); CP(pll_phase); CP(pll_freq); CP(pll_err); CP(t_phase); CP(t_freq); CP(t_prev);
	CP(inphase); CP(flags); CP(n_samples); CP(n_symbols); CP(first_lock); CP(sym_this_call); CP(ev_this_call); CP(overflow);
#undef CP
	HIP_TRY(hipMemcpyAsync(d.hist, s.hist, static_cast<size_t>(dst->tab.c.hpad) * n * (dst->tab.use_rw ? 8 : dst->sample_bytes),
	                       hipMemcpyDeviceToDevice, st));
	HIP_TRY(hipMemcpyAsync(d.events, s.events, sizeof(mdemod_lock_event) * MDEMOD_MAX_LOCK_EVENTS * n, hipMemcpyDeviceToDevice, st));
	return MDEMOD_OK;
} MDEMOD_API_CATCH

uint64_t
mdemod_nominal_pitch(const mdemod_ctx *ctx, uint64_t n_samples)
{
	return mdemod_nominal_symbols(ctx, n_samples);
}

int
mdemod_compact_soft(mdemod_ctx *ctx, const int8_t *soft_dev, uint64_t soft_stride_symbols,
                    int8_t *out_dev, uint64_t out_pitch_symbols, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!ctx || !soft_dev || !out_dev || (soft_stride_symbols & 7) || (out_pitch_symbols & 7)) return MDEMOD_ERR_PARAM;
	int rc = select_device(ctx);
	if (rc) return rc;
	HIP_TRY(mdemod_launch_compact_rows(soft_dev, soft_stride_symbols, out_dev, out_pitch_symbols, ctx->st.sym_this_call,
	                                   ctx->params.n_streams, static_cast<hipStream_t>(hip_stream)));
	return MDEMOD_OK;
} MDEMOD_API_CATCH

int
mdemod_fanin_peer(mdemod_ctx *src, const int8_t *soft_dev, uint64_t soft_stride_symbols, int dst_device,
                  int8_t *dst_soft_dev, uint64_t pitch_symbols, uint64_t first_row, uint32_t *dst_counts_dev, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!src || !soft_dev || !dst_soft_dev || (soft_stride_symbols & 7) || (pitch_symbols & 7) || dst_device < 0) return MDEMOD_ERR_PARAM;
	int rc = select_device(src);
	if (rc) return rc;
	const int src_device = src->params.device;
	if (dst_device != src_device) {
		/* the compaction kernel runs on src's GPU and stores into dst's memory: the stores themselves cross xGMI */
		int can = 0;
		HIP_TRY(hipDeviceCanAccessPeer(&can, src_device, dst_device));
		if (!can) {
			mdm_note_error("mdemod_fanin_peer: device %d cannot reach the memory of device %d (no peer access between them)", src_device, dst_device);
			return MDEMOD_ERR_HIP;
		}
		const hipError_t e = hipDeviceEnablePeerAccess(dst_device, 0);
		if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
		else HIP_TRY(e);
	}
	hipStream_t st = static_cast<hipStream_t>(hip_stream);
	const uint32_t n = src->params.n_streams;
	HIP_TRY(mdemod_launch_compact_rows(soft_dev, soft_stride_symbols, dst_soft_dev + 2 * first_row * pitch_symbols, pitch_symbols,
	                                   src->st.sym_this_call, n, st));
	if (dst_counts_dev)
		HIP_TRY(hipMemcpyPeerAsync(dst_counts_dev + first_row, dst_device, src->st.sym_this_call, src_device, sizeof(uint32_t) * n, st));
	return MDEMOD_OK;
} MDEMOD_API_CATCH

const char *
mdemod_kernel_name(const mdemod_ctx *ctx)
{
	if (!ctx) return "";
	if (wants_latency_kernel(ctx)) return "demod_kernel_lat (one stream per wave: FIR farm + serial scalar stage)";
	if (!ctx->tab.use_rw) return "demod_kernel (v1 LDS ring)";
	if (ctx->tab.rw_gather) return ctx->tab.c.taps > 65 ? "demod_kernel_gat (v3 gather: sample rates beyond every window, 129 taps, every firing loads its own taps)"
	                                                     : "demod_kernel_gat (v3 gather: sample rates beyond every window, 65 taps, every firing loads its own taps)";
	if (ctx->tab.rw_hyb && ctx->tab.rw_far) return "demod_kernel_roth (v3 hybrid window, far: float input, 65 taps at up to 54 samples per firing, 80 slots in VGPRs + 40 in AccVGPRs)";
	if (ctx->tab.rw_hyb) return ctx->tab.rw_mid ? "demod_kernel_roth (v3 hybrid window, mid: float input, 65 taps at up to 30 samples per firing, 80 slots in VGPRs + 16 in AccVGPRs)"
	                                            : "demod_kernel_roth (v3 hybrid window: float input, 129 taps, 80 slots in VGPRs + 80 in AccVGPRs)";
	if (ctx->tab.rw_compact4) return ctx->tab.rw_wide ? "demod_kernel_rotp (v3 rotating packed window, wide: 129 taps at up to 30 samples per firing)"
	                                 : (ctx->tab.rw_mid ? "demod_kernel_rotp (v3 rotating packed window, mid: 65 taps at up to 15 samples per firing)"
	                                                    : "demod_kernel_rotp (v3 rotating packed window, far: 65 taps at up to 46 samples per firing)");
	return "demod_kernel_rot (v3 rotating register window)";
}

int
mdemod_plan_kernel(const mdemod_params *params, char *name, uint32_t name_cap, uint32_t *lds_bytes, uint32_t *block_threads)
try { MDEMOD_API_ENTER
	if (!params || !name || name_cap == 0 || params->n_streams == 0) return MDEMOD_ERR_PARAM;
	mdemod_ctx *ctx = new (std::nothrow) mdemod_ctx();
	if (!ctx) return MDEMOD_ERR_NOMEM;
	ctx->params = *params;
	ctx->pipe = nullptr;
	const int rc = plan_context(ctx);
	if (rc == MDEMOD_OK) {
		snprintf(name, name_cap, "%s%s", mdemod_kernel_name(ctx), ctx->v1_global_table && !wants_latency_kernel(ctx) ? " [table in global memory]" : "");
		if (lds_bytes) *lds_bytes = static_cast<uint32_t>(wants_latency_kernel(ctx) ? ctx->lat_lds : ctx->lds_bytes);
		if (block_threads) *block_threads = static_cast<uint32_t>(ctx->tab.use_rw ? (ctx->tab.rw_wide ? MDEMOD_RW_WIDE_BLOCK : (ctx->tab.c.sin_lut ? MDEMOD_RW_LUT_BLOCK : MDEMOD_RW_BLOCK)) : ctx->block_threads);
	}
	delete ctx;
	return rc;
} MDEMOD_API_CATCH

uint32_t
mdemod_history_len(const mdemod_ctx *ctx)
{
	return ctx ? static_cast<uint32_t>(ctx->tab.c.hpad) : 0;
}

int
mdemod_get_history(mdemod_ctx *ctx, uint32_t stream, float *iq_pairs, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!ctx || !iq_pairs) return MDEMOD_ERR_PARAM;
	if (stream >= ctx->params.n_streams) return MDEMOD_ERR_RANGE;
	int rc = select_device(ctx);
	if (rc) return rc;
	hipStream_t st = static_cast<hipStream_t>(hip_stream);
	const int hfmt = ctx->tab.use_rw ? 32 : ctx->params.bps;     /* the register-window kernels keep converted floats */
	const size_t sb = 2 * static_cast<size_t>(hfmt) / 8, ns = ctx->params.n_streams;
	const int hpad = ctx->tab.c.hpad;
	std::vector<unsigned char> raw(static_cast<size_t>(hpad) * sb);
	if (ctx->tab.use_rw)      /* register-window layout: [stream][hpad] */
		HIP_TRY(hipMemcpyAsync(raw.data(), static_cast<const unsigned char *>(ctx->st.hist) + static_cast<size_t>(stream) * hpad * sb,
		                       static_cast<size_t>(hpad) * sb, hipMemcpyDeviceToHost, st));
	else
		HIP_TRY(hipMemcpy2DAsync(raw.data(), sb, static_cast<const unsigned char *>(ctx->st.hist) + stream * sb,
		                         ns * sb, sb, hpad, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	for (int k = 0; k < hpad; k++) {
		const unsigned char *p = &raw[k * sb];
		if (hfmt == 8) { iq_pairs[2*k] = (float)((int)p[0] - 128); iq_pairs[2*k+1] = (float)((int)p[1] - 128); }
		else if (hfmt == 16) { int16_t v[2]; memcpy(v, p, 4); iq_pairs[2*k] = v[0]; iq_pairs[2*k+1] = v[1]; }
		else memcpy(&iq_pairs[2*k], p, 8);
	}
	return MDEMOD_OK;
} MDEMOD_API_CATCH

int
mdemod_set_history(mdemod_ctx *ctx, uint32_t stream, const float *iq_pairs, void *hip_stream)
try { MDEMOD_API_ENTER
	if (!ctx || !iq_pairs) return MDEMOD_ERR_PARAM;
	if (stream >= ctx->params.n_streams) return MDEMOD_ERR_RANGE;
	int rc = select_device(ctx);
	if (rc) return rc;
	hipStream_t st = static_cast<hipStream_t>(hip_stream);
	const int hfmt = ctx->tab.use_rw ? 32 : ctx->params.bps;
	const size_t sb = 2 * static_cast<size_t>(hfmt) / 8, ns = ctx->params.n_streams;
	const int hpad = ctx->tab.c.hpad;
	std::vector<unsigned char> raw(static_cast<size_t>(hpad) * sb);
	for (int k = 0; k < hpad; k++) {
		unsigned char *p = &raw[k * sb];
		const float re = iq_pairs[2*k], im = iq_pairs[2*k+1];
		if (hfmt == 8) {
			if (re != floorf(re) || im != floorf(im) || re < -128 || re > 127 || im < -128 || im > 127) return MDEMOD_ERR_PARAM;
			p[0] = (unsigned char)((int)re + 128); p[1] = (unsigned char)((int)im + 128);
		} else if (hfmt == 16) {
			if (re != floorf(re) || im != floorf(im) || re < -32768 || re > 32767 || im < -32768 || im > 32767) return MDEMOD_ERR_PARAM;
			int16_t v[2] = { (int16_t)re, (int16_t)im }; memcpy(p, v, 4);
		} else memcpy(p, &iq_pairs[2*k], 8);
	}
	if (ctx->tab.use_rw)
		HIP_TRY(hipMemcpyAsync(static_cast<unsigned char *>(ctx->st.hist) + static_cast<size_t>(stream) * hpad * sb, raw.data(),
		                       static_cast<size_t>(hpad) * sb, hipMemcpyHostToDevice, st));
	else
		HIP_TRY(hipMemcpy2DAsync(static_cast<unsigned char *>(ctx->st.hist) + stream * sb, ns * sb, raw.data(), sb,
		                         sb, hpad, hipMemcpyHostToDevice, st));
	HIP_TRY(hipStreamSynchronize(st));
	return MDEMOD_OK;
} MDEMOD_API_CATCH

/* ---- tables -------------------------------------------------------------------- */

int
mdemod_get_rrc_table(const mdemod_ctx *ctx, float *out, uint32_t cap)
try { MDEMOD_API_ENTER
	if (!ctx || !out) return MDEMOD_ERR_PARAM;
	if (cap < ctx->tab.rrc.size()) return MDEMOD_ERR_PARAM;
	memcpy(out, ctx->tab.rrc.data(), ctx->tab.rrc.size() * sizeof(float));
	return static_cast<int>(ctx->tab.rrc.size());
} MDEMOD_API_CATCH

int
mdemod_get_loop_constants(const mdemod_ctx *ctx, float out[8])
try { MDEMOD_API_ENTER
	if (!ctx || !out) return MDEMOD_ERR_PARAM;
	const DemodConsts &c = ctx->tab.c;
	const float v[8] = { c.pll_alpha, c.pll_beta, c.pll_fmax, c.t_alpha, c.t_beta, c.t_center, c.t_maxdev, ctx->tab.osf };
	memcpy(out, v, sizeof(v));
	return MDEMOD_OK;
} MDEMOD_API_CATCH

int
mdemod_get_tanh_lut(const mdemod_ctx *ctx, float out[32])
try { MDEMOD_API_ENTER
	if (!ctx || !out) return MDEMOD_ERR_PARAM;
	memcpy(out, ctx->tab.tanh_lut, sizeof(ctx->tab.tanh_lut));
	return MDEMOD_OK;
} MDEMOD_API_CATCH

/* ---- device self-tests of the scalar primitives --------------------------------- */

int
mdemod_selftest_sincos(mdemod_ctx *ctx, const float *x, uint32_t n, float *sin_out, float *cos_out)
try { MDEMOD_API_ENTER
	if (!ctx || !x || !sin_out || !cos_out) return MDEMOD_ERR_PARAM;
	int rc = select_device(ctx);
	if (rc) return rc;
	float *d = nullptr;
	HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), sizeof(float) * 3 * static_cast<size_t>(n) + 16));
	hipError_t e = hipMemcpy(d, x, sizeof(float) * n, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = mdemod_launch_selftest_sincos(d, n, d + n, d + 2 * static_cast<size_t>(n), nullptr);
	if (e == hipSuccess) e = hipMemcpy(sin_out, d + n, sizeof(float) * n, hipMemcpyDeviceToHost);
	if (e == hipSuccess) e = hipMemcpy(cos_out, d + 2 * static_cast<size_t>(n), sizeof(float) * n, hipMemcpyDeviceToHost);
	(void)hipFree(d);
	return e == hipSuccess ? MDEMOD_OK : MDEMOD_ERR_HIP;
} MDEMOD_API_CATCH

int
mdemod_selftest_turncode(mdemod_ctx *ctx, uint64_t *n_checked, uint64_t *n_mismatch)
try { MDEMOD_API_ENTER
	if (!ctx || !n_mismatch) return MDEMOD_ERR_PARAM;
	int rc = select_device(ctx);
	if (rc) return rc;
	unsigned long long *d = nullptr, h = 0;
	HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), sizeof(h)));
	hipError_t e = hipMemset(d, 0, sizeof(h));
	if (e == hipSuccess) e = mdemod_launch_selftest_turncode(d, nullptr);
	if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
	(void)hipFree(d);
	if (e != hipSuccess) return MDEMOD_ERR_HIP;
	*n_mismatch = h;
	if (n_checked) *n_checked = 2ull * 0x41800000ull;
	return MDEMOD_OK;
} MDEMOD_API_CATCH

int
mdemod_selftest_cabsf(mdemod_ctx *ctx, uint64_t pairs, uint64_t *n_mismatch, uint64_t *n_fallback)
try { MDEMOD_API_ENTER
	if (!ctx || !n_mismatch || !pairs) return MDEMOD_ERR_PARAM;
	int rc = select_device(ctx);
	if (rc) return rc;
	unsigned long long *d = nullptr, h[2] = { 0, 0 };
	HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), sizeof(h)));
	hipError_t e = hipMemset(d, 0, sizeof(h));
	if (e == hipSuccess) e = mdemod_launch_selftest_cabsf(pairs, d, nullptr);
	if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	(void)hipFree(d);
	if (e != hipSuccess) return MDEMOD_ERR_HIP;
	*n_mismatch = h[0];
	if (n_fallback) *n_fallback = h[1];
	return MDEMOD_OK;
} MDEMOD_API_CATCH

int
mdemod_selftest_sinlut(mdemod_ctx *ctx, uint64_t *n_checked, uint64_t *n_mismatch)
try { MDEMOD_API_ENTER
	if (!ctx || !n_mismatch) return MDEMOD_ERR_PARAM;
	int rc = select_device(ctx);
	if (rc) return rc;
	unsigned long long *d = nullptr, h = 0;
	HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), sizeof(h)));
	hipError_t e = hipMemset(d, 0, sizeof(h));
	if (e == hipSuccess) e = mdemod_launch_selftest_sinlut(d, nullptr);
	if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
	(void)hipFree(d);
	if (e != hipSuccess) return MDEMOD_ERR_HIP;
	*n_mismatch = h;
	if (n_checked) *n_checked = 4ull * 65536ull;
	return MDEMOD_OK;
} MDEMOD_API_CATCH

int
mdemod_selftest_hypot(mdemod_ctx *ctx, const float *xy, uint32_t n_pairs, float *out)
try { MDEMOD_API_ENTER
	if (!ctx || !xy || !out) return MDEMOD_ERR_PARAM;
	int rc = select_device(ctx);
	if (rc) return rc;
	float *d = nullptr;
	HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), sizeof(float) * 3 * static_cast<size_t>(n_pairs) + 16));
	hipError_t e = hipMemcpy(d, xy, sizeof(float) * 2 * n_pairs, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = mdemod_launch_selftest_hypot(d, n_pairs, d + 2 * static_cast<size_t>(n_pairs), nullptr);
	if (e == hipSuccess) e = hipMemcpy(out, d + 2 * static_cast<size_t>(n_pairs), sizeof(float) * n_pairs, hipMemcpyDeviceToHost);
	(void)hipFree(d);
	return e == hipSuccess ? MDEMOD_OK : MDEMOD_ERR_HIP;
} MDEMOD_API_CATCH

} /* extern "C" */
