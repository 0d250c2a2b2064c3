/*
 * demod_aux.hip — small support kernels: power-on state reset and the device
 * self-tests of the scalar primitives (used by the parity tests to pin
 * fast_sin/fast_cos/cabsf on the GPU against the reference's values).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "demod_internal.h"
#include "demod_device.h"

#pragma clang fp contract(off)

namespace {

/* Power-on state of the reference's file-static globals:
 * agc.c:9-10, pll.c:33-36,112, timing.c:13-14,21,43, filter.c:16 (calloc). */
template <typename sample_t>
__global__ void
reset_kernel(DemodStateSoA st, float t_center, int hpad, uint32_t n_streams, sample_t zero, int stream_major)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_streams) return;
	st.agc_gain[s] = 1.0f; st.agc_bias_re[s] = 0.0f; st.agc_bias_im[s] = 0.0f;
	st.pll_phase[s] = 0.0f; st.pll_freq[s] = 0.0f; st.pll_err[s] = 1000.0f;
	st.t_phase[s] = 0.0f; st.t_freq[s] = t_center; st.t_prev[s] = 0.0f;
	st.inphase[s] = 0.0f;
	st.flags[s] = MDEMOD_FLAG_UPDOWN_POS | (1 << MDEMOD_FLAG_DUAL_SHIFT);
	st.n_samples[s] = 0; st.n_symbols[s] = 0; st.first_lock[s] = -1;
	st.sym_this_call[s] = 0; st.ev_this_call[s] = 0; st.overflow[s] = 0;
	sample_t *hist = reinterpret_cast<sample_t *>(st.hist);
	for (int k = 0; k < hpad; k++) hist[stream_major ? (size_t)s * hpad + k : (size_t)k * n_streams + s] = zero;
}

/* Broadcast one loop state to every stream (seed of overlapped tiles); history := zeros. */
template <typename sample_t>
__global__ void
seed_kernel(DemodStateSoA st, mdemod_stream_state v, int32_t flags, int hpad, uint32_t n_streams, sample_t zero, int stream_major)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_streams) return;
	st.agc_gain[s] = v.agc_gain; st.agc_bias_re[s] = v.agc_bias_re; st.agc_bias_im[s] = v.agc_bias_im;
	st.pll_phase[s] = v.pll_phase; st.pll_freq[s] = v.pll_freq; st.pll_err[s] = v.pll_err;
	st.t_phase[s] = v.t_phase; st.t_freq[s] = v.t_freq; st.t_prev[s] = v.t_prev;
	st.inphase[s] = v.oqpsk_inphase;
	st.flags[s] = flags;
	st.n_samples[s] = v.n_samples; st.n_symbols[s] = v.n_symbols; st.first_lock[s] = v.first_lock_symbol;
	st.sym_this_call[s] = 0; st.ev_this_call[s] = 0; st.overflow[s] = 0;
	sample_t *hist = reinterpret_cast<sample_t *>(st.hist);
	for (int k = 0; k < hpad; k++) hist[stream_major ? (size_t)s * hpad + k : (size_t)k * n_streams + s] = zero;
}

/* phase += k * pi/2, wrapped with the dividend's sign like pll.c:113.  OQPSK: an odd number of quarter turns also
 * swaps the rails, whose firings are half a symbol apart (demod.c:66-76): the firing the symbol clock is waiting for
 * changes its role (timing.c:41-57: state 1 fires at pi = in-phase, state 2 at 2*pi = quadrature + retime), i.e. the
 * clock phase moves by pi and the state toggles. */
__global__ void
rotate_kernel(DemodStateSoA st, const int32_t *quarter_turns, uint32_t n_streams, int oqpsk)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_streams) return;
	const int32_t k = quarter_turns[s] & 3;
	if (!k) return;
	const double p = (double)st.pll_phase[s] + (double)k * MD_HALF_PI_D;
	st.pll_phase[s] = (float)fmod(p, MD_TWO_PI_D);
	if (oqpsk && (k & 1)) {
		const int fl = st.flags[s];
		const int state = (fl >> MDEMOD_FLAG_DUAL_SHIFT) & 3;
		const float tp = st.t_phase[s];
		st.t_phase[s] = (state == 1) ? tp + MD_PI_F : tp - MD_PI_F;
		st.flags[s] = (fl & ~(3 << MDEMOD_FLAG_DUAL_SHIFT)) | ((state == 1 ? 2 : 1) << MDEMOD_FLAG_DUAL_SHIFT);
		/* the sample held for the next pairing changes rail too: out * j^k has I' = -Q, Q' = I (k = 1) or I' = Q,
		 * Q' = -I (k = 3); the last quadrature sample is t_prev (timing.c:65), the pending in-phase one is inphase */
		const float last_q = st.t_prev[s], pend_i = st.inphase[s];
		if (state == 1) st.inphase[s] = (k == 1) ? -last_q : last_q;      /* now waiting for Q': its I' is the old last Q */
		else st.t_prev[s] = (k == 1) ? pend_i : -pend_i;                   /* now waiting for I': the last Q' is the old pending I */
	}
}

__global__ void
carrier_seed_kernel(DemodStateSoA st, const float *freq, const int32_t *updown, uint32_t n_streams)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_streams) return;
	st.pll_freq[s] = freq[s];
	const int fl = st.flags[s];
	st.flags[s] = updown[s] > 0 ? (fl | MDEMOD_FLAG_UPDOWN_POS) : (fl & ~MDEMOD_FLAG_UPDOWN_POS);
}

__global__ void
gain_seed_kernel(DemodStateSoA st, const float *gain, uint32_t n_streams)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_streams) return;
	const float g = gain[s];
	st.agc_gain[s] = g > 0.0f ? g : 0.0f;                 /* the reference clamps at zero too (agc.c:23) */
}

/* Clock words from a device array (the stitcher's per-tile estimates): kept inside what timing.c:80-86 can hold, the range the
 * kernels' symbol clock counts on (step_fmax, clock_jump.h) - the same bounds mdemod_set_state checks on the host.  NaN -> the nominal rate. */
__global__ void
clock_seed_kernel(DemodStateSoA st, const float *t_freq, float lo, float hi, uint32_t n_streams)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_streams) return;
	const float f = t_freq[s];
	/* NaN is out of the domain like any other word the loop cannot hold (a NaN clock never fires: the kernel would spin to its
	 * watchdog): it becomes the nominal rate, the middle of the range */
	st.t_freq[s] = f != f ? 0.5f * (lo + hi) : (f < lo ? lo : (f > hi ? hi : f));
}

/* Host path: the demodulator writes its soft symbols with the hard-bound row pitch (one symbol per input sample);
 * what goes over PCIe is a copy with the nominal pitch.  One block per stream, 16-byte moves (pitches are multiples of
 * 8 symbols). */
__global__ void
compact_rows_kernel(const int8_t *src, uint64_t src_pitch, int8_t *dst, uint64_t dst_pitch, const uint32_t *counts, uint32_t n_streams)
{
	const uint32_t s = blockIdx.x;
	if (s >= n_streams) return;
	uint32_t m = counts[s];
	if (m > dst_pitch) m = (uint32_t)dst_pitch;
	const uint4 *a = reinterpret_cast<const uint4 *>(src + 2 * src_pitch * s);
	uint4 *b = reinterpret_cast<uint4 *>(dst + 2 * dst_pitch * s);
	for (uint32_t k = threadIdx.x; k < (m + 7) / 8; k += blockDim.x) b[k] = a[k];
}

__global__ void
selftest_sincos_kernel(const float *x, uint32_t n, float *s, float *c)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	s[i] = md_fast_sin(x[i]);
	c[i] = md_fast_cos(x[i]);
}

__global__ void
selftest_hypot_kernel(const float *xy, uint32_t n, float *out)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	out[i] = md_cabsf(xy[2 * i], xy[2 * i + 1]);
}

/* md_cabsf_short (short square root + exact fallback next to a rounding boundary) against md_cabsf_exact on pseudo-random pairs drawn like
 * the AGC's output (|re|, |im| up to a few hundred), on pairs across the whole float range, and on pairs with a zero: counts the
 * differences (must be 0) and the lanes that took the fallback. */
__global__ void
selftest_cabsf_kernel(uint64_t per_thread, unsigned long long *out /* [2]: mismatches, fallbacks */)
{
	uint64_t z = 0x9E3779B97F4A7C15ull * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1);
	unsigned long long bad = 0;
	uint32_t fb = 0;
	for (uint64_t k = 0; k < per_thread; k++) {
		z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; z += 0x9E3779B97F4A7C15ull;
		float re, im;
		const uint32_t a = (uint32_t)z, b = (uint32_t)(z >> 32);
		if ((k & 15) == 15) { re = __uint_as_float(a); im = __uint_as_float(b); }                       /* any two floats (NaN and inf among them) */
		else if ((k & 15) == 14) { re = (k & 16) ? 0.0f : -0.0f; im = __uint_as_float(b & 0x7FFFFFFFu) * 1e-30f; }
		else { re = ((float)(int32_t)a) * (400.0f / 2147483648.0f); im = ((float)(int32_t)b) * (400.0f / 2147483648.0f); }
		const float want = md_cabsf_exact(re, im), got = md_cabsf_short(re, im, &fb);
		bad += (__float_as_uint(want) != __float_as_uint(got)) && !(want != want && got != got);   /* (two NaNs may differ in payload: both are "NaN" to every consumer) */
	}
	if (bad) atomicAdd(&out[0], bad);
	if (fb) atomicAdd(&out[1], (unsigned long long)fb);
}

/* The lock events of this call, of the streams that have any, into a list of the same layout (mdemod_process_host keeps every
 * sub-block's events aside: the next launch overwrites the context's list).  Almost always nothing to do: streams lock once. */
__global__ void
copy_events_kernel(const mdemod_lock_event *src, const uint32_t *ev_this_call, mdemod_lock_event *dst, uint32_t n_streams,
                   const uint32_t *sym_this_call, uint32_t *sym_out, uint32_t *ev_out)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_streams) return;
	const uint32_t n_ev = ev_this_call[s];
	const uint32_t n = min(n_ev, (uint32_t)MDEMOD_MAX_LOCK_EVENTS);
	for (uint32_t e = 0; e < n; e++) dst[(size_t)s * MDEMOD_MAX_LOCK_EVENTS + e] = src[(size_t)s * MDEMOD_MAX_LOCK_EVENTS + e];
	/* the two counters of this launch, straight into the caller's (pinned host) arrays when it gives any */
	if (sym_out) sym_out[s] = sym_this_call[s];
	if (ev_out) ev_out[s] = n_ev;
}

/* Offsets and counts of a launch whose streams are all `count` samples long and `pitch` samples apart (mdemod_process_host on rows
 * the caller pinned: nothing to copy in besides the samples themselves). */
__global__ void
fill_uniform_rows_kernel(uint64_t *off, uint32_t *cnt, uint64_t pitch, uint32_t count, uint32_t n_streams)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_streams) return;
	off[s] = (uint64_t)s * pitch;
	cnt[s] = count;
}

/* Every float with |x| < 16, both signs: division-free turn code vs the real division. */
__global__ void
selftest_turncode_kernel(unsigned long long *mismatch)
{
	const uint32_t limit = 0x41800000u;                     /* bits of 16.0f */
	unsigned long long bad = 0;
	for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < limit; u += (uint64_t)gridDim.x * blockDim.x) {
		const float a = __uint_as_float((uint32_t)u), b = __uint_as_float((uint32_t)u | 0x80000000u);
		bad += (md_turn_code(a) != md_turn_code_div(a)) ? 1 : 0;
		bad += (md_turn_code(b) != md_turn_code_div(b)) ? 1 : 0;
	}
	if (bad) atomicAdd(mismatch, bad);
}

/* fast_sin's parabola from the table in LDS (md_sin_from_code_lut, what the LUT instances of the v3 kernels evaluate) against the
 * integer arithmetic of sincos.c:26-34 (md_sin_from_code), bit for bit: all 65 536 turn codes, with clean, sign-extended and
 * arbitrary upper halves of the 32-bit word the turn code arrives in. */
__global__ void
selftest_sinlut_kernel(unsigned long long *mismatch)
{
	extern __shared__ float sin_tab[];
	md_sin_lut_fill(sin_tab, (int)threadIdx.x, (int)blockDim.x);
	__syncthreads();
	unsigned long long bad = 0;
	for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < 65536u; c += gridDim.x * blockDim.x) {
		const uint32_t tops[4] = { 0u, (c & 0x8000u) ? 0xFFFF0000u : 0u, 0x5A5A0000u, 0xFFFF0000u ^ (c << 16) };
		for (int k = 0; k < 4; k++) {
			const int32_t wide = (int32_t)(tops[k] | c);
			bad += (__float_as_uint(md_sin_from_code(wide)) != __float_as_uint(md_sin_from_code_lut(sin_tab, wide))) ? 1 : 0;
		}
	}
	if (bad) atomicAdd(mismatch, bad);
}

} /* namespace */

hipError_t
mdemod_launch_selftest_cabsf(uint64_t pairs, unsigned long long *out_dev, hipStream_t stream)
{
	const unsigned blocks = 256 * 16, threads = 256;
	hipLaunchKernelGGL(selftest_cabsf_kernel, dim3(blocks), dim3(threads), 0, stream, (pairs + (uint64_t)blocks * threads - 1) / ((uint64_t)blocks * threads), out_dev);
	return hipGetLastError();
}

hipError_t
mdemod_launch_selftest_sinlut(unsigned long long *mismatch_dev, hipStream_t stream)
{
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(selftest_sinlut_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MDEMOD_SIN_LUT_BYTES);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(selftest_sinlut_kernel, dim3(64), dim3(256), MDEMOD_SIN_LUT_BYTES, stream, mismatch_dev);
	return hipGetLastError();
}

/* An empty launch: the first launch of a process loads the library's code objects (tens of milliseconds for the generated
 * assembly kernels); mdemod_init_device does it off the critical path of the first demodulation. */
namespace { __global__ void warm_kernel() {} }

hipError_t
mdemod_launch_warm(hipStream_t stream)
{
	hipLaunchKernelGGL(warm_kernel, dim3(1), dim3(64), 0, stream);
	return hipGetLastError();
}

hipError_t
mdemod_launch_selftest_turncode(unsigned long long *mismatch_dev, hipStream_t stream)
{
	hipLaunchKernelGGL(selftest_turncode_kernel, dim3(256 * 32), dim3(256), 0, stream, mismatch_dev);
	return hipGetLastError();
}

hipError_t
mdemod_launch_reset(const DemodStateSoA &st, const DemodConsts &c, int fmt, int float_history, uint32_t n_streams, hipStream_t stream)
{
	if (n_streams == 0) return hipSuccess;
	const dim3 block(256), grid((n_streams + 255) / 256);
	if (float_history) fmt = 32;        /* the register-window kernels keep their history as converted floats */
	switch (fmt) {
	case 16:
		hipLaunchKernelGGL(reset_kernel<uint32_t>, grid, block, 0, stream, st, c.t_center, c.hpad, n_streams, (uint32_t)0, float_history);
		break;
	case 8:   /* raw u8 encoding of 0.0 is 128 (wavfile.c:60) */
		hipLaunchKernelGGL(reset_kernel<uint16_t>, grid, block, 0, stream, st, c.t_center, c.hpad, n_streams, (uint16_t)0x8080, float_history);
		break;
	case 32:
		hipLaunchKernelGGL(reset_kernel<float2>, grid, block, 0, stream, st, c.t_center, c.hpad, n_streams, make_float2(0.0f, 0.0f), float_history);
		break;
	default:
		return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t
mdemod_launch_seed(const DemodStateSoA &st, const DemodConsts &c, const mdemod_stream_state &v, int32_t flags, int fmt,
                   int float_history, uint32_t n_streams, hipStream_t stream)
{
	if (n_streams == 0) return hipSuccess;
	const dim3 block(256), grid((n_streams + 255) / 256);
	if (float_history) fmt = 32;
	switch (fmt) {
	case 16:
		hipLaunchKernelGGL(seed_kernel<uint32_t>, grid, block, 0, stream, st, v, flags, c.hpad, n_streams, (uint32_t)0, float_history);
		break;
	case 8:
		hipLaunchKernelGGL(seed_kernel<uint16_t>, grid, block, 0, stream, st, v, flags, c.hpad, n_streams, (uint16_t)0x8080, float_history);
		break;
	case 32:
		hipLaunchKernelGGL(seed_kernel<float2>, grid, block, 0, stream, st, v, flags, c.hpad, n_streams, make_float2(0.0f, 0.0f), float_history);
		break;
	default:
		return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t
mdemod_launch_rotate(const DemodStateSoA &st, const int32_t *quarter_turns_dev, uint32_t n_streams, int oqpsk, hipStream_t stream)
{
	if (n_streams == 0) return hipSuccess;
	hipLaunchKernelGGL(rotate_kernel, dim3((n_streams + 255) / 256), dim3(256), 0, stream, st, quarter_turns_dev, n_streams, oqpsk);
	return hipGetLastError();
}

hipError_t
mdemod_launch_carrier_seeds(const DemodStateSoA &st, const float *freq_dev, const int32_t *updown_dev, uint32_t n_streams, hipStream_t stream)
{
	if (n_streams == 0) return hipSuccess;
	hipLaunchKernelGGL(carrier_seed_kernel, dim3((n_streams + 255) / 256), dim3(256), 0, stream, st, freq_dev, updown_dev, n_streams);
	return hipGetLastError();
}

hipError_t
mdemod_launch_gain_seeds(const DemodStateSoA &st, const float *gain_dev, uint32_t n_streams, hipStream_t stream)
{
	if (n_streams == 0) return hipSuccess;
	hipLaunchKernelGGL(gain_seed_kernel, dim3((n_streams + 255) / 256), dim3(256), 0, stream, st, gain_dev, n_streams);
	return hipGetLastError();
}

hipError_t
mdemod_launch_clock_seeds(const DemodStateSoA &st, const float *t_freq_dev, float lo, float hi, uint32_t n_streams, hipStream_t stream)
{
	if (n_streams == 0) return hipSuccess;
	hipLaunchKernelGGL(clock_seed_kernel, dim3((n_streams + 255) / 256), dim3(256), 0, stream, st, t_freq_dev, lo, hi, n_streams);
	return hipGetLastError();
}

hipError_t
mdemod_launch_compact_rows(const int8_t *src, uint64_t src_pitch_sym, int8_t *dst, uint64_t dst_pitch_sym, const uint32_t *counts_dev,
                           uint32_t n_streams, hipStream_t stream)
{
	if (n_streams == 0) return hipSuccess;
	hipLaunchKernelGGL(compact_rows_kernel, dim3(n_streams), dim3(128), 0, stream, src, src_pitch_sym, dst, dst_pitch_sym, counts_dev, n_streams);
	return hipGetLastError();
}

hipError_t
mdemod_launch_fill_uniform_rows(uint64_t *off_dev, uint32_t *cnt_dev, uint64_t pitch_samples, uint32_t count, uint32_t n_streams, hipStream_t stream)
{
	if (n_streams == 0) return hipSuccess;
	hipLaunchKernelGGL(fill_uniform_rows_kernel, dim3((n_streams + 255) / 256), dim3(256), 0, stream, off_dev, cnt_dev, pitch_samples, count, n_streams);
	return hipGetLastError();
}

hipError_t
mdemod_launch_copy_events(const DemodStateSoA &st, mdemod_lock_event *dst, uint32_t n_streams, uint32_t *sym_out, uint32_t *ev_out, hipStream_t stream)
{
	if (n_streams == 0) return hipSuccess;
	hipLaunchKernelGGL(copy_events_kernel, dim3((n_streams + 255) / 256), dim3(256), 0, stream, st.events, st.ev_this_call, dst, n_streams,
	                   st.sym_this_call, sym_out, ev_out);
	return hipGetLastError();
}

hipError_t
mdemod_launch_selftest_sincos(const float *x, uint32_t n, float *s, float *c, hipStream_t stream)
{
	if (!n) return hipSuccess;
	hipLaunchKernelGGL(selftest_sincos_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x, n, s, c);
	return hipGetLastError();
}

hipError_t
mdemod_launch_selftest_hypot(const float *xy, uint32_t n, float *out, hipStream_t stream)
{
	if (!n) return hipSuccess;
	hipLaunchKernelGGL(selftest_hypot_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, xy, n, out);
	return hipGetLastError();
}
