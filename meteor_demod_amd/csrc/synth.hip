/*
 * synth.hip — host and gfx950 front ends of the deterministic synthetic IQ
 * generator in synth_core.h.  Built into libmdemod_synth.so; used by tests,
 * bench.py and __graft_entry__.smoke() to create inputs (the reference ships no
 * signal source).  Host and device versions are bit-identical by construction.
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "synth_core.h"

#pragma clang fp contract(off)

namespace {

/* Unit-energy root-raised-cosine pulse (symbol period 1). */
double
rrc_pulse(double t, double alpha)
{
	const double pi = 3.14159265358979323846;
	if (fabs(t) < 1e-12) return 1.0 - alpha + 4.0 * alpha / pi;
	const double x = 4.0 * alpha * t;
	if (fabs(fabs(x) - 1.0) < 1e-9)
		return alpha / sqrt(2.0) * ((1.0 + 2.0 / pi) * sin(pi / (4.0 * alpha)) + (1.0 - 2.0 / pi) * cos(pi / (4.0 * alpha)));
	return (sin(pi * t * (1.0 - alpha)) + x * cos(pi * t * (1.0 + alpha))) / (pi * t * (1.0 - x * x));
}

__global__ void
synth_kernel(const synth_tables *tb, const synth_stream *streams, uint32_t n_streams,
             uint64_t n0, uint64_t count, unsigned char *dst, uint64_t stride_samples)
{
	const uint64_t total = (uint64_t)n_streams * count;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
	     i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t s = (uint32_t)(i / count);
		const uint64_t k = i - (uint64_t)s * count;
		const synth_stream st = streams[s];
		const size_t sb = 2 * (size_t)st.fmt / 8;
		synth_store(tb, &st, n0 + k, dst + ((uint64_t)s * stride_samples + k) * sb);
	}
}

} /* namespace */

extern "C" {

size_t mdemod_synth_tables_size(void) { return sizeof(synth_tables); }
size_t mdemod_synth_stream_size(void) { return sizeof(synth_stream); }

void
mdemod_synth_tables_init(synth_tables *tb, double alpha)
{
	const double two_pi = 6.283185307179586476925286766559;
	for (int i = 0; i < SYNTH_PULSE_LEN; i++)
		tb->pulse[i] = rrc_pulse(-(double)SYNTH_SPAN + (double)i / SYNTH_OS, alpha);
	for (int i = 0; i < SYNTH_TRIG_LEN; i++) {
		tb->cos_hi[i] = cos(two_pi * (double)i / 1024.0);
		tb->sin_hi[i] = sin(two_pi * (double)i / 1024.0);
		tb->cos_lo[i] = cos(two_pi * (double)i / 1048576.0);
		tb->sin_lo[i] = sin(two_pi * (double)i / 1048576.0);
	}
}

/* Host: samples [n0, n0+count) of one stream into dst (packed, stream format). */
void
mdemod_synth_host(const synth_tables *tb, const synth_stream *st, uint64_t n0, uint64_t count, void *dst)
{
	const size_t sb = 2 * (size_t)st->fmt / 8;
	for (uint64_t k = 0; k < count; k++)
		synth_store(tb, st, n0 + k, (unsigned char *)dst + k * sb);
}

/* Device: stream s writes samples [n0, n0+count) at dst_dev + s*stride_samples.
 * tables/streams are host pointers (uploaded here).  Synchronous. */
int
mdemod_synth_device(const synth_tables *tb, const synth_stream *streams, uint32_t n_streams,
                    uint64_t n0, uint64_t count, void *dst_dev, uint64_t stride_samples, int device)
{
	if (!n_streams || !count) return 0;
	synth_tables *d_tb = nullptr;
	synth_stream *d_st = nullptr;
	hipError_t e = hipSetDevice(device);
	if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_tb), sizeof(synth_tables));
	if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_st), sizeof(synth_stream) * n_streams);
	if (e == hipSuccess) e = hipMemcpy(d_tb, tb, sizeof(synth_tables), hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(d_st, streams, sizeof(synth_stream) * n_streams, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(synth_kernel, dim3(256 * 16), dim3(256), 0, nullptr, d_tb, d_st, n_streams, n0, count,
		                   static_cast<unsigned char *>(dst_dev), stride_samples);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipDeviceSynchronize();
	if (d_tb) (void)hipFree(d_tb);
	if (d_st) (void)hipFree(d_st);
	if (e != hipSuccess) fprintf(stderr, "mdemod_synth_device: %s\n", hipGetErrorString(e));
	return e == hipSuccess ? 0 : -3;
}

} /* extern "C" */
