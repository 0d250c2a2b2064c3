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

/* ---- truth check: the hard decisions of a demodulated recording against the symbols the generator transmitted ------------------
 * synth_symbol(seed, j) is a pure function, so the whole output of a multi-gigasample run can be checked in milliseconds without
 * any serial reference run.  A demodulator's output symbol m carries transmitted symbol m + lag on each rail, up to the PLL's
 * quarter-turn ambiguity (which rail is which, and each rail's sign) - for OQPSK the two rails may sit one symbol apart. */

/* agreements of received rail r (0: I, 1: Q) with transmitted rail t at lag lag_min + blockIdx.y over symbols [m0, m0 + count):
 * out[(r * 2 + t) * n_lags + blockIdx.y] */
__global__ void
truth_probe_kernel(uint64_t seed, const int8_t *soft, uint64_t m0, uint32_t count, int lag_min, int n_lags, unsigned *out)
{
	const int li = blockIdx.y;
	const int64_t lag = lag_min + li;
	unsigned c[4] = { 0, 0, 0, 0 };
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
		const uint64_t m = m0 + k;
		const int64_t j = (int64_t)m + lag;
		if (j < 0) continue;
		double si, sq;
		synth_symbol(seed, (uint64_t)j, &si, &sq);
		const bool ri = soft[2 * m] >= 0, rq = soft[2 * m + 1] >= 0, ti = si > 0, tq = sq > 0;
		c[0] += ri == ti; c[1] += ri == tq; c[2] += rq == ti; c[3] += rq == tq;
	}
	for (int q = 0; q < 4; q++) {
		unsigned v = c[q];
		for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
		if ((threadIdx.x & 63) == 0 && v) atomicAdd(&out[q * n_lags + li], v);
	}
}

/* decisions that differ from the transmitted symbols under one hypothesis, per block of block_syms output symbols:
 * hyp = { tx rail of the received I rail, its lag, 1 = inverted,  the same three for the received Q rail } */
struct TruthHyp { int rail_i, lag_i, inv_i, rail_q, lag_q, inv_q; };
__global__ void
truth_count_kernel(uint64_t seed, const int8_t *soft, uint64_t n_symbols, uint32_t block_syms, TruthHyp h, unsigned *err_i, unsigned *err_q)
{
	const uint64_t b = blockIdx.x, m_lo = b * block_syms;
	const uint64_t m_hi = m_lo + block_syms < n_symbols ? m_lo + block_syms : n_symbols;
	unsigned ei = 0, eq = 0;
	for (uint64_t m = m_lo + threadIdx.x; m < m_hi; m += blockDim.x) {
		double si, sq;
		const int64_t ji = (int64_t)m + h.lag_i, jq = (int64_t)m + h.lag_q;
		if (ji >= 0) {
			synth_symbol(seed, (uint64_t)ji, &si, &sq);
			const bool t = ((h.rail_i ? sq : si) > 0) != (h.inv_i != 0);
			ei += (soft[2 * m] >= 0) != t;
		}
		if (jq >= 0) {
			synth_symbol(seed, (uint64_t)jq, &si, &sq);
			const bool t = ((h.rail_q ? sq : si) > 0) != (h.inv_q != 0);
			eq += (soft[2 * m + 1] >= 0) != t;
		}
	}
	__shared__ unsigned acc[2];
	if (threadIdx.x < 2) acc[threadIdx.x] = 0;
	__syncthreads();
	for (int o = 32; o > 0; o >>= 1) { ei += __shfl_xor(ei, o); eq += __shfl_xor(eq, o); }
	if ((threadIdx.x & 63) == 0) { atomicAdd(&acc[0], ei); atomicAdd(&acc[1], eq); }
	__syncthreads();
	if (threadIdx.x == 0) { err_i[b] = acc[0]; err_q[b] = acc[1]; }
}

} /* namespace */

extern "C" {

/* out_host[(r * 2 + t) * n_lags + l]: symbols of [m0, m0 + count) on which received rail r agrees with transmitted rail t at lag
 * lag_min + l (about count / 2 for a wrong pairing, near count or near 0 - inverted - for the right one).  Synchronous. */
int
mdemod_synth_truth_probe(uint64_t seed, const void *soft_dev, uint64_t m0, uint32_t count, int lag_min, int n_lags, uint32_t *out_host, int device)
{
	if (n_lags < 1 || n_lags > 256 || !count) return -1;
	unsigned *d = nullptr;
	hipError_t e = hipSetDevice(device);
	if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d), sizeof(unsigned) * 4 * n_lags);
	if (e == hipSuccess) e = hipMemset(d, 0, sizeof(unsigned) * 4 * n_lags);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(truth_probe_kernel, dim3(64, n_lags), dim3(256), 0, nullptr, seed, static_cast<const int8_t *>(soft_dev), m0, count, lag_min, n_lags, d);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpy(out_host, d, sizeof(unsigned) * 4 * n_lags, hipMemcpyDeviceToHost);
	if (d) (void)hipFree(d);
	return e == hipSuccess ? 0 : -3;
}

/* err_host[0 .. n_blocks): wrong decisions of the received I rail per block of block_syms symbols, err_host[n_blocks .. 2 n_blocks):
 * of the Q rail, under hyp = { tx rail of rx I, lag, inverted, tx rail of rx Q, lag, inverted }.  Synchronous. */
int
mdemod_synth_truth_count(uint64_t seed, const void *soft_dev, uint64_t n_symbols, uint32_t block_syms, const int32_t hyp[6], uint32_t *err_host, int device)
{
	if (!n_symbols || !block_syms) return -1;
	const uint64_t nb = (n_symbols + block_syms - 1) / block_syms;
	if (nb > 0x7FFFFFFFull) return -1;
	unsigned *d = nullptr;
	hipError_t e = hipSetDevice(device);
	if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d), sizeof(unsigned) * 2 * nb);
	if (e == hipSuccess) {
		const TruthHyp h = { hyp[0], hyp[1], hyp[2], hyp[3], hyp[4], hyp[5] };
		hipLaunchKernelGGL(truth_count_kernel, dim3(static_cast<unsigned>(nb)), dim3(256), 0, nullptr, seed, static_cast<const int8_t *>(soft_dev), n_symbols, block_syms, h, d, d + nb);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpy(err_host, d, sizeof(unsigned) * 2 * nb, hipMemcpyDeviceToHost);
	if (d) (void)hipFree(d);
	return e == hipSuccess ? 0 : -3;
}

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
