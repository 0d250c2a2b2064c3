/*
 * demod_kernel_gat.hip — v3 "gather" geometry: sample rates at which a firing's taps are a minority of the samples that pass
 * (s16 input; up to 65 taps above 46 samples per firing, 66..129 taps above 30: an Airspy's 6 or 10 MS/s).
 *
 * The other geometries slide a register window over EVERY sample.  Here there is no window: the kernel body (rotwin_body.h: symbol
 * clock, AGC, NCO, loops, output ring - the same code) asks the policy for the filter output of the firing at sample v, and the
 * policy loads that firing's own taps: NS raw samples starting at the 16-byte step below the oldest tap (17 or 33
 * global_load_dwordx4 per lane, 272 or 528 contiguous bytes), and sums them oldest first with the row of the compact4 coefficient
 * table that is shifted by the 0..3 samples between that step and the oldest tap (filter.c:46-65: two rounded products and two
 * rounded sums per tap; the padding slots multiply finite samples by zero).  A firing whose taps reach into the history of the
 * previous block, or within NS samples of the block's end, takes a tap-by-tap path.
 */
#include "rotwin_body.h"

namespace {

typedef float gpair_t __attribute__((ext_vector_type(2)));
typedef float gquad_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const gquad_t lds_gquad;
typedef __attribute__((address_space(3))) const float lds_gfloat;

/* KT: embedded taps (65 or 129); NS = KT + 3 slots loaded per firing */
template <int KT>
struct WinG {
	static constexpr int kTaps = KT, kBack = KT - 1, NS = KT + 3, AMAX = 3, NW = 4 * ((NS + 3) / 4), SLIDE = 4, MAXSL = 1, BLOCK = MDEMOD_RW_BLOCK,
	                     ROTN = 1, RING = 32, REGSLOTS = 0;
	static constexpr bool GATHER = true;
	static_assert(NS % 4 == 0, "whole 16-byte loads");
	__device__ __forceinline__ void setup(uint32_t) {}
	/* (never called: the body skips the window in GATHER mode) */
	__device__ static __forceinline__ void put_history(const float2 *, bool, int) {}
	__device__ static __forceinline__ void put_init(const RGran<16> (&)[1], int) {}
	__device__ static __forceinline__ void put(const RGran<16> (&)[1], int) {}
	__device__ static __forceinline__ void fir(uint32_t, int, int, const DemodConsts &, int, float &, float &) {}

	/* filter.c:46-65 for the firing on virtual sample v (history ++ block): taps on block samples [v - kBack - (KT - 1), v - kBack] */
	__device__ static __forceinline__ void
	fir_gather(uint32_t ctab_addr, const uint32_t *src, const float2 *hist, int v, int n, int bank, const DemodConsts &C, float &re, float &im)
	{
		const int oldest = v - kBack - (KT - 1);          /* block index of the oldest tap's sample; negative: in the history */
		const int d = oldest & 3;                         /* samples between the 16-byte step below it and the oldest tap */
		const int start = oldest - d;
		/* compact4 table with three alignments' worth of padding: copy (3 - d) of the bank holds the taps d slots in */
		const uint32_t row = ctab_addr + 4u * (uint32_t)__mul24(bank * 4 + (AMAX - d), C.ctab_row_stride);
		gpair_t acc = { 0.0f, 0.0f };
		if (__builtin_expect(__all(start >= 0 && start + NS <= n), 1)) {
			uint4 raw[NS / 4];
#pragma unroll
			for (int g = 0; g < NS / 4; g++) __builtin_memcpy(&raw[g], src + start + 4 * g, 16);
#pragma unroll
			for (int g = 0; g < NS / 4; g++) {
				const gquad_t c = *(lds_gquad *)(row + 16u * (uint32_t)g);
				const uint32_t w[4] = { raw[g].x, raw[g].y, raw[g].z, raw[g].w };
				const float cf[4] = { c.x, c.y, c.z, c.w };
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const gpair_t s = { (float)(int)(int16_t)(w[k] & 0xFFFFu), (float)((int)w[k] >> 16) };
					const gpair_t p = s * cf[k];
					acc = acc + p;
				}
			}
		} else {
			/* the first firings of a block (taps in the previous block's history) and the last ones (within NS samples of its end) */
#pragma unroll 1
			for (int k = 0; k < KT; k++) {
				const int idx = oldest + k;
				gpair_t s;
				if (idx < 0) { const float2 h = hist[kBack + idx]; s.x = h.x; s.y = h.y; }
				else { const uint32_t w = src[idx]; s.x = (float)(int)(int16_t)(w & 0xFFFFu); s.y = (float)((int)w >> 16); }
				const float cf = *(lds_gfloat *)(row + 4u * (uint32_t)(d + k));
				const gpair_t p = s * cf;
				acc = acc + p;
			}
		}
		re = acc.x; im = acc.y;
	}
};

template <int OQPSK, int KT>
__global__ void __launch_bounds__(MDEMOD_RW_BLOCK, 2)
demod_kernel_gat(const DemodLaunch L)
{
	rotwin_demod<WinG<KT>, 16, OQPSK, 0>(L);
}

template <int OQPSK, int KT>
hipError_t
launch_gat(const DemodLaunch &L, size_t lds_bytes, hipStream_t stream)
{
	const uint32_t blocks = (L.n_streams + MDEMOD_RW_BLOCK - 1) / MDEMOD_RW_BLOCK;
	auto kfn = demod_kernel_gat<OQPSK, KT>;
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(kfn, dim3(blocks), dim3(MDEMOD_RW_BLOCK), lds_bytes, stream, L);
	return hipGetLastError();
}

} /* namespace */

hipError_t
mdemod_launch_demod_gat(const DemodLaunch &L, int long_filter, size_t lds_bytes, hipStream_t stream)
{
	if (long_filter) return L.c.oqpsk ? launch_gat<1, 129>(L, lds_bytes, stream) : launch_gat<0, 129>(L, lds_bytes, stream);
	return L.c.oqpsk ? launch_gat<1, 65>(L, lds_bytes, stream) : launch_gat<0, 65>(L, lds_bytes, stream);
}
