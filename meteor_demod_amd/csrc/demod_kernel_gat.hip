/*
 * demod_kernel_gat.hip — v3 "gather" geometry: sample rates at which a firing's taps are a minority of the samples that pass
 * (up to 65 taps above 46 samples per firing - float input 54 -, 66..129 taps above 30 - not float -: an Airspy's 6 or 10 MS/s,
 * a HackRF's 8-bit 8..20 MS/s).
 *
 * The other geometries slide a register window over EVERY sample.  Here there is no window: the kernel body (rotwin_body.h: symbol
 * clock, AGC, NCO, loops, output ring - the same code) asks the policy for the filter output of the firing at sample v, and the
 * policy loads that firing's own taps: NS raw samples starting at the 16-byte step below the oldest tap (s16: 17 or 33
 * global_load_dwordx4 per lane, 272 or 528 contiguous bytes), and sums them oldest first with the row of the compact4 coefficient
 * table that is shifted by the samples between that step and the oldest tap (0..3 for s16, 0..7 for u8, 0..1 for float) (filter.c:46-65: two rounded products and two
 * rounded sums per tap; the padding slots multiply finite samples by zero).  A firing whose taps reach into the history of the
 * previous block, or within NS samples of the block's end, takes a tap-by-tap path.
 */
#include "rotwin_body.h"

namespace {

typedef float gpair_t __attribute__((ext_vector_type(2)));
typedef float gquad_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const gquad_t lds_gquad;
typedef __attribute__((address_space(3))) const float lds_gfloat;

/* one 16-byte load = G samples; sample k of it */
template <int FMT> struct GFmt;
template <> struct GFmt<16> {
	enum { G = 4 };
	__device__ static __forceinline__ gpair_t get(const uint4 &q, int k) { const uint32_t w = (&q.x)[k]; return gpair_t{ (float)(int)(int16_t)(w & 0xFFFFu), (float)((int)w >> 16) }; }
	__device__ static __forceinline__ gpair_t one(uint32_t w) { return gpair_t{ (float)(int)(int16_t)(w & 0xFFFFu), (float)((int)w >> 16) }; }
};
template <> struct GFmt<8> {          /* wavfile.c:61: (int)byte - 128 */
	enum { G = 8 };
	__device__ static __forceinline__ gpair_t get(const uint4 &q, int k) { const uint32_t w = (&q.x)[k >> 1] >> (16 * (k & 1)); return gpair_t{ (float)((int)(w & 0xFFu) - 128), (float)((int)((w >> 8) & 0xFFu) - 128) }; }
	__device__ static __forceinline__ gpair_t one(uint16_t w) { return gpair_t{ (float)((int)(w & 0xFFu) - 128), (float)((int)(w >> 8) - 128) }; }
};
template <> struct GFmt<32> {
	enum { G = 2 };
	__device__ static __forceinline__ gpair_t get(const uint4 &q, int k) { return gpair_t{ __uint_as_float((&q.x)[2 * k]), __uint_as_float((&q.x)[2 * k + 1]) }; }
	__device__ static __forceinline__ gpair_t one(float2 w) { return gpair_t{ w.x, w.y }; }
};

/* KT: embedded taps (65 or 129); NS = KT + G - 1 slots loaded per firing (G = samples per 16-byte load: 4 / 8 / 2) */
template <int KT, int FMT>
struct WinG {
	static constexpr int G = GFmt<FMT>::G, kTaps = KT, kBack = KT - 1, AMAX = G - 1, NS = KT + AMAX, NW = 4 * ((NS + 3) / 4), SLIDE = 4, MAXSL = 1, DEPTH = 1,
	                     BLOCK = MDEMOD_RW_BLOCK, ROTN = 1, RING = 32, REGSLOTS = 0;
	static constexpr bool GATHER = true;
	static_assert(NS % G == 0 && NS / G <= 36, "whole 16-byte loads, and few enough of them to stay in registers");
	typedef typename RFmt<FMT>::sample_t sample_t;
	__device__ __forceinline__ void setup(uint32_t) {}
	/* (never called: the body skips the window in GATHER mode) */
	__device__ static __forceinline__ void put_history(const float2 *, bool, int) {}
	__device__ static __forceinline__ void put_init(const RGran<FMT> (&)[1], int) {}
	__device__ static __forceinline__ void put(const RGran<FMT> (&)[1], int) {}
	__device__ static __forceinline__ void fir(uint32_t, int, int, const DemodConsts &, int, float &, float &) {}

	/* filter.c:46-65 for the firing on virtual sample v (history ++ block): taps on block samples [v - kBack - (KT - 1), v - kBack] */
	__device__ static __forceinline__ void
	fir_gather(uint32_t ctab_addr, const sample_t *src, const float2 *hist, int v, int n, int bank, const DemodConsts &C, float &re, float &im)
	{
		const int oldest = v - kBack - (KT - 1);          /* block index of the oldest tap's sample; negative: in the history */
		const int d = oldest & (G - 1);                   /* samples between the 16-byte step below it and the oldest tap */
		const int start = oldest - d;
		/* compact4 table with G - 1 alignments' worth of padding: the lane reads copy ((AMAX - d) & 3) from index ((AMAX - d) & ~3) on */
		const int o = AMAX - d;
		const uint32_t row = ctab_addr + 4u * (uint32_t)(__mul24(bank * 4 + (o & 3), C.ctab_row_stride) + (o & ~3));
		gpair_t acc = { 0.0f, 0.0f };
		if (__builtin_expect(md_all(start >= 0 && start + NS <= n), 1)) {
			uint4 raw[NS / G];
#pragma unroll
			for (int g = 0; g < NS / G; g++) __builtin_memcpy(&raw[g], src + start + G * g, 16);
#pragma unroll
			for (int s4 = 0; s4 < NS / 4; s4++) {                        /* four slots = one 16-byte read of coefficients */
				gquad_t c = { 0.0f, 0.0f, 0.0f, 0.0f };
				c = *(lds_gquad *)(row + 16u * (uint32_t)s4);
				const float cf[4] = { c.x, c.y, c.z, c.w };
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const int slot = 4 * s4 + k;
					const gpair_t smp = GFmt<FMT>::get(raw[slot / G], slot % G);
					const gpair_t p = smp * cf[k];
					acc = acc + p;
				}
			}
			if constexpr (NS % 4 != 0) {                                 /* float input: 66 slots */
#pragma unroll
				for (int slot = NS / 4 * 4; slot < NS; slot++) {
					const float cf = *(lds_gfloat *)(row + 4u * (uint32_t)slot);
					const gpair_t p = GFmt<FMT>::get(raw[slot / G], slot % G) * cf;
					acc = acc + p;
				}
			}
		} else {
			/* the first firings of a block (taps in the previous block's history) and the last ones (within NS samples of its end) */
#pragma unroll 1
			for (int k = 0; k < KT; k++) {
				const int idx = oldest + k;
				gpair_t smp;
				if (idx < 0) { const float2 h = hist[kBack + idx]; smp.x = h.x; smp.y = h.y; }
				else smp = GFmt<FMT>::one(src[idx]);
				const float cf = *(lds_gfloat *)(row + 4u * (uint32_t)(d + k));
				const gpair_t p = smp * cf;
				acc = acc + p;
			}
		}
		re = acc.x; im = acc.y;
	}
};

template <int FMT, int OQPSK, int KT>
__global__ void __launch_bounds__(MDEMOD_RW_BLOCK, 2)
demod_kernel_gat(const DemodLaunch L)
{
	rotwin_demod<WinG<KT, FMT>, FMT, OQPSK, 0>(L);
}

template <int FMT, int OQPSK, int KT>
hipError_t
launch_gat(const DemodLaunch &L, size_t lds_bytes, hipStream_t stream)
{
	const uint32_t blocks = (L.n_streams + MDEMOD_RW_BLOCK - 1) / MDEMOD_RW_BLOCK;
	auto kfn = demod_kernel_gat<FMT, OQPSK, KT>;
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(kfn, dim3(blocks), dim3(MDEMOD_RW_BLOCK), lds_bytes, stream, L);
	return hipGetLastError();
}

} /* namespace */

template <int FMT>
static hipError_t
launch_gat_fmt(const DemodLaunch &L, int long_filter, size_t lds_bytes, hipStream_t stream)
{
	if constexpr (FMT != 32)
		if (long_filter) return L.c.oqpsk ? launch_gat<FMT, 1, 129>(L, lds_bytes, stream) : launch_gat<FMT, 0, 129>(L, lds_bytes, stream);
	if (long_filter) return hipErrorInvalidValue;                    /* (float input with the long filter: 65 loads per firing do not stay in registers) */
	return L.c.oqpsk ? launch_gat<FMT, 1, 65>(L, lds_bytes, stream) : launch_gat<FMT, 0, 65>(L, lds_bytes, stream);
}

hipError_t
mdemod_launch_demod_gat(const DemodLaunch &L, int fmt, int long_filter, size_t lds_bytes, hipStream_t stream)
{
	switch (fmt) {
	case 16: return launch_gat_fmt<16>(L, long_filter, lds_bytes, stream);
	case 8:  return launch_gat_fmt<8>(L, long_filter, lds_bytes, stream);
	case 32: return launch_gat_fmt<32>(L, long_filter, lds_bytes, stream);
	default: return hipErrorInvalidValue;
	}
}
