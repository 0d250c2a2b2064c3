/*
 * demod_kernel_rotp.hip — v3 rotating PACKED register window: the wide (up to 129 taps, <= 15 samples per firing), mid
 * (<= 65 taps, <= 15) and far (<= 65 taps, <= 46: 47 alignments) geometries for s16 and u8 input; the wide window also takes 66..129 taps at
 * 15..30 samples per firing (two loop iterations per firing).
 *
 * The kernel body is rotwin_body.h (shared with demod_kernel_rot.hip); this file is the window policy: 96..160 slots of RAW
 * samples in registers that belong to the generated assembly of rotpk_asm.h (gen_rotpk_asm.py: ring of per-chunk FIR code,
 * 4 instructions per tap, nothing moves on a slide), compact4 coefficient table (demod_host.cpp).  What it replaces is the
 * packed-window C++ of round 2's v2 kernel (retired): 6 instructions per tap, 144 v_mov_b32 per slide (one slide per firing at 1 MS/s)
 * and the PHI copies around them - 1 502 VALU instructions per wave-firing on configs[3] where the reference's arithmetic
 * needs 516 (profiles/r02_kernels.md).
 */
#include "rotwin_body.h"
#ifdef ROTPK_ASM_HEADER
#include ROTPK_ASM_HEADER        /* experimental builds (tools/build_exp_rotp.sh) */
#else
#include "rotpk_asm.h"
#endif

namespace {

typedef float pair_t __attribute__((ext_vector_type(2)));

template <int GEO> struct GeoP;
#ifndef ROTP_WIDE_DEPTH
#define ROTP_WIDE_DEPTH 1
#endif
#ifndef ROTP_MID_DEPTH
#define ROTP_MID_DEPTH 1
#endif
#ifndef ROTP_FAR_DEPTH
#define ROTP_FAR_DEPTH 2
#endif
template <> struct GeoP<0> { static constexpr int kTaps = 129, NW = MDEMOD_RW_WIDE_NW, MAXSL = 1, DEPTH = ROTP_WIDE_DEPTH, BLOCK = MDEMOD_RW_WIDE_BLOCK; };
template <> struct GeoP<1> { static constexpr int kTaps = 65, NW = MDEMOD_RW_MID_NW, MAXSL = 1, DEPTH = ROTP_MID_DEPTH, BLOCK = MDEMOD_RW_BLOCK; };
#ifndef ROTP_FAR_MAXSL
#define ROTP_FAR_MAXSL 2
#endif
template <> struct GeoP<2> { static constexpr int kTaps = 65, NW = MDEMOD_RW_FAR_NW, MAXSL = ROTP_FAR_MAXSL, DEPTH = ROTP_FAR_DEPTH, BLOCK = MDEMOD_RW_BLOCK; };
static_assert(MDEMOD_RW_WIDE_NW == 160 && MDEMOD_RW_MID_NW == 96 && MDEMOD_RW_FAR_NW == 112, "gen_rotpk_asm.py: GEOS");

/* the assembly of one (geometry, format): the FIR from group `sub` of chunk `entry` of the ring until `cnt` + 1 exit points have
 * gone by, and the slide */
template <int GEO, int FMT> struct AsmP;
#if ROTPK_FORM == 0             /* plain f32 products and sums: the accumulator is two registers */
#define ROTP_FIR_CALL(TEXT, CLOB)                                                                                             \
	float ar = 0.0f, ai = 0.0f;                                                                                               \
	asm volatile(TEXT : [ar] "+v"(ar), [ai] "+v"(ai), [addr] "+v"(addr), [cnt] "+s"(cnt), [tmp] "=&s"(tmp)                    \
	             : [entry] "s"(entry), [sub] "s"(sub) : "vcc", "scc", CLOB);                                                \
	re = ar; im = ai;
#else                           /* packed products and sums: the accumulator is an aligned register pair */
#define ROTP_FIR_CALL(TEXT, CLOB)                                                                                             \
	pair_t acc = { 0.0f, 0.0f };                                                                                              \
	asm volatile(TEXT : [acc] "+v"(acc), [addr] "+v"(addr), [cnt] "+s"(cnt), [tmp] "=&s"(tmp)                                 \
	             : [entry] "s"(entry), [sub] "s"(sub) : "vcc", "scc", CLOB);                                                \
	re = acc.x; im = acc.y;
#endif
#define ROTP_ASM(GEO, NAME, FMT)                                                                                              \
	template <> struct AsmP<GEO, FMT> {                                                                                       \
		__device__ static __forceinline__ void fir(uint32_t addr, int entry, int sub, int cnt, float &re, float &im)          \
		{                                                                                                                     \
			int tmp;                                                                                                          \
			ROTP_FIR_CALL(ROTPK_##NAME##_##FMT##_FIR_ASM, ROTPK_##NAME##_##FMT##_CLOBBERS)                                     \
		}                                                                                                                     \
		__device__ static __forceinline__ void put(const uint32_t (&g)[FMT == 16 ? 16 : 8], int q);                           \
	};
ROTP_ASM(0, WIDE, 16) ROTP_ASM(0, WIDE, 8) ROTP_ASM(1, MID, 16) ROTP_ASM(1, MID, 8) ROTP_ASM(2, FAR, 16) ROTP_ASM(2, FAR, 8)

/* "memory": the refill loads that follow a slide may not be hoisted above it (see demod_kernel_rot.hip) */
#define ROTP_PUT16(GEO, NAME)                                                                                                 \
	__device__ __forceinline__ void AsmP<GEO, 16>::put(const uint32_t (&g)[16], int q)                                        \
	{                                                                                                                         \
		int tmp;                                                                                                              \
		asm volatile(ROTPK_##NAME##_16_PUT_ASM                                                                                \
		             : [tmp] "=&s"(tmp)                                                                                       \
		             : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]), [g6] "v"(g[6]),  \
		               [g7] "v"(g[7]), [g8] "v"(g[8]), [g9] "v"(g[9]), [g10] "v"(g[10]), [g11] "v"(g[11]), [g12] "v"(g[12]),  \
		               [g13] "v"(g[13]), [g14] "v"(g[14]), [g15] "v"(g[15]), [rot] "s"(q)                                     \
		             : "vcc", "scc", "v255", "memory");                                                                       \
	}
#define ROTP_PUT8(GEO, NAME)                                                                                                  \
	__device__ __forceinline__ void AsmP<GEO, 8>::put(const uint32_t (&g)[8], int q)                                          \
	{                                                                                                                         \
		int tmp;                                                                                                              \
		asm volatile(ROTPK_##NAME##_8_PUT_ASM                                                                                 \
		             : [tmp] "=&s"(tmp)                                                                                       \
		             : [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]), [g6] "v"(g[6]),  \
		               [g7] "v"(g[7]), [rot] "s"(q)                                                                           \
		             : "vcc", "scc", "v255", "memory");                                                                       \
	}
ROTP_PUT16(0, WIDE) ROTP_PUT16(1, MID) ROTP_PUT16(2, FAR) ROTP_PUT8(0, WIDE) ROTP_PUT8(1, MID) ROTP_PUT8(2, FAR)

/* window policy: NW raw samples, slides of 16 slots, AMAX + 1 alignments, compact4 coefficient table */
template <int GEO, int FMT>
struct WinP {
	static constexpr int kTaps = GeoP<GEO>::kTaps, kBack = kTaps - 1, NW = GeoP<GEO>::NW, SLIDE = 16, AMAX = NW - kTaps,
	                     MAXSL = GeoP<GEO>::MAXSL, DEPTH = GeoP<GEO>::DEPTH, BLOCK = GeoP<GEO>::BLOCK, NCH = NW / SLIDE, GD = FMT == 16 ? 16 : 8, REGSLOTS = 0, ROTN = NCH, RING = 32;
	static constexpr bool GATHER = false;
	__device__ __forceinline__ void setup(uint32_t) {}
	static_assert(kBack % SLIDE == 0 && NW % SLIDE == 0, "history and window are whole chunks");

	/* wavfile.c:58-69 backwards: the raw sample a history value (an exactly converted input sample) came from */
	__device__ static __forceinline__ void put_history(const float2 *hist, bool valid, int q)
	{
		uint32_t g[GD];
#pragma unroll
		for (int k = 0; k < 16; k++) {
			float2 h = hist[k];
			if (!valid) h = make_float2(0.0f, 0.0f);
			if (FMT == 16) {
				g[k] = ((uint32_t)(int)h.x & 0xFFFFu) | ((uint32_t)(int)h.y << 16);
			} else {
				const uint32_t s = (uint32_t)(((int)h.x + 128) & 0xFF) | ((uint32_t)(((int)h.y + 128) & 0xFF) << 8);
				if (k & 1) g[k / 2] |= s << 16; else g[k / 2] = s;
			}
		}
		AsmP<GEO, FMT>::put(g, q);
	}
	__device__ static __forceinline__ void put(const RGran<FMT> (&gr)[4], int q)
	{
		uint32_t g[GD];
#pragma unroll
		for (int i = 0; i < 4; i++)
#pragma unroll
			for (int j = 0; j < GD / 4; j++) g[i * (GD / 4) + j] = gr[i].w[j];
		AsmP<GEO, FMT>::put(g, q);
	}
	__device__ static __forceinline__ void put_init(const RGran<FMT> (&gr)[4], int q) { put(gr, q); }
	__device__ static __forceinline__ void fir(uint32_t ctab_addr, int a, int bank, const DemodConsts &C, int rot, float &re, float &im)
	{
		/* compact4 table: per bank the taps padded with AMAX zeros either side, four copies shifted by 0..3 floats: the lane at
		 * alignment a reads P[(AMAX - a) + s] for slot s = copy ((AMAX - a) & 3) at the 16-byte aligned index ((AMAX - a) & ~3) + s */
		const int o = AMAX - a;
		uint32_t addr = rot_mad24((uint32_t)(bank * 4 + (o & 3)), 4u * (uint32_t)C.ctab_row_stride, ctab_addr + 4u * (uint32_t)(o & ~3));
		/* groups of four slots that are padding for every lane of the wave are not run: the first tap of a lane is in group a / 4,
		 * its last in group (a + kTaps - 1) / 4; the wave's extremes by bisection on votes (one comparison each) */
		constexpr int QB = AMAX / 4 >= 8 ? 4 : 3;                 /* bits of a / 4 */
		static_assert(AMAX / 4 < (1 << QB) && (kTaps - 1) % 4 == 0, "alignments in groups of four");
		int q_lo = 0, q_hi = (1 << QB) - 1;
#pragma unroll
		for (int b = QB - 1; b >= 0; b--) {
			const int t = q_lo + (1 << b);
			q_lo = md_all(a >= 4 * t) ? t : q_lo;
			const int u = q_hi - (1 << b);
			q_hi = md_all(a < 4 * (u + 1)) ? u : q_hi;
		}
		q_lo = __builtin_amdgcn_readfirstlane(q_lo); q_hi = __builtin_amdgcn_readfirstlane(q_hi);
		const int c_lo = q_lo >> 2;                               /* the chunk of the first group */
		int chunk = rot + c_lo;
		chunk = chunk >= NCH ? chunk - NCH : chunk;
		const int sub = q_lo & 3;                                 /* the first group's place in its chunk */
		/* exit points of the ring (ROTPK_EXIT slots apart) to pass before leaving: the last tap's group is q_hi + (kTaps - 1) / 4 */
		constexpr int XS = ROTPK_EXIT / 4;                        /* groups per exit point */
		const int cnt = (q_hi + (kTaps - 1) / 4) / XS - q_lo / XS;
		AsmP<GEO, FMT>::fir(addr + 64u * (uint32_t)c_lo, __builtin_amdgcn_readfirstlane(chunk), __builtin_amdgcn_readfirstlane(sub), __builtin_amdgcn_readfirstlane(cnt), re, im);
	}
};

/* amdgpu_num_vgpr wants a literal: one kernel per (geometry, format, modulation), all of them the shared body */
#define ROTP_KERNEL(GEO, NAME, FMT, OQ)                                                                                       \
	__global__ void __launch_bounds__(GeoP<GEO>::BLOCK, 2) __attribute__((amdgpu_num_vgpr(ROTPK_##NAME##_##FMT##_LIMIT / 2)))  \
	demod_kernel_rotp_##NAME##_##FMT##_##OQ(const DemodLaunch L) { rotwin_demod<WinP<GEO, FMT>, FMT, OQ, 0>(L); }
/* configs[3] (72k QPSK in 1 MS/s, -O 8): its 109 blind symbol-clock steps compiled in, like the two LRPT settings of the std kernel */
__global__ void __launch_bounds__(GeoP<0>::BLOCK, 2) __attribute__((amdgpu_num_vgpr(ROTPK_WIDE_16_LIMIT / 2)))
demod_kernel_rotp_WIDE_16_0_ks109(const DemodLaunch L) { rotwin_demod<WinP<0, 16>, 16, 0, 109, 1>(L); }        /* ... and the sine table in LDS */
ROTP_KERNEL(0, WIDE, 16, 0) ROTP_KERNEL(0, WIDE, 16, 1) ROTP_KERNEL(0, WIDE, 8, 0) ROTP_KERNEL(0, WIDE, 8, 1)
ROTP_KERNEL(1, MID, 16, 0) ROTP_KERNEL(1, MID, 16, 1) ROTP_KERNEL(1, MID, 8, 0) ROTP_KERNEL(1, MID, 8, 1)
ROTP_KERNEL(2, FAR, 16, 0) ROTP_KERNEL(2, FAR, 16, 1) ROTP_KERNEL(2, FAR, 8, 0) ROTP_KERNEL(2, FAR, 8, 1)

typedef void (*rotp_kernel_t)(const DemodLaunch);

hipError_t
launch_rotp(rotp_kernel_t kfn, int block, const DemodLaunch &L, size_t lds_bytes, hipStream_t stream)
{
	const uint32_t blocks = (L.n_streams + block - 1) / block;
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(kfn, dim3(blocks), dim3(block), lds_bytes, stream, L);
	return hipGetLastError();
}

} /* namespace */

hipError_t
mdemod_launch_demod_rotp(const DemodLaunch &L, int fmt, int geom /* 0 wide, 1 mid, 2 far */, size_t lds_bytes, hipStream_t stream)
{
	if ((fmt != 16 && fmt != 8) || geom < 0 || geom > 2) return hipErrorInvalidValue;
	static const rotp_kernel_t table[3][2][2] = {
		{ { demod_kernel_rotp_WIDE_16_0, demod_kernel_rotp_WIDE_16_1 }, { demod_kernel_rotp_WIDE_8_0, demod_kernel_rotp_WIDE_8_1 } },
		{ { demod_kernel_rotp_MID_16_0, demod_kernel_rotp_MID_16_1 }, { demod_kernel_rotp_MID_8_0, demod_kernel_rotp_MID_8_1 } },
		{ { demod_kernel_rotp_FAR_16_0, demod_kernel_rotp_FAR_16_1 }, { demod_kernel_rotp_FAR_8_0, demod_kernel_rotp_FAR_8_1 } },
	};
	static const int blocks[3] = { GeoP<0>::BLOCK, GeoP<1>::BLOCK, GeoP<2>::BLOCK };
	if (geom == 0 && fmt == 16 && !L.c.oqpsk && L.c.step_safe == 109 && L.c.sin_lut) return launch_rotp(demod_kernel_rotp_WIDE_16_0_ks109, blocks[0], L, lds_bytes, stream);
	return launch_rotp(table[geom][fmt == 16 ? 0 : 1][L.c.oqpsk ? 1 : 0], blocks[geom], L, lds_bytes, stream);
}
