// Micro-benchmark (round 4): what one wave-instruction of each kind the v3 demodulator kernels issue costs a SIMD of gfx950, at the
// occupancy those kernels run at (2 waves per SIMD; 1 and 3 for comparison).  One workgroup of 256*w threads per CU (96 KB of LDS
// keeps a second one out), every wave runs `iters` x REP x 8 instructions of ONE kind (independent chains unless the name says
// "chain"), timed with events.  Output: nanoseconds of SIMD time per wave-instruction and the same in cycles of a 2.4 GHz clock,
// as JSON lines on stdout (tools/valu_peak.py folds them into the measured peak of a kernel's instruction mix: bench.py
// roofline.valu.peak_measured_top_s).
//
//     hipcc -O2 --offload-arch=gfx950 -o tools/ubench/valu_mix tools/ubench/valu_mix.hip && tools/ubench/valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#pragma clang fp contract(off)

#define REP 32
typedef float f2 __attribute__((ext_vector_type(2)));

enum Op { MUL_F32, ADD_F32_CHAIN, PK_MUL_BCAST, PK_ADD, FIR_TAPS, FIR_TAPS_PLAIN, CVT_SDWA, MUL_F64, FMA_F64, ADD_F64, CVT_F64_F32, CVT_F32_F64, CVT_I32_F64, RSQ_F64,
          CNDMASK, CMP_SGPR, MUL_LO_U32, MUL_I24, MUL_HI_U32, MED3, CVT_I32_F32, LSHL_ADD, PERM, LDS_GATHER_B32, LDS_B128, ACC_READ, FMA_F64_CHAIN, MUL_F32_CHAIN,
          CNDMASK_E64, CMP_CNDMASK, AND_B32, ADD_U32, SUB_U32, LSHLREV, MOV_B32, XOR_B32, MAX_F32, SUB_F32, FMA_F32, BFE_U32, AND_OR, SAD_U32, ADD3_U32, MAD_U24, CVT_F32_I32, RCP_F32, FLOOR_F32,
          ADD_F32_LIT, MUL_F32_SGPR, LDS_ROW_BCAST, N_OPS };
static const char *op_name[N_OPS] = { "v_mul_f32", "v_add_f32 chain", "v_pk_mul_f32 op_sel bcast", "v_pk_add_f32", "fir taps pk (mul+add, one acc chain)", "fir taps plain (2 mul + 2 add, two acc chains)",
          "v_cvt_f32_i32_sdwa", "v_mul_f64", "v_fma_f64", "v_add_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_cvt_i32_f64", "v_rsq_f64",
          "v_cndmask_b32", "v_cmp_lt_f32 -> sgpr", "v_mul_lo_u32", "v_mul_i32_i24", "v_mul_hi_u32", "v_med3_f32", "v_cvt_i32_f32", "v_lshl_add_u32", "v_perm_b32",
          "ds_read_b32 gather", "ds_read_b128 row", "v_accvgpr_read_b32", "v_fma_f64 chain", "v_mul_f32 chain",
          "v_cndmask_b32_e64 sgpr mask", "v_cmp_lt_f32 vcc + v_cndmask_b32 (pairs)", "v_and_b32", "v_add_u32", "v_sub_u32", "v_lshlrev_b32", "v_mov_b32", "v_xor_b32", "v_max_f32", "v_sub_f32", "v_fma_f32",
          "v_bfe_u32", "v_and_or_b32", "v_sad_u32", "v_add3_u32", "v_mad_u32_u24", "v_cvt_f32_i32", "v_rcp_f32", "v_floor_f32", "v_add_f32 literal", "v_mul_f32 sgpr", "ds_read_b128 16 distinct rows (as the FIR)" };

#define I8(s) s s s s s s s s
template <int OP>
__global__ void __launch_bounds__(1024) k(float *out, int iters, float seed)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 }, w0 = { 1.5f, 2.5f }, w1 = { 3.5f, 0.5f };
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, db = 1.0000000001;
    float b = 1.0000001f;
    int i0 = threadIdx.x * 65537 + 12345, i1 = i0 * 3, i2 = i0 ^ 0x5555, i3 = i1 + 77;
    uint32_t addr = (uint32_t)((threadIdx.x * 2654435761u) >> 18) * 4u;       /* pseudo-random word of the first 64 KB */
    int i4 = i0 | 1, i5 = i1 | 3; const int sconst = 16384;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if (OP == MUL_F32) asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            else if (OP == MUL_F32_CHAIN) asm volatile(I8("v_mul_f32 %0, %0, %1\n") : "+v"(a0) : "v"(b));
            else if (OP == ADD_F32_CHAIN) asm volatile(I8("v_add_f32 %0, %0, %1\n") : "+v"(a0) : "v"(b));
            else if (OP == PK_MUL_BCAST) asm volatile("v_pk_mul_f32 %0, %4, %5 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %4, %5 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %2, %4, %6 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %4, %6 op_sel:[0,1] op_sel_hi:[1,1]\n"
                                                      "v_pk_mul_f32 %0, %4, %5 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %4, %5 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %2, %4, %6 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %4, %6 op_sel:[0,1] op_sel_hi:[1,1]\n"
                                                      : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(w0), "v"(w1), "v"(w0));
            else if (OP == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                                                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(w0));
            else if (OP == FIR_TAPS) asm volatile("v_pk_mul_f32 %1, %3, %4 op_sel_hi:[1,0]\n v_pk_add_f32 %0, %0, %1\n v_pk_mul_f32 %2, %3, %4 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_add_f32 %0, %0, %2\n"
                                                  "v_pk_mul_f32 %1, %3, %5 op_sel_hi:[1,0]\n v_pk_add_f32 %0, %0, %1\n v_pk_mul_f32 %2, %3, %5 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_add_f32 %0, %0, %2\n"
                                                  : "+v"(p0), "=&v"(p1), "=&v"(p2) : "v"(p3), "v"(w0), "v"(w1));
            else if (OP == FIR_TAPS_PLAIN) asm volatile("v_mul_f32 %2, %4, %6\n v_mul_f32 %3, %5, %6\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n v_mul_f32 %2, %4, %7\n v_mul_f32 %3, %5, %7\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n"
                                                  : "+v"(a0), "+v"(a1), "=&v"(a2), "=&v"(a3) : "v"(a4), "v"(a5), "v"(b), "v"(a6));
            else if (OP == CVT_SDWA) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %1, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                                                  "v_cvt_f32_i32_sdwa %2, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %3, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                                                  "v_cvt_f32_i32_sdwa %4, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %5, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                                                  "v_cvt_f32_i32_sdwa %6, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %7, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                                                  : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(i0), "v"(i1));
            else if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                                                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));
            else if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3\n v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3\n"
                                                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));
            else if (OP == FMA_F64_CHAIN) asm volatile(I8("v_fma_f64 %0, %0, %1, %0\n") : "+v"(d0) : "v"(db));
            else if (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                                                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));
            else if (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n v_cvt_f64_f32 %0, %5\n v_cvt_f64_f32 %1, %6\n v_cvt_f64_f32 %2, %7\n v_cvt_f64_f32 %3, %4\n"
                                                     : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            else if (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n v_cvt_f32_f64 %0, %5\n v_cvt_f32_f64 %1, %6\n v_cvt_f32_f64 %2, %7\n v_cvt_f32_f64 %3, %4\n"
                                                     : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
            else if (OP == CVT_I32_F64) asm volatile("v_cvt_i32_f64 %0, %4\n v_cvt_i32_f64 %1, %5\n v_cvt_i32_f64 %2, %6\n v_cvt_i32_f64 %3, %7\n v_cvt_i32_f64 %0, %5\n v_cvt_i32_f64 %1, %6\n v_cvt_i32_f64 %2, %7\n v_cvt_i32_f64 %3, %4\n"
                                                     : "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
            else if (OP == RSQ_F64) asm volatile("v_rsq_f64 %0, %4\n v_rsq_f64 %1, %5\n v_rsq_f64 %2, %6\n v_rsq_f64 %3, %7\n v_rsq_f64 %0, %5\n v_rsq_f64 %1, %6\n v_rsq_f64 %2, %7\n v_rsq_f64 %3, %4\n"
                                                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(db), "v"(db), "v"(db), "v"(db));
            else if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");
            else if (OP == CMP_SGPR) { uint64_t s0, s1;
                asm volatile("v_cmp_lt_f32 %0, %2, %3\n v_cmp_lt_f32 %1, %3, %4\n v_cmp_lt_f32 %0, %4, %5\n v_cmp_lt_f32 %1, %5, %2\n v_cmp_lt_f32 %0, %2, %3\n v_cmp_lt_f32 %1, %3, %4\n v_cmp_lt_f32 %0, %4, %5\n v_cmp_lt_f32 %1, %5, %2\n"
                             : "=&s"(s0), "=&s"(s1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3)); }
            else if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                                                    : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0 | 1));
            else if (OP == MUL_I24) asm volatile("v_mul_i32_i24 %0, %0, %4\n v_mul_i32_i24 %1, %1, %4\n v_mul_i32_i24 %2, %2, %4\n v_mul_i32_i24 %3, %3, %4\n v_mul_i32_i24 %0, %0, %4\n v_mul_i32_i24 %1, %1, %4\n v_mul_i32_i24 %2, %2, %4\n v_mul_i32_i24 %3, %3, %4\n"
                                                 : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0 | 1));
            else if (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n"
                                                    : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0 | 1));
            else if (OP == MED3) asm volatile("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5\n v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5\n"
                                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(a7));
            else if (OP == CVT_I32_F32) asm volatile("v_cvt_i32_f32 %0, %4\n v_cvt_i32_f32 %1, %5\n v_cvt_i32_f32 %2, %6\n v_cvt_i32_f32 %3, %7\n v_cvt_i32_f32 %0, %5\n v_cvt_i32_f32 %1, %6\n v_cvt_i32_f32 %2, %7\n v_cvt_i32_f32 %3, %4\n"
                                                     : "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            else if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_lshl_add_u32 %2, %2, 2, %4\n v_lshl_add_u32 %3, %3, 2, %4\n v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_lshl_add_u32 %2, %2, 2, %4\n v_lshl_add_u32 %3, %3, 2, %4\n"
                                                  : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0));
            else if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5\n v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5\n"
                                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i0), "v"(0x05040100));
            else if (OP == LDS_GATHER_B32) asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:4\n ds_read_b32 %2, %4 offset:8\n ds_read_b32 %3, %4 offset:12\n ds_read_b32 %0, %4 offset:16\n ds_read_b32 %1, %4 offset:20\n ds_read_b32 %2, %4 offset:24\n ds_read_b32 %3, %4 offset:28\n s_waitcnt lgkmcnt(0)\n"
                                                        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(addr));
            else if (OP == LDS_B128) { typedef float f4 __attribute__((ext_vector_type(4))); f4 q0, q1;
                asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n ds_read_b128 %0, %2 offset:32\n ds_read_b128 %1, %2 offset:48\n ds_read_b128 %0, %2 offset:64\n ds_read_b128 %1, %2 offset:80\n ds_read_b128 %0, %2 offset:96\n ds_read_b128 %1, %2 offset:112\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(q0), "=&v"(q1) : "v"((addr & 0x7ff0u) + (threadIdx.x & 15) * 336)); }
#define ONE4(name, fmt, regs) else if (OP == name) asm volatile(fmt fmt : "+v"(regs##0), "+v"(regs##1), "+v"(regs##2), "+v"(regs##3) : "v"(regs##4), "v"(regs##5), "s"(sconst))
            ONE4(AND_B32, "v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4\n", i);
            ONE4(ADD_U32, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n", i);
            ONE4(SUB_U32, "v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4\n", i);
            ONE4(LSHLREV, "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n", i);
            ONE4(MOV_B32, "v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %4\n v_mov_b32 %3, %5\n", i);
            ONE4(XOR_B32, "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n", i);
            ONE4(MAX_F32, "v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4\n", a);
            ONE4(SUB_F32, "v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_sub_f32 %2, %2, %4\n v_sub_f32 %3, %3, %4\n", a);
            ONE4(FMA_F32, "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n", a);
            ONE4(BFE_U32, "v_bfe_u32 %0, %0, 1, 15\n v_bfe_u32 %1, %1, 1, 15\n v_bfe_u32 %2, %2, 1, 15\n v_bfe_u32 %3, %3, 1, 15\n", i);
            ONE4(AND_OR, "v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5\n", i);
            ONE4(SAD_U32, "v_sad_u32 %0, %0, %6, 0\n v_sad_u32 %1, %1, %6, 0\n v_sad_u32 %2, %2, %6, 0\n v_sad_u32 %3, %3, %6, 0\n", i);
            ONE4(ADD3_U32, "v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5\n", i);
            ONE4(MAD_U24, "v_mad_u32_u24 %0, %0, %6, %5\n v_mad_u32_u24 %1, %1, %6, %5\n v_mad_u32_u24 %2, %2, %6, %5\n v_mad_u32_u24 %3, %3, %6, %5\n", i);
            ONE4(CVT_F32_I32, "v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n", i);
            ONE4(RCP_F32, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n", a);
            ONE4(FLOOR_F32, "v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n", a);
            ONE4(ADD_F32_LIT, "v_add_f32 %0, 0x40c90fdb, %0\n v_add_f32 %1, 0x40c90fdb, %1\n v_add_f32 %2, 0x40c90fdb, %2\n v_add_f32 %3, 0x40c90fdb, %3\n", a);
            ONE4(MUL_F32_SGPR, "v_mul_f32 %0, %6, %0\n v_mul_f32 %1, %6, %1\n v_mul_f32 %2, %6, %2\n v_mul_f32 %3, %6, %3\n", a);
            else if (OP == CNDMASK_E64) { uint64_t m = 0x5555aaaa3333ccccull;
                asm volatile("v_cndmask_b32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, %5\n v_cndmask_b32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %4, %5\n v_cndmask_b32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, %5\n v_cndmask_b32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %4, %5\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "s"(m)); }
            else if (OP == CMP_CNDMASK) asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %4, vcc\n"
                                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");
            else if (OP == LDS_ROW_BCAST) { typedef float f4 __attribute__((ext_vector_type(4))); f4 q0, q1;
                asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n ds_read_b128 %0, %2 offset:32\n ds_read_b128 %1, %2 offset:48\n ds_read_b128 %0, %2 offset:64\n ds_read_b128 %1, %2 offset:80\n ds_read_b128 %0, %2 offset:96\n ds_read_b128 %1, %2 offset:112\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(q0), "=&v"(q1) : "v"(((threadIdx.x * 7u) & 15u) * 336u + ((threadIdx.x >> 4) & 3u) * 5376u)); }
            else if (OP == ACC_READ) asm volatile("v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3\n v_accvgpr_read_b32 %0, a4\n v_accvgpr_read_b32 %1, a5\n v_accvgpr_read_b32 %2, a6\n v_accvgpr_read_b32 %3, a7\n"
                                                  : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) :: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + (float)(d0 + d1 + d2 + d3) + (float)(i0 + i1 + i2 + i3);
}

template <int OP>
static void run(int w, int cus, float *out)
{
    const int threads = 256 * w, blocks = cus, iters = 400;
    const size_t lds = 96 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), lds, 0, out, 4, 1.0f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int t = 0; t < 3; t++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), lds, 0, out, iters, 1.0f); hipEventRecord(e1);
        hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    /* fixed cost of the launch and the LDS fill: a run of 4 iterations */
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), lds, 0, out, 4, 1.0f); hipEventRecord(e1);
    hipDeviceSynchronize(); float ms0; hipEventElapsedTime(&ms0, e0, e1);
    const double inst_per_simd = (double)(iters - 4) * REP * 8 * w;
    const double ns = (best - ms0) * 1e6 / inst_per_simd;
    printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ns_per_wave_instr\": %.4f, \"cycles_at_2p4GHz\": %.3f}\n", op_name[OP], w, ns, ns * 2.4);
    fflush(stdout);
}

template <int OP> static void run_all(int cus, float *out, const int *ws, int nw)
{
    for (int i = 0; i < nw; i++) run<OP>(ws[i], cus, out);
    if constexpr (OP + 1 < N_OPS) run_all<OP + 1>(cus, out, ws, nw);
}

int main(int argc, char **argv)
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    float *out; hipMalloc(&out, 1 << 26);
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    const int ws[3] = { 2, 1, 3 };
    run_all<0>(p.multiProcessorCount, out, ws, argc > 1 ? atoi(argv[1]) : 3);
    return 0;
}
