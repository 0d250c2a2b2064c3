// Micro-benchmark: cycles per wave-instruction for the VALU ops the FIR is made of (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#pragma clang fp contract(off)

#define REP 64
template <int OP>
__global__ void k(float *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0+4, a5=a0+5, a6=a0+6, a7=a0+7;
    float b = 1.0000001f;
    int   i0 = threadIdx.x * 65537 + 12345, i1 = i0 * 3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if (OP == 0) { // 8 independent v_mul_f32
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if (OP == 1) { // 4 independent v_pk_mul_f32 (8 floats)
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                             "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                             : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b));
            } else if (OP == 2) { // dependent chain v_add_f32 x8 (one accumulator)
                asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                             "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                             : "+v"(a0) : "v"(b));
            } else if (OP == 3) { // dependent chain v_pk_add_f32 x8
                asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n"
                             "v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n"
                             : "+v"(*(double*)&a0) : "v"(*(double*)&b));
            } else if (OP == 4) { // 8 independent sdwa converts
                asm volatile("v_cvt_f32_i32_sdwa %0, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                             "v_cvt_f32_i32_sdwa %1, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             "v_cvt_f32_i32_sdwa %2, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                             "v_cvt_f32_i32_sdwa %3, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             "v_cvt_f32_i32_sdwa %4, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                             "v_cvt_f32_i32_sdwa %5, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             "v_cvt_f32_i32_sdwa %6, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                             "v_cvt_f32_i32_sdwa %7, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(i0), "v"(i1));
            } else if (OP == 5) { // FIR-like mix: per tap 2 cvt + pk_mul + pk_add (dependent acc), x4 taps
                asm volatile(
                    "v_cvt_f32_i32_sdwa %2, sext(%6) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                    "v_cvt_f32_i32_sdwa %3, sext(%6) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                    "v_cvt_f32_i32_sdwa %4, sext(%7) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                    "v_cvt_f32_i32_sdwa %5, sext(%7) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                    "v_pk_mul_f32 %1, %1, %8\n"
                    "v_pk_add_f32 %0, %0, %1\n"
                    "v_pk_mul_f32 %1, %1, %8\n"
                    "v_pk_add_f32 %0, %0, %1\n"
                    : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(i0), "v"(i1), "v"(*(double*)&b));
            } else if (OP == 6) { // 8 independent v_mov_b64 (shift cost)
                asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0\n"
                             "v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0\n"
                             : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6));
            } else if (OP == 7) { // scalar FIR tap: 2 cvt + 2 mul + 2 add (2 independent chains), x2 taps = 12 instr? keep 8: 1 tap + extra
                asm volatile(
                    "v_cvt_f32_i32_sdwa %2, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                    "v_cvt_f32_i32_sdwa %3, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                    "v_mul_f32 %2, %2, %5\n v_mul_f32 %3, %3, %5\n"
                    "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n"
                    "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n"
                    : "+v"(a0), "+v"(a1), "=v"(a2), "=v"(a3) : "v"(i0), "v"(b));
            } else if (OP == 8) { // f64 fma dependent-ish: 4 independent v_fma_f64 x2
                double *d0=(double*)&a0,*d1=(double*)&a2,*d2=(double*)&a4,*d3=(double*)&a6; double db = 1.0000000001;
                asm volatile("v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3\n"
                             "v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3\n"
                             : "+v"(*d0), "+v"(*d1), "+v"(*d2), "+v"(*d3) : "v"(db));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP> void run(const char *name, int waves_per_simd) {
    int dev_cus = 256; float *out; hipMalloc(&out, 1 << 24);
    int threads = 256 * waves_per_simd;            // block = waves_per_simd waves per SIMD on one CU
    if (threads > 1024) threads = 1024;
    int blocks = dev_cus * (256 * waves_per_simd / threads);
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f); hipEventRecord(e1);
    hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst_per_simd = (double)iters * REP * 8 * waves_per_simd;     // wave-instructions issued per SIMD
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-34s waves/SIMD=%d  %.3f ms  %.2f cycles/wave-instr (at 2.4 GHz)\n", name, waves_per_simd, ms, cyc / inst_per_simd);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_mul_f32 x8 indep", w); run<1>("v_pk_mul_f32 x8 (4 indep)", w);
        run<2>("v_add_f32 dep chain", w); run<3>("v_pk_add_f32 dep chain", w);
        run<4>("v_cvt_f32_i32_sdwa x8", w); run<5>("FIR pk mix (4cvt+2pkmul+2pkadd)", w);
        run<7>("FIR scalar mix (2cvt+2mul+4add)", w);
        run<6>("v_mov_b64 x8", w); run<8>("v_fma_f64 x8 (4 indep)", w);
    }
    return 0;
}
