// Does a wave64 VALU instruction cost less when most of EXEC is off?  (round 5 question for the latency kernel's serial stage:
// wave-uniform arithmetic that only needs one lane.)  One wave, a dependent chain and an independent stream of v_add_f32 /
// v_fma_f64 / v_pk_add_f32, timed with s_memtime under EXEC = all, low 16 lanes, lane 0.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/exec_skip.cpp -o /tmp/exec_skip && /tmp/exec_skip
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ void chain(float *out, uint64_t *cycles, int lanes, int reps)
{
	float a = threadIdx.x * 1e-9f, b = 1.0f, c = 2.0f, d = (MODE == 6 || MODE == 7) ? -3.0f : 3.0f;
	double x = threadIdx.x * 1e-9, y = 1.5;
	uint64_t t0 = 0, t1 = 0;
	if ((int)threadIdx.x < lanes) {
		t0 = __builtin_readcyclecounter();
		for (int r = 0; r < reps; r++) {
			if (MODE == 0) {               /* dependent f32 chain, 16 per trip */
#pragma unroll
				for (int u = 0; u < 16; u++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
			} else if (MODE == 1) {        /* four independent f32 chains, 16 per trip */
#pragma unroll
				for (int u = 0; u < 4; u++) {
					asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
					asm volatile("v_add_f32 %0, %0, %1" : "+v"(b) : "v"(c));
					asm volatile("v_add_f32 %0, %0, %1" : "+v"(c) : "v"(d));
					asm volatile("v_add_f32 %0, %0, %1" : "+v"(d) : "v"(a));
				}
			} else if (MODE == 2) {        /* dependent f64 chain */
#pragma unroll
				for (int u = 0; u < 16; u++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y));
			} else if (MODE == 3) {        /* f32 alternating with a scalar op */
#pragma unroll
				for (int u = 0; u < 8; u++) {
					asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
					asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");
				}
			} else if (MODE == 4) {        /* v_cmp -> s_and on the result -> v_cndmask: the crossing */
#pragma unroll
				for (int u = 0; u < 4; u++) {
					asm volatile("v_cmp_lt_f32 vcc, %1, %2\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %0, %1, %2, vcc\n v_add_f32 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c) : "vcc", "scc");
				}
			} else if (MODE == 5) {        /* the same without the scalar step */
#pragma unroll
				for (int u = 0; u < 4; u++) {
					asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c) : "vcc");
				}
			} else if (MODE == 6) {        /* a branch that is never taken behind every add (condition from a VALU compare) */
#pragma unroll
				for (int u = 0; u < 4; u++)
					asm volatile("v_add_f32 %0, %0, %1\n v_cmp_lt_f32 vcc, %0, %2\n s_cbranch_vccnz 1f\n v_add_f32 %0, %0, %1\n1:" : "+v"(a) : "v"(b), "v"(d) : "vcc");
			} else if (MODE == 7) {        /* the same instructions with an s_nop where the branch was */
#pragma unroll
				for (int u = 0; u < 4; u++)
					asm volatile("v_add_f32 %0, %0, %1\n v_cmp_lt_f32 vcc, %0, %2\n s_nop 0\n v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b), "v"(d) : "vcc");
			} else if (MODE == 8) {        /* a never-taken branch on a SCALAR condition */
#pragma unroll
				for (int u = 0; u < 4; u++)
					asm volatile("v_add_f32 %0, %0, %1\n s_cmp_eq_u32 s20, 12345\n s_cbranch_scc1 1f\n v_add_f32 %0, %0, %1\n1:" : "+v"(a) : "v"(b) : "s20", "scc");
			} else if (MODE == 9) {        /* a TAKEN branch over nothing */
#pragma unroll
				for (int u = 0; u < 4; u++)
					asm volatile("v_add_f32 %0, %0, %1\n s_cmp_lg_u32 s20, 12345\n s_cbranch_scc1 1f\n1:\n v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b) : "s20", "scc");
			}
		}
		t1 = __builtin_readcyclecounter();
	}
	out[threadIdx.x] = a + b + c + d + (float)(x + y);
	if (threadIdx.x == 0) *cycles = t1 - t0;
}

template <int MODE>
static void run(const char *what, int per_trip)
{
	float *out; uint64_t *cyc, h;
	hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 8);
	const int reps = 100000;
	for (int lanes : { 64, 32, 16, 1 }) {
		chain<MODE><<<1, 64>>>(out, cyc, lanes, reps);
		chain<MODE><<<1, 64>>>(out, cyc, lanes, reps);
		hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
		printf("%-44s lanes %2d: %.2f counter ticks per instruction\n", what, lanes, (double)h / ((double)reps * per_trip));
		fflush(stdout);
	}
	hipFree(out); hipFree(cyc);
}

int main()
{
	int clk = 0, wall = 0;
	hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
	hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
	printf("clock %d kHz, wall clock %d kHz (s_memtime counts the latter or a fixed 100 MHz)\n", clk, wall);
	run<0>("dependent v_add_f32", 16);
	run<1>("four chains of v_add_f32", 16);
	run<2>("dependent v_add_f64", 16);
	run<3>("v_add_f32 ; s_add_u32 alternating", 16);
	run<4>("v_cmp ; s_and ; v_cndmask ; v_add", 16);
	run<5>("v_cmp ; v_cndmask ; v_add ; v_add", 16);
	run<6>("v_add ; v_cmp ; s_cbranch_vccnz (never) ; v_add", 16);
	run<7>("v_add ; v_cmp ; s_nop ; v_add", 16);
	run<8>("v_add ; s_cmp ; s_cbranch_scc1 (never) ; v_add", 16);
	run<9>("v_add ; s_cmp ; s_cbranch_scc1 (taken) ; v_add", 16);
	return 0;
}
