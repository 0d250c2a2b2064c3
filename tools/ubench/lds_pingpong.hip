// Micro-benchmark: round-trip time of a 16-byte message between two waves of one workgroup through an LDS mailbox (polling).
// Variants: who writes (all lanes / lane 0), how the poll reads (4 dwords / one b128), s_sleep in the poll loop.
// Build: hipcc --offload-arch=gfx950 -O3 lds_pingpong.hip -o lds_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u4 *lds_u4_ptr;

template <int MODE>
__device__ __forceinline__ u4 poll(u4 *box, uint32_t want, uint32_t &spins)
{
	u4 v;
	const uint32_t addr = (uint32_t)(uintptr_t)(lds_u4_ptr)box;
	do {
		asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
		spins++;
		if (MODE & 4) __builtin_amdgcn_s_sleep(1);
	} while (v.x != want && spins < (1u << 24));
	return v;
}
template <int MODE>
__device__ __forceinline__ void post(u4 *box, u4 m)
{
	const uint32_t addr = (uint32_t)(uintptr_t)(lds_u4_ptr)box;
	if (!(MODE & 1) || (threadIdx.x & 63) == 0)
		asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(m) : "memory");
}

template <int MODE>
__global__ void __launch_bounds__(128) pingpong(unsigned long long *out, int n)
{
	__shared__ u4 box[2];
	const int wave = threadIdx.x >> 6;
	if (threadIdx.x == 0) { box[0] = (u4){0, 0, 0, 0}; box[1] = (u4){0, 0, 0, 0}; }
	__syncthreads();
	float acc = (float)threadIdx.x;
	uint32_t spins = 0;
	if (MODE & 8) { if (wave == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
	const unsigned long long t0 = clock64();
	if (wave == 0) {
		for (int i = 1; i <= n; i++) {
			post<MODE>(&box[0], (u4){(uint32_t)i, __float_as_uint(acc), 0, 0});
			const u4 r = poll<MODE>(&box[1], (uint32_t)i, spins);
			acc += __uint_as_float(r.y) * 1e-9f;
		}
	} else {
		for (int i = 1; i <= n; i++) {
			const u4 r = poll<MODE>(&box[0], (uint32_t)i, spins);
			post<MODE>(&box[1], (u4){(uint32_t)i, r.y + 1, 0, 0});
		}
	}
	const unsigned long long t1 = clock64();
	if ((threadIdx.x & 63) == 0) { out[2 * wave] = t1 - t0; out[2 * wave + 1] = spins + (unsigned long long)(acc == 12345.f); }
}

template <int MODE> void run(const char *what)
{
	unsigned long long *d, h[4];
	hipMalloc(&d, 32);
	const int n = 100000;
	hipLaunchKernelGGL(pingpong<MODE>, dim3(1), dim3(128), 0, 0, d, n); hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
	printf("%-60s %.1f cycles per round trip, %.2f / %.2f polls per message\n", what, (double)h[0] / n, (double)h[1] / n, (double)h[3] / n);
	hipFree(d);
}

int main()
{
	run<0>("all lanes write, b128 poll");
	run<1>("lane 0 writes, b128 poll");
	run<5>("lane 0 writes, b128 poll, s_sleep 1 between polls");
	run<9>("lane 0 writes, b128 poll, wave 0 at priority 3");
	run<4>("all lanes write, s_sleep 1");
	return 0;
}
