// How fast can rows of a caller's host buffers reach the device without a CPU pack?  (round 5, host_pipe.cpp)
// The shape of bench.py host_fed: NS streams x N samples x 4 B, rows at a uniform stride in ONE host allocation; every sub-block k of K
// takes bytes [k W, (k + 1) W) of every row (W = N 4 / K) into a packed device buffer.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/h2d_rect.cpp -o /tmp/h2d_rect -pthread && /tmp/h2d_rect [NS=16384] [N=32768] [K=16]
// Prints: hipHostRegister cost, then GB/s of (a) contiguous pinned copies of the same bytes, (b) hipMemcpy2DAsync from the registered
// rows, (c) a gather kernel that reads the registered rows over the link itself, each with and without a device-to-host copy of
// 13.5 % of the bytes running the other way, and (d) the CPU pack into a pinned ring with T threads (plain and non-temporal stores).
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// one block per (row, 4 KB piece): 256 threads x 16 B
__global__ void
gather_rows(const uint4 *__restrict__ src, size_t src_pitch16, size_t col16, uint4 *__restrict__ dst, size_t w16, unsigned rows)
{
	const size_t pieces = (w16 + 255) / 256;
	for (size_t b = blockIdx.x; b < (size_t)rows * pieces; b += gridDim.x) {
		const size_t r = b / pieces, c = (b % pieces) * 256 + threadIdx.x;
		if (c < w16) dst[r * w16 + c] = src[r * src_pitch16 + col16 + c];
	}
}

static void
nt_copy(unsigned char *dst, const unsigned char *src, size_t n)          /* dst 32-byte aligned, n % 32 == 0 */
{
	for (size_t i = 0; i < n; i += 32) _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i)));
}

int main(int argc, char **argv)
{
	const size_t NS = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 32768, K = argc > 3 ? atoi(argv[3]) : 16;
	const size_t row = N * 4, total = NS * row, W = row / K, sub = NS * W;
	CK(hipSetDevice(0)); CK(hipFree(nullptr));
	unsigned char *h = static_cast<unsigned char *>(aligned_alloc(4096, total));
	for (size_t i = 0; i < total; i += 4096) h[i] = (unsigned char)i;
	memset(h, 3, total);
	unsigned char *d[2]; CK(hipMalloc(&d[0], sub)); CK(hipMalloc(&d[1], sub));
	const size_t out_bytes = total * 135 / 1000;
	unsigned char *h_out, *d_out; CK(hipHostMalloc(reinterpret_cast<void **>(&h_out), out_bytes / K + 64, hipHostMallocDefault)); CK(hipMalloc(&d_out, out_bytes / K + 64));
	hipStream_t s_in, s_out; CK(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
	printf("%zu streams x %zu samples = %.3f GB, %zu sub-blocks of %.1f MB (%zu B per row)\n", NS, N, total / 1e9, K, sub / 1e6, W);

	// (d) CPU pack into a pinned ring, T threads
	{
		unsigned char *p[2]; CK(hipHostMalloc(reinterpret_cast<void **>(&p[0]), sub, hipHostMallocDefault)); CK(hipHostMalloc(reinterpret_cast<void **>(&p[1]), sub, hipHostMallocDefault));
		for (int nt_store = 0; nt_store < 2; nt_store++)
			for (unsigned T : { 4u, 8u, 12u, 16u, 24u }) {
				hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
				double best = 1e9, best_pack = 1e9;
				for (int rep = 0; rep < 3; rep++) {
					const double t0 = now();
					double t_pack = 0;
					for (size_t k = 0; k < K; k++) {
						if (k >= 2) CK(hipEventSynchronize(ev[k & 1]));
						const double tp = now();
						std::vector<std::thread> th;
						for (unsigned t = 0; t < T; t++) th.emplace_back([&, t] {
							for (size_t r = NS * t / T; r < NS * (t + 1) / T; r++) {
								if (nt_store) nt_copy(p[k & 1] + r * W, h + r * row + k * W, W);
								else memcpy(p[k & 1] + r * W, h + r * row + k * W, W);
							}
							if (nt_store) _mm_sfence();
						});
						for (auto &x : th) x.join();
						t_pack += now() - tp;
						CK(hipMemcpyAsync(d[k & 1], p[k & 1], sub, hipMemcpyHostToDevice, s_in));
						CK(hipEventRecord(ev[k & 1], s_in));
					}
					CK(hipStreamSynchronize(s_in));
					best = std::min(best, now() - t0); best_pack = std::min(best_pack, t_pack);
				}
				printf("CPU pack, %2u threads, %s stores: %.1f ms = %.1f GB/s (pack alone %.1f ms = %.1f GB/s)\n", T, nt_store ? "non-temporal" : "plain       ", best * 1e3, total / best / 1e9, best_pack * 1e3, total / best_pack / 1e9);
			}
		CK(hipHostFree(p[0])); CK(hipHostFree(p[1]));
	}

	double t0 = now(); CK(hipHostRegister(h, total, hipHostRegisterDefault)); printf("hipHostRegister of %.2f GB: %.1f ms\n", total / 1e9, (now() - t0) * 1e3);
	void *h_dev = nullptr; CK(hipHostGetDevicePointer(&h_dev, h, 0));
	for (int with_out = 0; with_out < 2; with_out++) {
		auto other_way = [&](size_t k) { if (with_out) (void)hipMemcpyAsync(h_out, d_out, out_bytes / K, hipMemcpyDeviceToHost, s_out); (void)k; };
		double best[3] = { 1e9, 1e9, 1e9 };
		for (int rep = 0; rep < 3; rep++) {
			t0 = now();
			for (size_t k = 0; k < K; k++) { CK(hipMemcpyAsync(d[k & 1], h + k * sub, sub, hipMemcpyHostToDevice, s_in)); other_way(k); }
			CK(hipStreamSynchronize(s_in)); CK(hipStreamSynchronize(s_out));
			best[0] = std::min(best[0], now() - t0);
			t0 = now();
			for (size_t k = 0; k < K; k++) { CK(hipMemcpy2DAsync(d[k & 1], W, h + k * W, row, W, NS, hipMemcpyHostToDevice, s_in)); other_way(k); }
			CK(hipStreamSynchronize(s_in)); CK(hipStreamSynchronize(s_out));
			best[1] = std::min(best[1], now() - t0);
			t0 = now();
			for (size_t k = 0; k < K; k++) {
				hipLaunchKernelGGL(gather_rows, dim3(4096), dim3(256), 0, s_in, static_cast<const uint4 *>(h_dev), row / 16, k * W / 16, reinterpret_cast<uint4 *>(d[k & 1]), W / 16, (unsigned)NS);
				other_way(k);
			}
			CK(hipStreamSynchronize(s_in)); CK(hipStreamSynchronize(s_out));
			best[2] = std::min(best[2], now() - t0);
		}
		printf("%s: contiguous registered %.1f ms = %.1f GB/s | hipMemcpy2DAsync rows %.1f ms = %.1f GB/s | gather kernel %.1f ms = %.1f GB/s\n",
		       with_out ? "with 13.5 % going out" : "input only          ", best[0] * 1e3, total / best[0] / 1e9, best[1] * 1e3, total / best[1] / 1e9, best[2] * 1e3, total / best[2] / 1e9);
	}
	// after an idle gap: is it the first milliseconds of 2-D copies that are slow?  (events around every sub-block's copy)
	for (int kind = 0; kind < 2; kind++) {
		std::vector<hipEvent_t> e(K + 1);
		for (auto &x : e) CK(hipEventCreate(&x));
		std::this_thread::sleep_for(std::chrono::milliseconds(300));
		CK(hipEventRecord(e[0], s_in));
		for (size_t k = 0; k < K; k++) {
			if (kind == 0) CK(hipMemcpy2DAsync(d[k & 1], W, h + k * W, row, W, NS, hipMemcpyHostToDevice, s_in));
			else CK(hipMemcpyAsync(d[k & 1], h + k * sub, sub, hipMemcpyHostToDevice, s_in));
			CK(hipEventRecord(e[k + 1], s_in));
		}
		CK(hipStreamSynchronize(s_in));
		printf("%s after 300 ms of idle, ms per sub-block:", kind == 0 ? "2-D copies from registered rows" : "contiguous registered copies  ");
		for (size_t k = 0; k < K; k++) { float ms = 0; CK(hipEventElapsedTime(&ms, e[k], e[k + 1])); printf(" %.2f", ms); }
		printf("\n");
	}
	t0 = now(); CK(hipHostUnregister(h)); printf("unregister %.1f ms\n", (now() - t0) * 1e3);
	return 0;
}
