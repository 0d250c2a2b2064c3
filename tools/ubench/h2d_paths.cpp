// Host-to-device paths for a pageable buffer (what mdemod_demodulate_recording_host is handed): rates and first-call costs.
//   hipcc -O2 tools/ubench/h2d_paths.cpp -o /tmp/h2d && /tmp/h2d [MB=256]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv)
{
	const size_t mb = argc > 1 ? atoi(argv[1]) : 256, n = mb << 20;
	double t0 = now();
	CK(hipSetDevice(0)); CK(hipFree(nullptr));
	printf("runtime up: %.1f ms\n", (now() - t0) * 1e3);
	unsigned char *h = static_cast<unsigned char *>(malloc(n));
	memset(h, 1, n);
	unsigned char *d; CK(hipMalloc(&d, n));
	hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	t0 = now(); CK(hipMemcpy(d, h, 4 << 20, hipMemcpyHostToDevice)); printf("first pageable hipMemcpy of 4 MB: %.1f ms\n", (now() - t0) * 1e3);
	t0 = now(); CK(hipMemcpy(d, h, 4 << 20, hipMemcpyHostToDevice)); printf("second: %.2f ms\n", (now() - t0) * 1e3);
	t0 = now(); CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); printf("pageable hipMemcpy %zu MB: %.1f ms = %.1f GB/s\n", mb, (now() - t0) * 1e3, n / (now() - t0) / 1e9);
	t0 = now();
	for (size_t at = 0; at < n; at += 32u << 20) { CK(hipMemcpyAsync(d + at, h + at, std::min<size_t>(32u << 20, n - at), hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }
	printf("pageable hipMemcpyAsync in 32 MB chunks: %.1f ms = %.1f GB/s\n", (now() - t0) * 1e3, n / (now() - t0) / 1e9);
	// own pinned ring: CPU memcpy into pinned chunks, DMA behind it
	for (size_t chunk_mb : { 4, 16 }) {
		const size_t c = chunk_mb << 20;
		unsigned char *p[2];
		t0 = now();
		CK(hipHostMalloc(reinterpret_cast<void **>(&p[0]), c, hipHostMallocDefault)); CK(hipHostMalloc(reinterpret_cast<void **>(&p[1]), c, hipHostMallocDefault));
		const double t_alloc = (now() - t0) * 1e3;
		hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
		t0 = now();
		int k = 0;
		for (size_t at = 0; at < n; at += c, k ^= 1) {
			const size_t m = std::min(c, n - at);
			if (at >= 2 * c) CK(hipEventSynchronize(ev[k]));
			memcpy(p[k], h + at, m);
			CK(hipMemcpyAsync(d + at, p[k], m, hipMemcpyHostToDevice, s));
			CK(hipEventRecord(ev[k], s));
		}
		CK(hipStreamSynchronize(s));
		printf("pinned ring 2 x %zu MB (alloc %.1f ms): %.1f ms = %.1f GB/s\n", chunk_mb, t_alloc, (now() - t0) * 1e3, n / (now() - t0) / 1e9);
		CK(hipHostFree(p[0])); CK(hipHostFree(p[1]));
	}
	t0 = now(); CK(hipHostRegister(h, n, hipHostRegisterDefault)); const double t_reg = (now() - t0) * 1e3;
	t0 = now(); CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
	printf("hipHostRegister %.1f ms, then copy %.1f ms = %.1f GB/s", t_reg, (now() - t0) * 1e3, n / (now() - t0) / 1e9);
	t0 = now(); CK(hipHostUnregister(h)); printf(", unregister %.1f ms\n", (now() - t0) * 1e3);
	// register in pieces on a second thread while copying (pipelined)
	t0 = now();
	{
		const size_t c = 32u << 20;
		for (size_t at = 0; at < n; at += c) { const size_t m = std::min(c, n - at); CK(hipHostRegister(h + at, m, hipHostRegisterDefault)); CK(hipMemcpyAsync(d + at, h + at, m, hipMemcpyHostToDevice, s)); }
		CK(hipStreamSynchronize(s));
		printf("register 32 MB pieces + async copies: %.1f ms = %.1f GB/s\n", (now() - t0) * 1e3, n / (now() - t0) / 1e9);
		for (size_t at = 0; at < n; at += c) CK(hipHostUnregister(h + at));
	}
	return 0;
}
