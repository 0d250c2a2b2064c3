// Exhaustive proof that the division-free turn-code computation used by the v2 kernel
// (demod_device.h: md_turn_code) equals the reference's
//     (int16)(fx * 0x10000 / (2*M_PI))            dsp/sincos.c:24
// for EVERY float fx with |fx| < 16 (the PLL can only produce |fx| < 2*pi + 1 + pi/2 < 8.9).
// Build: g++ -O2 -ffp-contract=off -pthread verify_sincos_shortcut.cpp -o verify && ./verify
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>

static const double TWO_PI = 6.283185307179586476925286766559;
static const double INV_TWO_PI = 1.0 / TWO_PI;     // RN(1/(2*pi_d))

static inline int32_t ref_code(float fx) {
	const double xd = (double)(fx * 65536.0f) / TWO_PI;
	return (int32_t)xd;                              // truncation toward zero; low 16 bits taken by caller
}

static inline int32_t fast_code(float fx) {
	const double xd = (double)(fx * 65536.0f);
	const double ax = std::fabs(xd);
	int32_t n = (int32_t)(ax * INV_TWO_PI);          // within +-1 of trunc(ax / 2pi)
	double r = std::fma(-(double)n, TWO_PI, ax);     // exact: multiple of 2^-50 below 2^3
	if (r < 0.0) n -= 1;
	else if (r >= TWO_PI) n += 1;
	return xd < 0.0 ? -n : n;
}

int main() {
	const unsigned nthreads = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 4;
	std::atomic<uint64_t> bad{0}, total{0};
	std::vector<std::thread> th;
	const uint32_t limit = 0x41800000u;              // bit pattern of 16.0f: all |fx| < 16
	for (unsigned t = 0; t < nthreads; t++) {
		th.emplace_back([&, t]() {
			uint64_t lb = 0, lt = 0;
			for (uint64_t u = t; u < limit; u += nthreads) {
				for (uint32_t sign = 0; sign < 2; sign++) {
					uint32_t bits = (uint32_t)u | (sign << 31);
					float fx; std::memcpy(&fx, &bits, 4);
					const int32_t a = ref_code(fx), b = fast_code(fx);
					lt++;
					if (a != b) { if (lb < 5) std::printf("MISMATCH fx=%a ref=%d fast=%d\n", fx, a, b); lb++; }
				}
			}
			bad += lb; total += lt;
		});
	}
	for (auto &x : th) x.join();
	std::printf("checked %llu floats, mismatches %llu\n", (unsigned long long)total.load(), (unsigned long long)bad.load());
	return bad.load() ? 1 : 0;
}
