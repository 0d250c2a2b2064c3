// Exhaustive search/proof: is the reference's turn code
//     (int32)((double)(fx * 0x10000) / (2*M_PI))            dsp/sincos.c:24
// equal to a single double MULTIPLICATION  (int32)((double)(fx * 0x10000) * K)  for every float |fx| < LIMIT, for some
// constant K near 1/(2*pi)?  The quotient and the product are two roundings of (nearly) the same real number, so they
// truncate differently only if an integer lies between them: for the ~2.2e9 floats below 16 the closest approach of
// x*65536/(2pi) to an integer decides.  Prints the number of mismatches for K = RN(1/2pi) + k ulp, k = -3..3.
// Build: g++ -O2 -ffp-contract=off -pthread verify_turncode_mul.cpp -o verify_mul && ./verify_mul [limit_bits_hex]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>

static const double TWO_PI = 6.283185307179586476925286766559;

int main(int argc, char **argv) {
	const unsigned nthreads = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 4;
	const uint32_t limit = argc > 1 ? (uint32_t)strtoul(argv[1], nullptr, 16) : 0x41800000u;   // bit pattern of 16.0f
	const int NK = 7;
	double K[NK];
	{
		double k0 = 1.0 / TWO_PI;
		for (int i = 0; i < NK; i++) {
			double k = k0;
			for (int s = 0; s < std::abs(i - 3); s++) k = std::nextafter(k, i < 3 ? 0.0 : 1.0);
			K[i] = k;
		}
	}
	std::atomic<uint64_t> bad[NK];
	for (auto &b : bad) b = 0;
	std::atomic<uint64_t> total{0};
	std::vector<std::thread> th;
	for (unsigned t = 0; t < nthreads; t++) {
		th.emplace_back([&, t]() {
			uint64_t lb[NK] = {0}, lt = 0;
			for (uint64_t u = t; u < limit; u += nthreads) {
				uint32_t bits = (uint32_t)u;
				float fx; std::memcpy(&fx, &bits, 4);
				const double ax = (double)(fx * 65536.0f);          // the sign is symmetric: both sides truncate toward zero
				const int32_t ref = (int32_t)(ax / TWO_PI);
				lt++;
				for (int i = 0; i < NK; i++) {
					const int32_t n = (int32_t)(ax * K[i]);
					if (n != ref) { if (lb[i] < 3 && i == 3) std::printf("MISMATCH k=%d fx=%a ref=%d got=%d\n", i - 3, fx, ref, n); lb[i]++; }
				}
			}
			for (int i = 0; i < NK; i++) bad[i] += lb[i];
			total += lt;
		});
	}
	for (auto &x : th) x.join();
	std::printf("checked %llu non-negative floats below %#x\n", (unsigned long long)total.load(), limit);
	int rc = 1;
	for (int i = 0; i < NK; i++) {
		std::printf("K = RN(1/2pi) %+d ulp = %a : mismatches %llu\n", i - 3, K[i], (unsigned long long)bad[i].load());
		if (i == 3 && bad[i].load() == 0) rc = 0;
	}
	return rc;
}
