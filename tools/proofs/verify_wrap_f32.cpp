// Exhaustive proof that the two "subtract one period" steps of the PLL can be done in float arithmetic:
//     pll.c:60-61   phase += freq; if ((double)phase >= 2*M_PI) phase = (float)((double)phase - 2*M_PI);
//     pll.c:113     phase = (float)fmod((double)phase', 2*M_PI)   for 2pi <= |phase'| < 4pi  (== phase' -+ 2pi, exact in double)
// Claim: for every float x with 2pi <= |x| < 4pi,
//     (float)((double)x - copysign(2pi_d, x))  ==  (x - C_HI) - C_LO  for x > 0,  (x + C_HI) + C_LO  for x < 0      in float arithmetic,
// with C_HI = (float)2pi = 6.2831855f (x - C_HI is exact: Sterbenz) and C_LO = (float)(2pi_d - C_HI) = -1.7484555e-07f.
// Build: g++ -O2 -ffp-contract=off verify_wrap_f32.cpp -o verify_wrap && ./verify_wrap
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

int main() {
	const double TWO_PI = 6.283185307179586476925286766559;
	const float C_HI = 6.28318548202514648437500f;
	const float C_LO = (float)(TWO_PI - (double)C_HI);
	std::printf("C_HI = %a, C_LO = %a (%.9g)\n", C_HI, C_LO, C_LO);
	uint32_t lo, hi; float f = C_HI; std::memcpy(&lo, &f, 4); f = 12.56637096405029296875f /* the float just above 4pi */; std::memcpy(&hi, &f, 4);
	uint64_t n = 0, bad = 0;
	for (uint32_t b = lo; b < hi; b++) {
		for (int s = 0; s < 2; s++) {
			uint32_t bits = b | ((uint32_t)s << 31);
			float x; std::memcpy(&x, &bits, 4);
			const float ref = (float)((double)x - std::copysign(TWO_PI, (double)x));
			volatile float d = x - std::copysign(C_HI, x);
			const float got = (x < 0.0f) ? d + C_LO : d - C_LO;
			n++;
			if (std::memcmp(&ref, &got, 4)) { if (bad < 5) std::printf("MISMATCH x=%a ref=%a got=%a\n", x, ref, got); bad++; }
		}
	}
	std::printf("checked %llu floats with 2pi <= |x| < 4pi, mismatches %llu\n", (unsigned long long)n, (unsigned long long)bad);
	return bad ? 1 : 0;
}
