// Host check of csrc/clock_jump.h: the closed-form blind steps of configs[3]'s symbol clock against the reference's sequential
// rounded additions (timing.c:32-38), for EVERY clock word f the instance can be launched with (step_safe == 109) and a dense
// sample of starting phases, including the ones that make ties (f's low bits = half an ulp of a binade) - the case the
// argument in clock_jump.h needs the "one real addition inside the binade" for.
//   g++ -O2 -ffp-contract=off -o /tmp/vcj tools/proofs/verify_clock_jump.cpp && /tmp/vcj [stride of f = 1] [starting phases per f = 512]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../../meteor_demod_amd/csrc/clock_jump.h"

static float next_up(float x) { uint32_t u; memcpy(&u, &x, 4); u++; memcpy(&x, &u, 4); return x; }

/* any rate (clock_jump_run + cj_schedule): a list of sample rates / symbol rates / -O, every run the host would schedule, clock
 * words sampled over the whole range the loop allows (plus the words that make ties in some binade), starting phases over the window */
static int
generic(int n_f, int n_p0)
{
	struct Cfg { double fs, sym; int interp, oqpsk; };
	std::vector<Cfg> cfgs;
	const double rates[] = { 230e3, 288e3, 900001, 1e6, 1.024e6, 1.4e6, 1.8e6, 2.048e6, 2.4e6, 2.56e6, 3.2e6, 6e6, 10e6, 20e6 };
	for (double fs : rates) for (int oq = 0; oq < 2; oq++) for (int interp : { 1, 2, 3, 4, 5, 8 })
		cfgs.push_back({ fs, oq ? 80e3 : 72e3, interp, oq });
	unsigned long long bad = 0, cases = 0, runs = 0, unused = 0, short_by[4] = { 0, 0, 0, 0 };
	double worst_ratio = 0;
	for (const Cfg &c : cfgs) {
		/* demod_host.cpp: sym_freq, its allowed deviation, the launch bound */
		const float center = (float)(2.0 * M_PI * c.sym / (c.fs * c.interp)), maxdev = center / (float)(1 << 12);
		const float f_hi = nextafterf((float)(((double)center + (double)maxdev) * (1.0 + 1e-6)), 1e30f);
		const float inv = (float)((1.0 - 1.0 / 4096.0) / (double)f_hi);
		const float pi_f = (float)M_PI, two_pi_f = 2.0f * pi_f;
		for (int run = 0; run < (c.oqpsk ? 2 : 1); run++) {
			const float S = run ? pi_f : 0.0f, thr = (c.oqpsk && !run) ? pi_f : two_pi_f;
			const cj_sched J = cj_schedule(S, thr, f_hi);
			if (!J.nb) { unused++; continue; }
			runs++;
			std::vector<float> fs;
			const float f_lo = center - maxdev;
			for (int i = 0; i < n_f; i++) fs.push_back(f_lo + (center + maxdev - f_lo) * (float)i / (float)(n_f - 1));
			for (int i = 0; i < n_f / 4; i++) {          /* words with a tie in some binade: low bits 1000.. at each length */
				float f = fs[(size_t)i * 4]; uint32_t u; memcpy(&u, &f, 4);
				const int bits = 1 + i % 8; u = (u & ~((1u << bits) - 1)) | (1u << (bits - 1)); memcpy(&f, &u, 4);
				if (f >= f_lo && f <= center + maxdev) fs.push_back(f);
			}
			const unsigned nt = std::max(1u, std::thread::hardware_concurrency());
			std::vector<unsigned long long> b(nt, 0), n(nt, 0), sb(nt * 4, 0);
			std::vector<double> wr(nt, 0);
			std::vector<std::thread> th;
			for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t] {
				uint64_t rng = 0x9E3779B97F4A7C15ull * (t + 1);
				for (size_t i = t; i < fs.size(); i += nt) {
					const float f = fs[i];
					for (int j = 0; j < n_p0; j++) {
						rng = rng * 6364136223846793005ull + 1442695040888963407ull;
						float p0;
						if (j < 8) p0 = S + (j & 1 ? 1 : -1) * (j / 2) * 0.1f * f;
						else if (j < 16) p0 = j & 1 ? nextafterf(J.floor, 10.0f) + 1e-6f * (j - 8) : nextafterf(J.hi, -10.0f) - 1e-6f * (j - 8);
						else if (j < 24) p0 = J.lo + (j & 1 ? 1e-6f : -1e-6f) * (j - 16);
						else p0 = J.floor + (J.hi - J.floor) * (float)((rng >> 40) * (1.0 / 16777216.0));
						if (!(p0 > J.floor && p0 < J.hi)) continue;
						float ps = p0; int ms = 0;
						while (!(ps >= thr)) { ps = ps + f; ms++; }
						float p = p0;
						const int k = clock_jump_run(p, f, thr, f_hi, inv, J);
						const float p1 = p + f, p2 = p1 + f, p3 = p2 + f, p4 = p3 + f;
						const bool c1 = p1 >= thr, c2 = p2 >= thr, c3 = p3 >= thr, c4 = p4 >= thr;
						const int m = k + 1 + (c1 ? 0 : 1) + (c2 ? 0 : 1) + (c3 ? 0 : 1);
						const float ph = c1 ? p1 : (c2 ? p2 : (c3 ? p3 : p4));
						n[t]++;
						if (p >= thr || !c4 || m != ms || memcmp(&ph, &ps, 4) != 0 || k > J.max_steps) b[t]++;
						sb[t * 4 + (c1 ? 0 : c2 ? 1 : c3 ? 2 : 3)]++;
						const double r = (double)(J.ra + 3 * (J.nb - 1) + 12 * J.nb) / ms;       /* rough cost: instructions per step replaced */
						if (r > wr[t]) wr[t] = r;
					}
				}
			});
			for (auto &x : th) x.join();
			unsigned long long bb = 0;
			for (unsigned t = 0; t < nt; t++) { bb += b[t]; cases += n[t]; for (int q = 0; q < 4; q++) short_by[q] += sb[t * 4 + q]; if (wr[t] > worst_ratio) worst_ratio = wr[t]; }
			if (bb) printf("  %.0f S/s %s -O %d run %d (ra %d, %d binades from %g): %llu mismatches\n", c.fs, c.oqpsk ? "oqpsk" : "qpsk", c.interp, run, J.ra, J.nb, J.B0, bb);
			bad += bb;
		}
	}
	printf("any rate: %zu configurations, %llu runs with a schedule (%llu too short for one), %llu cases: %llu mismatches; firing found by checked addition 1..4: %llu %llu %llu %llu; worst instructions per step replaced %.2f\n",
	       cfgs.size(), runs, unused, cases, bad, short_by[0], short_by[1], short_by[2], short_by[3], worst_ratio);
	return bad ? 1 : 0;
}

int main(int argc, char **argv)
{
	if (argc > 1 && !strcmp(argv[1], "any")) return generic(argc > 2 ? atoi(argv[2]) : 400, argc > 3 ? atoi(argv[3]) : 64);
	const int stride = argc > 1 ? atoi(argv[1]) : 1, n_p0 = argc > 2 ? atoi(argv[2]) : 512;
	const float thr = 6.28318548202514648437500f;          /* 2 * (float)M_PI: timing.c:37 */
	/* every f_hi that gives step_safe == 109 (demod_host.cpp: ks = floor((2 pi - f_hi - 0.051) / f_hi)), and under each the clock
	   words f in [center (1 - 2^-12), center (1 + 2^-12)] it allows: all in all f in [0.05609, 0.05666] */
	const float f_min = 0.05609f, f_max = 0.05666f;
	std::vector<float> fs;
	{ int i = 0; for (float f = f_min; f <= f_max; f = next_up(f), i++) if (i % stride == 0) fs.push_back(f); }
	const unsigned nt = std::max(1u, std::thread::hardware_concurrency());
	std::vector<unsigned long long> bad(nt, 0), cases(nt, 0), short_by(nt * 4, 0), ties(nt, 0);
	std::vector<std::thread> th;
	for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t] {
		uint64_t rng = 0x9E3779B97F4A7C15ull * (t + 1);
		for (size_t i = t; i < fs.size(); i += nt) {
			const float f = fs[i];
			/* the launcher's f_hi for this f is anywhere in [f, f (1 + 4.9e-4)]; the least conservative inv is the largest: f_hi = f */
			for (int which = 0; which < 2; which++) {
				const float f_hi = which ? f * (1.0f + 1.0f / 1024.0f) : f;       /* (demod_host.cpp: f_hi / f <= (1 + 2^-12)(1 + 1e-6) / (1 - 2^-12) = 1 + 4.9e-4) */
				const float inv = (float)((1.0 - 1.0 / 4096.0) / (double)f_hi);
				uint32_t fb; memcpy(&fb, &f, 4);
				if ((fb & 0x7F) == 0x40 || (fb & 0x3F) == 0x20 || (fb & 0x1F) == 0x10) ties[t]++;
				for (int j = 0; j < n_p0; j++) {
					rng = rng * 6364136223846793005ull + 1442695040888963407ull;
					float p0;
					if (j < 8) p0 = (j & 1 ? 1 : -1) * (j / 2) * 0.1f * f;             /* 0, +-0.1 f, ... */
					else if (j < 16) p0 = j & 1 ? CJ109_P_LO + 1e-6f * j : CJ109_P_HI - 1e-6f * j;
					else p0 = CJ109_P_LO + (CJ109_P_HI - CJ109_P_LO) * (float)((rng >> 40) * (1.0 / 16777216.0));
					if (!(p0 > CJ109_P_LO && p0 < CJ109_P_HI)) continue;
					/* the reference: one rounded addition per step until the threshold */
					float ps = p0; int ms = 0;
					while (!(ps >= thr)) { ps = ps + f; ms++; }
					/* the kernel's way: closed form, then four checked additions */
					float p = p0;
					const int k = clock_jump_109(p, f, thr, inv);
					const float p1 = p + f, p2 = p1 + f, p3 = p2 + f, p4 = p3 + f;
					const bool c1 = p1 >= thr, c2 = p2 >= thr, c3 = p3 >= thr, c4 = p4 >= thr;
					const int m = k + 1 + (c1 ? 0 : 1) + (c2 ? 0 : 1) + (c3 ? 0 : 1);
					const float ph = c1 ? p1 : (c2 ? p2 : (c3 ? p3 : p4));
					cases[t]++;
					if (p >= thr || !c4 || m != ms || memcmp(&ph, &ps, 4) != 0 || k > CJ109_MAX_STEPS) bad[t]++;
					short_by[t * 4 + (c1 ? 0 : c2 ? 1 : c3 ? 2 : 3)]++;
				}
			}
		}
	});
	for (auto &x : th) x.join();
	unsigned long long b = 0, c = 0, ti = 0, s[4] = { 0, 0, 0, 0 };
	for (unsigned t = 0; t < nt; t++) { b += bad[t]; c += cases[t]; ti += ties[t]; for (int k = 0; k < 4; k++) s[k] += short_by[t * 4 + k]; }
	printf("%zu clock words x 2 launch bounds x %d phases = %llu cases (%llu words with a tie in some binade): %llu mismatches; firing found by checked addition 1..4: %llu %llu %llu %llu\n",
	       fs.size(), n_p0, c, ti / 2, b, s[0], s[1], s[2], s[3]);
	return b ? 1 : 0;
}
