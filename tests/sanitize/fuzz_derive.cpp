// Sanitizer + fuzz gate of the HIP-free init-time host code: mdemod_host_derive (csrc/demod_host.cpp) with everything it calls
// (csrc/clock_jump.h: cj_schedule), over random and edge `mdemod_params` - what the reference takes without looking at it
// (main.c:109-111,118-123: any -O / -r / -s; demod.c:8-15).  Built by tests/test_sanitize.py with
//     g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -ffp-contract=off
// Test infrastructure; nothing here is part of the product.
//
//   every call returns MDEMOD_OK or MDEMOD_ERR_PARAM within the wall-clock bound (a watchdog thread aborts with the parameters of a
//   call that does not: round 4's cj_schedule looped forever for symrate >= 2 fs O (OQPSK) / 4 fs O (QPSK))
//   an accepted configuration has a coefficient table of the size its geometry says, sane clock constants, and - where
//   cj_schedule returned a schedule - clock_jump_run agrees with the reference's sequential rounded additions (timing.c:32-38)
//   on clock words across the loop's range and starts across the schedule's window: the proof of tools/proofs/verify_clock_jump.cpp
//   at rates nobody listed
//
//   usage: fuzz_derive <cases> [seed] [threads] [seconds per call]
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../meteor_demod_amd/csrc/demod_host.h"

namespace {

struct Rng {
	uint64_t s;
	uint64_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; uint64_t x = s; x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; return x; }
	double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
	int range(int lo, int hi) { return lo + (int)(next() % (uint64_t)(hi - lo + 1)); }
	double logu(double lo, double hi) { return lo * std::pow(hi / lo, uni()); }
};

struct Slot {
	std::atomic<int64_t> started_ms{ -1 };
	mdemod_params p;                      /* what the call in flight was given (read by the watchdog only after a timeout) */
	int generation;
};

int64_t
now_ms()
{
	return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void
print_params(FILE *f, const mdemod_params &p, int generation)
{
	fprintf(f, "pll_bw=%g sym_bw=%g samplerate=%d symrate=%d interp=%d order=%d oqpsk=%d freq_max=%g bps=%d reserved=0x%x generation=%d\n",
	        p.pll_bw, p.sym_bw, p.samplerate, p.symrate, p.interp_factor, p.rrc_order, p.oqpsk, p.freq_max, p.bps, p.reserved, generation);
}

float
odd_float(Rng &r, float nominal)
{
	switch (r.range(0, 15)) {
	case 0: return 0.0f;
	case 1: return -nominal;
	case 2: return nominal * 1e6f;
	case 3: return nominal * 1e-6f;
	case 4: return NAN;
	case 5: return INFINITY;
	case 6: return -INFINITY;
	case 7: return 1e-42f;                /* denormal */
	default: return nominal * (float)r.logu(0.01, 100.0);
	}
}

mdemod_params
draw(Rng &r, uint64_t i)
{
	mdemod_params p;
	memset(&p, 0, sizeof(p));
	p.pll_bw = r.range(0, 3) ? 1.0f : odd_float(r, 1.0f);
	p.sym_bw = r.range(0, 3) ? 0.00005f : odd_float(r, 0.00005f);
	p.freq_max = r.range(0, 2) ? -1.0f : odd_float(r, 0.3f);
	p.oqpsk = r.range(0, 1);
	p.bps = r.range(0, 19) ? (r.range(0, 2) == 0 ? 8 : (r.range(0, 1) ? 16 : 32)) : r.range(-8, 64);
	p.interp_factor = r.range(0, 7) ? r.range(1, 64) : r.range(-2, 70);
	if (r.range(0, 3) == 0) p.interp_factor = r.range(1, 8);                       /* the everyday ones more often: small tables, more cases per second */
	p.rrc_order = r.range(0, 3) ? r.range(1, 64) : (r.range(0, 7) ? r.range(1, 256) : r.range(-2, 260));
	p.symrate = r.range(0, 2) ? (p.oqpsk ? 80000 : 72000) : (int)r.logu(50.0, 3e7);
	if (r.range(0, 63) == 0) p.symrate = r.range(-1, 1);
	const double fire = p.oqpsk ? 2.0 : 1.0;                                           /* firings per symbol */
	switch (r.range(0, 9)) {
	case 0: case 1: case 2: p.samplerate = (int)r.logu(1e3, 2e8); break;
	case 3: p.samplerate = (int)(p.symrate * r.logu(0.05, 8.0)); break;              /* around and below one sample per symbol */
	case 4: {                                                                          /* the edge of demod_host.cpp's accepted region: fs * fire >= symrate / 4 */
		const double edge = 0.25 * p.symrate / fire;
		p.samplerate = (int)edge + r.range(-2, 2);
		break;
	}
	case 5: {                                                                          /* symrate / (fs O) on and next to whole numbers: several firings per step (cj_schedule's round-4 hang) */
		const int ratio = r.range(1, 8), o = p.interp_factor > 0 ? p.interp_factor : 1;
		p.samplerate = (int)((double)p.symrate / ((double)ratio * o)) + r.range(-1, 1);
		break;
	}
	case 6: p.samplerate = (int)(p.symrate * r.logu(1.0, 300.0)); break;              /* everyday oversampling up to SDR rates */
	case 7: p.samplerate = r.range(0, 1) ? 230000 : 1000000; break;
	case 8: p.samplerate = r.range(-1, 2); break;
	default: p.samplerate = INT32_MAX - r.range(0, 2); break;
	}
	p.device = 0;
	p.n_streams = 1;
	p.reserved = r.range(0, 7) ? 0u : (uint32_t)r.next();
	(void)i;
	return p;
}

/* clock_jump_run against timing.c:32-38's additions, as tools/proofs/verify_clock_jump.cpp does for its listed rates */
bool
check_schedule(const DemodConsts &c, const cj_sched &J, float S, float thr, Rng &r, const mdemod_params &p, int generation)
{
	if (J.nb < 0 || J.nb > 8 || J.ra < 0 || J.max_steps < 0 || J.need < 0 || J.up_max < 0) return false;
	if (J.nb == 0) return true;
	if (!(J.floor <= J.lo && J.lo < J.hi) || !std::isfinite(J.lo) || !std::isfinite(J.hi) || !(J.B0 > 0.0f)) return false;
	if (J.max_steps > (1 << 24) || J.up_max > (1 << 24)) return false;
	/* a clock word from outside the loop's range (the API keeps them out; this is the second fence): the run returns, however wrong */
	for (float bad : { 0.0f, -c.t_center, 1e-30f, NAN, -INFINITY }) {
		float q = J.lo;
		const int k = clock_jump_run(q, bad, thr, c.step_fmax, c.step_inv, J);
		if (k < 0 && !(bad != bad)) return false;
	}
	const float f_min = c.t_center - c.t_maxdev, f_max = c.t_center + c.t_maxdev;
	for (int a = 0; a < 6; a++) {
		float f = a == 0 ? f_min : (a == 1 ? f_max : f_min + (f_max - f_min) * (float)r.uni());
		if (a >= 4) {                                                                  /* a word whose low bits make ties in some binade */
			uint32_t u; memcpy(&u, &f, 4);
			const int bits = 1 + r.range(0, 7);
			u = (u & ~((1u << bits) - 1)) | (1u << (bits - 1));
			float g; memcpy(&g, &u, 4);
			if (g >= f_min && g <= f_max) f = g;
		}
		for (int b = 0; b < 8; b++) {
			float p0;
			if (b == 0) p0 = nextafterf(J.floor, 10.0f);
			else if (b == 1) p0 = nextafterf(J.hi, -10.0f);
			else if (b == 2) p0 = J.lo;
			else if (b == 3) p0 = S;
			else p0 = J.floor + (J.hi - J.floor) * (float)r.uni();
			if (!(p0 > J.floor && p0 < J.hi)) continue;
			float ps = p0; long ms = 0;
			while (!(ps >= thr) && ms < (1 << 25)) { ps = ps + f; ms++; }
			float q = p0;
			const int k = clock_jump_run(q, f, thr, c.step_fmax, c.step_inv, J);
			const float p1 = q + f, p2 = p1 + f, p3 = p2 + f, p4 = p3 + f;
			const bool c1 = p1 >= thr, c2 = p2 >= thr, c3 = p3 >= thr, c4 = p4 >= thr;
			const long m = (long)k + 1 + (c1 ? 0 : 1) + (c2 ? 0 : 1) + (c3 ? 0 : 1);
			const float ph = c1 ? p1 : (c2 ? p2 : (c3 ? p3 : p4));
			if (q >= thr || !c4 || m != ms || memcmp(&ph, &ps, 4) != 0 || k > J.max_steps) {
				fprintf(stderr, "clock_jump_run differs from sequential stepping: S=%g thr=%g f=%.9g p0=%.9g: %ld steps to %.9g against %ld to %.9g (k=%d, max_steps=%d)\n  ",
				        S, thr, f, p0, m, ph, ms, ps, k, J.max_steps);
				print_params(stderr, p, generation);
				return false;
			}
		}
	}
	return true;
}

bool
check_tables(const mdemod_params &p, const HostTables &t, int generation, Rng &r)
{
	const DemodConsts &c = t.c;
	bool ok = true;
	ok = ok && c.interp == p.interp_factor && c.taps == 2 * p.rrc_order + 1 && (c.oqpsk == 0 || c.oqpsk == 1);
	ok = ok && t.rrc.size() == (size_t)c.taps * (size_t)c.interp;
	ok = ok && !t.ctab.empty() && c.ctab_row_stride >= c.ctab_row_floats && c.ctab_row_floats > 0 && c.ctab_row_stride % 4 == 0;
	ok = ok && c.hpad >= c.taps - 1 && c.win_granules > 0 && c.step_safe >= 0 && c.step_check == 4;
	ok = ok && c.t_center > 0.0f && c.step_fmax > c.t_center && std::isfinite(c.step_inv);
	/* x / interp == mulhi(x, magic) on the range the kernels use */
	for (uint32_t x : { 0u, 1u, (uint32_t)c.interp - 1u, (uint32_t)c.interp, 1000u * (uint32_t)c.interp + 7u, 200000u })
		ok = ok && (c.interp == 1 || (uint32_t)(((uint64_t)x * c.interp_magic) >> 32) == x / (uint32_t)c.interp);
	/* exactly one geometry, or the v1 ring */
	const int geoms = (t.rw_wide ? 1 : 0) + (t.rw_mid ? 1 : 0) + (t.rw_far ? 1 : 0) + (t.rw_gather ? 1 : 0);
	ok = ok && (t.use_rw || geoms == 0) && geoms <= 1 && (generation != 0 || !t.use_rw);
	if (!ok) { fprintf(stderr, "inconsistent tables for: "); print_params(stderr, p, generation); return false; }
	const float pi_f = (float)M_PI, two_pi_f = 2.0f * pi_f;
	if (!check_schedule(c, c.jump[0], 0.0f, p.oqpsk ? pi_f : two_pi_f, r, p, generation)) return false;
	if (!check_schedule(c, c.jump[1], pi_f, two_pi_f, r, p, generation)) return false;
	if (!p.oqpsk && c.jump[1].nb != 0) return false;
	if ((p.reserved & MDEMOD_FLAG_NO_CLOCK_JUMP) && (c.jump[0].nb || c.jump[1].nb)) return false;
	return true;
}

} /* namespace */

int
main(int argc, char **argv)
{
	const uint64_t cases = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000;
	const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
	unsigned nt = argc > 3 ? (unsigned)atoi(argv[3]) : std::thread::hardware_concurrency();
	const double bound_s = argc > 4 ? atof(argv[4]) : 5.0;
	if (nt < 1) nt = 1;
	if (nt > 64) nt = 64;
	std::vector<Slot> slots(nt);
	std::atomic<uint64_t> accepted{ 0 }, refused{ 0 }, scheduled{ 0 }, sub_sample{ 0 };
	std::atomic<bool> done{ false }, failed{ false };
	std::thread watchdog([&] {
		while (!done.load()) {
			std::this_thread::sleep_for(std::chrono::milliseconds(50));
			const int64_t t = now_ms();
			for (unsigned k = 0; k < nt; k++) {
				const int64_t s = slots[k].started_ms.load();
				if (s >= 0 && (double)(t - s) > bound_s * 1e3) {
					fprintf(stderr, "mdemod_host_derive has not returned after %.1f s for: ", bound_s);
					print_params(stderr, slots[k].p, slots[k].generation);
					fflush(stderr);
					_Exit(3);
				}
			}
		}
	});
	std::vector<std::thread> th;
	for (unsigned k = 0; k < nt; k++) th.emplace_back([&, k] {
		Rng r{ seed * 0x9E3779B97F4A7C15ull + k * 0xD1B54A32D192ED03ull + 1 };
		for (uint64_t i = k; i < cases && !failed.load(); i += nt) {
			const mdemod_params p = draw(r, i);
			const int generation = r.range(0, 3) ? 2 : 0;
			slots[k].p = p; slots[k].generation = generation;
			slots[k].started_ms.store(now_ms());
			HostTables t;
			const int rc = mdemod_host_derive(p, t, generation);
			slots[k].started_ms.store(-1);
			if (rc == MDEMOD_ERR_PARAM) { refused++; continue; }
			if (rc != MDEMOD_OK) { fprintf(stderr, "return code %d for: ", rc); print_params(stderr, p, generation); failed = true; break; }
			accepted++;
			if (t.c.jump[0].nb || t.c.jump[1].nb) scheduled++;
			if ((double)p.samplerate * (p.oqpsk ? 2.0 : 1.0) < (double)p.symrate) sub_sample++;
			if (!check_tables(p, t, generation, r)) { failed = true; break; }
		}
	});
	for (auto &t : th) t.join();
	done = true;
	watchdog.join();
	printf("{\"cases\": %llu, \"accepted\": %llu, \"refused\": %llu, \"with_clock_schedule\": %llu, \"below_one_sample_per_firing\": %llu, \"ok\": %s}\n",
	       (unsigned long long)cases, (unsigned long long)accepted.load(), (unsigned long long)refused.load(),
	       (unsigned long long)scheduled.load(), (unsigned long long)sub_sample.load(), failed.load() ? "false" : "true");
	return failed.load() ? 1 : 0;
}
