/*
 * ref_harness.c — drives the REAL reference implementation (objects compiled by
 * oracle/Makefile straight from /root/reference, nothing copied) so that the
 * restatement in lrpt_oracle.c can be pinned against it.
 *
 * TEST INFRASTRUCTURE ONLY.  Own code; it only calls the reference's public
 * entry points: demod_init/demod_qpsk/demod_oqpsk (demod.h:29-50), the getters
 * pll_get_freq/pll_get_locked/pll_did_lock_once (dsp/pll.h:20-34), mm_omega
 * (dsp/timing.h:32), agc_get_gain (dsp/agc.h:18), filter_init_rrc
 * (dsp/filter.h:25) and fast_sin/fast_cos (dsp/sincos.h:4-5).
 *
 * The reference keeps its state in file-static globals that cannot be
 * re-initialised, so one process == one stream.
 *
 * usage:
 *   ref_harness run  MODE FS SYMRATE INTERP ORDER PLL_BW FREQ_MAX FMT in.raw out.soft [out.trace]
 *   ref_harness time MODE FS SYMRATE INTERP ORDER PLL_BW FREQ_MAX FMT in.raw      (prints seconds, samples)
 *   ref_harness rrc  FS SYMRATE INTERP ORDER out.f32
 *   ref_harness sin  in.f32 out.f32          (fast_sin of each input, then fast_cos of each input)
 */
#include <complex.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "demod.h"
#include "dsp/sincos.h"

#pragma pack(push, 1)
struct trace_rec {
	uint64_t sample_index;
	float re, im, pll_freq, omega, gain;
	int32_t locked;
};
#pragma pack(pop)

static void *
slurp(const char *path, size_t *len)
{
	FILE *f = fopen(path, "rb");
	if (!f) { perror(path); exit(2); }
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	void *buf = malloc(n > 0 ? (size_t)n : 1);
	if (n > 0 && fread(buf, 1, (size_t)n, f) != (size_t)n) { perror("fread"); exit(2); }
	fclose(f);
	*len = (size_t)n;
	return buf;
}

static int8_t
quantise(float v)
{
	/* same expression shape as the reference's writer (main.c:305) */
	float h = v / 2;
	h = (127 < h) ? 127 : h;
	h = (-127 > h) ? -127 : h;
	return (int8_t)h;
}

static int
cmd_run(int argc, char **argv, int timing_only)
{
	if (argc < (timing_only ? 9 : 10)) return 1;
	const int oqpsk = !strcmp(argv[0], "oqpsk");
	const int fs = atoi(argv[1]), symrate = atoi(argv[2]);
	const int interp = atoi(argv[3]), order = atoi(argv[4]);
	const float pll_bw = (float)atof(argv[5]);
	const float freq_max = (float)atof(argv[6]);
	const int fmt = atoi(argv[7]);
	size_t nbytes;
	uint8_t *raw = slurp(argv[8], &nbytes);
	const size_t n = nbytes / (2 * (size_t)fmt / 8);

	int (*demod)(float complex *) = oqpsk ? demod_oqpsk : demod_qpsk;
	demod_init(pll_bw, SYM_BW, fs, symrate, interp, order, oqpsk, freq_max);

	FILE *fsoft = NULL, *ftrace = NULL;
	if (!timing_only) {
		fsoft = fopen(argv[9], "wb");
		if (!fsoft) { perror(argv[9]); return 2; }
		if (argc > 10) ftrace = fopen(argv[10], "wb");
	}

	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	size_t nsym = 0;
	unsigned sink = 0;
	for (size_t k = 0; k < n; k++) {
		float complex s;
		if (fmt == 8) s = ((int)raw[2*k] - 128) + I * ((int)raw[2*k+1] - 128);
		else if (fmt == 16) s = ((int16_t *)raw)[2*k] + I * ((int16_t *)raw)[2*k+1];
		else s = ((float *)raw)[2*k] + I * ((float *)raw)[2*k+1];
		if (!demod(&s)) continue;
		int8_t q[2] = { quantise(crealf(s)), quantise(cimagf(s)) };
		nsym++;
		if (timing_only) { sink += (unsigned)q[0] + (unsigned)q[1]; continue; }
		fwrite(q, 1, 2, fsoft);
		if (ftrace) {
			struct trace_rec r = { k, crealf(s), cimagf(s), pll_get_freq(), mm_omega(),
			                       agc_get_gain(), pll_get_locked() };
			fwrite(&r, sizeof(r), 1, ftrace);
		}
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	if (timing_only) {
		double dt = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
		printf("%.6f %zu %zu %u\n", dt, n, nsym, sink);
	} else {
		fclose(fsoft);
		if (ftrace) fclose(ftrace);
		fprintf(stderr, "symbols=%zu locked_once=%d\n", nsym, pll_did_lock_once());
	}
	demod_deinit();
	free(raw);
	return 0;
}

static int
cmd_rrc(int argc, char **argv)
{
	if (argc < 5) return 1;
	const int fs = atoi(argv[0]), symrate = atoi(argv[1]);
	const int interp = atoi(argv[2]), order = atoi(argv[3]);
	Filter flt;
	memset(&flt, 0, sizeof(flt));
	/* same argument expressions as demod.c:14 */
	if (filter_init_rrc(&flt, order, (float)fs / symrate, RRC_ALPHA, interp)) return 2;
	FILE *f = fopen(argv[4], "wb");
	if (!f) { perror(argv[4]); return 2; }
	fwrite(flt.coeffs, sizeof(float), (size_t)flt.size * (size_t)interp, f);
	fclose(f);
	filter_deinit(&flt);
	return 0;
}

static int
cmd_sin(int argc, char **argv)
{
	if (argc < 2) return 1;
	size_t nbytes;
	float *in = slurp(argv[0], &nbytes);
	const size_t n = nbytes / sizeof(float);
	float *out = malloc(2 * n * sizeof(float));
	for (size_t k = 0; k < n; k++) { out[k] = fast_sin(in[k]); out[n + k] = fast_cos(in[k]); }
	FILE *f = fopen(argv[1], "wb");
	if (!f) { perror(argv[1]); return 2; }
	fwrite(out, sizeof(float), 2 * n, f);
	fclose(f);
	return 0;
}

int
main(int argc, char **argv)
{
	int rc = 1;
	if (argc >= 2) {
		if (!strcmp(argv[1], "run")) rc = cmd_run(argc - 2, argv + 2, 0);
		else if (!strcmp(argv[1], "time")) rc = cmd_run(argc - 2, argv + 2, 1);
		else if (!strcmp(argv[1], "rrc")) rc = cmd_rrc(argc - 2, argv + 2);
		else if (!strcmp(argv[1], "sin")) rc = cmd_sin(argc - 2, argv + 2);
	}
	if (rc == 1) fprintf(stderr, "usage: see header of oracle/ref_harness.c\n");
	return rc;
}
