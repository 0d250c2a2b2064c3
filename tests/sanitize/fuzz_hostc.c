/*
 * fuzz_hostc.c — the C host's file-level helpers (host/meteor_demod_amd.c) under ASan + UBSan with random input.  The host source is
 * compiled in as it is (its main renamed) and linked against tests/sanitize/stub_backend.c.  Test infrastructure.
 *
 *   write_gated     random block partitions and first-lock positions against main.c:305-315 written one symbol at a time
 *   parse_wav       random and almost-valid 44-byte headers (fmemopen), short files
 *   human_number    digits, dots, suffixes, junk (values kept where (int) of the product is defined: the reference's own cast)
 *   parse_devices   lists with junk, overlong lists
 */
#define main cli_main
#include "../../host/meteor_demod_amd.c"
#undef main

static uint64_t rng_state = 0x243F6A8885A308D3ull;
static uint64_t rnd(void) { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; uint64_t x = rng_state; x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; return x; }
static unsigned rnd_below(unsigned n) { return (unsigned)(rnd() % n); }

/* main.c:305-315 as written there: one symbol at a time, the chunk goes out when it completes iff the PLL has locked once by then */
struct model { int8_t ring[2 * RINGSIZE]; unsigned idx; uint64_t symbols; unsigned char *out; size_t len, cap; };

static void
model_push(struct model *m, int8_t re, int8_t im, int64_t first_lock)
{
	m->ring[m->idx++] = re;
	m->ring[m->idx++] = im;
	m->symbols++;
	if (m->idx >= 2 * RINGSIZE) {
		m->idx = 0;
		if (first_lock >= 0 && (uint64_t)first_lock <= m->symbols - 1) {
			if (m->len + 2 * RINGSIZE > m->cap) abort();
			memcpy(m->out + m->len, m->ring, 2 * RINGSIZE);
			m->len += 2 * RINGSIZE;
		}
	}
}

static int
fuzz_write_gated(int rounds)
{
	for (int r = 0; r < rounds; r++) {
		const uint32_t total = rnd_below(6) == 0 ? rnd_below(700) : rnd_below(40000);
		int8_t *soft = malloc(2 * (size_t)total + 2);
		for (uint32_t k = 0; k < 2 * total; k++) soft[k] = (int8_t)rnd();
		int64_t lock = rnd_below(5) == 0 ? -1 : (int64_t)rnd_below(total + 600);
		if (rnd_below(7) == 0) lock = (int64_t)(rnd_below(40) * RINGSIZE) + (int64_t)rnd_below(3) - 1;      /* on and next to chunk edges */
		struct model m;
		memset(&m, 0, sizeof(m));
		m.cap = 2 * (size_t)total + 4096; m.out = malloc(m.cap);
		char *got = NULL; size_t got_len = 0;
		struct stream_io io;
		memset(&io, 0, sizeof(io));
		io.out = open_memstream(&got, &got_len);
		/* the blocks the library hands over: any sizes; first_lock is -1 until the block that contains the lock has been processed
		   (the status snapshot after that call), as in run_exact */
		uint32_t done = 0;
		while (done < total) {
			uint32_t n = rnd_below(4) == 0 ? rnd_below(5) : (rnd_below(3) == 0 ? RINGSIZE * rnd_below(6) : rnd_below(3000));
			if (n > total - done) n = total - done;
			const int64_t known = (lock >= 0 && (uint64_t)lock < (uint64_t)done + n) ? lock : -1;
			write_gated(&io, soft + 2 * (size_t)done, n, known);
			for (uint32_t k = 0; k < n; k++) model_push(&m, soft[2 * (size_t)(done + k)], soft[2 * (size_t)(done + k) + 1], known);
			done += n;
		}
		fflush(io.out);
		int bad = got_len != m.len || (m.len && memcmp(got, m.out, m.len)) || io.ring_idx != m.idx || io.symbols != m.symbols
		          || memcmp(io.ring, m.ring, sizeof(m.ring)) || io.bytes_out != m.len;
		fclose(io.out);
		free(got); free(m.out); free(soft);
		if (bad) { fprintf(stderr, "write_gated differs from the per-symbol model: total %u, lock %lld\n", total, (long long)lock); return 1; }
	}
	return 0;
}

static int
fuzz_parse_wav(int rounds)
{
	for (int r = 0; r < rounds; r++) {
		unsigned char h[64];
		const unsigned len = rnd_below(8) == 0 ? rnd_below(44) : 44 + rnd_below(20);
		for (unsigned i = 0; i < sizeof(h); i++) h[i] = (unsigned char)rnd();
		const int riff = rnd_below(4) != 0, wave = rnd_below(4) != 0;
		if (riff) memcpy(h, "RIFF", 4);
		if (wave) memcpy(h + 8, "WAVE", 4);
		if (rnd_below(3)) { h[22] = (unsigned char)(rnd_below(4) ? 2 : rnd_below(4)); h[23] = 0; }
		if (rnd_below(3)) { const unsigned char bits[] = { 0, 8, 16, 24, 32, 12, 64 }; h[34] = bits[rnd_below(7)]; h[35] = 0; }
		FILE *f = len ? fmemopen(h, len, "rb") : fmemopen(h, 1, "rb");
		if (!f) return 1;
		int sr = -7, bps = -9;
		const int rc = parse_wav(f, &sr, &bps);
		fclose(f);
		/* wavfile.c:16-48 */
		const unsigned channels = h[22] | (h[23] << 8), bits = h[34] | (h[35] << 8);
		int want_rc = 1, want_sr = -7, want_bps = -9;
		if (len >= 44 && riff && wave && channels == 2) {
			want_bps = (int)bits;
			if (bits) { want_rc = 0; want_sr = (int)(h[24] | (h[25] << 8) | (h[26] << 16) | ((unsigned)h[27] << 24)); }
		}
		if (rc != want_rc || sr != want_sr || bps != want_bps) { fprintf(stderr, "parse_wav: rc %d sr %d bps %d, expected %d %d %d\n", rc, sr, bps, want_rc, want_sr, want_bps); return 1; }
	}
	return 0;
}

static int
fuzz_numbers(int rounds)
{
	static const struct { const char *s; float v; } known[] = {
		{ "72000", 72000 }, { "72k", 72000 }, { "72K", 72000 }, { "1M", 1000000 }, { "1.024M", 1024000 }, { "0.5k", 500 }, { "2.5", 2 },
		{ "", 0 }, { "k", 0 }, { "abc", 0 }, { "12abc", 12 }, { "1.5.2k", 1500 }, { "137.1", 137 }, { "-3k", -3 }, { "1m", 1 },
	};
	for (unsigned i = 0; i < sizeof(known) / sizeof(known[0]); i++)
		if (human_number(known[i].s) != known[i].v) { fprintf(stderr, "human_number(\"%s\") = %g, expected %g\n", known[i].s, human_number(known[i].s), known[i].v); return 1; }
	for (int r = 0; r < rounds; r++) {
		char s[24];
		const unsigned n = rnd_below(10);
		static const char alphabet[] = "0123456789..kKMm-+e xz";
		unsigned digits = 0;
		for (unsigned i = 0; i < n; i++) { s[i] = alphabet[rnd_below(sizeof(alphabet) - 1)]; if (s[i] >= '0' && s[i] <= '9') digits++; }
		s[n] = 0;
		if (digits > 3) continue;                  /* keeps (int)(v * 1e6) inside int: beyond it the reference's cast is undefined too (utils.c:60-86) */
		if (strchr(s, 'e')) continue;              /* atof reads exponents */
		(void)human_number(s);
	}
	for (int r = 0; r < rounds; r++) {
		char s[40];
		unsigned n = 0;
		const unsigned items = rnd_below(70);
		for (unsigned i = 0; i < items && n + 6 < sizeof(s); i++) n += (unsigned)snprintf(s + n, sizeof(s) - n, rnd_below(9) ? "%u," : "x%u", rnd_below(300));
		s[n] = 0;
		int out[8];
		const int got = parse_devices(s, out, 8);
		if (got < 0 || got > 8) return 1;
	}
	int devs[MAX_DEVICES];
	if (parse_devices("0,2,3", devs, MAX_DEVICES) != 3 || devs[0] != 0 || devs[1] != 2 || devs[2] != 3) return 1;
	if (parse_devices("0,,1", devs, MAX_DEVICES) != 0 || parse_devices("-1", devs, MAX_DEVICES) != 0 || parse_devices("a", devs, MAX_DEVICES) != 0) return 1;
	return 0;
}

int
main(int argc, char **argv)
{
	const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
	if (argc > 2) rng_state ^= strtoull(argv[2], NULL, 10) * 0x9E3779B97F4A7C15ull;
	(void)cli_main;
	if (fuzz_write_gated(rounds)) return 1;
	if (fuzz_parse_wav(rounds * 10)) return 1;
	if (fuzz_numbers(rounds * 10)) return 1;
	printf("fuzz_hostc: %d rounds ok\n", rounds);
	return 0;
}
