/*
 * meteor_demod_amd — C host for the MI355X LRPT demodulator.
 *
 * Same command line as the reference's CLI for the path this repository replaces
 * (main.c:19,35-51,82-152): -b -d -f -m -o -O -q -B -R -r -s -S/--bps --stdout -h -v,
 * same defaults (demod.h:8-15), k/M suffixes (utils.c:60-86), WAV header overriding
 * -s/--bps with raw fallback (main.c:164-166, wavfile.c:34-48), and the same file-level
 * behaviour: input consumed in whole 32768-byte reads (wavfile.c:6,55), 1024-byte
 * chunks written only once the PLL has locked (main.c:308-315), final flush of
 * 2*ring_idx bytes (main.c:321).
 *
 * All demodulation happens on the GPU through the C-ABI (include/meteor_demod_amd.h);
 * there is no CPU demodulator in this program.  Extensions: several input files are
 * demodulated as one batch, one stream per file (outputs <input>.s); --tiled demodulates
 * ONE file on many GPU lanes as overlapped tiles (mdemod_demodulate_recording_host: the head
 * up to PLL lock + settling is the reference's own serial run, the rest agrees with it
 * statistically, NOTEBOOK.md 3.1).  --devices a,b,... (default: every GPU of the node when there is more than one file) starts
 * one worker thread and one library context per GPU: file i goes to GPU i mod G, each worker demodulates its files as its own
 * batch (exact mode) or one after the other (--tiled) and writes its own outputs - no data crosses GPUs.
 *
 * Status line: the reference's "(%5.1f%%) Carrier: ... Symbol rate: ... Locked: ..." (main.c:249-261) from the status
 * snapshot of stream 0, at most once per -R milliseconds (default 2000 with -B, 50 without: main.c:144), "\n" separated
 * with -B and redrawn in place otherwise.  On a terminal, without -B / -q / --tiled, the full-screen display of the reference
 * (tui.c; main.c:197,224-245) is drawn instead: host/tui.c, built in when ncurses is there (as the reference's ENABLE_TUI).
 * Unlike the reference it is not started when stdin or stdout is not a terminal, unless --tui asks for it.
 * Known deviation: if the final flush would read past the 1024-byte ring (ring_idx >
 * 512, where the reference reads out of bounds) only the bytes inside the ring are
 * written.
 */
#include <getopt.h>
#include <math.h>
#include <errno.h>
#include <stdint.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "meteor_demod_amd.h"
#ifdef MDEMOD_TUI
#include "tui.h"
#endif

#define FILE_BUFFER_SIZE 32768          /* wavfile.c:6 */
#define RINGSIZE 512                    /* main.c:20   */
#define BLOCK_BUFFERS 128               /* 4 MiB of input per stream per GPU call (files) */
#define PIPE_BUFFERS  8                 /* 256 KiB when reading a pipe: ~0.3 s of a live 230 kS/s s16 stream */
#define MAX_JOBS 16                     /* --tiled: files of one GPU in flight at once, at most */

struct stream_io {
	const char *in_name;
	char       *out_name;
	FILE       *in, *out;
	int8_t      ring[2 * RINGSIZE];      /* main.c:34 (static => zero initialised) */
	unsigned    ring_idx;
	uint64_t    symbols;                 /* symbols emitted so far */
	unsigned long bytes_out;
	unsigned long file_len;              /* 0 = unknown (pipe): main.c:192 */
	int         eof;
};

static const struct option longopts[] = {
	{ "batch", 0, NULL, 'B' },      { "pll-bw", 1, NULL, 'b' },   { "freq-delta", 1, NULL, 'd' },
	{ "fir-order", 1, NULL, 'f' },  { "help", 0, NULL, 'h' },     { "mode", 1, NULL, 'm' },
	{ "output", 1, NULL, 'o' },     { "oversamp", 1, NULL, 'O' }, { "quiet", 0, NULL, 'q' },
	{ "refresh-rate", 1, NULL, 'R' }, { "symrate", 1, NULL, 'r' }, { "stdout", 0, NULL, 0x00 },
	{ "samplerate", 1, NULL, 's' }, { "bps", 1, NULL, 'S' },      { "version", 0, NULL, 'v' },
	{ "device", 1, NULL, 0x01 },    { "tiled", 0, NULL, 0x02 },   { "tile-samples", 1, NULL, 0x03 },
	{ "pilot-margin", 1, NULL, 0x04 }, { "carrier-seed", 1, NULL, 0x05 }, { "devices", 1, NULL, 0x06 }, { "plan", 0, NULL, 0x07 },
	{ "tui-selftest", 0, NULL, 0x08 }, { "tui", 0, NULL, 0x09 }, { "jobs", 1, NULL, 0x0a },
	{ NULL, 0, NULL, 0 }
};

/* utils.c:60-86: number with optional k/M suffix, truncated to int, returned as float */
static float
human_number(const char *s)
{
	const float v = (float)atof(s);
	const char *p = s;
	int out;
	while ((*p >= '0' && *p <= '9') || *p == '.') p++;
	if (*p == 'k' || *p == 'K') out = (int)(v * 1000);
	else if (*p == 'M') out = (int)(v * 1000000);
	else out = (int)v;
	return (float)out;
}

static void
usage(const char *prog)
{
	fprintf(stderr,
	        "Usage: %s [options] file_in [file_in ...]\n"
	        "   -B, --batch             No full-screen display, status lines one below the other\n"
	        "       --tui               Full-screen display even when stdin / stdout are not a terminal\n"
	        "   -b, --pll-bw <bw>       PLL bandwidth (default: 1)\n"
	        "   -d, --freq-delta <hz>   Max carrier deviation in Hz (default: +-3.5 kHz at 72 ksym/s)\n"
	        "   -f, --fir-order <ord>   RRC filter order (default: 32)\n"
	        "   -m, --mode <mode>       qpsk (default) or oqpsk\n"
	        "   -o, --output <file>     Output file (single input only; default LRPT_<date>.s)\n"
	        "   -O, --oversamp <mult>   Interpolation factor (default: 5)\n"
	        "   -q, --quiet             No status output\n"
	        "   -r, --symrate <rate>    Symbol rate (default: 72000)\n"
	        "   -s, --samplerate <rate> Sample rate of raw input\n"
	        "       --bps <bits>        Bits per sample of raw input (8, 16, 32)\n"
	        "       --stdout            Write soft symbols to stdout (implies -B -q)\n"
	        "       --device <n>        HIP device ordinal (one GPU)\n"
	        "       --devices <a,b,..>  GPUs to spread the input files over, file i on GPU i mod G (default: all\n"
	        "                           GPUs of the node when several files are given); --plan prints the assignment\n"
	        "       --tiled             Each file on many lanes as overlapped tiles (fast, not bit-exact\n"
	        "                           after the head); --tile-samples <n>, --pilot-margin <symbols>,\n"
	        "                           --carrier-seed spectrum|pilot (default spectrum: tiles follow Doppler);\n"
	        "                           --jobs <n>: files of one GPU in flight at once (default 4: their serial heads\n"
	        "                           and file reads overlap)\n"
	        "   -h, --help   -v, --version\n", prog);
}

/* wavfile.c:16-48: canonical 44-byte header, two channels */
static int
parse_wav(FILE *f, int *samplerate, int *bps)
{
	unsigned char h[44];
	if (fread(h, sizeof(h), 1, f) != 1) return 1;
	if (memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) return 1;
	const unsigned channels = h[22] | (h[23] << 8);
	const unsigned bits = h[34] | (h[35] << 8);
	if (channels != 2) return 1;
	*bps = (int)bits;                    /* wavfile.c:44: assigned before the zero test, so a 0 here overrides --bps */
	if (!bits) return 1;
	*samplerate = (int)(h[24] | (h[25] << 8) | (h[26] << 16) | ((unsigned)h[27] << 24));
	return 0;
}

/* main.c:305-315: ring of 512 symbols, a chunk is written when it completes iff the PLL has
 * locked at least once by then, i.e. first_lock <= index of the chunk's last symbol. */
static void
write_gated(struct stream_io *io, const int8_t *soft, uint32_t n, int64_t first_lock)
{
	for (uint32_t k = 0; k < n; k++) {
		/* whole chunks with the gate open (it never closes again): straight from the caller's buffer */
		if (io->ring_idx == 0 && n - k >= RINGSIZE && first_lock >= 0 && (uint64_t)first_lock <= io->symbols + RINGSIZE - 1) {
			const uint32_t chunks = (n - k) / RINGSIZE;
			fwrite(soft + 2 * (size_t)k, 2 * RINGSIZE, chunks, io->out);
			io->symbols += (uint64_t)chunks * RINGSIZE;
			io->bytes_out += 2ull * RINGSIZE * chunks;
			k += chunks * RINGSIZE;
			memcpy(io->ring, soft + 2 * (size_t)(k - RINGSIZE), 2 * RINGSIZE);     /* the ring holds the last chunk: the final flush writes stale bytes of it (main.c:321) */
			if (k >= n) break;
		}
		io->ring[io->ring_idx++] = soft[2 * k];
		io->ring[io->ring_idx++] = soft[2 * k + 1];
		io->symbols++;
		if (io->ring_idx >= 2 * RINGSIZE) {
			io->ring_idx = 0;
			if (first_lock >= 0 && (uint64_t)first_lock <= io->symbols - 1) {
				fwrite(io->ring, RINGSIZE, 2, io->out);
				io->bytes_out += 2 * RINGSIZE;
			}
		}
	}
}

/* The reference never looks at what fwrite / fclose return (main.c:314,321,274): a full disk ends in a short file and exit status 0.
 * The status stays the reference's; the loss is said, once per file, on stderr. */
static void
close_output(struct stream_io *o)
{
	if (!o->out) return;
	int failed = ferror(o->out);
	errno = 0;
	if (o->out == stdout) failed |= fflush(stdout) != 0;
	else failed |= fclose(o->out) != 0;
	const int why = errno;                                       /* (of the flush / close; an earlier fwrite's is long overwritten) */
	if (failed) fprintf(stderr, "%s: writing the soft symbols failed (%s): the output is incomplete\n", o->out_name ? o->out_name : "(stdout)",
	                    why ? strerror(why) : "a write was refused");
	o->out = NULL;
}

/* error exits: whatever was written so far is flushed and closed */
static void
close_all(struct stream_io *io, int n)
{
	for (int i = 0; i < n; i++) {
		close_output(&io[i]);
		if (io[i].in && io[i].in != stdin) fclose(io[i].in);
		io[i].in = NULL;
	}
}

static void *
init_device_thread(void *arg)        /* the HIP runtime comes up (0.1-0.2 s) while --tiled reads its file */
{
	(void)mdemod_init_device(*(int *)arg);
	return NULL;
}

static double
now_ms(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}


/* main.c:150: messages go to stdout, or into the display's log pane while that is up */
static int (*say)(const char *, ...) = printf;

#ifdef MDEMOD_TUI
/* --tui-selftest: the display's formatting helpers as text, and - on a terminal - one frame of made-up values (no GPU call) */
static int
tui_selftest(int force)
{
	int8_t sym[2 * RINGSIZE];
	unsigned lcg = 12345;
	for (int k = 0; k < RINGSIZE; k++) {
		lcg = lcg * 1664525u + 1013904223u;
		sym[2 * k] = (int8_t)(((lcg >> 8) & 1 ? 90 : -90) + (int)((lcg >> 16) % 21) - 10);
		sym[2 * k + 1] = (int8_t)(((lcg >> 9) & 1 ? 90 : -90) + (int)((lcg >> 24) % 21) - 10);
	}
	if ((force || (isatty(STDIN_FILENO) && isatty(STDOUT_FILENO))) && tui_open(50) == 0) {
		const struct tui_frame f = { 1234.5, 72000.1, 0.031, 1, 60ul << 20, 110ul << 20, 920000, 23456789ul, sym, RINGSIZE };
		tui_log("Input: %s, output: %s\n", "selftest.wav", "selftest.s");
		tui_log("Demodulator initialized\n");
		tui_draw(&f);
		tui_log("Demodulation complete\n");
		tui_log("Press any key to exit...\n");
		tui_wait_key();
		tui_close();
	}
	const unsigned long sizes[] = { 0, 999, 1000, 1001, 12345, 123456, 23456789ul, 4000000000ul };
	for (unsigned i = 0; i < sizeof(sizes) / sizeof(sizes[0]); i++) { char b[16]; tui_fmt_size(sizes[i], b); printf("size %lu -> [%sB]\n", sizes[i], b); }
	const unsigned long secs[] = { 0, 59, 3725, 356400, 360000 };
	for (unsigned i = 0; i < sizeof(secs) / sizeof(secs[0]); i++) { char b[16]; tui_fmt_clock(secs[i], b); printf("clock %lu -> %s\n", secs[i], b); }
	unsigned char hits[9 * 19];
	tui_constellation(sym, RINGSIZE, 9, 19, hits);
	for (int r = 0; r < 9; r++) { printf("plot |"); for (int c = 0; c < 19; c++) putchar(tui_glyph(hits[r * 19 + c])); printf("|\n"); }
	return 0;
}
#endif

/* What one worker (= one GPU) needs: its files and a copy of the options. */
struct worker {
	pthread_t   thr;
	int         index;                   /* worker 0 prints the status line (stream 0 of ITS batch, as the reference prints its one stream) */
	int         n_files;
	struct stream_io *io;                /* this worker's files (a contiguous copy; the originals are not touched again) */
	mdemod_params p;                     /* p.device, p.n_streams are this worker's */
	int         tiled, quiet, batch, update_interval, tile_samples, pilot_margin, carrier_seed;
	int         tui;                     /* worker 0 only: the full-screen display is up */
	int         jobs;                    /* --tiled: files of this worker in flight at once */
	int         rc;                      /* exit code of this worker: 0 ok, 1 host error, 2 library error */
};

/* the code in words plus what the library has to add (mdemod_last_error: it prints nothing itself), taken at once - the text belongs
 * to the failing call and the next call into the library may replace it */
static const char *
why_of(int rc)
{
	static __thread char buf[640];
	const char *more = mdemod_last_error();
	if (more && *more) snprintf(buf, sizeof buf, "%s: %s", mdemod_strerror(rc), more);
	else snprintf(buf, sizeof buf, "%s", mdemod_strerror(rc));
	return buf;
}

/* ---- --tiled: each file on many lanes: read it whole (32768-byte buffers only, wavfile.c:55), one library call per file.  The
 * serial head of a recording keeps one wavefront busy for ~0.1 s and its file takes as long to read: up to `jobs` files of a worker
 * are in flight at once, each on a host thread of its own (the library calls are independent: own contexts, own streams).
 * Measured on page-cached files (tools/cli_jobs_time.py, 8 x 2^25 samples, 0.45 s of it process and runtime start): 1.66 s with one
 * job, 1.25 with two, 1.13 with four (the default), 1.15 with eight - the heads and the file reads overlap, the tile phases (each
 * fills the GPU) do not. ---- */
struct tiled_pool {
	struct worker  *w;
	pthread_mutex_t lock;
	int             next;                /* next file of the worker nobody has taken */
	int             rc;                  /* worst exit code so far */
};

static int
tiled_one_file(struct worker *w, int f)
{
	struct stream_io *io = w->io;
	const int quiet = w->quiet, bps = w->p.bps, samplerate = w->p.samplerate;
	const float symrate = (float)w->p.symrate;
	const int tile_samples = w->tile_samples, pilot_margin = w->pilot_margin, carrier_seed = w->carrier_seed;
	mdemod_params p = w->p;
	const int timing = getenv("MDEMOD_CLI_TIMING") != NULL;       /* where the wall time of a --tiled run goes (stderr) */
	const double t_begin = now_ms();
	size_t cap_bytes = 1u << 26, len = 0;
	if (io[f].file_len + 2 * FILE_BUFFER_SIZE > cap_bytes) cap_bytes = io[f].file_len + 2 * FILE_BUFFER_SIZE;   /* a regular file: one allocation, no copies */
	unsigned char *data = malloc(cap_bytes);
	for (;;) {
		if (len + FILE_BUFFER_SIZE > cap_bytes) {
			unsigned char *grown = realloc(data, cap_bytes * 2);
			if (!grown) { free(data); data = NULL; break; }
			data = grown; cap_bytes *= 2;
		}
		if (!data) break;
		if (fread(data + len, FILE_BUFFER_SIZE, 1, io[f].in) != 1) break;
		len += FILE_BUFFER_SIZE;
	}
	if (!data) { fprintf(stderr, "out of memory reading %s\n", io[f].in_name); return 1; }
	const uint64_t n_samples = len / (2 * (size_t)bps / 8);
	/* (rates the library will refuse anyway must not size an allocation: a WAV header may say 0 Hz) */
	const double nominal = samplerate > 0 && symrate > 0 ? (double)n_samples * symrate / samplerate * 1.02 : -1.0;
	if (!(nominal >= 0.0 && nominal < 1e15)) {
		fprintf(stderr, "mdemod_demodulate_recording_host: %s\n", mdemod_strerror(MDEMOD_ERR_PARAM));
		free(data);
		return 2;
	}
	const uint64_t cap_sym = (uint64_t)nominal + 4096;
	int8_t *soft_all = malloc(cap_sym * 2);
	if (!soft_all) { free(data); return 1; }
	mdemod_recording_opts ro;
	mdemod_recording_default_opts(&ro);
	if (getenv("MDEMOD_RECORDING_DEBUG")) ro.debug = atoi(getenv("MDEMOD_RECORDING_DEBUG")) > 0 ? atoi(getenv("MDEMOD_RECORDING_DEBUG")) : 1;     /* the library reads no environment: the CLI does */
	if (tile_samples > 0) ro.tile_samples = (uint32_t)tile_samples;
	if (pilot_margin >= 0) ro.pilot_margin_symbols = (uint32_t)pilot_margin;
	if (carrier_seed >= 0) ro.carrier_seed = (uint32_t)carrier_seed;
	mdemod_recording_report rr;
	const double t_read = now_ms();
	int rc2 = mdemod_demodulate_recording_host(&p, &ro, data, n_samples, soft_all, cap_sym, &rr);
	const double t_lib = now_ms();
	if (rc2 != MDEMOD_OK) {
		fprintf(stderr, "mdemod_demodulate_recording_host: %s\n", why_of(rc2));
		free(data); free(soft_all);
		return 2;
	}
	if (rr.pilot_locked == 2)
		fprintf(stderr, "%s: note: the reference's PLL reports lock far from this signal's carrier (a false lock, common with "
		        "-m oqpsk): the exact mode would write what it produces from there on; --tiled demodulates the signal\n", io[f].in_name);
	if (!quiet)
		fprintf(stderr, "%s: %llu samples: %llu serial (pilot) + %u tiles, %llu symbols, first lock at symbol %lld, %u seam fixes, "
		        "%u weak seams, %u rotation jumps, %u tiles without a carrier line, %.2f s\n", io[f].in_name, (unsigned long long)n_samples,
		        (unsigned long long)rr.pilot_samples, rr.n_tiles, (unsigned long long)rr.n_symbols, (long long)rr.first_lock_symbol,
		        rr.seam_fixes, rr.weak_seams, rr.rotation_jumps, rr.weak_carrier_tiles, rr.pilot_seconds + rr.tiles_seconds);
	for (uint64_t k = 0; k < rr.n_symbols; k += 1u << 20)
		write_gated(&io[f], soft_all + 2 * k, (uint32_t)((rr.n_symbols - k < (1u << 20)) ? rr.n_symbols - k : (1u << 20)), rr.first_lock_symbol);
	size_t tail = 2 * (size_t)io[f].ring_idx;                       /* main.c:321 */
	if (tail > sizeof(io[f].ring)) tail = sizeof(io[f].ring);
	fwrite(io[f].ring, 1, tail, io[f].out);
	close_output(&io[f]);
	if (io[f].in != stdin) fclose(io[f].in);
	io[f].in = NULL;
	free(data); free(soft_all);
	if (timing)
		fprintf(stderr, "%s: read %.0f ms, library call %.0f ms (pilot %.0f + tiles %.0f on the device), write %.0f ms\n", io[f].in_name,
		        t_read - t_begin, t_lib - t_read, rr.pilot_seconds * 1e3, rr.tiles_seconds * 1e3, now_ms() - t_lib);
	return 0;
}

static void *
tiled_job(void *arg)
{
	struct tiled_pool *pool = arg;
	for (;;) {
		pthread_mutex_lock(&pool->lock);
		const int f = pool->next < pool->w->n_files && pool->rc == 0 ? pool->next++ : -1;
		pthread_mutex_unlock(&pool->lock);
		if (f < 0) return NULL;
		const int rc = tiled_one_file(pool->w, f);
		if (rc) {
			pthread_mutex_lock(&pool->lock);
			if (rc > pool->rc) pool->rc = rc;
			pthread_mutex_unlock(&pool->lock);
		}
	}
}

static int
run_tiled(struct worker *w)
{
	int device = w->p.device;
	/* the HIP runtime comes up (0.1-0.2 s) on a thread of its own while the first file is read */
	pthread_t init_thr;
	const int init_started = pthread_create(&init_thr, NULL, init_device_thread, &device) == 0;
	struct tiled_pool pool = { w, PTHREAD_MUTEX_INITIALIZER, 0, 0 };
	int jobs = w->jobs < 1 ? 1 : w->jobs;
	if (jobs > w->n_files) jobs = w->n_files;
	if (jobs > MAX_JOBS) jobs = MAX_JOBS;
	for (int i = 0; i < w->n_files; i++) if (w->io[i].in == stdin || w->io[i].out == stdout) jobs = 1;
	if (jobs == 1) {
		tiled_job(&pool);
		if (init_started) pthread_join(init_thr, NULL);
	} else {
		pthread_t thr[MAX_JOBS];
		int started = 0;
		for (; started < jobs; started++) if (pthread_create(&thr[started], NULL, tiled_job, &pool)) break;
		if (!started) tiled_job(&pool);
		for (int i = 0; i < started; i++) pthread_join(thr[i], NULL);
		if (init_started) pthread_join(init_thr, NULL);
	}
	if (pool.rc) close_all(w->io, w->n_files);
	return pool.rc;
}

/* an error path of run_exact: the full-screen display comes down first (the reference's main.c:241-244 tears it down on every exit),
 * so that the message lands on a sane terminal */
static int
exact_failed(struct worker *w, int code, const char *what, const char *why)
{
#ifdef MDEMOD_TUI
	if (w->tui) { tui_close(); say = printf; }
#endif
	fprintf(stderr, "%s: %s\n", what, why);
	close_all(w->io, w->n_files);
	return code;
}

/* ---- exact mode: this worker's files as ONE batch, one stream per file, block by block (main.c:303-316) ---- */
static int
run_exact(struct worker *w)
{
	struct stream_io *io = w->io;
	const int n_files = w->n_files, quiet = w->quiet || w->index != 0, batch = w->batch, update_interval = w->update_interval;
	const int bps = w->p.bps, samplerate = w->p.samplerate, interp = w->p.interp_factor, oqpsk = w->p.oqpsk;
	const float symrate = (float)w->p.symrate;
	mdemod_params p = w->p;
	mdemod_ctx *ctx = NULL;
	int rc = mdemod_create(&p, &ctx);
	if (rc != MDEMOD_OK) return exact_failed(w, 2, "mdemod_create", why_of(rc));
	if (!quiet) say("Demodulator initialized\n");                                    /* main.c:219 */
	if (!quiet && !w->tui && n_files < 64 && io[0].file_len > (64ul << 20))
		fprintf(stderr, "note: %d file%s demodulated exactly = %d serial stream%s, one GPU wavefront each (about 3.6 MS/s: slower than the "
		        "reference on one host core); --tiled puts a long recording on many lanes (50x faster, same symbols, soft values within "
		        "+-1 LSB of these on 99.6-99.9 %%), and a batch of many files fills the GPU in exact mode\n",
		        n_files, n_files == 1 ? "" : "s", n_files, n_files == 1 ? "" : "s");

	size_t block_buffers = BLOCK_BUFFERS;
	for (int i = 0; i < n_files; i++) if (io[i].in == stdin) block_buffers = PIPE_BUFFERS;     /* live input: short blocks */
	const size_t block_bytes = block_buffers * FILE_BUFFER_SIZE;
	const uint32_t block_samples = (uint32_t)(block_bytes / (2 * (size_t)bps / 8));
	const uint32_t cap = (uint32_t)mdemod_max_symbols(ctx, block_samples);
	unsigned char *in_buf = malloc(block_bytes * (size_t)n_files);
	int8_t *soft = malloc((size_t)cap * 2 * (size_t)n_files);
	const void **iq = malloc(sizeof(*iq) * (size_t)n_files);
	int8_t **outp = malloc(sizeof(*outp) * (size_t)n_files);
	uint32_t *n_in = malloc(sizeof(uint32_t) * (size_t)n_files), *caps = malloc(sizeof(uint32_t) * (size_t)n_files);
	uint32_t *n_out = malloc(sizeof(uint32_t) * (size_t)n_files);
	mdemod_status *st = malloc(sizeof(*st) * (size_t)n_files);
#define FREE_BLOCKS() do { free(in_buf); free(soft); free(iq); free(outp); free(n_in); free(caps); free(n_out); free(st); } while (0)
	if (!in_buf || !soft || !iq || !outp || !n_in || !caps || !n_out || !st) { mdemod_destroy(ctx); FREE_BLOCKS(); return exact_failed(w, 1, "meteor_demod_amd", "out of memory"); }

	/* the read buffer is the same for every block: pinned once, the batch then goes to the GPU from where fread put it
	   (a batch of files; one file is a few MiB per call either way.  A refusal - no memory to pin - only means the library stages the
	   blocks itself.  The context is destroyed, which unpins, BEFORE the buffer is freed on every way out.) */
	if (n_files >= 2 && block_buffers == BLOCK_BUFFERS) (void)mdemod_pin_host_buffer(ctx, in_buf, block_bytes * (size_t)n_files);

	double last_status = -1e18;
#ifdef MDEMOD_TUI
	int8_t shown[2 * RINGSIZE];                 /* the latest symbols of stream 0 for the constellation (main.c:238 shows its ring) */
	unsigned n_shown = 0;
#endif
	for (;;) {
		int active = 0;
		for (int i = 0; i < n_files; i++) {
			iq[i] = in_buf + block_bytes * (size_t)i;
			outp[i] = soft + (size_t)cap * 2 * (size_t)i;
			caps[i] = cap;
			n_in[i] = 0;
			if (io[i].eof) continue;
			/* whole 32768-byte buffers only: a short trailing read ends the stream (wavfile.c:55) */
			const size_t got = fread(in_buf + block_bytes * (size_t)i, FILE_BUFFER_SIZE, block_buffers, io[i].in);
			if (got < block_buffers) io[i].eof = 1;
			n_in[i] = (uint32_t)(got * FILE_BUFFER_SIZE / (2 * (size_t)bps / 8));
			if (got) active = 1;
		}
		if (!active) break;
		rc = mdemod_process_host(ctx, iq, n_in, outp, caps, n_out);          /* demod(&sample) x n: main.c:304 */
		if (rc != MDEMOD_OK) { const char *why = why_of(rc); mdemod_destroy(ctx); FREE_BLOCKS(); return exact_failed(w, 2, "mdemod_process_host", why); }
		rc = mdemod_get_status(ctx, 0, (uint32_t)n_files, st, NULL);
		if (rc != MDEMOD_OK) { const char *why = why_of(rc); mdemod_destroy(ctx); FREE_BLOCKS(); return exact_failed(w, 2, "mdemod_get_status", why); }
		for (int i = 0; i < n_files; i++)
			write_gated(&io[i], outp[i], n_out[i], st[i].first_lock_symbol);
#ifdef MDEMOD_TUI
		if (w->tui && n_out[0]) {
			n_shown = n_out[0] < RINGSIZE ? n_out[0] : RINGSIZE;
			memcpy(shown, outp[0] + 2 * (size_t)(n_out[0] - n_shown), 2 * (size_t)n_shown);
		}
#endif
		if (!quiet && now_ms() - last_status >= update_interval) {
			/* main.c:249-261: status line from the snapshot of stream 0, at most once per refresh period */
			last_status = now_ms();
			const double freq_hz = st[0].pll_freq * symrate / (2 * M_PI) * (oqpsk ? 2 : 1);
			const double rate_hz = st[0].omega * ((double)samplerate * interp) / (2 * M_PI);
			const long pos = io[0].in != stdin ? ftell(io[0].in) : 0;
#ifdef MDEMOD_TUI
			if (w->tui) {
				/* main.c:224-239: the display instead of the line; q ends the run after this block (the reference's `done = 1`) */
				const struct tui_frame f = { freq_hz, rate_hz, st[0].gain, st[0].locked, pos > 0 ? (unsigned long)pos : 0, io[0].file_len,
				                             (unsigned)(2 * (size_t)samplerate * (size_t)bps / 8), io[0].bytes_out, shown, n_shown };
				if (tui_draw(&f)) for (int i = 0; i < n_files; i++) io[i].eof = 1;
				continue;
			}
#endif
			printf(batch ? "\n" : "\033[1K\r");
			printf("(%5.1f%%) Carrier: %+7.1f Hz, Symbol rate: %.1f Hz, Locked: %s",
			       io[0].file_len && pos > 0 ? 100.0 * (double)pos / (double)io[0].file_len : 0.0, freq_hz, rate_hz, st[0].locked ? "Yes" : "No");
			fflush(stdout);
		}
	}
	if (!quiet && !w->tui) printf("\n");

	for (int i = 0; i < n_files; i++) {
		/* main.c:321: fwrite(ring, ring_idx, 2, f) */
		size_t tail = 2 * (size_t)io[i].ring_idx;
		if (tail > sizeof(io[i].ring)) tail = sizeof(io[i].ring);
		fwrite(io[i].ring, 1, tail, io[i].out);
		io[i].bytes_out += io[i].ring_idx;
		close_output(&io[i]);
		if (io[i].in != stdin) fclose(io[i].in);
		io[i].in = NULL;
	}
	mdemod_destroy(ctx);                                                          /* demod_deinit: main.c:273 */
	FREE_BLOCKS();
#undef FREE_BLOCKS
#ifdef MDEMOD_TUI
	if (w->tui) {                                                                 /* main.c:241-244 */
		say("Demodulation complete\n");
		say("Press any key to exit...\n");
		tui_wait_key();
		tui_close();
		say = printf;
	}
#endif
	return 0;
}

static void *
worker_main(void *arg)
{
	struct worker *w = arg;
	w->rc = w->tiled ? run_tiled(w) : run_exact(w);
	return NULL;
}

/* "0,2,3" -> device ordinals; returns the count (0 on a malformed list) */
static int
parse_devices(const char *s, int *out, int cap)
{
	int n = 0;
	while (*s && n < cap) {
		char *end;
		const long v = strtol(s, &end, 10);
		if (end == s || v < 0) return 0;
		out[n++] = (int)v;
		if (*end == ',') end++;
		else if (*end) return 0;
		s = end;
	}
	return n;
}

#define MAX_DEVICES 64

int
main(int argc, char **argv)
{
	float pll_bw = MDEMOD_DEFAULT_PLL_BW, symrate = MDEMOD_DEFAULT_SYM_RATE, freq_max_delta = -1;
	int rrc_order = MDEMOD_DEFAULT_RRC_ORDER, interp = MDEMOD_DEFAULT_INTERP;
	int quiet = 0, batch = 0, oqpsk = 0, bps = 0, samplerate = -1, stdout_mode = 0, device = 0, tiled = 0;
	int tile_samples = 0, pilot_margin = -1, carrier_seed = -1, update_interval = -1;
	const char *output_fname = NULL;
	int devs[MAX_DEVICES], n_dev = 0, plan = 0, jobs = 4;
#ifdef MDEMOD_TUI
	int force_tui = 0;
#endif
	int c;

	while ((c = getopt_long(argc, argv, "a:Bb:d:f:hm:o:O:qR:r:s:S:v", longopts, NULL)) != -1) {
		switch (c) {
		case 0x00: stdout_mode = 1; break;
		case 0x01: device = atoi(optarg); devs[0] = device; n_dev = 1; break;
		case 0x06:
			n_dev = parse_devices(optarg, devs, MAX_DEVICES);
			if (!n_dev) { fprintf(stderr, "--devices: a comma separated list of GPU ordinals\n"); return 1; }
			break;
		case 0x07: plan = 1; break;
#ifdef MDEMOD_TUI
		case 0x09: force_tui = 1; break;
#else
		case 0x09: break;                             /* --tui in a build without the display: accepted, nothing to draw */
#endif
		case 0x0a: jobs = atoi(optarg); if (jobs < 1) { fprintf(stderr, "--jobs: a positive number\n"); return 1; } break;
		case 0x08:
#ifdef MDEMOD_TUI
			return tui_selftest(force_tui);
#else
			fprintf(stderr, "built without ncurses\n"); return 1;
#endif
		case 0x02: tiled = 1; break;
		case 0x03: tile_samples = (int)human_number(optarg); break;
		case 0x04: pilot_margin = (int)human_number(optarg); break;
		case 0x05:
			if (!strcmp(optarg, "spectrum")) carrier_seed = 1;
			else if (!strcmp(optarg, "pilot")) carrier_seed = 0;
			else { fprintf(stderr, "--carrier-seed: spectrum or pilot\n"); return 1; }
			break;
		case 'b': pll_bw = human_number(optarg); break;
		case 'B': batch = 1; break;
		case 'd': freq_max_delta = human_number(optarg); break;
		case 'f': rrc_order = atoi(optarg); break;
		case 'h': usage(argv[0]); return 0;
		case 'm': if (!strcmp(optarg, "oqpsk")) oqpsk = 1; break;     /* unknown modes stay QPSK, main.c:104 */
		case 'o': output_fname = optarg; break;
		case 'O': interp = atoi(optarg); break;
		case 'q': quiet = 1; break;
		case 'R': update_interval = atoi(optarg); break;              /* main.c:116 */
		case 'r': symrate = human_number(optarg); break;
		case 's': samplerate = (int)human_number(optarg); break;
		case 'S': bps = atoi(optarg); break;
		case 'v': printf("meteor_demod_amd (MI355X) ABI %u\n", mdemod_abi_version()); return 0;
		default: usage(argv[0]); return 1;
		}
	}
	freq_max_delta = (float)(freq_max_delta * (2 * M_PI) / symrate);       /* main.c:136 */
	if (argc - optind < 1) { usage(argv[0]); return 1; }
	if (update_interval < 0) update_interval = batch ? 2000 : 50;                 /* main.c:144 (before batch is forced below) */
	if (stdout_mode) { batch = 1; quiet = 1; }
	for (int i = optind; i < argc; i++) if (!strcmp(argv[i], "-")) batch = 1;     /* stdin forces batch: main.c:157 */

	const int n_files = argc - optind;
	if (n_files > 1 && (output_fname || stdout_mode)) {
		fprintf(stderr, "-o/--stdout need a single input file\n");
		return 1;
	}
	if (plan) {
		/* the sharding arithmetic, without touching files or GPUs: file i on the (i mod G)-th device of the list */
		if (!n_dev) { fprintf(stderr, "--plan needs --devices\n"); return 1; }
		const int g = n_dev > n_files ? n_files : n_dev;
		for (int d = 0; d < g; d++) {
			printf("device %d:", devs[d]);
			for (int i = d; i < n_files; i += g) printf(" %s", argv[optind + i]);
			printf("\n");
		}
		return 0;
	}
	struct stream_io *io = calloc((size_t)n_files, sizeof(*io));
	if (!io) return 1;
	struct worker *ws = NULL;
	int n_workers = 0;
	/* every way out from here on: files closed, nothing left allocated (the sanitizer builds of tests/test_sanitize.py look) */
#define LEAVE(code) do { \
		close_all(io, n_files); \
		for (int d_ = 0; d_ < n_workers; d_++) { free(ws[d_].io); } \
		free(ws); \
		for (int i_ = 0; i_ < n_files; i_++) { free(io[i_].out_name); } \
		free(io); \
		return (code); \
	} while (0)

	for (int i = 0; i < n_files; i++) {
		io[i].in_name = argv[optind + i];
		io[i].in = !strcmp(io[i].in_name, "-") ? stdin : fopen(io[i].in_name, "rb");
		if (!io[i].in) { fprintf(stderr, "Could not open input file\n"); LEAVE(1); }
		int sr = samplerate, b = bps;
		if (parse_wav(io[i].in, &sr, &b)) fseek(io[i].in, 0, SEEK_SET);    /* raw: main.c:164-166 */
		if (i == 0) { samplerate = sr; bps = b; }
		else if (sr != samplerate || b != bps) { fprintf(stderr, "all inputs of a batch must share rate and format\n"); LEAVE(1); }
	}
	if (samplerate < 0) {
		fprintf(stderr, "Could not auto-detect sample rate. Please specify it with -s <samplerate>\n");
		usage(argv[0]);                                                                /* main.c:170 */
		LEAVE(1);
	}
	if (!bps) { fprintf(stderr, "Could not auto-detect bits per sample, assuming 16\n"); bps = 16; }
	/* any other sample size: the reference's reader returns 0 on the first sample (wavfile.c:71-73) and it writes an empty
	 * output file; same here, without touching the GPU */
	const int bps_ok = (bps == 8 || bps == 16 || bps == 32);
	if (!bps_ok) fprintf(stderr, "%d bits per sample: nothing to demodulate (8, 16 or 32 expected)\n", bps);

	for (int i = 0; i < n_files; i++) {
		if (stdout_mode) { io[i].out = stdout; continue; }
		if (n_files == 1 && output_fname) io[i].out_name = strdup(output_fname);
		else if (n_files == 1) {                                           /* utils.c:8: LRPT_%Y_%m_%d-%H_%M.s */
			char buf[64]; time_t t = time(NULL);
			strftime(buf, sizeof(buf), "LRPT_%Y_%m_%d-%H_%M.s", localtime(&t));
			io[i].out_name = strdup(buf);
		} else {
			io[i].out_name = malloc(strlen(io[i].in_name) + 3);
			sprintf(io[i].out_name, "%s.s", io[i].in_name);
		}
		io[i].out = fopen(io[i].out_name, "wb");
		if (!io[i].out) { fprintf(stderr, "Could not open output file\n"); LEAVE(1); }
	}

	int use_tui = 0;
#ifdef MDEMOD_TUI
	/* main.c:197: the display unless -B (or a mode that forces it); here also only when both ends are a terminal, or with --tui
	   (the reference draws into whatever stdout is) */
	if (!batch && !quiet && !tiled && bps_ok && (force_tui || (isatty(STDIN_FILENO) && isatty(STDOUT_FILENO))) && tui_open(update_interval) == 0) {
		use_tui = 1;
		say = tui_log;
	}
#endif
	if (!quiet)                                                                        /* main.c:200 */
		for (int i = 0; i < n_files; i++) say("Input: %s, output: %s\n", io[i].in_name, stdout_mode ? "(stdout)" : io[i].out_name);
	if (!bps_ok) {
		LEAVE(0);
	}
	/* file lengths for the progress figure of the status line (main.c:189-193) */
	for (int i = 0; i < n_files; i++) {
		if (io[i].in == stdin) continue;
		const long here = ftell(io[i].in);
		if (here < 0 || fseek(io[i].in, 0, SEEK_END)) continue;
		const long end = ftell(io[i].in);
		io[i].file_len = end > 0 ? (unsigned long)end : 0;
		fseek(io[i].in, here, SEEK_SET);
	}
	/* demod_init(pll_bw, SYM_BW, samplerate, symrate, interp, order, oqpsk, freq_max): main.c:187 */
	mdemod_params p;
	memset(&p, 0, sizeof(p));
	p.pll_bw = pll_bw; p.sym_bw = MDEMOD_DEFAULT_SYM_BW; p.samplerate = samplerate; p.symrate = (int)symrate;
	p.interp_factor = interp; p.rrc_order = rrc_order; p.oqpsk = oqpsk; p.freq_max = freq_max_delta;
	p.bps = bps; p.device = device; p.n_streams = (uint32_t)n_files;
	/* ---- one worker per GPU: file i on GPU i mod G (streams shard, nothing crosses GPUs) ---- */
	if (n_dev == 0) {
		/* (one file: one GPU, and no question to the HIP runtime before its input is being read - the runtime takes 50-100 ms to
		   come up, which --tiled hides behind the file read) */
		const int have = n_files < 2 ? 1 : mdemod_device_count();
		if (n_files < 2 || have < 2) { devs[0] = device; n_dev = 1; }
		else for (n_dev = 0; n_dev < have && n_dev < MAX_DEVICES; n_dev++) devs[n_dev] = n_dev;
	}
	if (n_dev > n_files) n_dev = n_files;
	ws = calloc((size_t)n_dev, sizeof(*ws));
	if (!ws) LEAVE(1);
	n_workers = n_dev;
	for (int d = 0; d < n_dev; d++) {
		struct worker *w = &ws[d];
		w->index = d; w->p = p; w->p.device = devs[d];
		w->tiled = tiled; w->quiet = quiet; w->batch = batch; w->update_interval = update_interval;
		w->tile_samples = tile_samples; w->pilot_margin = pilot_margin; w->carrier_seed = carrier_seed;
		w->tui = use_tui && d == 0;
		w->jobs = jobs;
		for (int i = d; i < n_files; i += n_dev) w->n_files++;
		w->io = calloc((size_t)w->n_files, sizeof(*w->io));
		if (!w->io) LEAVE(1);
		for (int i = d, k = 0; i < n_files; i += n_dev, k++) w->io[k] = io[i];
		w->p.n_streams = (uint32_t)w->n_files;
	}
	if (n_dev == 1) {
		worker_main(&ws[0]);
	} else {
		for (int d = 0; d < n_dev; d++)
			if (pthread_create(&ws[d].thr, NULL, worker_main, &ws[d])) { fprintf(stderr, "could not start the worker of device %d\n", devs[d]); for (int k = 0; k < d; k++) pthread_join(ws[k].thr, NULL); for (int i = 0; i < n_files; i++) { io[i].in = NULL; io[i].out = NULL; } LEAVE(1); }
		for (int d = 0; d < n_dev; d++) pthread_join(ws[d].thr, NULL);
	}
	int rc_all = 0;
	for (int d = 0; d < n_dev; d++) if (ws[d].rc > rc_all) rc_all = ws[d].rc;
	/* (the workers closed their files through their own copies of the stream_io entries: nothing of the originals is open any more) */
	for (int i = 0; i < n_files; i++) { io[i].in = NULL; io[i].out = NULL; }
	LEAVE(rc_all);
}
