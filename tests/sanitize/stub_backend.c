/*
 * stub_backend.c — a stand-in for libmeteor_demod_amd.so behind the C host (host/meteor_demod_amd.c), for the sanitizer builds of
 * tests/test_sanitize.py ONLY.  Test infrastructure: it demodulates nothing.  It implements the entries the host calls with a
 * trivially predictable "demodulator", so that the host's own logic - option parsing, the WAV header, whole-32768-byte reads, the
 * 1024-byte gated ring and its final flush (main.c:303-322), the per-device worker threads and the --tiled job threads - can run
 * under ASan / UBSan / TSan on a box without a GPU and be compared byte for byte with a model in the test.
 *
 *   stream of samples -> one "symbol" per STUB_DECIM (3) samples: the first two bytes of that sample, as they are
 *   first lock at symbol STUB_LOCK (environment, default 1000; -1: never)
 *   mdemod_create refuses what mdemod_host_derive refuses (same checks, no tables)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "meteor_demod_amd.h"

#define STUB_DECIM 3

struct mdemod_ctx {
	mdemod_params p;
	uint64_t *n_samples, *n_symbols;
	uint32_t *this_call;
	int64_t lock_at;
};

static int64_t
lock_symbol(void)
{
	const char *e = getenv("STUB_LOCK");
	return e ? atoll(e) : 1000;
}

static int
params_ok(const mdemod_params *p)
{
	if (p->interp_factor < 1 || p->interp_factor > 64 || p->rrc_order < 1 || p->rrc_order > 256) return 0;
	if (p->samplerate <= 0 || p->symrate <= 0) return 0;
	if (p->bps != 8 && p->bps != 16 && p->bps != 32) return 0;
	if ((double)p->samplerate * (p->oqpsk ? 2.0 : 1.0) < (double)p->symrate * 0.25) return 0;
	return 1;
}

uint32_t mdemod_abi_version(void) { return MDEMOD_ABI_VERSION; }
const char *mdemod_strerror(int code) { return code == MDEMOD_OK ? "ok" : (code == MDEMOD_ERR_PARAM ? "bad parameter" : "stub error"); }
const char *mdemod_last_error(void) { return getenv("STUB_LAST_ERROR") ? getenv("STUB_LAST_ERROR") : ""; }
int mdemod_init_device(int device) { (void)device; return MDEMOD_OK; }
int mdemod_device_count(void) { const char *e = getenv("STUB_DEVICES"); return e ? atoi(e) : 1; }

int
mdemod_create(const mdemod_params *params, mdemod_ctx **out)
{
	if (!params || !out || !params_ok(params) || params->n_streams == 0) return MDEMOD_ERR_PARAM;
	mdemod_ctx *c = calloc(1, sizeof(*c));
	if (!c) return MDEMOD_ERR_NOMEM;
	c->p = *params;
	c->n_samples = calloc(params->n_streams, sizeof(uint64_t));
	c->n_symbols = calloc(params->n_streams, sizeof(uint64_t));
	c->this_call = calloc(params->n_streams, sizeof(uint32_t));
	c->lock_at = lock_symbol();
	if (!c->n_samples || !c->n_symbols || !c->this_call) { mdemod_destroy(c); return MDEMOD_ERR_NOMEM; }
	*out = c;
	return MDEMOD_OK;
}

void
mdemod_destroy(mdemod_ctx *c)
{
	if (!c) return;
	free(c->n_samples); free(c->n_symbols); free(c->this_call); free(c);
}

int mdemod_pin_host_buffer(mdemod_ctx *c, const void *base, size_t bytes) { return (c && base && bytes) ? MDEMOD_OK : MDEMOD_ERR_PARAM; }
int mdemod_unpin_host_buffer(mdemod_ctx *c, const void *base) { return (c && base) ? MDEMOD_OK : MDEMOD_ERR_PARAM; }

uint64_t mdemod_max_symbols(const mdemod_ctx *ctx, uint64_t n_samples) { (void)ctx; return n_samples + 8; }

/* symbols of samples [first, first + n) of a stream whose bytes start at iq: sample j (absolute) makes one iff j % STUB_DECIM == 0 */
static uint32_t
fake_demod(const unsigned char *iq, size_t sample_bytes, uint64_t first, uint64_t n, int8_t *soft, uint64_t cap, int *overflow)
{
	uint32_t made = 0;
	for (uint64_t k = 0; k < n; k++) {
		if ((first + k) % STUB_DECIM) continue;
		if (made >= cap) { *overflow = 1; break; }
		soft[2 * (size_t)made] = (int8_t)iq[k * sample_bytes];
		soft[2 * (size_t)made + 1] = (int8_t)iq[k * sample_bytes + 1];
		made++;
	}
	return made;
}

int
mdemod_process_host(mdemod_ctx *c, const void *const *iq_host, const uint32_t *n_samples, int8_t *const *soft_host,
                    const uint32_t *soft_cap_symbols, uint32_t *n_symbols)
{
	if (!c || !iq_host || !n_samples || !soft_host || !soft_cap_symbols) return MDEMOD_ERR_PARAM;
	const size_t sb = 2 * (size_t)c->p.bps / 8;
	int overflow = 0;
	for (uint32_t s = 0; s < c->p.n_streams; s++) {
		const uint32_t made = fake_demod(iq_host[s], sb, c->n_samples[s], n_samples[s], soft_host[s], soft_cap_symbols[s], &overflow);
		c->n_samples[s] += n_samples[s];
		c->n_symbols[s] += made;
		c->this_call[s] = made;
		if (n_symbols) n_symbols[s] = made;
	}
	return overflow ? MDEMOD_ERR_OVERFLOW : MDEMOD_OK;
}

int
mdemod_get_status(mdemod_ctx *c, uint32_t first, uint32_t count, mdemod_status *out, void *hip_stream)
{
	(void)hip_stream;
	if (!c || !out) return MDEMOD_ERR_PARAM;
	if ((uint64_t)first + count > c->p.n_streams) return MDEMOD_ERR_RANGE;
	for (uint32_t i = 0; i < count; i++) {
		const uint32_t s = first + i;
		memset(&out[i], 0, sizeof(out[i]));
		out[i].n_samples = c->n_samples[s];
		out[i].n_symbols = c->n_symbols[s];
		const int locked = c->lock_at >= 0 && c->n_symbols[s] > (uint64_t)c->lock_at;
		out[i].first_lock_symbol = locked ? c->lock_at : -1;
		out[i].symbols_this_call = c->this_call[s];
		out[i].locked = out[i].locked_once = locked;
		out[i].pll_freq = 0.01f; out[i].omega = 0.39f; out[i].gain = 0.03f;
	}
	return MDEMOD_OK;
}

void
mdemod_recording_default_opts(mdemod_recording_opts *o)
{
	memset(o, 0, sizeof(*o));
	o->tile_samples = 1u << 20;
	o->pilot_margin_symbols = 15000;
	o->carrier_seed = 1;
}

int
mdemod_demodulate_recording_host(const mdemod_params *params, const mdemod_recording_opts *opts, const void *iq_host, uint64_t n_samples,
                                 int8_t *soft_host, uint64_t soft_cap_symbols, mdemod_recording_report *report)
{
	(void)opts;
	if (!params || !iq_host || !soft_host || !report || !params_ok(params)) return MDEMOD_ERR_PARAM;
	int overflow = 0;
	memset(report, 0, sizeof(*report));
	/* the real entry's capacity rule is the nominal symbol count plus slack; the stub's decimation is not the symbol rate, so it
	 * only reports what fits (the host sizes its buffer from -r / -s) */
	const uint64_t made = fake_demod(iq_host, 2 * (size_t)params->bps / 8, 0, n_samples, soft_host, soft_cap_symbols > 0xFFFFFFFFull ? 0xFFFFFFFFull : soft_cap_symbols, &overflow);
	report->n_symbols = made;
	const int64_t lock_at = lock_symbol();
	report->first_lock_symbol = (lock_at >= 0 && made > (uint64_t)lock_at) ? lock_at : -1;
	report->pilot_locked = report->first_lock_symbol >= 0;
	report->pilot_samples = n_samples;
	return MDEMOD_OK;
}
