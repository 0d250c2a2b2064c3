/*
 * host_pipe.cpp — the host-buffer entry (mdemod_process_host) as a three-stage pipeline.
 *
 * The reference reads its input with fread into a 32 KiB buffer and converts sample by
 * sample (wavfile.c:55-69); a GPU fed from host memory is bound by PCIe, so the work is to
 * keep the link busy: each stream's block is cut into K consecutive sub-blocks (chained
 * calls are exact, the state lives in the context), and sub-block k+1 is packed into pinned
 * memory and copied in while sub-block k is demodulated and sub-block k-1 is copied out and
 * handed back to the caller's buffers:
 *
 *     CPU pack(k+1) | H2D(k+1)  [stream in]  |  kernel(k) [stream cmp]  |  D2H(k-1) [stream out] | CPU unpack(k-1)
 *
 * (the host unpacks k-1 after it has enqueued kernel k and its copies, so the GPU never waits for the CPU; every sub-block's
 * lock events are copied aside on the compute stream because the next launch overwrites the context's list).
 * Two sets of pinned + device staging buffers (grow only), three HIP streams, events for the
 * four hand-offs.  Packing and unpacking are spread over a few host threads.
 * After the last sub-block the per-call counters of the context (symbols / lock events of "this
 * call") are rewritten with the totals over all sub-blocks, so the status snapshot means what the
 * header says.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <unistd.h>

#include "demod_internal.h"
#include "pack_pool.h"

namespace {

#define PIPE_TRY(expr)                                                                          \
	do {                                                                                        \
		hipError_t e_ = (expr);                                                                 \
		if (e_ != hipSuccess) {                                                                 \
			fprintf(stderr, "meteor_demod_amd: %s failed: %s (%s:%d)\n", #expr,                 \
			        hipGetErrorString(e_), __FILE__, __LINE__);                                 \
			return e_ == hipErrorOutOfMemory ? MDEMOD_ERR_NOMEM : MDEMOD_ERR_HIP;               \
		}                                                                                       \
	} while (0)

struct Slot {                    /* one of the two staging sets */
	unsigned char *h_iq = nullptr, *d_iq = nullptr;   size_t iq_bytes = 0;
	int8_t *h_soft = nullptr;  size_t h_soft_bytes = 0;
	int8_t *d_soft = nullptr;  size_t d_soft_bytes = 0;          /* kernel output, hard-bound pitch              */
	int8_t *d_pack = nullptr;  size_t d_pack_bytes = 0;          /* the same rows at the nominal pitch: what is copied out */
	uint32_t pitch = 0;          /* nominal pitch (symbols) of d_pack / h_soft for the sub-block in flight */
	uint64_t *h_off = nullptr, *d_off = nullptr;
	uint32_t *h_cnt = nullptr, *d_cnt = nullptr, *h_prod = nullptr, *h_ev = nullptr;
	mdemod_lock_event *d_events = nullptr;   /* this sub-block's lock events: the next launch overwrites the context's list */
	hipEvent_t ev_in = nullptr, ev_k = nullptr, ev_out = nullptr;
	bool used_in = false, used_k = false, used_out = false;
	uint32_t cap = 0;            /* symbol stride of the DEVICE soft buffer for the sub-block in flight: one symbol per input
	                              * sample is the hard bound (a full-scale transient can exceed mdemod_max_symbols) */
	uint32_t width = 0;          /* symbols per stream actually copied out = the largest count of the sub-block */
};

struct HostPipe {
	Slot slot[2];
	hipStream_t s_in = nullptr, s_cmp = nullptr, s_out = nullptr;
	uint32_t ns = 0;
	bool ready = false;
};

template <typename T>
int
grow_pair(T **host, T **dev, size_t *have, size_t need)
{
	if (need <= *have) return MDEMOD_OK;
	if (*host) (void)hipHostFree(*host);
	if (*dev) (void)hipFree(*dev);
	*host = nullptr; *dev = nullptr; *have = 0;
	need += need / 8;                                        /* a little headroom: fewer re-allocations */
	PIPE_TRY(hipHostMalloc(reinterpret_cast<void **>(host), need, hipHostMallocDefault));
	PIPE_TRY(hipMalloc(reinterpret_cast<void **>(dev), need));
	*have = need;
	return MDEMOD_OK;
}

int
grow_host(int8_t **host, size_t *have, size_t need)
{
	if (need <= *have) return MDEMOD_OK;
	if (*host) (void)hipHostFree(*host);
	*host = nullptr; *have = 0;
	need += need / 8;
	PIPE_TRY(hipHostMalloc(reinterpret_cast<void **>(host), need, hipHostMallocDefault));
	*have = need;
	return MDEMOD_OK;
}

int
grow_dev(int8_t **dev, size_t *have, size_t need)
{
	if (need <= *have) return MDEMOD_OK;
	if (*dev) (void)hipFree(*dev);
	*dev = nullptr; *have = 0;
	need += need / 8;
	PIPE_TRY(hipMalloc(reinterpret_cast<void **>(dev), need));
	*have = need;
	return MDEMOD_OK;
}

int
pipe_init(HostPipe *p, uint32_t ns)
{
	if (p->ready) return MDEMOD_OK;
	/* every resource is created only if it is missing: a call that failed part-way is picked up where it stopped, nothing leaks */
	p->ns = ns;
	if (!p->s_in) PIPE_TRY(hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking));
	if (!p->s_cmp) PIPE_TRY(hipStreamCreateWithFlags(&p->s_cmp, hipStreamNonBlocking));
	if (!p->s_out) PIPE_TRY(hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
	for (Slot &s : p->slot) {
		if (!s.ev_in) PIPE_TRY(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
		if (!s.ev_k) PIPE_TRY(hipEventCreateWithFlags(&s.ev_k, hipEventDisableTiming));
		if (!s.ev_out) PIPE_TRY(hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming));
		if (!s.h_off) PIPE_TRY(hipHostMalloc(reinterpret_cast<void **>(&s.h_off), sizeof(uint64_t) * ns, hipHostMallocDefault));
		if (!s.h_cnt) PIPE_TRY(hipHostMalloc(reinterpret_cast<void **>(&s.h_cnt), sizeof(uint32_t) * ns, hipHostMallocDefault));
		if (!s.h_prod) PIPE_TRY(hipHostMalloc(reinterpret_cast<void **>(&s.h_prod), sizeof(uint32_t) * ns, hipHostMallocDefault));
		if (!s.h_ev) PIPE_TRY(hipHostMalloc(reinterpret_cast<void **>(&s.h_ev), sizeof(uint32_t) * ns, hipHostMallocDefault));
		if (!s.d_off) PIPE_TRY(hipMalloc(reinterpret_cast<void **>(&s.d_off), sizeof(uint64_t) * ns));
		if (!s.d_cnt) PIPE_TRY(hipMalloc(reinterpret_cast<void **>(&s.d_cnt), sizeof(uint32_t) * ns));
		if (!s.d_events) PIPE_TRY(hipMalloc(reinterpret_cast<void **>(&s.d_events), sizeof(mdemod_lock_event) * MDEMOD_MAX_LOCK_EVENTS * ns));
	}
	p->ready = true;
	return MDEMOD_OK;
}

} /* namespace */

void
mdemod_hostpipe_free(void *opaque)
{
	HostPipe *p = static_cast<HostPipe *>(opaque);
	if (!p) return;
	for (Slot &s : p->slot) {
		if (s.h_iq) (void)hipHostFree(s.h_iq);
		if (s.d_iq) (void)hipFree(s.d_iq);
		if (s.h_soft) (void)hipHostFree(s.h_soft);
		if (s.d_soft) (void)hipFree(s.d_soft);
		if (s.d_pack) (void)hipFree(s.d_pack);
		if (s.h_off) (void)hipHostFree(s.h_off);
		if (s.h_cnt) (void)hipHostFree(s.h_cnt);
		if (s.h_prod) (void)hipHostFree(s.h_prod);
		if (s.h_ev) (void)hipHostFree(s.h_ev);
		if (s.d_off) (void)hipFree(s.d_off);
		if (s.d_cnt) (void)hipFree(s.d_cnt);
		if (s.d_events) (void)hipFree(s.d_events);
		if (s.ev_in) (void)hipEventDestroy(s.ev_in);
		if (s.ev_k) (void)hipEventDestroy(s.ev_k);
		if (s.ev_out) (void)hipEventDestroy(s.ev_out);
	}
	if (p->s_in) (void)hipStreamDestroy(p->s_in);
	if (p->s_cmp) (void)hipStreamDestroy(p->s_cmp);
	if (p->s_out) (void)hipStreamDestroy(p->s_out);
	delete p;
}

int
mdemod_hostpipe_run(mdemod_ctx *ctx, void **pipe_slot, const DemodStateSoA &st, uint32_t ns, size_t sb,
                    const void *const *iq_host, const uint32_t *n_samples,
                    int8_t *const *soft_host, const uint32_t *soft_cap, uint32_t *n_symbols)
{
	if (!*pipe_slot) *pipe_slot = new HostPipe();
	HostPipe *p = static_cast<HostPipe *>(*pipe_slot);
	int rc = pipe_init(p, ns);
	if (rc) return rc;

	uint64_t total = 0; uint32_t n_max = 0;
	for (uint32_t s = 0; s < ns; s++) {
		if (n_samples[s] > 0x3FFFFF00u) return MDEMOD_ERR_PARAM;
		total += n_samples[s];
		n_max = std::max(n_max, n_samples[s]);
	}
	/* sub-blocks of ~32 MiB of input, at most 16, at least 4096 samples of the longest stream each */
	uint32_t K = static_cast<uint32_t>(std::min<uint64_t>(16, std::max<uint64_t>(1, total * sb / (32u << 20))));
	while (K > 1 && n_max / K < 4096) K--;

	std::vector<uint32_t> produced(ns, 0), events(ns, 0);
	std::vector<mdemod_lock_event> ev_store;                  /* merged lock events: [stream][32] (only if any) */
	int result = MDEMOD_OK;
	auto sub_lo = [&](uint32_t s, uint32_t k) { return static_cast<uint32_t>(static_cast<uint64_t>(n_samples[s]) * k / K); };

	auto unpack = [&](Slot &sl) -> int {
		PIPE_TRY(hipEventSynchronize(sl.ev_out));
		/* common case: every row fits the nominal pitch and h_soft holds it.  A stream that fired on (almost) every sample
		 * (full-scale transient) exceeds it: then the rows are fetched again, 2-D, from the hard-pitch buffer. */
		uint32_t widest = 0;
		for (uint32_t s = 0; s < ns; s++) {
			if (sl.h_prod[s] > sl.cap) return MDEMOD_ERR_OVERFLOW;       /* cannot happen: <= 1 symbol per input sample */
			widest = std::max(widest, sl.h_prod[s]);
		}
		sl.width = sl.pitch;
		if (widest > sl.pitch) {
			int rcg = grow_host(&sl.h_soft, &sl.h_soft_bytes, static_cast<size_t>(widest) * 2 * ns + 16);
			if (rcg) return rcg;
			PIPE_TRY(hipMemcpy2DAsync(sl.h_soft, static_cast<size_t>(widest) * 2, sl.d_soft, static_cast<size_t>(sl.cap) * 2,
			                          static_cast<size_t>(widest) * 2, ns, hipMemcpyDeviceToHost, p->s_out));
			PIPE_TRY(hipStreamSynchronize(p->s_out));
			sl.width = widest;
		}
		std::vector<uint64_t> w(ns);
		uint64_t acc = 0;
		for (uint32_t s = 0; s < ns; s++) { acc += sl.h_prod[s]; w[s] = acc; }
		parallel_streams(ns, w, [&](uint32_t a, uint32_t b) {
			for (uint32_t s = a; s < b; s++) {
				uint32_t m = sl.h_prod[s];
				const uint32_t room = soft_cap[s] > produced[s] ? soft_cap[s] - produced[s] : 0;
				if (m > room) m = room;                           /* overflow is reported below */
				if (m) memcpy(soft_host[s] + 2 * static_cast<size_t>(produced[s]), sl.h_soft + static_cast<size_t>(s) * sl.width * 2, static_cast<size_t>(m) * 2);
			}
		});
		for (uint32_t s = 0; s < ns; s++) {
			if (static_cast<uint64_t>(produced[s]) + sl.h_prod[s] > soft_cap[s]) { result = MDEMOD_ERR_OVERFLOW; produced[s] = soft_cap[s]; }
			else produced[s] += sl.h_prod[s];
		}
		/* lock events of this sub-block (rare): keep the first 32 per stream over the whole call */
		bool any = false;
		for (uint32_t s = 0; s < ns && !any; s++) any = sl.h_ev[s] != 0;
		if (any && K > 1) {
			if (ev_store.empty()) ev_store.resize(static_cast<size_t>(ns) * MDEMOD_MAX_LOCK_EVENTS);
			for (uint32_t s = 0; s < ns; s++) {
				const uint32_t n_new = std::min<uint32_t>(sl.h_ev[s], MDEMOD_MAX_LOCK_EVENTS);
				if (!n_new) continue;
				const uint32_t have = std::min<uint32_t>(events[s], MDEMOD_MAX_LOCK_EVENTS);
				const uint32_t take = std::min<uint32_t>(n_new, MDEMOD_MAX_LOCK_EVENTS - have);
				if (take) {
					/* from the slot's own copy (taken on the compute stream right after its kernel; ev_out follows it) */
					PIPE_TRY(hipMemcpy(&ev_store[static_cast<size_t>(s) * MDEMOD_MAX_LOCK_EVENTS + have],
					                   sl.d_events + static_cast<size_t>(s) * MDEMOD_MAX_LOCK_EVENTS, sizeof(mdemod_lock_event) * take, hipMemcpyDeviceToHost));
				}
			}
		}
		for (uint32_t s = 0; s < ns; s++) events[s] += sl.h_ev[s];
		return MDEMOD_OK;
	};

	for (uint32_t k = 0; k < K; k++) {
		Slot &sl = p->slot[k & 1];
		/* ---- layout of sub-block k ---- */
		std::vector<uint64_t> wpre(ns);
		uint64_t pos = 0; uint32_t sub_max = 0;
		if (sl.used_in) PIPE_TRY(hipEventSynchronize(sl.ev_in));          /* h_iq / h_off / h_cnt of this slot are free again */
		for (uint32_t s = 0; s < ns; s++) {
			const uint32_t lo = sub_lo(s, k), hi = sub_lo(s, k + 1);
			sl.h_off[s] = pos; sl.h_cnt[s] = hi - lo;
			pos += (static_cast<uint64_t>(hi - lo) + 7) & ~7ull;          /* keep streams 16-byte aligned */
			wpre[s] = pos;
			sub_max = std::max(sub_max, hi - lo);
		}
		const size_t iq_bytes = static_cast<size_t>(pos) * sb + 64;
		const uint32_t cap = ((sub_max + 8 + 7) / 8) * 8;                   /* hard bound: one symbol per input sample */
		const uint32_t pitch = static_cast<uint32_t>(std::min<uint64_t>(cap, mdemod_nominal_symbols(ctx, sub_max)));
		const size_t soft_bytes = static_cast<size_t>(cap) * 2 * ns, pack_bytes = static_cast<size_t>(pitch) * 2 * ns;
		if (iq_bytes > sl.iq_bytes || soft_bytes > sl.d_soft_bytes || pack_bytes > sl.d_pack_bytes || pack_bytes > sl.h_soft_bytes) {
			/* the slot's previous sub-block must be completely through before its buffers are replaced */
			if (sl.used_out) PIPE_TRY(hipEventSynchronize(sl.ev_out));
			rc = grow_pair(&sl.h_iq, &sl.d_iq, &sl.iq_bytes, iq_bytes);
			if (rc) return rc;
			rc = grow_dev(&sl.d_soft, &sl.d_soft_bytes, soft_bytes);
			if (rc) return rc;
			rc = grow_dev(&sl.d_pack, &sl.d_pack_bytes, pack_bytes);
			if (rc) return rc;
			rc = grow_host(&sl.h_soft, &sl.h_soft_bytes, pack_bytes);
			if (rc) return rc;
		}
		/* ---- pack (CPU) ---- */
		parallel_streams(ns, wpre, [&](uint32_t a, uint32_t b) {
			for (uint32_t s = a; s < b; s++)
				if (sl.h_cnt[s])
					memcpy(sl.h_iq + sl.h_off[s] * sb, static_cast<const unsigned char *>(iq_host[s]) + static_cast<size_t>(sub_lo(s, k)) * sb,
					       static_cast<size_t>(sl.h_cnt[s]) * sb);
		});
		/* ---- H2D: after the kernel that last read this slot's device input ---- */
		if (sl.used_k) PIPE_TRY(hipStreamWaitEvent(p->s_in, sl.ev_k, 0));
		if (pos) PIPE_TRY(hipMemcpyAsync(sl.d_iq, sl.h_iq, static_cast<size_t>(pos) * sb, hipMemcpyHostToDevice, p->s_in));
		PIPE_TRY(hipMemcpyAsync(sl.d_off, sl.h_off, sizeof(uint64_t) * ns, hipMemcpyHostToDevice, p->s_in));
		PIPE_TRY(hipMemcpyAsync(sl.d_cnt, sl.h_cnt, sizeof(uint32_t) * ns, hipMemcpyHostToDevice, p->s_in));
		PIPE_TRY(hipEventRecord(sl.ev_in, p->s_in)); sl.used_in = true;
		/* ---- kernel: after the copy-in, and after the copy-out that last read this slot's device output ---- */
		PIPE_TRY(hipStreamWaitEvent(p->s_cmp, sl.ev_in, 0));
		if (sl.used_out) PIPE_TRY(hipStreamWaitEvent(p->s_cmp, sl.ev_out, 0));
		sl.cap = cap; sl.pitch = pitch;
		rc = mdemod_process_device(ctx, sl.d_iq, sl.d_off, sl.d_cnt, sl.d_soft, cap, cap, p->s_cmp);
		if (rc) return rc;
		PIPE_TRY(mdemod_launch_compact_rows(sl.d_soft, cap, sl.d_pack, pitch, st.sym_this_call, ns, p->s_cmp));
		PIPE_TRY(hipMemcpyAsync(sl.h_prod, st.sym_this_call, sizeof(uint32_t) * ns, hipMemcpyDeviceToHost, p->s_cmp));
		PIPE_TRY(hipMemcpyAsync(sl.h_ev, st.ev_this_call, sizeof(uint32_t) * ns, hipMemcpyDeviceToHost, p->s_cmp));
		if (K > 1) PIPE_TRY(hipMemcpyAsync(sl.d_events, st.events, sizeof(mdemod_lock_event) * MDEMOD_MAX_LOCK_EVENTS * ns, hipMemcpyDeviceToDevice, p->s_cmp));
		PIPE_TRY(hipEventRecord(sl.ev_k, p->s_cmp)); sl.used_k = true;
		/* ---- D2H of the nominal-pitch copy ---- */
		PIPE_TRY(hipStreamWaitEvent(p->s_out, sl.ev_k, 0));
		PIPE_TRY(hipMemcpyAsync(sl.h_soft, sl.d_pack, pack_bytes, hipMemcpyDeviceToHost, p->s_out));
		PIPE_TRY(hipEventRecord(sl.ev_out, p->s_out)); sl.used_out = true;
		/* ---- hand sub-block k-1 back to the caller while the GPU works on k ---- */
		if (k >= 1) { rc = unpack(p->slot[(k - 1) & 1]); if (rc) return rc; }
	}
	rc = unpack(p->slot[(K - 1) & 1]);
	if (rc) return rc;

	/* ---- "this call" counters := totals over the sub-blocks ---- */
	if (K > 1) {
		PIPE_TRY(hipMemcpyAsync(st.sym_this_call, produced.data(), sizeof(uint32_t) * ns, hipMemcpyHostToDevice, p->s_cmp));
		PIPE_TRY(hipMemcpyAsync(st.ev_this_call, events.data(), sizeof(uint32_t) * ns, hipMemcpyHostToDevice, p->s_cmp));
		if (!ev_store.empty())
			PIPE_TRY(hipMemcpyAsync(st.events, ev_store.data(), ev_store.size() * sizeof(mdemod_lock_event), hipMemcpyHostToDevice, p->s_cmp));
	}
	PIPE_TRY(hipStreamSynchronize(p->s_cmp));
	PIPE_TRY(hipStreamSynchronize(p->s_out));
	if (n_symbols) for (uint32_t s = 0; s < ns; s++) n_symbols[s] = produced[s];
	return result;
}
