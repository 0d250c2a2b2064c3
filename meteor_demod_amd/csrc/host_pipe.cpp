/*
 * host_pipe.cpp — the host-buffer entry (mdemod_process_host) as a three-stage pipeline.
 *
 * The reference reads its input with fread into a 32 KiB buffer and converts sample by
 * sample (wavfile.c:55-69); a GPU fed from host memory is bound by PCIe, so the work is to
 * keep the link busy: each stream's block is cut into K consecutive sub-blocks (chained
 * calls are exact, the state lives in the context), and sub-block k+1 is packed into pinned
 * memory and copied in while sub-block k is demodulated and sub-block k-1 is copied out;
 * the host hands sub-block k-2 back to the caller's buffers:
 *
 *     CPU pack(k+1) | H2D(k+1)  [stream in]  |  kernel(k) [stream cmp]  |  D2H(k-1) [stream out] | CPU unpack(k-2)
 *
 * Round 5 (the link is the bound, so everything else has to stay off its critical path; tools/ubench/h2d_rect.cpp has the rates):
 *   - THREE staging sets: the host unpacks k-2, whose copy-out finished long ago, instead of waiting for k-1 with its hands in
 *     its pockets while the copy-in engine runs dry (r04: 40 of 57 GB/s);
 *   - the pack writes the pinned ring with non-temporal stores (no read-for-ownership of lines the CPU never reads again: the pack
 *     alone 65 -> 105 GB/s on 16 threads of an EPYC 9575F) on a pool of 12 threads;
 *   - up to 16 full-size sub-blocks of >= 32 MiB between two short ones at either end (a quarter, a half): what the pipeline cannot
 *     overlap is one pack / copy-in at the start and one kernel + copy-out + unpack at the end, and both are as long as a sub-block;
 *   - rows inside a range the caller pinned (mdemod_pin_host_buffer) are copied from where they are, one 2-D copy per sub-block.
 * A call that fails part-way waits for whatever it has queued before it returns (Drain below): the copy engine may be reading the
 * caller's rows and writing this function's vectors.
 * Every sub-block's lock events are copied aside on the compute stream because the next launch overwrites the context's list.
 * Pinned + device staging buffers grow only; three HIP streams, events for the hand-offs.
 * After the last sub-block the per-call counters of the context (symbols / lock events of "this
 * call") are rewritten with the totals over all sub-blocks, so the status snapshot means what the
 * header says.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
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
			mdm_note_error("%s failed: %s (%s:%d)", #expr,                                   \
			        hipGetErrorString(e_), __FILE__, __LINE__);                                 \
			(void)hipGetLastError();                                                            \
			return e_ == hipErrorOutOfMemory ? MDEMOD_ERR_NOMEM : MDEMOD_ERR_HIP;               \
		}                                                                                       \
	} while (0)

constexpr int kSlots = 3;        /* staging sets: one being filled / copied in, one under the kernel / copied out, one being handed back */

struct Slot {                    /* one of the staging sets */
	unsigned char *h_iq = nullptr;  size_t h_iq_bytes = 0;       /* pinned ring (staged path only)                 */
	unsigned char *d_iq = nullptr;  size_t d_iq_bytes = 0;
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

struct Pin { const unsigned char *base; size_t bytes; bool ours; };      /* ours: registered by this library (memory the caller got from hipHostMalloc is pinned already) */

struct HostPipe;
/* every way out of mdemod_hostpipe_run, the failing ones too, leaves nothing in flight */
struct Drain {
	HostPipe *p;
	explicit Drain(HostPipe *pipe) : p(pipe) {}
	~Drain();
	Drain(const Drain &) = delete;
	Drain &operator=(const Drain &) = delete;
};

struct HostPipe {
	Slot slot[kSlots];
	std::vector<Pin> pins;       /* ranges of the caller's memory registered with the HIP runtime (mdemod_pin_host_buffer) */
	hipStream_t s_in = nullptr, s_cmp = nullptr, s_out = nullptr;
	uint32_t ns = 0;
	bool ready = false;
};

Drain::~Drain()
{
	if (p->s_in) (void)hipStreamSynchronize(p->s_in);
	if (p->s_cmp) (void)hipStreamSynchronize(p->s_cmp);
	if (p->s_out) (void)hipStreamSynchronize(p->s_out);
}

template <typename T>
int
grow_host_any(T **host, size_t *have, size_t need)
{
	if (need <= *have) return MDEMOD_OK;
	if (*host) (void)hipHostFree(*host);
	*host = nullptr; *have = 0;
	need += need / 8;                                        /* a little headroom: fewer re-allocations */
	PIPE_TRY(hipHostMalloc(reinterpret_cast<void **>(host), need, hipHostMallocDefault));
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

template <typename T>
int
grow_dev(T **dev, size_t *have, size_t need)
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
#ifdef MDEMOD_PIPE_PRIO
	{
		int lo = 0, hi = 0;
		(void)hipDeviceGetStreamPriorityRange(&lo, &hi);          /* lo: numerically greatest = least urgent */
		if (!p->s_in) PIPE_TRY(hipStreamCreateWithPriority(&p->s_in, hipStreamNonBlocking, lo));
		if (!p->s_cmp) PIPE_TRY(hipStreamCreateWithPriority(&p->s_cmp, hipStreamNonBlocking, hi));
		if (!p->s_out) PIPE_TRY(hipStreamCreateWithPriority(&p->s_out, hipStreamNonBlocking, hi));
	}
#endif
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
	for (const Pin &pin : p->pins) if (pin.ours) (void)hipHostUnregister(const_cast<unsigned char *>(pin.base));
	if (p->s_in) (void)hipStreamDestroy(p->s_in);
	if (p->s_cmp) (void)hipStreamDestroy(p->s_cmp);
	if (p->s_out) (void)hipStreamDestroy(p->s_out);
	delete p;
}

/* Does the pinning that holds `b` hold all of [b, b + bytes)?  The runtime knows the allocation a pointer belongs to
 * (hipMemGetAddressRange: base and size) - for hipHostMalloc memory; for memory somebody hipHostRegister'ed, ROCm 7.2 answers with a
 * NULL base and the registration's size (measured, r06: gpurun_exp/pin_probe.py), which says nothing about where `b` lies in it.
 * Then every 4 KiB page up to the end is asked for its own type: slow (a fraction of a microsecond per page) but this is a
 * once-per-buffer call and the answer decides between a copy and a fault. */
static bool
pinned_extent_covers(const unsigned char *b, size_t bytes)
{
	hipDeviceptr_t abase = nullptr;
	size_t asize = 0;
	if (hipMemGetAddressRange(&abase, &asize, const_cast<unsigned char *>(b)) == hipSuccess && abase && asize) {
		const unsigned char *a0 = static_cast<const unsigned char *>(abase);
		return a0 <= b && b + bytes <= a0 + asize;
	}
	(void)hipGetLastError();
	const uintptr_t page = 4096;
	for (uintptr_t at = (reinterpret_cast<uintptr_t>(b) & ~(page - 1)) + page; at < reinterpret_cast<uintptr_t>(b) + bytes; at += page) {
		hipPointerAttribute_t a;
		if (hipPointerGetAttributes(&a, reinterpret_cast<const void *>(at)) != hipSuccess || a.type != hipMemoryTypeHost) { (void)hipGetLastError(); return false; }
	}
	return true;
}

/* mdemod_pin_host_buffer / mdemod_unpin_host_buffer (include/meteor_demod_amd.h) */
int
mdemod_hostpipe_pin(void **pipe_slot, const void *base, size_t bytes)
{
	if (!base || !bytes) return MDEMOD_ERR_PARAM;
	if (!*pipe_slot) *pipe_slot = new HostPipe();
	HostPipe *p = static_cast<HostPipe *>(*pipe_slot);
	const unsigned char *b = static_cast<const unsigned char *>(base);
	for (const Pin &pin : p->pins)
		if (b < pin.base + pin.bytes && pin.base < b + bytes) {                                /* overlaps a range that is pinned already */
			mdm_note_error("mdemod_pin_host_buffer: [%p, +%zu) overlaps [%p, +%zu), pinned through this context already", base, bytes, static_cast<const void *>(pin.base), pin.bytes);
			return MDEMOD_ERR_PARAM;
		}
	/* memory that is pinned already (hipHostMalloc, or registered by the caller) is taken as it is and left as it is - if the
	 * pinning covers ALL of [base, base + bytes): a caller who registered a part of the buffer, or registered it in pieces, gets
	 * MDEMOD_ERR_PARAM (the direct path would hand hipMemcpy2DAsync pages the runtime never locked); rows outside any pinned
	 * range take the staged path anyway */
	hipPointerAttribute_t attr;
	const hipError_t q = hipPointerGetAttributes(&attr, b);
	if (q == hipSuccess && attr.type == hipMemoryTypeHost) {
		if (!pinned_extent_covers(b, bytes)) {
			mdm_note_error("mdemod_pin_host_buffer: %p is pinned already, but that pinning does not cover all %zu bytes asked for", base, bytes);
			return MDEMOD_ERR_PARAM;
		}
		p->pins.push_back({ b, bytes, false });
		return MDEMOD_OK;
	}
	if (q != hipSuccess) (void)hipGetLastError();                 /* (pageable memory: an error or "unregistered", by runtime version) */
	const hipError_t e = hipHostRegister(const_cast<unsigned char *>(b), bytes, hipHostRegisterDefault);
	if (e == hipErrorHostMemoryAlreadyRegistered) {
		/* a part of the range (not its first byte: that case is above) belongs to somebody's registration */
		(void)hipGetLastError();
		mdm_note_error("mdemod_pin_host_buffer: a part of [%p, +%zu) is registered already, the range as a whole is not", base, bytes);
		return MDEMOD_ERR_PARAM;
	}
	PIPE_TRY(e);
	p->pins.push_back({ b, bytes, true });
	return MDEMOD_OK;
}

int
mdemod_hostpipe_unpin(void *opaque, const void *base)
{
	HostPipe *p = static_cast<HostPipe *>(opaque);
	if (!p || !base) return MDEMOD_ERR_PARAM;
	for (size_t i = 0; i < p->pins.size(); i++)
		if (p->pins[i].base == base) {
			/* (nothing of this context is in flight: mdemod_process_host is synchronous) */
			if (p->pins[i].ours) PIPE_TRY(hipHostUnregister(const_cast<unsigned char *>(p->pins[i].base)));
			p->pins.erase(p->pins.begin() + static_cast<long>(i));
			return MDEMOD_OK;
		}
	mdm_note_error("mdemod_unpin_host_buffer: %p is not the base of a range pinned through this context", base);
	return MDEMOD_ERR_PARAM;
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
	/* Rows the copy engine can take straight from the caller's pages: the whole batch inside a range the caller pinned
	 * (mdemod_pin_host_buffer), every stream as long as the others and one stride apart - a batch read into one buffer (the C
	 * host's, a [streams][samples] array).  Then a sub-block is ONE two-dimensional copy and the CPU does not touch the input at all
	 * (r05, tools/ubench/h2d_rect.cpp: 57 GB/s, what the link gives a contiguous pinned copy; the staged path's pack competes with the
	 * copy engine for the host's memory and holds it at 53).  Anything else is staged through the pinned ring as before. */
	bool direct_all = false;
	size_t row_stride = 0;
	if (!p->pins.empty() && ns >= 1 && n_max > 0 && iq_host[0]) {
		/* (addresses compared as integers: the rows need not belong to one object as far as the language is concerned) */
		const uintptr_t first = reinterpret_cast<uintptr_t>(iq_host[0]);
		const size_t row_bytes = static_cast<size_t>(n_max) * sb;
		direct_all = true;
		for (uint32_t s = 0; s < ns && direct_all; s++) direct_all = n_samples[s] == n_max;
		if (direct_all && ns > 1) {
			const uintptr_t second = reinterpret_cast<uintptr_t>(iq_host[1]);
			direct_all = second > first && second - first >= row_bytes;
			row_stride = direct_all ? static_cast<size_t>(second - first) : 0;
			for (uint32_t s = 2; s < ns && direct_all; s++) direct_all = reinterpret_cast<uintptr_t>(iq_host[s]) == first + static_cast<uintptr_t>(s) * row_stride;
		} else if (direct_all) row_stride = row_bytes;
		if (direct_all) {
			const uintptr_t end = first + static_cast<uintptr_t>(ns - 1) * row_stride + row_bytes;
			bool inside = false;
			for (const Pin &pin : p->pins) {
				const uintptr_t lo = reinterpret_cast<uintptr_t>(pin.base);
				inside = inside || (first >= lo && end <= lo + pin.bytes);
			}
			direct_all = inside;
		}
	}
	/* Sub-blocks: K consecutive pieces of every stream's block, 16 full-size ones at most (each stream's piece should stay a few
	 * KB: both host copies pay a DRAM round trip per piece; and a launch hands the filter history over, taps - 1 samples per
	 * stream, again and again), at least MINSAMP samples of the longest stream each.  What the pipeline cannot overlap is the first
	 * pack and the last kernel + copy-out + unpack, so the first two and the last two sub-blocks are a quarter and a half of the
	 * others (r05: 16 equal ones left 3.5 ms of a 41 ms call outside the overlap). */
#ifndef MDEMOD_PIPE_KMAX
#define MDEMOD_PIPE_KMAX 16
#define MDEMOD_PIPE_MINSAMP 2048
#define MDEMOD_PIPE_SUBBYTES (32u << 20)
#define MDEMOD_PIPE_RAMP 1
#endif
	uint32_t K = static_cast<uint32_t>(std::min<uint64_t>(MDEMOD_PIPE_KMAX, std::max<uint64_t>(1, total * sb / MDEMOD_PIPE_SUBBYTES)));
	while (K > 1 && n_max / K < MDEMOD_PIPE_MINSAMP) K--;
	std::vector<uint64_t> cumw(1, 0);                         /* sub-block k covers [cumw[k], cumw[k + 1]) / cumw[K] of every stream */
	if (MDEMOD_PIPE_RAMP && K >= 8) {
		K += 2;                                                  /* (the four short ones together are one and a half full ones) */
		for (uint32_t k = 0; k < K; k++) cumw.push_back(cumw.back() + ((k == 0 || k == K - 1) ? 1 : (k == 1 || k == K - 2) ? 2 : 4));
	} else {
		for (uint32_t k = 0; k < K; k++) cumw.push_back(cumw.back() + 1);
	}
#ifdef MDEMOD_PIPE_TRACE
	/* diagnosis only (results are wrong with any of them): MDEMOD_PIPE_SKIP bit 0 no kernel / compaction, bit 1 no copy-out, bit 2 no
	   CPU unpack, bit 3 no compaction only, bit 4 no demodulation kernel only - which of them slows the copy-ins from 2.36 ms alone to 2.48 ms in the pipeline? */
	const int tr_skip = getenv("MDEMOD_PIPE_SKIP") ? atoi(getenv("MDEMOD_PIPE_SKIP")) : 0;
	double tr_pack = 0, tr_unpack = 0, tr_wait_in = 0, tr_enq = 0, tr_layout = 0, tr_wait_out = 0;
	std::vector<hipEvent_t> tr_e0(80), tr_e1(80), tr_k0(80), tr_k1(80), tr_o1(80);
	for (int i = 0; i < 80; i++) { (void)hipEventCreate(&tr_e0[i]); (void)hipEventCreate(&tr_e1[i]); (void)hipEventCreate(&tr_k0[i]); (void)hipEventCreate(&tr_k1[i]); (void)hipEventCreate(&tr_o1[i]); }
	std::vector<double> tr_enq_at(80, 0);
	auto tr_now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double tr_t0 = tr_now();
#define TR(acc, stmt) do { const double t_ = tr_now(); stmt; acc += tr_now() - t_; } while (0)
#else
#define TR(acc, stmt) do { stmt; } while (0)
#endif

#ifdef MDEMOD_PIPE_TRACE_LIGHT
	/* three events and two host times per call: the copy-ins' span without the per-sub-block events of MDEMOD_PIPE_TRACE, which move things */
	hipEvent_t lt_first, lt_last_in, lt_last_k;
	(void)hipEventCreate(&lt_first); (void)hipEventCreate(&lt_last_in); (void)hipEventCreate(&lt_last_k);
	const auto lt_t0 = std::chrono::steady_clock::now();
#endif
	std::vector<uint32_t> produced(ns, 0), events(ns, 0);
	std::vector<mdemod_lock_event> ev_store;                  /* merged lock events: [stream][32] (only if any) */
	int result = MDEMOD_OK;
	const Drain drain(p);                                     /* declared after the vectors the queued copies read: destroyed before them */
	/* (Starting the sub-blocks on multiples of 128 or 1 024 samples of a stream - whole bus transactions for the copy engine - was
	 * measured in r05: the same 55.7 GB/s per 2-D copy as with pieces that begin anywhere.) */
	auto sub_lo = [&](uint32_t s, uint32_t k) { return static_cast<uint32_t>(static_cast<uint64_t>(n_samples[s]) * cumw[k] / cumw[K]); };

	auto unpack = [&](Slot &sl) -> int {
		TR(tr_wait_out, PIPE_TRY(hipEventSynchronize(sl.ev_out)));
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
#ifdef MDEMOD_PIPE_TRACE
				if (tr_skip & 4) m = 0;
#endif
				const uint32_t room = soft_cap[s] > produced[s] ? soft_cap[s] - produced[s] : 0;
				if (m > room) m = room;                           /* overflow is reported below */
				if (m) stream_copy(reinterpret_cast<unsigned char *>(soft_host[s]) + 2 * static_cast<size_t>(produced[s]),
				                   reinterpret_cast<const unsigned char *>(sl.h_soft) + static_cast<size_t>(s) * sl.width * 2, static_cast<size_t>(m) * 2);
			}
			_mm_sfence();                                             /* the caller may read its rows from any thread once the call returns */
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
		Slot &sl = p->slot[k % kSlots];
		/* ---- layout of sub-block k ---- */
		std::vector<uint64_t> wpre(ns);
		uint64_t pos = 0; uint32_t sub_max = 0;
		TR(tr_wait_in, if (sl.used_in) PIPE_TRY(hipEventSynchronize(sl.ev_in)));          /* h_iq / h_off / h_cnt of this slot are free again */
		for (uint32_t s = 0; s < ns; s++) {
			const uint32_t lo = sub_lo(s, k), hi = sub_lo(s, k + 1);
			sl.h_off[s] = pos; sl.h_cnt[s] = hi - lo;
			pos += (static_cast<uint64_t>(hi - lo) + 7) & ~7ull;          /* keep streams 16-byte aligned */
			wpre[s] = pos;
			sub_max = std::max(sub_max, hi - lo);
		}
		/* experiment knob: the first sub-blocks of a pinned batch staged like any other (see MDEMOD_PIPE_DIRECT_FROM below) */
#ifndef MDEMOD_PIPE_DIRECT_FROM
#define MDEMOD_PIPE_DIRECT_FROM 0
#endif
		const bool direct_k = direct_all && (K < 8 || k >= MDEMOD_PIPE_DIRECT_FROM);
		/* the launch's offsets and counts travel behind the samples in the same ring and the same copy (staged path) or are written by
		   a kernel (pinned rows: every stream as long as the others, one pitch apart) - as two copies of their own on the copy-in
		   stream they sat between every two sub-blocks' samples, 0.03-0.04 ms of idle link each time (r05) */
		const size_t meta_at = (static_cast<size_t>(pos) * sb + 63) & ~static_cast<size_t>(63);
		const size_t meta_bytes = static_cast<size_t>(ns) * (sizeof(uint64_t) + sizeof(uint32_t));
		const size_t iq_bytes = meta_at + meta_bytes + 64;
		const uint32_t cap = ((sub_max + 8 + 7) / 8) * 8;                   /* hard bound: one symbol per input sample */
		const uint32_t pitch = static_cast<uint32_t>(std::min<uint64_t>(cap, mdemod_nominal_symbols(ctx, sub_max)));
		const size_t soft_bytes = static_cast<size_t>(cap) * 2 * ns, pack_bytes = static_cast<size_t>(pitch) * 2 * ns;
		if (iq_bytes > sl.d_iq_bytes || (!direct_k && iq_bytes > sl.h_iq_bytes) || soft_bytes > sl.d_soft_bytes || pack_bytes > sl.d_pack_bytes || pack_bytes > sl.h_soft_bytes) {
			/* the slot's previous sub-block must be completely through before its buffers are replaced */
			if (sl.used_out) PIPE_TRY(hipEventSynchronize(sl.ev_out));
			rc = grow_dev(&sl.d_iq, &sl.d_iq_bytes, iq_bytes);
			if (rc) return rc;
			if (!direct_k) { rc = grow_host_any(&sl.h_iq, &sl.h_iq_bytes, iq_bytes); if (rc) return rc; }
			rc = grow_dev(&sl.d_soft, &sl.d_soft_bytes, soft_bytes);
			if (rc) return rc;
			rc = grow_dev(&sl.d_pack, &sl.d_pack_bytes, pack_bytes);
			if (rc) return rc;
			rc = grow_host(&sl.h_soft, &sl.h_soft_bytes, pack_bytes);
			if (rc) return rc;
		}
		/* ---- pack (CPU): not for rows the copy engine reads where they are ---- */
		if (!direct_k) TR(tr_pack, parallel_streams(ns, wpre, [&](uint32_t a, uint32_t b) {
			for (uint32_t s = a; s < b; s++) {
				if (s + 1 < b && sl.h_cnt[s + 1]) prefetch_piece(static_cast<const unsigned char *>(iq_host[s + 1]) + static_cast<size_t>(sub_lo(s + 1, k)) * sb);
				if (sl.h_cnt[s])
					stream_copy(sl.h_iq + sl.h_off[s] * sb, static_cast<const unsigned char *>(iq_host[s]) + static_cast<size_t>(sub_lo(s, k)) * sb,
					            static_cast<size_t>(sl.h_cnt[s]) * sb);
			}
			_mm_sfence();                                                  /* the non-temporal stores of this thread are out before the copy engine reads the ring */
		}));
		/* ---- H2D: after the kernel that last read this slot's device input ---- */
		if (sl.used_k) PIPE_TRY(hipStreamWaitEvent(p->s_in, sl.ev_k, 0));
		/* Never more than two copy-ins queued: a 2-D copy that is enqueued while two are still ahead of it runs at 38 GB/s instead of
		 * 56, and so does the one behind it (r05 timeline: the first three of a pinned batch, which the host enqueues within 0.2 ms
		 * of each other; any later one that finds two ahead of it).  Waiting for the copy before the previous one costs nothing:
		 * the link has a whole sub-block queued meanwhile. */
#ifndef MDEMOD_PIPE_MAX_QUEUED
#define MDEMOD_PIPE_MAX_QUEUED 2
#endif
		if (k >= MDEMOD_PIPE_MAX_QUEUED) PIPE_TRY(hipEventSynchronize(p->slot[(k - MDEMOD_PIPE_MAX_QUEUED) % kSlots].ev_in));
#ifdef MDEMOD_PIPE_TRACE
		tr_enq_at[k] = (tr_now() - tr_t0) * 1e3;
		(void)hipEventRecord(tr_e0[k], p->s_in);
#endif
#ifdef MDEMOD_PIPE_TRACE_LIGHT
		if (k == 0) (void)hipEventRecord(lt_first, p->s_in);
#endif
		if (pos && direct_k) {
			/* every stream's piece is [lo, lo + cnt) of its row: device rows at the ring's own pitch (8-sample multiples, h_off) */
			const uint32_t lo = sub_lo(0, k), cnt = sl.h_cnt[0];
			const size_t dev_pitch = ns > 1 ? static_cast<size_t>(sl.h_off[1] - sl.h_off[0]) * sb : static_cast<size_t>(cnt) * sb;
			if (cnt) PIPE_TRY(hipMemcpy2DAsync(sl.d_iq, dev_pitch, static_cast<const unsigned char *>(iq_host[0]) + static_cast<size_t>(lo) * sb, row_stride,
			                                   static_cast<size_t>(cnt) * sb, ns, hipMemcpyHostToDevice, p->s_in));
		} else if (!direct_k) {
			memcpy(sl.h_iq + meta_at, sl.h_off, sizeof(uint64_t) * ns);
			memcpy(sl.h_iq + meta_at + sizeof(uint64_t) * ns, sl.h_cnt, sizeof(uint32_t) * ns);
			PIPE_TRY(hipMemcpyAsync(sl.d_iq, sl.h_iq, meta_at + meta_bytes, hipMemcpyHostToDevice, p->s_in));
		}
#ifdef MDEMOD_PIPE_TRACE
		(void)hipEventRecord(tr_e1[k], p->s_in);
#endif
#ifdef MDEMOD_PIPE_TRACE_LIGHT
		if (k + 1 == K) (void)hipEventRecord(lt_last_in, p->s_in);
#endif
		PIPE_TRY(hipEventRecord(sl.ev_in, p->s_in)); sl.used_in = true;

		/* ---- kernel: after the copy-in, and after the copy-out that last read this slot's device output ---- */
		PIPE_TRY(hipStreamWaitEvent(p->s_cmp, sl.ev_in, 0));
		if (sl.used_out) PIPE_TRY(hipStreamWaitEvent(p->s_cmp, sl.ev_out, 0));
		sl.cap = cap; sl.pitch = pitch;
		const uint64_t *k_off = reinterpret_cast<const uint64_t *>(sl.d_iq + meta_at);
		const uint32_t *k_cnt = reinterpret_cast<const uint32_t *>(sl.d_iq + meta_at + sizeof(uint64_t) * ns);
		if (direct_k) {
			/* (after the kernel that last read this slot's arrays: the compute stream is in order) */
			PIPE_TRY(mdemod_launch_fill_uniform_rows(sl.d_off, sl.d_cnt, ns > 1 ? sl.h_off[1] - sl.h_off[0] : 0, sl.h_cnt[0], ns, p->s_cmp));
			k_off = sl.d_off; k_cnt = sl.d_cnt;
		}
#ifdef MDEMOD_PIPE_TRACE
		(void)hipEventRecord(tr_k0[k], p->s_cmp);
#endif
#ifdef MDEMOD_PIPE_TRACE
		if (!(tr_skip & (1 | 16)))
#endif
		rc = mdemod_process_device(ctx, sl.d_iq, k_off, k_cnt, sl.d_soft, cap, cap, p->s_cmp);
		if (rc) return rc;
		/* The rows of the LAST sub-blocks go to the pinned host buffer straight from this kernel (posted writes over the link), not
		 * through the copy engine: a copy-out lands on whichever engine is free NEXT, which is the one busy with the copy-in in flight,
		 * so the last three of them queue up behind the last copy-ins and then behind each other - 2.4 ms after the last kernel (r05
		 * timeline).  For every sub-block it is no gain: the kernel's writes slow each copy-in by 4 %, the same 42.6 ms in total. */
#ifndef MDEMOD_PIPE_ZC_LAST
#define MDEMOD_PIPE_ZC_LAST 3
#endif
#ifndef MDEMOD_PIPE_ZC_FIRST
#define MDEMOD_PIPE_ZC_FIRST 0
#endif
		const bool zc_out = K >= 8 && (k + MDEMOD_PIPE_ZC_LAST >= K || k < MDEMOD_PIPE_ZC_FIRST);
#ifdef MDEMOD_PIPE_TRACE
		if (!(tr_skip & (1 | 8)))
#endif
		PIPE_TRY(mdemod_launch_compact_rows(sl.d_soft, cap, zc_out ? sl.h_soft : sl.d_pack, pitch, st.sym_this_call, ns, p->s_cmp));
#ifdef MDEMOD_PIPE_COUNTS_BY_COPY
		PIPE_TRY(hipMemcpyAsync(sl.h_prod, st.sym_this_call, sizeof(uint32_t) * ns, hipMemcpyDeviceToHost, p->s_cmp));
		PIPE_TRY(hipMemcpyAsync(sl.h_ev, st.ev_this_call, sizeof(uint32_t) * ns, hipMemcpyDeviceToHost, p->s_cmp));
#endif
		/* (only the events there are: the whole list is 512 bytes per stream - 8 MB and 0.46 ms of blit kernel per sub-block at 16 384
		   streams, the last of them in the call's tail; r05) */
#ifdef MDEMOD_PIPE_COUNTS_BY_COPY
		if (K > 1) PIPE_TRY(mdemod_launch_copy_events(st, sl.d_events, ns, nullptr, nullptr, p->s_cmp));
#else
		/* ... and the launch's two counters per stream into the slot's pinned arrays by the same kernel: as copies they sat in a copy
		   engine's queue until the kernel was through, and the runtime may put the next 2-D copy-in behind them (r05) */
		PIPE_TRY(mdemod_launch_copy_events(st, sl.d_events, ns, sl.h_prod, sl.h_ev, p->s_cmp));
#endif
#ifdef MDEMOD_PIPE_TRACE
		(void)hipEventRecord(tr_k1[k], p->s_cmp);
#endif
#ifdef MDEMOD_PIPE_TRACE_LIGHT
		if (k + 1 == K) (void)hipEventRecord(lt_last_k, p->s_cmp);
#endif
		PIPE_TRY(hipEventRecord(sl.ev_k, p->s_cmp)); sl.used_k = true;
		/* ---- D2H of the nominal-pitch copy ---- */
		PIPE_TRY(hipStreamWaitEvent(p->s_out, sl.ev_k, 0));
#ifdef MDEMOD_PIPE_TRACE
		if (!(tr_skip & 2))
#endif
		if (!zc_out) PIPE_TRY(hipMemcpyAsync(sl.h_soft, sl.d_pack, pack_bytes, hipMemcpyDeviceToHost, p->s_out));
#ifdef MDEMOD_PIPE_TRACE
		(void)hipEventRecord(tr_o1[k], p->s_out);
#endif
		PIPE_TRY(hipEventRecord(sl.ev_out, p->s_out)); sl.used_out = true;
		/* ---- hand sub-block k-2 back to the caller while k-1 is under the kernel and k on the link (its copy-out is done: no wait) ---- */
		if (k >= 2) { TR(tr_unpack, rc = unpack(p->slot[(k - 2) % kSlots])); if (rc) return rc; }
	}
	if (K >= 2) { TR(tr_unpack, rc = unpack(p->slot[(K - 2) % kSlots])); if (rc) return rc; }
	TR(tr_unpack, rc = unpack(p->slot[(K - 1) % kSlots]));
	if (rc) return rc;
#ifdef MDEMOD_PIPE_TRACE
	fprintf(stderr, "[host_pipe] K %u, total %.2f ms: pack %.2f, unpack (with its waits) %.2f, wait for the ring %.2f, layout %.2f ms; pool %u\n", K, (tr_now() - tr_t0) * 1e3,
	        tr_pack * 1e3, tr_unpack * 1e3, tr_wait_in * 1e3, tr_layout * 1e3, PackPool::get().size());
	fprintf(stderr, "[host_pipe]   waits for the copy-out inside unpack: %.2f ms\n", tr_wait_out * 1e3);
	(void)hipDeviceSynchronize();
	for (uint32_t k = 0; k < K && k < 80; k++) {
		float a = 0, b = 0, c = 0, d = 0, e = 0;
		(void)hipEventElapsedTime(&a, tr_e0[0], tr_e0[k]); (void)hipEventElapsedTime(&b, tr_e0[0], tr_e1[k]);
		(void)hipEventElapsedTime(&c, tr_e0[0], tr_k0[k]); (void)hipEventElapsedTime(&d, tr_e0[0], tr_k1[k]); (void)hipEventElapsedTime(&e, tr_e0[0], tr_o1[k]);
		fprintf(stderr, "[host_pipe]   k %2u: enqueued at %6.2f (host clock); H2D %6.2f .. %6.2f, kernel+compact %6.2f .. %6.2f, copy-out done %6.2f (ms after the first H2D began)\n", k, tr_enq_at[k], a, b, c, d, e);
	}
	for (int i = 0; i < 80; i++) { (void)hipEventDestroy(tr_e0[i]); (void)hipEventDestroy(tr_e1[i]); (void)hipEventDestroy(tr_k0[i]); (void)hipEventDestroy(tr_k1[i]); (void)hipEventDestroy(tr_o1[i]); }
	(void)tr_enq;
#endif

#ifdef MDEMOD_PIPE_TRACE_LIGHT
	{
		const double host_ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - lt_t0).count() * 1e3;
		(void)hipDeviceSynchronize();
		float a = 0, b = 0;
		(void)hipEventElapsedTime(&a, lt_first, lt_last_in); (void)hipEventElapsedTime(&b, lt_first, lt_last_k);
		fprintf(stderr, "[host_pipe] K %u: copy-ins span %.2f ms, last kernel done %.2f ms after the first copy-in began; host loop + unpack %.2f ms\n", K, a, b, host_ms);
		(void)hipEventDestroy(lt_first); (void)hipEventDestroy(lt_last_in); (void)hipEventDestroy(lt_last_k);
	}
#endif
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
