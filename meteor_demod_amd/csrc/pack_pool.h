/*
 * pack_pool.h — the host threads behind mdemod_process_host's packing and unpacking (host_pipe.cpp).  No HIP in here: the pool is
 * plain C++ threads and the streaming copy they run, so that tests/sanitize/pool_test.cpp can run exactly this code under ThreadSanitizer and
 * AddressSanitizer.
 */
#ifndef MDEMOD_PACK_POOL_H
#define MDEMOD_PACK_POOL_H

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <unistd.h>
#include <sched.h>
#include <cstdio>
#include <cstring>
#include <emmintrin.h>

namespace {

/* A few persistent worker threads for the packing and unpacking of the sub-blocks (memcpy between the caller's buffers and the
 * pinned ring).  Round 3 started fresh std::threads for every sub-block - 2 x 16 spawns of 4..8 threads per call, ~10 ms of a 64 ms
 * call, and the reason MORE threads were slower (measured r04: 4 threads 34.5 GB/s of input, 8..24 threads 30..31).  Pools are created
 * on first use and parked on a condition variable between jobs.  ONE pool serves a process with one caller; callers that arrive while
 * it is busy - one library context per GPU is one host thread each, all packing at once - get a pool of their own, up to as many as
 * the CPUs this process may use have room for (usable_cpus() / threads per pool, at most 8; MDEMOD_PACK_POOLS overrides): round 6,
 * ADVICE r05 - until then every caller queued on the one pool's mutex and eight GPUs shared one pack at a time. */
class PackPool {
public:
	static constexpr unsigned kMaxPools = 8;
	/* pool i, created on first use (0: the one a process with a single caller ever sees) */
	static PackPool &get(unsigned i = 0)
	{
		static Slots slots;
		PackPool *q = slots.p[i].load(std::memory_order_acquire);
		if (!q) {
			PackPool *fresh = new PackPool;
			if (slots.p[i].compare_exchange_strong(q, fresh, std::memory_order_acq_rel)) q = fresh;
			else delete fresh;                                   /* somebody else was first: theirs it is */
		}
		return *q;
	}
	static unsigned max_pools()
	{
		static const unsigned n = [] {
			const char *e = getenv("MDEMOD_PACK_POOLS");
			const int want = e ? atoi(e) : 0;
			if (want >= 1 && want <= static_cast<int>(kMaxPools)) return static_cast<unsigned>(want);
			return std::max(1u, std::min(kMaxPools, usable_cpus() / std::max(1u, get(0).size())));
		}();
		return n;
	}
	/* a pool nobody else is running a job on right now (its run mutex held until the lease goes): the first idle one, created if need
	   be, or - all busy - a wait for the first.  A forked child has the pool objects but not their threads, and their mutexes in
	   whatever state the fork caught them: it takes no lock and works alone. */
	class Lease {
	public:
		unsigned size() const { return alone ? 1u : pool->size(); }
		void run(unsigned n_jobs, const std::function<void(unsigned)> &job)
		{
			if (alone) { for (unsigned i = 0; i < n_jobs; i++) job(i); return; }
			pool->run_locked(n_jobs, job);
		}
	private:
		friend class PackPool;
		PackPool *pool = nullptr;
		std::unique_lock<std::mutex> lk;
		bool alone = false;
	};
	static Lease lease()
	{
		Lease l;
		PackPool &first = get(0);
		if (getpid() != first.owner) { l.alone = true; return l; }
		const unsigned np = max_pools();
		for (unsigned i = 0; i < np; i++) {
			PackPool &p = get(i);
			std::unique_lock<std::mutex> lk(p.sh->run_m, std::try_to_lock);
			if (lk.owns_lock()) { l.pool = &p; l.lk = std::move(lk); return l; }
		}
		l.pool = &first;
		l.lk = std::unique_lock<std::mutex>(first.sh->run_m);
		return l;
	}
	unsigned size() const { return static_cast<unsigned>(workers.size()) + 1; }          /* + the calling thread */
	/* run job(i) for i in [0, n_jobs): the workers take jobs off a shared counter, the caller takes its share too and returns when all are done */
	void run(unsigned n_jobs, const std::function<void(unsigned)> &job)
	{
		if (n_jobs == 0) return;
		/* (a forked child has the pool object but not its threads: it works alone) */
		if (n_jobs == 1 || workers.empty() || getpid() != owner) { for (unsigned i = 0; i < n_jobs; i++) job(i); return; }
		std::lock_guard<std::mutex> one_at_a_time(sh->run_m);          /* callers on several host threads may share this pool */
		run_locked(n_jobs, job);
	}
private:
	struct Slots {
		std::atomic<PackPool *> p[kMaxPools];
		Slots() { for (auto &q : p) q.store(nullptr, std::memory_order_relaxed); }
		~Slots() { for (auto &q : p) delete q.load(std::memory_order_acquire); }          /* (~PackPool knows what to do in a forked child) */
	};
	void run_locked(unsigned n_jobs, const std::function<void(unsigned)> &job)           /* sh->run_m is held by the caller */
	{
		if (n_jobs == 0) return;
		if (n_jobs == 1 || workers.empty()) { for (unsigned i = 0; i < n_jobs; i++) job(i); return; }
		Shared &s = *sh;
		{
			std::lock_guard<std::mutex> lk(s.m);
			s.cur = &job; s.total = n_jobs; s.next = 0; s.pending = n_jobs; s.generation++;
		}
		s.cv.notify_all();
		drain(s);
		std::unique_lock<std::mutex> lk(s.m);
		s.done_cv.wait(lk, [&] { return s.pending == 0; });
		s.cur = nullptr;
	}
private:
	/* everything the threads share lives on the heap, so that a forked child can walk away from it (see ~PackPool) */
	struct Shared {
		std::mutex run_m;
		std::mutex m;
		std::condition_variable cv, done_cv;
		const std::function<void(unsigned)> *cur = nullptr;
		unsigned total = 0, next = 0, pending = 0;
		uint64_t generation = 0;
		bool stop = false;
	};
public:
	/* CPUs this process may actually run on at once: the affinity mask (a container's cpuset, taskset) and the cgroup's CPU quota
	 * (cpu.max "quota period"), not the machine's core count - 12 pack threads time-slicing on a 2-CPU quota are slower than 2. */
	static unsigned usable_cpus()
	{
		unsigned n = std::max(1u, std::thread::hardware_concurrency());
		cpu_set_t set;
		CPU_ZERO(&set);
		if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = std::min<unsigned>(n, static_cast<unsigned>(CPU_COUNT(&set)));
		if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       /* cgroup v2 */
			char q[32] = "";
			long long period = 0;
			if (fscanf(f, "%31s %lld", q, &period) == 2 && period > 0 && strcmp(q, "max") != 0) {
				const long long quota = atoll(q);
				if (quota > 0) n = std::min<unsigned>(n, static_cast<unsigned>(std::max<long long>(1, (quota + period - 1) / period)));
			}
			fclose(f);
		} else {                                                                     /* cgroup v1 */
			long long quota = -1, period = 0;
			if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &quota) != 1) quota = -1; fclose(fq); }
			if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
			if (quota > 0 && period > 0) n = std::min<unsigned>(n, static_cast<unsigned>(std::max<long long>(1, (quota + period - 1) / period)));
		}
		return std::max(1u, n);
	}
private:
	PackPool() : sh(new Shared)
	{
		const char *e = getenv("MDEMOD_PACK_THREADS");
		int want = e ? atoi(e) : 0;
		if (want <= 0 || want > 64) want = 12;          /* r05, tools/ubench/h2d_rect.cpp: the pack into the pinned ring keeps up with the link from 8 threads on, best at 12..16 */
		const unsigned n = std::min<unsigned>(static_cast<unsigned>(want), usable_cpus());
		Shared *s = sh;
		for (unsigned i = 1; i < n; i++) workers.emplace_back([s] { loop(*s); });
	}
	~PackPool()
	{
		/* A forked child that leaves through exit() has this object but none of its threads, the mutex in whatever state a worker
		 * held it at the fork, and a condition variable with the parent's parked workers on its books (destroying that one waits
		 * for them: for ever).  Nothing to stop, nothing to join, nothing to destroy: the thread handles and the shared state are
		 * leaked on purpose (a joinable std::thread must not be destroyed either). */
		if (getpid() != owner) { new std::vector<std::thread>(std::move(workers)); return; }
		{ std::lock_guard<std::mutex> lk(sh->m); sh->stop = true; }
		sh->cv.notify_all();
		for (auto &t : workers) t.join();
		delete sh;
	}
	static void drain(Shared &s)
	{
		for (;;) {
			unsigned i;
			const std::function<void(unsigned)> *job;
			{
				std::lock_guard<std::mutex> lk(s.m);
				if (!s.cur || s.next >= s.total) return;
				i = s.next++; job = s.cur;
			}
			(*job)(i);
			std::lock_guard<std::mutex> lk(s.m);
			if (--s.pending == 0) s.done_cv.notify_all();
		}
	}
	static void loop(Shared &s)
	{
		uint64_t seen = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> lk(s.m);
				s.cv.wait(lk, [&] { return s.stop || s.generation != seen; });
				if (s.stop) return;
				seen = s.generation;
			}
			drain(s);
		}
	}
	std::vector<std::thread> workers;
	const pid_t owner = getpid();
	Shared *const sh;
};

/* run fn(first, last) over the streams [0, n) on the pool, split by the weights' prefix sums into a few pieces per thread */
template <typename F>
void
parallel_streams(uint32_t n, const std::vector<uint64_t> &weight_prefix, F fn)
{
	const uint64_t total = weight_prefix.empty() ? 0 : weight_prefix.back();
	/* small jobs are not worth waking anybody - but "small" is a megabyte, not eight: the last (quarter) sub-block of a pipelined call is
	   4.6 MB of 640-byte pieces at the bench shape, and one thread took a millisecond over it with nothing left to hide it behind (r05) */
#ifndef MDEMOD_POOL_MIN_BYTES
#define MDEMOD_POOL_MIN_BYTES (1u << 20)
#endif
	if (total < MDEMOD_POOL_MIN_BYTES) { fn(0u, n); return; }
	PackPool::Lease pool = PackPool::lease();                /* an idle pool (this caller's alone until the lease goes out of scope) */
	if (pool.size() == 1) { fn(0u, n); return; }
	const unsigned pieces = pool.size() * 4;
	std::vector<uint32_t> cut(pieces + 1, n);
	cut[0] = 0;
	uint32_t at = 0;
	for (unsigned w = 1; w < pieces; w++) {
		const uint64_t goal = total * w / pieces;
		while (at < n && weight_prefix[at] < goal) at++;
		cut[w] = at;
	}
	pool.run(pieces, [&](unsigned i) { if (cut[i + 1] > cut[i]) fn(cut[i], cut[i + 1]); });
}


/* memcpy with non-temporal stores: into the pinned ring (the CPU never reads it again) and into the caller's output rows (640 B ..
 * a few KB each, every one on another page: a plain memcpy first READS the destination lines it is about to overwrite - one DRAM
 * round trip per piece that nothing hides; tools/ubench/h2d_rect.cpp: the pack alone 65 -> 105 GB/s).  Whoever calls this fences
 * (_mm_sfence) before the bytes are handed on. */
inline void
stream_copy(unsigned char *dst, const unsigned char *src, size_t n)
{
	size_t head = (16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15;
	if (head > n) head = n;
	if (head) { memcpy(dst, src, head); dst += head; src += head; n -= head; }
	size_t i = 0;
	for (; i + 64 <= n; i += 64) {
		const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i)), b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i + 16));
		const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i + 32)), d = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i + 48));
		_mm_stream_si128(reinterpret_cast<__m128i *>(dst + i), a); _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i + 16), b);
		_mm_stream_si128(reinterpret_cast<__m128i *>(dst + i + 32), c); _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i + 48), d);
	}
	if (i < n) memcpy(dst + i, src + i, n - i);
}

/* the first lines of the NEXT piece a thread will read (another row of the caller's, far from this one): on their way while this
 * piece is copied */
inline void
prefetch_piece(const unsigned char *src)
{
	_mm_prefetch(reinterpret_cast<const char *>(src), _MM_HINT_NTA); _mm_prefetch(reinterpret_cast<const char *>(src + 64), _MM_HINT_NTA);
	_mm_prefetch(reinterpret_cast<const char *>(src + 128), _MM_HINT_NTA); _mm_prefetch(reinterpret_cast<const char *>(src + 192), _MM_HINT_NTA);
}

} /* namespace */

#endif
