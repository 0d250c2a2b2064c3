// The host thread pool of mdemod_process_host (csrc/pack_pool.h: PackPool, parallel_streams) on its own, built twice by
// tests/test_sanitize.py: with -fsanitize=thread (`pool_test race`) and with -fsanitize=address,undefined (`pool_test fork`).
// Test infrastructure; nothing here is part of the product.
//
//   race: several caller threads - one library context per GPU is one host thread each - share the pool and run jobs that write
//         disjoint pieces of their own buffers; every piece must have been written exactly once, and TSan must stay silent.
//   fork: a child forked while the pool exists (a) still gets its jobs done, alone, and (b) leaves through exit() - static
//         destructors run - without hanging on threads that do not exist in it (~PackPool joined them).
#include <atomic>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <sys/wait.h>

#include "../../meteor_demod_amd/csrc/pack_pool.h"

static int
race(int callers, int rounds)
{
	std::atomic<int> bad{ 0 };
	std::vector<std::thread> th;
	for (int c = 0; c < callers; c++) th.emplace_back([&, c] {
		for (int r = 0; r < rounds; r++) {
			/* pool.run directly */
			const unsigned n_jobs = 1 + (unsigned)((c * 7 + r * 13) % 40);
			std::vector<int> hits(n_jobs, 0);
			PackPool::get().run(n_jobs, [&](unsigned i) { hits[i]++; });
			for (int h : hits) if (h != 1) bad++;
			/* parallel_streams: above the threshold (1 MB), ragged weights (some streams empty) */
			const uint32_t n = 50 + (uint32_t)((c + r) % 200);
			std::vector<uint64_t> prefix(n);
			uint64_t acc = 0;
			for (uint32_t s = 0; s < n; s++) { acc += (s % 7 == 3) ? 0 : (400000 + 1000 * ((s * 31 + r) % 97)); prefix[s] = acc; }
			std::vector<unsigned char> seen(n, 0);
			parallel_streams(n, prefix, [&](uint32_t first, uint32_t last) { for (uint32_t s = first; s < last; s++) seen[s]++; });
			for (unsigned char v : seen) if (v != 1) bad++;
			/* below the threshold: the caller alone */
			std::vector<uint64_t> small(4, 100);
			std::partial_sum(small.begin(), small.end(), small.begin());
			int calls = 0;
			parallel_streams(4, small, [&](uint32_t first, uint32_t last) { calls += (int)(last - first); });
			if (calls != 4) bad++;
		}
	});
	for (auto &t : th) t.join();
	printf("race: %d callers x %d rounds on up to %u pools of %u, %d bad\n", callers, rounds, PackPool::max_pools(), PackPool::get().size(), bad.load());
	return bad.load() ? 1 : 0;
}

/* `callers` threads each hold a lease while all of them are inside a job at the same time (a barrier inside the jobs): possible only
 * if every caller got a pool of its own - with one shared pool the second caller would wait for the first one's job to end, which
 * waits for the second to arrive: this test would hang (the harness bounds it). */
static int
overlap(int callers)
{
	std::atomic<int> inside{ 0 }, bad{ 0 };
	std::vector<std::thread> th;
	for (int c = 0; c < callers; c++) th.emplace_back([&] {
		PackPool::Lease l = PackPool::lease();
		std::vector<int> hits(8, 0);
		l.run(8, [&](unsigned i) {
			if (i == 0) { inside++; while (inside.load() < callers) std::this_thread::yield(); }
			hits[i]++;
		});
		for (int h : hits) if (h != 1) bad++;
	});
	for (auto &t : th) t.join();
	printf("overlap: %d callers inside their jobs at once on %u pools, %d bad\n", callers, PackPool::max_pools(), bad.load());
	return bad.load() ? 1 : 0;
}

static int
fork_exit()
{
	std::vector<int> warm(64, 0);
	PackPool::get().run(64, [&](unsigned i) { warm[i] = 1; });          /* the pool exists, its workers are parked */
	fflush(stdout);
	const pid_t pid = fork();
	if (pid < 0) { perror("fork"); return 1; }
	if (pid == 0) {
		std::vector<int> hits(32, 0);
		PackPool::get().run(32, [&](unsigned i) { hits[i]++; });        /* no workers here: the child works alone */
		for (int h : hits) if (h != 1) _exit(7);
		exit(0);                                                        /* static destructors run: ~PackPool in a process without its threads */
	}
	for (int waited = 0; waited < 100; waited++) {
		int status = 0;
		const pid_t r = waitpid(pid, &status, WNOHANG);
		if (r == pid) {
			const int rc = WIFEXITED(status) ? WEXITSTATUS(status) : 100 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0);
			printf("fork: child left with %d\n", rc);
			return rc;
		}
		usleep(100000);
	}
	kill(pid, SIGKILL);
	waitpid(pid, nullptr, 0);
	printf("fork: the child hung at exit\n");
	return 1;
}

/* stream_copy (non-temporal stores behind an aligning head and a tail) against memcpy: every destination alignment, source alignment
 * and length around its 16- and 64-byte steps, with guard bytes either side */
static int
copy_check()
{
	std::vector<unsigned char> src(4096), dst(4096), ref(4096);
	for (size_t i = 0; i < src.size(); i++) src[i] = (unsigned char)(i * 131 + 7);
	unsigned long long cases = 0;
	for (size_t da = 0; da < 32; da++)
		for (size_t sa = 0; sa < 17; sa += 1)
			for (size_t n : { 0ul, 1ul, 2ul, 15ul, 16ul, 17ul, 31ul, 47ul, 48ul, 63ul, 64ul, 65ul, 79ul, 80ul, 127ul, 128ul, 129ul, 640ul, 1280ul, 1281ul, 2047ul }) {
				std::fill(dst.begin(), dst.end(), 0xEE); std::fill(ref.begin(), ref.end(), 0xEE);
				unsigned char *d = dst.data() + 64 + da;
				/* (the vectors' storage is 16-byte aligned by the allocator: da really is the misalignment) */
				stream_copy(d, src.data() + sa, n);
				_mm_sfence();
				memcpy(ref.data() + 64 + da, src.data() + sa, n);
				if (dst != ref) { printf("copy: differs at dst+%zu src+%zu n=%zu\n", da, sa, n); return 1; }
				cases++;
			}
	prefetch_piece(src.data());
	printf("copy: %llu cases equal memcpy, guards intact\n", cases);
	return 0;
}

int
main(int argc, char **argv)
{
	if (argc > 1 && !strcmp(argv[1], "copy")) return copy_check();
	if (argc > 1 && !strcmp(argv[1], "race")) return race(argc > 2 ? atoi(argv[2]) : 6, argc > 3 ? atoi(argv[3]) : 200);
	if (argc > 1 && !strcmp(argv[1], "fork")) return fork_exit();
	if (argc > 1 && !strcmp(argv[1], "cpus")) { printf("usable %u pool %u pools %u\n", PackPool::usable_cpus(), PackPool::get().size(), PackPool::max_pools()); return 0; }
	if (argc > 1 && !strcmp(argv[1], "overlap")) return overlap(argc > 2 ? atoi(argv[2]) : 3);
	fprintf(stderr, "usage: pool_test race [callers] [rounds] | fork | copy\n");
	return 2;
}
