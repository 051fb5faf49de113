/* tools/first_touch_bench.cpp THREADS MiB_PER_THREAD -- what writing into memory that has never been touched costs on this
 * host, against writing into a small buffer that is reused (the two situations of early_ingest.hpp and of the pinned lane
 * slots): N threads each write M MiB in 3 MiB pieces, (a) every piece freshly allocated with malloc, (b) the same with a
 * 2 MiB-aligned allocation advised MADV_HUGEPAGE, (c) one piece per thread, reused, (d) pieces from one big mmap populated
 * beforehand by the same threads (time of the population reported separately).
 * g++ -O2 -std=c++17 -pthread tools/first_touch_bench.cpp -o build/first_touch_bench */
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
	const int n = argc > 1 ? atoi(argv[1]) : 16;
	const size_t mib = argc > 2 ? (size_t) atol(argv[2]) : 140, piece = 3u << 20, pieces = mib * (1u << 20) / piece;
	auto run = [&](const char *name, int mode) {
		std::vector<void *> big(n, nullptr);
		double t_pop = 0;
		if (mode == 3) {
			const double t0 = now();
			std::vector<std::thread> th;
			for (int t = 0; t < n; ++t) th.emplace_back([&, t]() {
				big[t] = mmap(nullptr, pieces * piece, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
			});
			for (auto &x : th) x.join();
			t_pop = now() - t0;
		}
		const double t0 = now();
		std::vector<std::thread> th;
		for (int t = 0; t < n; ++t) th.emplace_back([&, t]() {
			void *reuse = mode == 2 ? malloc(piece) : nullptr;
			std::vector<void *> keep;
			for (size_t p = 0; p < pieces; ++p) {
				void *m;
				if (mode == 0) m = malloc(piece);
				else if (mode == 1) { m = aligned_alloc(2u << 20, 4u << 20); madvise(m, 4u << 20, MADV_HUGEPAGE); }
				else if (mode == 2) m = reuse;
				else m = (char *) big[t] + p * piece;
				memset(m, (int) p + 1, piece);
				if (mode < 2) keep.push_back(m);
			}
			for (void *m : keep) free(m);
			free(reuse);
		});
		for (auto &x : th) x.join();
		const double dt = now() - t0;
		printf("%-44s %2d threads x %zu MiB: %.3f s = %.1f GB/s%s", name, n, mib, dt, n * pieces * piece / dt / 1e9, mode == 3 ? "" : "\n");
		if (mode == 3) { printf("  (+ %.3f s to populate)\n", t_pop); for (void *b : big) munmap(b, pieces * piece); }
	};
	run("fresh malloc per 3 MiB piece", 0);
	run("fresh 2 MiB-aligned + MADV_HUGEPAGE per piece", 1);
	run("one reused 3 MiB piece per thread", 2);
	run("pieces of a pre-populated mapping", 3);
	return 0;
}
