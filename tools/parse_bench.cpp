/* tools/parse_bench.cpp FILE.fq [reps] -- the plain-FASTQ parse loop alone on one thread, file in memory: record boundaries
 * (ParallelFastq::strict_record: four memchr calls) and boundaries + 2-bit packing (pack2_append).  MI355X box (EPYC 9575F),
 * 31 MB of text (cache resident): 13 + 13 ns per 315-byte record; 1.9 GB: 13-17 + 14-18 ns.  A predict-and-verify
 * scanner (line ends where the previous record had them, checked with straight-line vector compares) did the boundaries in
 * 6 ns in cache and in the same 14-16 ns out of it, and the two together were no faster than this (profiles/r04_parse/):
 * the loop waits for memory, not for memchr -- not adopted.
 * g++ -O3 -std=c++17 -I ntsm_amd/csrc/host tools/parse_bench.cpp ntsm_amd/csrc/host/{parallel_fastq,pack2}.cpp -pthread -o build/parse_bench */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pack2.hpp"
#include "parallel_fastq.hpp"
using namespace ntsm;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
	if (argc < 2) return 1;
	const int reps = argc > 2 ? atoi(argv[2]) : 3;
	FILE *f = fopen(argv[1], "rb");
	if (!f) return 2;
	std::vector<char> d;
	{ char buf[1 << 16]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n); }
	fclose(f);
	const char *b = d.data(), *e = b + d.size();
	std::vector<uint8_t> codes((8u << 20) / 4 + 64), valid((8u << 20) / 8 + 64);
	for (int r = 0; r < reps; ++r) {
		uint64_t n = 0, bases = 0;
		double t0 = now();
		for (const char *p = b; p < e;) { const char *s; uint64_t l; p = ParallelFastq::strict_record(p, e, &s, &l); if (!p) break; ++n; bases += l; }
		const double t1 = now() - t0;
		t0 = now();
		uint64_t pos = 0, n2 = 0;
		for (const char *p = b; p < e;) {
			const char *s; uint64_t l;
			p = ParallelFastq::strict_record(p, e, &s, &l);
			if (!p) break;
			if (pack2_extent(pos, l) > (8u << 20)) pos = 0;
			pos = pack2_append(codes.data(), valid.data(), pos, s, l);
			++n2;
		}
		const double t2 = now() - t0;
		printf("%llu records, %llu bases: boundaries %.1f ns/record (%.2f GB/s of text), boundaries + pack %.1f ns/record (%.2f GB/s) [%s]\n", (unsigned long long) n, (unsigned long long) bases,
		       1e9 * t1 / n, d.size() / t1 / 1e9, 1e9 * t2 / n2, d.size() / t2 / 1e9, pack2_impl());
	}
	return 0;
}
