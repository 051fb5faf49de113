/* tools/inflate_bench.cpp FILE.gz [reps] -- the two DEFLATE inner loops alone, steady state (buffers reused, one thread):
 *   run    the in-order decoder (inflate.hpp), bytes out;
 *   run16  the speculative decoder (inflate_spec.hpp) from the first dynamic block it finds, 16-bit symbols out, plus
 *          resolve() + CRC-32 of the result (what a chunk worker of gz_parallel.cpp does per chunk), in chunks of CAP_M M
 *          symbols (default 6: what 1 MiB of compressed FASTQ inflates to).
 * g++ -O3 -std=c++17 -I ntsm_amd/csrc/host tools/inflate_bench.cpp ntsm_amd/csrc/host/{inflate,inflate_spec,crc32_fast}.cpp -o build/inflate_bench -lz */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "crc32_fast.hpp"
#include "inflate_spec.hpp"

using namespace ntsm;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
	if (argc < 2) return 1;
	const int reps = argc > 2 ? atoi(argv[2]) : 3;
	FILE *f = fopen(argv[1], "rb");
	if (!f) return 2;
	std::vector<uint8_t> d;
	{
		uint8_t buf[1 << 16];
		size_t n;
		while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
	}
	fclose(f);
	d.resize(d.size() + 64);
	const uint8_t *base = d.data(), *end = d.data() + d.size() - 64;
	const size_t W = 32768, P = 1u << 20;
	std::vector<uint8_t> work(W + P + 1024);
	for (int r = 0; r < reps; ++r) {
		Inflate inf;
		inf.reset(base + 10, end);                            /* plain 10-byte gzip header (what bench.pigz_like writes) */
		size_t out = 0;
		uint64_t total = 0, sum = 0;
		const double t0 = now();
		for (;;) {
			const Inflate::Status st = inf.run(work.data(), &out, W + P);
			if (st == Inflate::MORE) {
				if (out >= W + P) { sum += work[out - 1]; total += out - W; memmove(work.data(), work.data() + out - W, W); out = W; }
				continue;
			}
			total += out - (total ? W : 0);
			if (st != Inflate::STREAM_END) { printf("run: status %d\n", (int) st); return 3; }
			break;
		}
		const double t = now() - t0;
		printf("run   : %.0f MB in %.3f s = %.3f GB/s  (%llu)\n", total / 1e6, t, total / t / 1e9, (unsigned long long) sum);
	}
	const size_t cap = W + ((size_t) (getenv("CAP_M") ? atoi(getenv("CAP_M")) : 6) << 20);   /* symbols per chunk: 1 MiB of compressed FASTQ is about 6 M */
	std::vector<uint16_t> sym(cap + 1024);
	std::vector<uint8_t> bytes(cap), window(W, 'A');
	SpecInflate::fill_markers(sym.data());
	for (int r = 0; r < reps; ++r) {
		SpecInflate sp;
		uint64_t from = 10 * 8 + 8 * 1024, total = 0, crc = 0;
		double t_find = 0, t_dec = 0, t_res = 0;
		for (;;) {
			double t0 = now();
			const uint64_t b = sp.find(base, end, from, (uint64_t) (end - base) * 8);
			t_find += now() - t0;
			if (b == ~0ull) break;
			sp.set_stop(base, ~0ull);
			size_t out = W;
			t0 = now();
			const Inflate::Status st = sp.run16(sym.data(), &out, cap);
			t_dec += now() - t0;
			const size_t n = out - W;
			t0 = now();
			uint32_t c = 0;
			SpecInflate::resolve(sym.data() + W, n, window.data(), W, bytes.data(), &c);
			crc ^= c;
			t_res += now() - t0;
			total += n;
			if (st != Inflate::MORE) break;
			from = sp.bit_pos(base);                            /* buffer full: next chunk from the next block the finder sees */
		}
		printf("run16 : %.0f M symbols, decode %.3f s = %.3f G/s, resolve + crc %.3f s = %.2f G/s, find %.3f s  (%llx)\n", total / 1e6, t_dec, total / t_dec / 1e9,
		       t_res, total / t_res / 1e9, t_find, (unsigned long long) crc);
	}
	return 0;
}
