/* tools/gunzip_bench.cpp FILE.gz [threads ...] -- throughput of the gzip ingest decoder (gz_stream.hpp) alone: bytes of text
 * per second through GzStream::read with 1 (in order) or n decoder threads, and zlib's gzread beside it.
 * g++ -O3 -std=c++17 -I ntsm_amd/csrc/host tools/gunzip_bench.cpp ntsm_amd/csrc/host/{gz_stream,gz_parallel,inflate,inflate_spec,crc32_fast}.cpp -lz -pthread */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <zlib.h>
#include "gz_stream.hpp"

int main(int argc, char **argv)
{
	if (argc < 2) return 1;
	if (getenv("CHUNK")) ntsm::GzStream::set_parallel_chunk((size_t) atol(getenv("CHUNK")));
	std::vector<unsigned char> buf(4u << 20);
	for (int a = 2; a < argc || a == 2; ++a) {
		const int n = a < argc ? atoi(argv[a]) : 1;
		const auto t0 = std::chrono::steady_clock::now();
		unsigned long long total = 0, sum = 0;
		int r;
		if (n == 0) {
			gzFile f = gzopen(argv[1], "r");
			gzbuffer(f, 1 << 20);
			while ((r = gzread(f, buf.data(), (unsigned) buf.size())) > 0) { total += (unsigned) r; sum += buf[0]; }
			gzclose(f);
		} else {
			ntsm::GzStream::set_decoder_threads((unsigned) n);
			ntsm::GzStream gz;
			if (!gz.open(argv[1])) return 2;
			while ((r = gz.read(buf.data(), (unsigned) buf.size())) > 0) { total += (unsigned) r; sum += buf[0]; }
		}
		const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		uint64_t st[2] = { 0, 0 };
		ntsm::GzStream::last_parallel_stats(st);
		printf("threads %2d (%s): %llu bytes, rc %d, %.3f s, %.2f GB/s of text, spliced %llu dropped %llu (checksum %llu)\n", n, n == 0 ? "zlib gzread" : n == 1 ? "in order" : "parallel",
		       total, r, s, total / s / 1e9, (unsigned long long) st[0], (unsigned long long) st[1], sum);
	}
	return 0;
}
