/* Runs the gzip decoder thread over every file given (tests/test_host_cpu.py builds this with ASan + UBSan and feeds
 * it corrupted streams): prints "<bytes> <status>" per file. */
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../ntsm_amd/csrc/host/gz_stream.hpp"

int main(int argc, char **argv)
{
	std::vector<unsigned char> buf(1 << 16);
	if (const char *t = getenv("NTSM_DECODER_THREADS")) ntsm::GzStream::set_decoder_threads((unsigned) atoi(t));   /* BGZF: block-parallel; plain gzip: chunk-parallel */
	if (const char *c = getenv("NTSM_PARALLEL_CHUNK")) ntsm::GzStream::set_parallel_chunk((size_t) atol(c));     /* compressed bytes per chunk of the latter */
	const unsigned long long stop_after = getenv("NTSM_STOP_AFTER") ? strtoull(getenv("NTSM_STOP_AFTER"), nullptr, 10) : ~0ull;   /* reader walks away early */
	for (int i = 1; i < argc; ++i) {
		ntsm::GzStream gz;
		if (!gz.open(argv[i])) { printf("open-failed\n"); continue; }
		unsigned long long total = 0;
		int r;
		while ((r = gz.read(buf.data(), (unsigned) buf.size())) > 0) { total += (unsigned long long) r; if (total >= stop_after) break; }
		printf("%llu %d\n", total, r);
	}
	return 0;
}
