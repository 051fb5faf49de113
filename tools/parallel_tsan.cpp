/* Driver for running the block-parallel FASTQ ingest under ThreadSanitizer (tests/test_host_cpu.py):
 *   parallel_tsan FILE THREADS BLOCK_BYTES        ->  "reads=N bytes=M parallel=P resume=R fnv=..."
 *   parallel_tsan FILE.gz DECODERS PARSERS SINK CHUNK  (five arguments: the parallel gzip ingest, parallel_gz_fastq.hpp)
 *                                                  ->  "reads=N bytes=M parallel=P pieces=Q status=S sum=..." (sum: order-free)
 *   parallel_tsan SITES.fa sites K                 (the site loader's threads: k-merising, bucketed sorts, allele lists)
 *                                                  ->  "keys=N sites=S erased=E fnv=..." */
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../include/ntsm_host.h"

int main(int argc, char **argv)
{
	if (argc < 4) return 2;
	uint8_t *bases = nullptr;
	uint64_t *ends = nullptr, nb = 0, nr = 0, nblk = 0, npar = 0, resume = 0;
	if (!strcmp(argv[2], "sites")) {
		ntsm_sites *st = nullptr;
		int rc = ntsm_sites_load(argv[1], (unsigned) atoi(argv[3]), argc > 4 ? atoi(argv[4]) : 0, &st);
		if (rc) { fprintf(stderr, "rc=%d\n", rc); return 1; }
		uint64_t h = 1469598103934665603ull;
		const uint64_t *keys = ntsm_sites_keys(st);
		for (uint64_t i = 0; i < ntsm_sites_n_keys(st); ++i) h = (h ^ keys[i]) * 1099511628211ull;
		printf("keys=%llu sites=%llu erased=%llu fnv=%016llx\n", (unsigned long long) ntsm_sites_n_keys(st), (unsigned long long) ntsm_sites_n_sites(st),
				(unsigned long long) ntsm_sites_n_erased(st), (unsigned long long) h);
		ntsm_sites_free(st);
		return 0;
	}
	if (argc >= 8) {   /* parallel_tsan FILE early PARSERS DECODERS BLOCK CHUNK_POSITIONS BUDGET CONSUMERS: early_ingest.hpp */
		uint8_t *text = nullptr;
		uint64_t nt = 0, nreads = 0, nbases = 0, npar2 = 0;
		uint64_t nrest = 0;                                   /* a tenth argument: hand the gzip stream over after that many chunks */
		int rc = ntsm_host_early_ingest_hand_over(argv[1], (unsigned) atoi(argv[3]), (unsigned) atoi(argv[4]), strtoull(argv[5], nullptr, 10), strtoull(argv[6], nullptr, 10),
				strtoull(argv[7], nullptr, 10), argc > 8 ? (unsigned) atoi(argv[8]) : 2u, argc > 9 ? strtoull(argv[9], nullptr, 10) : ~0ull, &text, &nt, &nreads, &nbases, &npar2, &nrest);
		if (rc) { fprintf(stderr, "rc=%d\n", rc); return 1; }
		uint64_t letters = 0;
		for (uint64_t i = 0; i < nt; ++i) letters += text[i] != 'N';
		printf("reads=%llu bases=%llu letters=%llu\n", (unsigned long long) nreads, (unsigned long long) nbases, (unsigned long long) letters);
		ntsm_host_free(text);
		return 0;
	}
	if (argc >= 6) {
		int st = 0;
		ntsm_host_gunzip_parallel_chunk(strtoull(argv[5], nullptr, 10));
		int rc = ntsm_host_flatten_parallel_gz(argv[1], (unsigned) atoi(argv[2]), (unsigned) atoi(argv[3]), strtoull(argv[4], nullptr, 10), &bases, &nb, &ends, &nr,
				&nblk, &npar, &st);
		if (rc) { fprintf(stderr, "rc=%d\n", rc); return 1; }
		/* order-free digest: sum over reads of an FNV hash of the read */
		uint64_t sum = 0, start = 0;
		for (uint64_t r = 0; r < nr; ++r) {
			uint64_t h = 1469598103934665603ull;
			for (uint64_t i = start; i < ends[r]; ++i) h = (h ^ bases[i]) * 1099511628211ull;
			sum += h;
			start = ends[r] + 1;
		}
		printf("reads=%llu bytes=%llu parallel=%llu pieces=%llu status=%d sum=%016llx\n", (unsigned long long) nr, (unsigned long long) nb,
				(unsigned long long) npar, (unsigned long long) nblk, st, (unsigned long long) sum);
		ntsm_host_free(bases);
		ntsm_host_free(ends);
		return 0;
	}
	int rc = ntsm_host_flatten_parallel(argv[1], (unsigned) atoi(argv[2]), strtoull(argv[3], nullptr, 10), &bases, &nb, &ends, &nr,
			&nblk, &npar, &resume);
	if (rc) { fprintf(stderr, "rc=%d\n", rc); return 1; }
	uint64_t h = 1469598103934665603ull;
	for (uint64_t i = 0; i < nb; ++i) h = (h ^ bases[i]) * 1099511628211ull;
	printf("reads=%llu bytes=%llu parallel=%llu resume=%llu fnv=%016llx\n", (unsigned long long) nr, (unsigned long long) nb,
			(unsigned long long) npar, (unsigned long long) resume, (unsigned long long) h);
	ntsm_host_free(bases);
	ntsm_host_free(ends);
	return 0;
}
