/*
 * oracle/ntsm_oracle_main.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Command-line front end of the CPU restatement; same observable behaviour as the reference
 * main for the flags that reach the counting path (src/ntSeqMatchCount.cpp:53-185):
 *   ntsm_oracle -s sites.fa [-k K] [-m M] [-t T] [-d] [-o summary] [-v] reads...
 * stdout = counts.txt, stderr = warnings + summary.  `--time-scan` additionally prints
 * "SCAN_SECONDS <s> BASES <n>" for bench.py's cpu_baseline leg (site-table build excluded).
 * `--insert-multiplier M` replaces computeCounts by "every record through insertCount(seq, len, M)" -- the reference's
 * own third parameter (src/FingerPrint.hpp:89) -- exactly like oracle/ref_driver.cpp under NTSM_REF_INSERT_MULTIPLIER:
 * the way to per-k-mer counts beyond 2^32 on a small input (tests/golden/make_wrap.py, printCountsMax's `unsigned`).
 */
#define _POSIX_C_SOURCE 200809L
#include "ntsm_oracle.h"
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

int main(int argc, char **argv)
{
	const char *sites = NULL, *summary = NULL;
	unsigned k = 19;                              /* Options.h:24 */
	double cov = DBL_MAX;                         /* Options.h:32 */
	int dupes = 0, time_scan = 0, n_files = 0, use_mult = 0;
	unsigned mult = 1;
	const char **files = (const char **) calloc((size_t) argc, sizeof(char *));
	for (int i = 1; i < argc; ++i) {
		if (!strcmp(argv[i], "-s") && i + 1 < argc) sites = argv[++i];
		else if (!strcmp(argv[i], "-k") && i + 1 < argc) k = (unsigned) strtoul(argv[++i], NULL, 10);
		else if (!strcmp(argv[i], "-m") && i + 1 < argc) cov = strtod(argv[++i], NULL);
		else if (!strcmp(argv[i], "-t") && i + 1 < argc) ++i;   /* one thread is the defined schedule */
		else if (!strcmp(argv[i], "-o") && i + 1 < argc) summary = argv[++i];
		else if (!strcmp(argv[i], "-d")) dupes = 1;
		else if (!strcmp(argv[i], "-v")) { }
		else if (!strcmp(argv[i], "--time-scan")) time_scan = 1;
		else if (!strcmp(argv[i], "--insert-multiplier") && i + 1 < argc) { use_mult = 1; mult = (unsigned) strtoul(argv[++i], NULL, 10); }
		else files[n_files++] = argv[i];
	}
	if (!sites || n_files == 0) {
		fprintf(stderr, "usage: ntsm_oracle -s sites.fa [-k K] [-m M] [-d] [-o F] reads...\n");
		return 1;
	}
	ntsm_oracle_fp *fp = ntsm_oracle_fp_create(sites, k, cov, dupes, stderr);
	if (!fp) return 1;
	double t0 = now_s();
	if (use_mult) {
		for (int i = 0; i < n_files; ++i) {
			ntsm_oracle_reader *r = ntsm_oracle_reader_open(files[i]);
			if (!r) { fprintf(stderr, "file %s cannot be opened\n", files[i]); return 1; }
			for (int64_t l = ntsm_oracle_reader_next(r); l >= 0; l = ntsm_oracle_reader_next(r))
				ntsm_oracle_fp_insert_count_mult(fp, ntsm_oracle_reader_seq(r), (uint64_t) l, mult);
			ntsm_oracle_reader_close(r);
		}
	} else if (ntsm_oracle_fp_compute_counts(fp, files, n_files, stderr)) return 1;
	double t1 = now_s();
	if (ntsm_oracle_fp_print_counts(fp, stdout) != 0) {
		fflush(stdout);
		fprintf(stderr, "terminate: Couldn't find key.\n");   /* reference aborts here (exit 134) */
		return 134;
	}
	char buf[1024];
	ntsm_oracle_fp_info_summary(fp, buf, sizeof buf, stderr);
	if (summary) {
		FILE *fh = fopen(summary, "w");
		if (fh) { fputs(buf, fh); fclose(fh); }
	}
	fprintf(stderr, "%s\n", buf);
	if (time_scan)
		fprintf(stderr, "SCAN_SECONDS %.6f BASES %llu\n", t1 - t0,
				(unsigned long long) ntsm_oracle_fp_total_bases(fp));
	ntsm_oracle_fp_destroy(fp);
	free(files);
	return 0;
}
