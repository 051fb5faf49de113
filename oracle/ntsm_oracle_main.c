/*
 * oracle/ntsm_oracle_main.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Command-line front end of the CPU restatement; same observable behaviour as the reference
 * main for the flags that reach the counting path (src/ntSeqMatchCount.cpp:53-185):
 *   ntsm_oracle -s sites.fa [-k K] [-m M] [-t T] [-d] [-o summary] [-v] reads...
 * stdout = counts.txt, stderr = warnings + summary.  `--time-scan` additionally prints
 * "SCAN_SECONDS <s> BASES <n>" for bench.py's cpu_baseline leg (site-table build excluded).
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
	int dupes = 0, time_scan = 0, n_files = 0;
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
		else files[n_files++] = argv[i];
	}
	if (!sites || n_files == 0) {
		fprintf(stderr, "usage: ntsm_oracle -s sites.fa [-k K] [-m M] [-d] [-o F] reads...\n");
		return 1;
	}
	ntsm_oracle_fp *fp = ntsm_oracle_fp_create(sites, k, cov, dupes, stderr);
	if (!fp) return 1;
	double t0 = now_s();
	if (ntsm_oracle_fp_compute_counts(fp, files, n_files, stderr)) return 1;
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
