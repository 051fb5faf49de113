/* oracle/ntsm_eval_oracle_main.c -- TEST INFRASTRUCTURE ONLY: command-line front of ntsm_eval_oracle.c with the flags of
 * ntsmEval that the restated path reads (src/ntSeqMatchEval.cpp:97-157): -s score threshold, -a all, -w skew, -c minimum
 * coverage, -g genome size, -e FILE merge (after the analysis), -o merge only.  stdout = what ntsmEval prints on stdout. */
#include <stdlib.h>
#include <string.h>
#include "ntsm_eval_oracle.h"

int main(int argc, char **argv)
{
	double thresh = 0.5, skew = 0.2;                 /* src/Options.h:48-49 */
	unsigned min_cov = 1;                            /* :52 */
	unsigned long long genome = 6200000000ull;       /* :55 */
	int all = 0, i = 1, only_merge = 0;
	const char *merge = NULL;
	for (; i < argc && argv[i][0] == '-' && argv[i][1]; ++i) {
		if (!strcmp(argv[i], "-a")) all = 1;
		else if (!strcmp(argv[i], "-o")) only_merge = 1;
		else if (i + 1 < argc && !strcmp(argv[i], "-e")) merge = argv[++i];
		else if (i + 1 < argc && !strcmp(argv[i], "-s")) thresh = atof(argv[++i]);
		else if (i + 1 < argc && !strcmp(argv[i], "-w")) skew = atof(argv[++i]);
		else if (i + 1 < argc && !strcmp(argv[i], "-c")) min_cov = (unsigned) strtoul(argv[++i], NULL, 10);
		else if (i + 1 < argc && !strcmp(argv[i], "-g")) genome = strtoull(argv[++i], NULL, 10);
		else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
	}
	if (i >= argc) { fprintf(stderr, "Error: Need Input File\n"); return 1; }
	ntsm_eval_oracle *e = ntsm_eval_oracle_load((const char *const *) (argv + i), (unsigned) (argc - i));
	if (!e) { fprintf(stderr, "parse error\n"); return 134; }
	const unsigned n = ntsm_eval_oracle_samples(e);
	if (n == 1 || !only_merge) ntsm_eval_oracle_print(e, stdout, min_cov, thresh, all, skew, genome);
	if (n > 1 && only_merge && !merge) { fprintf(stderr, "(-l) cannot be used without --merge (-e) option.\n"); return 1; }
	if (n > 1 && merge) {                                 /* src/ntSeqMatchEval.cpp:310-341: only with more than one file */
		FILE *out = fopen(merge, "w");
		if (!out || ntsm_eval_oracle_merge(e, out)) return 134;
		fclose(out);
	}
	ntsm_eval_oracle_free(e);
	return 0;
}
