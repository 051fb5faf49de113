/*
 * oracle/ntsm_eval_oracle.h -- TEST INFRASTRUCTURE ONLY: CPU restatement of ntsmEval's all-pairs scoring
 * (src/CompareCounts.hpp).  PARITY UNPINNED: see ntsm_eval_oracle.c.  Only tests/ may use it.
 */
#ifndef NTSM_EVAL_ORACLE_H
#define NTSM_EVAL_ORACLE_H
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ntsm_eval_oracle ntsm_eval_oracle;

/* what one pair of samples yields before the score is formed (same fields as include/ntsm_eval_hip.h's record) */
typedef struct ntsm_eval_pair {
	double sum_joint, sum_single1, sum_single2;
	uint64_t n_valid;
	uint32_t hets1, homs1, hets2, homs2, shared_hets, shared_homs, ibs0, ibs2;
} ntsm_eval_pair;

ntsm_eval_oracle *ntsm_eval_oracle_load(const char *const *files, unsigned n_files);   /* NULL: a parse error the reference would throw on */
void ntsm_eval_oracle_free(ntsm_eval_oracle *e);
unsigned ntsm_eval_oracle_samples(const ntsm_eval_oracle *e);
unsigned ntsm_eval_oracle_sites(const ntsm_eval_oracle *e);
const unsigned *ntsm_eval_oracle_counts(const ntsm_eval_oracle *e);      /* [sample][site][2] */
const unsigned *ntsm_eval_oracle_sums(const ntsm_eval_oracle *e);
const unsigned *ntsm_eval_oracle_distinct(const ntsm_eval_oracle *e);    /* [site][2] */
uint64_t ntsm_eval_oracle_total(const ntsm_eval_oracle *e, unsigned i);
uint64_t ntsm_eval_oracle_raw_total(const ntsm_eval_oracle *e, unsigned i);
unsigned ntsm_eval_oracle_kmer_size(const ntsm_eval_oracle *e, unsigned i);
void ntsm_eval_oracle_genotype(const ntsm_eval_oracle *e, unsigned i, unsigned min_cov, unsigned out[3]);   /* hets, homs, miss */
double ntsm_eval_oracle_error_rate(const ntsm_eval_oracle *e, unsigned i, uint64_t genome_size);
void ntsm_eval_oracle_pair(const ntsm_eval_oracle *e, unsigned i1, unsigned i2, unsigned min_cov, ntsm_eval_pair *r);
double ntsm_eval_oracle_score(const ntsm_eval_pair *r, double cov1, double cov2, double cov_skew);
int ntsm_eval_oracle_merge(const ntsm_eval_oracle *e, FILE *out);       /* mergeCounts; -1: samples with different k */
int ntsm_eval_oracle_print(const ntsm_eval_oracle *e, FILE *out, unsigned min_cov, double score_thresh, int all, double cov_skew, uint64_t genome_size);

#ifdef __cplusplus
}
#endif
#endif
