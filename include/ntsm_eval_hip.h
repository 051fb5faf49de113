/*
 * include/ntsm_eval_hip.h -- C ABI of the MI355X (gfx950) all-pairs scoring of ntsmEval.
 *
 * The reference (src/CompareCounts.hpp) scores every pair of samples by walking all sites three times per pair:
 *   gatherValidEntries(i, j)                       :1057-1078   sites where both samples have an allele above min_cov
 *   computeLogLikelihood(i, j, valid)              :1093-1099   -2 * (joint - (single_i + single_j)),
 *     computeSumLogPJoint / computeSumLogPSingle   :1013-1033, :968-989   three sequential double sums over the valid sites
 *   calcRelatedness(i, j, valid)                   :1144-1196   genotype tallies (het / hom / shared / ibs0 / ibs2)
 * inside  for i: for j > i  (computeScore, :591-624, an OpenMP loop over i).  This library replaces exactly those calls:
 * one launch returns, for every pair, the three sums, the number of valid sites and the eight tallies; the caller forms
 * score = skew(-2 * (joint - (single1 + single2)), cov1, cov2) / n_valid  (:611-615, :1081-1083), relatedness and
 * homConcord (:1190-1194) and prints (resultsStr, :843-905) -- ntsm_amd/csrc/host/ntsm_eval_main.cpp does.
 * The sums are accumulated in site order with IEEE double operations and no contraction, i.e. bit for bit what one
 * thread of the reference computes.  Parity with the reference itself is UNPINNED.
 */
#ifndef NTSM_EVAL_HIP_H
#define NTSM_EVAL_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ntsm_eval_record {        /* one pair (i < j); 64 bytes */
	double sum_joint, sum_single1, sum_single2;
	uint64_t n_valid;
	uint32_t hets1, homs1, hets2, homs2, shared_hets, shared_homs, ibs0, ibs2;
} ntsm_eval_record;

/* index of pair (i, j), i < j < n, in the output: rows i in order, j ascending inside a row (the reference's loop order) */
static inline uint64_t ntsm_eval_pair_index(uint32_t i, uint32_t j, uint32_t n)
{
	return (uint64_t) i * n - (uint64_t) i * (i + 1) / 2 + (j - i - 1);
}

/* counts: host array [n_samples][n_sites][2] = m_counts (countAT, countCG per site, src/CompareCounts.hpp:99-101).
 * out: host array of n_samples * (n_samples - 1) / 2 records.  kernel_ms (may be NULL): duration of the pair kernel
 * from HIP events.  Returns 0, -1 bad argument, -2 HIP error. */
int ntsm_eval_pairs(int device, const uint32_t *counts, uint32_t n_samples, uint32_t n_sites, uint32_t min_cov,
		ntsm_eval_record *out, double *kernel_ms);

#ifdef __cplusplus
}
#endif
#endif
