/*
 * ntsm_eval.hip -- all-pairs scoring of ntsmEval on one MI355X (include/ntsm_eval_hip.h; reference:
 * src/CompareCounts.hpp:591-624 computeScore and the functions it calls).
 *
 * Layout.  The reference keeps m_counts[sample][site]; a pair walks two rows.  Here the counts are transposed once to
 * [site][sample] and one thread owns one pair (i, j): a 256-thread workgroup takes 256 consecutive j against TI = 4
 * consecutive i, so at every site the j side is one coalesced 4-byte load per array and the i side is wave-uniform
 * (scalar loads), and each j value loaded is used for four pairs.  The single-sample term of a site,
 *   first * freqAT + second * freqCG   (computeSumLogPSingle, :971-987),
 * does not depend on the partner, so it is computed once per (sample, site) by the prepare kernel with the reference's
 * expression and the pair kernel only adds it when the site is valid for the pair -- the same additions in the same
 * order as the reference's loop over validIndexes.
 * Arithmetic: IEEE double, __dadd_rn / __dmul_rn / __ddiv_rn (no fused multiply-add): the reference binary is built
 * without FMA contraction on x86-64 and every sum is sequential, so the results are bit-identical by construction.
 * Bound: the vector ALUs (two correctly rounded double divisions per pair and site); the 12 bytes per j and site come
 * from L2 for all but the first of the workgroups that share a j range.
 */
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../include/ntsm_eval_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTI = 4;                       /* samples i per workgroup row */

/* [sample][site][2] -> at[site][sample], cg[site][sample], term[site][sample] */
__global__ __launch_bounds__(kThreads) void ntsm_eval_prepare(const uint32_t *counts, uint32_t n_samples, uint32_t n_sites, uint32_t min_cov,
		uint32_t *at, uint32_t *cg, double *term)
{
	const uint64_t cell = (uint64_t) blockIdx.x * kThreads + threadIdx.x;       /* site-major: neighbouring threads = neighbouring samples */
	if (cell >= (uint64_t) n_samples * n_sites) return;
	const uint32_t site = (uint32_t) (cell / n_samples), s = (uint32_t) (cell % n_samples);
	const uint32_t a0 = counts[((uint64_t) s * n_sites + site) * 2], a1 = counts[((uint64_t) s * n_sites + site) * 2 + 1];
	/* src/CompareCounts.hpp:971-987 */
	double fAT = 0, fCG = 0;
	if (a0 > min_cov) fAT = __ddiv_rn((double) a0, (double) (a0 + a1));
	if (a1 > min_cov) fCG = __ddiv_rn((double) a1, (double) (a0 + a1));
	at[cell] = a0;
	cg[cell] = a1;
	term[cell] = __dadd_rn(__dmul_rn((double) a0, fAT), __dmul_rn((double) a1, fCG));
}

struct PairAcc {
	double joint = 0, s1 = 0, s2 = 0;
	uint32_t n = 0, hets1 = 0, homs1 = 0, hets2 = 0, homs2 = 0, sh_het = 0, sh_hom = 0, ibs0 = 0;
};

__global__ __launch_bounds__(kThreads) void ntsm_eval_pair_kernel(const uint32_t *__restrict__ at, const uint32_t *__restrict__ cg, const double *__restrict__ term,
		uint32_t n_samples, uint32_t n_sites, uint32_t min_cov, ntsm_eval_record *__restrict__ out)
{
	const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t i0 = blockIdx.y * kTI;
	if (blockIdx.x * blockDim.x + (blockDim.x - 1) <= i0) return;               /* the whole tile lies on or below the diagonal */
	const uint32_t jc = j < n_samples ? j : n_samples - 1;                      /* lanes past the end compute a copy of the last sample, not stored */
	PairAcc acc[kTI];
	for (uint32_t site = 0; site < n_sites; ++site) {
		const uint64_t row = (uint64_t) site * n_samples;
		const uint32_t b0 = at[row + jc], b1 = cg[row + jc];
		const double tj = term[row + jc];
		const bool vj = b0 > min_cov || b1 > min_cov;
		const bool hetj = b0 > min_cov && b1 > min_cov;
#pragma unroll
		for (int ii = 0; ii < kTI; ++ii) {
			const uint32_t i = i0 + ii < n_samples ? i0 + ii : n_samples - 1;   /* wave-uniform: scalar loads */
			const uint32_t a0 = at[row + i], a1 = cg[row + i];
			const double ti = term[row + i];
			const bool vi = a0 > min_cov || a1 > min_cov;
			if (!(vi && vj)) continue;                                          /* gatherValidEntries, :1057-1078 */
			PairAcc &A = acc[ii];
			A.n++;
			/* computeSumLogPJoint, :1018-1031 */
			const uint32_t cAT = a0 + b0, cCG = a1 + b1;
			const double den = (double) (cAT + cCG);
			double fAT = 0, fCG = 0;
			if (cAT > min_cov) fAT = __ddiv_rn((double) cAT, den);
			if (cCG > min_cov) fCG = __ddiv_rn((double) cCG, den);
			A.joint = __dadd_rn(A.joint, __dadd_rn(__dmul_rn((double) cAT, fAT), __dmul_rn((double) cCG, fCG)));
			A.s1 = __dadd_rn(A.s1, ti);
			A.s2 = __dadd_rn(A.s2, tj);
			/* calcRelatedness, :1151-1188 */
			const bool heti = a0 > min_cov && a1 > min_cov;
			const bool i_at = a0 > min_cov, j_at = b0 > min_cov;               /* for a homozygous site: which allele */
			A.hets1 += heti; A.homs1 += !heti;
			A.hets2 += hetj; A.homs2 += !hetj;
			if (heti && hetj) A.sh_het++;
			else if (!heti && !hetj) { if (i_at == j_at) A.sh_hom++; else A.ibs0++; }
		}
	}
	if (j >= n_samples) return;
#pragma unroll
	for (int ii = 0; ii < kTI; ++ii) {
		const uint32_t i = i0 + ii;
		if (i >= n_samples || j <= i) continue;
		const PairAcc &A = acc[ii];
		ntsm_eval_record r;
		r.sum_joint = A.joint; r.sum_single1 = A.s1; r.sum_single2 = A.s2;
		r.n_valid = A.n;
		r.hets1 = A.hets1; r.homs1 = A.homs1; r.hets2 = A.hets2; r.homs2 = A.homs2;
		r.shared_hets = A.sh_het; r.shared_homs = A.sh_hom; r.ibs0 = A.ibs0; r.ibs2 = A.sh_het + A.sh_hom;
		out[(uint64_t) i * n_samples - (uint64_t) i * (i + 1) / 2 + (j - i - 1)] = r;        /* ntsm_eval_pair_index */
	}
}

}  // namespace

#define EVCHK(x) do { if ((x) != hipSuccess) { rc = -2; goto done; } } while (0)

extern "C" int ntsm_eval_pairs(int device, const uint32_t *counts, uint32_t n_samples, uint32_t n_sites, uint32_t min_cov,
		ntsm_eval_record *out, double *kernel_ms)
{
	if (!counts || (!out && n_samples > 1)) return -1;
	if (kernel_ms) *kernel_ms = 0;
	if (n_samples < 2 || n_sites == 0) {
		for (uint64_t p = 0; n_samples >= 2 && p < (uint64_t) n_samples * (n_samples - 1) / 2; ++p) out[p] = ntsm_eval_record {};
		return 0;
	}
	int rc = 0;
	const uint64_t cells = (uint64_t) n_samples * n_sites, pairs = (uint64_t) n_samples * (n_samples - 1) / 2;
	uint32_t *d_counts = nullptr, *d_at = nullptr, *d_cg = nullptr;
	double *d_term = nullptr;
	ntsm_eval_record *d_out = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	float ms = 0;
	EVCHK(hipSetDevice(device));
	EVCHK(hipMalloc(&d_counts, cells * 2 * sizeof(uint32_t)));
	EVCHK(hipMalloc(&d_at, cells * sizeof(uint32_t)));
	EVCHK(hipMalloc(&d_cg, cells * sizeof(uint32_t)));
	EVCHK(hipMalloc(&d_term, cells * sizeof(double)));
	EVCHK(hipMalloc(&d_out, pairs * sizeof(ntsm_eval_record)));
	EVCHK(hipMemcpy(d_counts, counts, cells * 2 * sizeof(uint32_t), hipMemcpyHostToDevice));
	EVCHK(hipEventCreate(&e0));
	EVCHK(hipEventCreate(&e1));
	hipLaunchKernelGGL(ntsm_eval_prepare, dim3((unsigned) ((cells + kThreads - 1) / kThreads)), dim3(kThreads), 0, 0,
			d_counts, n_samples, n_sites, min_cov, d_at, d_cg, d_term);
	EVCHK(hipGetLastError());
	EVCHK(hipEventRecord(e0, 0));
	{   /* the sums of a pair are sequential over the sites, so the only parallelism is across pairs: few samples get
	     * one-wave workgroups (S = 256: 256 workgroups instead of 64) */
		const unsigned bt = n_samples <= 1024 ? 64 : kThreads;
		hipLaunchKernelGGL(ntsm_eval_pair_kernel, dim3((n_samples + bt - 1) / bt, (n_samples + kTI - 1) / kTI), dim3(bt), 0, 0,
				d_at, d_cg, d_term, n_samples, n_sites, min_cov, d_out);
	}
	EVCHK(hipGetLastError());
	EVCHK(hipEventRecord(e1, 0));
	EVCHK(hipMemcpy(out, d_out, pairs * sizeof(ntsm_eval_record), hipMemcpyDeviceToHost));
	EVCHK(hipEventElapsedTime(&ms, e0, e1));
	if (kernel_ms) *kernel_ms = ms;
done:
	if (e0) (void) hipEventDestroy(e0);
	if (e1) (void) hipEventDestroy(e1);
	(void) hipFree(d_counts); (void) hipFree(d_at); (void) hipFree(d_cg); (void) hipFree(d_term); (void) hipFree(d_out);
	return rc;
}
