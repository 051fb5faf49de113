/*
 * oracle/ntsm_eval_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the all-pairs scoring of the reference's ntsmEval:
 * reading counts files, the per-sample genotype summary and error rate, and for every pair of samples
 * the log-likelihood score, the relatedness tallies and the result line.
 *
 * PARITY UNPINNED.  The reference class (src/CompareCounts.hpp) cannot be compiled in this image:
 * its line 19 includes vendor/kfunc.c, whose line 28 includes the autoconf-generated config.h, and the
 * reference ships neither that file, nor tests, nor recorded outputs for this path.  What is here follows
 * the reference text function by function (file:line cited at each), in the same order of floating-point
 * operations (sequential double sums over the sites in file order, no contraction), but it has never been
 * compared with the reference's own output -- except for the genotype tallies and the two ratios derived from them,
 * which tests/test_eval.py checks against the six example rows of the reference's README (README.md:143-150); the
 * log-likelihood score has no such anchor.
 *
 * Covered: CompareCounts::CompareCounts (:30-114), computeScoreSingle (:541-585, without PCA columns),
 * computeScore (:591-624, one thread: pairs in i < j order), calcHomHetMiss (:742-767), loadPair (:934-940),
 * computeSumLogPSingle (:968-989), computeSumLogPJoint (:1013-1033), gatherValidEntries (:1057-1078),
 * skew (:1081-1083), computeLogLikelihood (:1093-1099), calcRelatedness (:1144-1196), computeErrorRate
 * (:1198-1216), resultsStr (:843-905), mergeCounts (:626-674).  Not covered: PCA projection and kd-tree search (-p).
 */
#define _GNU_SOURCE
#include "ntsm_eval_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct ntsm_eval_oracle {
	unsigned n_samples, n_sites;
	char **names;                 /* file names as given (m_filenames) */
	char **locus;                 /* m_locusIDs, order of the first file */
	unsigned *distinct;           /* [site][2]  m_distinct */
	unsigned *counts;             /* [sample][site][2]  m_counts */
	unsigned *sums;               /* [sample][site][2]  m_sum */
	uint64_t *raw_total;          /* m_rawTotalCounts (#@TK) */
	unsigned *kmer_size;          /* m_kmerSize (#@KS) */
	uint64_t *total;              /* m_totalCounts */
	/* locus id -> index: open addressing over the first file's ids */
	unsigned *slot;
	unsigned n_slots;
};

static uint64_t str_hash(const char *s, size_t n)
{
	uint64_t h = 1469598103934665603ull;
	for (size_t i = 0; i < n; ++i) { h ^= (unsigned char) s[i]; h *= 1099511628211ull; }
	return h;
}

static int locus_find(const ntsm_eval_oracle *e, const char *id, size_t n)
{
	for (uint64_t h = str_hash(id, n);; ++h) {
		const unsigned s = e->slot[h & (e->n_slots - 1)];
		if (s == 0xFFFFFFFFu) return -1;
		if (strlen(e->locus[s]) == n && memcmp(e->locus[s], id, n) == 0) return (int) s;
	}
}

/* one tab-separated field of a line: [*p, end of field); advances *p past the tab.  std::getline(ss, item, '\t')
 * returns an empty item once the stream is exhausted; std::stoul on it throws -- mirrored as a parse error. */
static int next_field(const char **p, const char *end, const char **f, size_t *n)
{
	if (*p > end) { *f = end; *n = 0; return 0; }
	const char *t = memchr(*p, '\t', (size_t) (end - *p));
	*f = *p;
	*n = (size_t) ((t ? t : end) - *p);
	*p = t ? t + 1 : end + 1;
	return 1;
}

/* std::stoul / std::stoull: optional white space and sign, digits, trailing characters ignored; no digits = error */
static int parse_ul(const char *f, size_t n, unsigned long long *out)
{
	char buf[64];
	if (n == 0 || n >= sizeof buf) return -1;
	memcpy(buf, f, n);
	buf[n] = 0;
	char *endp = NULL;
	*out = strtoull(buf, &endp, 10);
	return endp == buf ? -1 : 0;
}

/* src/CompareCounts.hpp:934-940: two consecutive fields, `item` already holds the first */
static int load_pair(const char **p, const char *end, const char *f, size_t n, unsigned *a, unsigned *b, const char **nf, size_t *nn)
{
	unsigned long long v;
	if (parse_ul(f, n, &v)) return -1;
	*a = (unsigned) v;
	next_field(p, end, &f, &n);
	if (parse_ul(f, n, &v)) return -1;
	*b = (unsigned) v;
	next_field(p, end, nf, nn);
	return 0;
}

static char *read_file(const char *path, size_t *len)
{
	FILE *fh = fopen(path, "rb");
	if (!fh) return NULL;
	size_t cap = 1 << 20, n = 0;
	char *buf = malloc(cap);
	for (;;) {
		if (n == cap) buf = realloc(buf, cap *= 2);
		const size_t got = fread(buf + n, 1, cap - n, fh);
		if (got == 0) break;
		n += got;
	}
	fclose(fh);
	*len = n;
	return buf;
}

/* src/CompareCounts.hpp:30-114 */
ntsm_eval_oracle *ntsm_eval_oracle_load(const char *const *files, unsigned n_files)
{
	if (n_files == 0) return NULL;
	ntsm_eval_oracle *e = calloc(1, sizeof *e);
	e->n_samples = n_files;
	e->names = calloc(n_files, sizeof *e->names);
	for (unsigned i = 0; i < n_files; ++i) e->names[i] = strdup(files[i]);
	/* :38-63 the first file fixes the loci, their order and the distinct-k-mer columns */
	{
		size_t len = 0;
		char *buf = read_file(files[0], &len);
		if (buf) {
			unsigned cap = 0;
			const char *p = buf, *stop = buf + len;
			while (p < stop) {
				const char *nl = memchr(p, '\n', (size_t) (stop - p));
				const char *end = nl ? nl : stop;
				if (end > p && p[0] != '#') {
					const char *q = p, *f; size_t n;
					next_field(&q, end, &f, &n);
					if (e->n_sites == cap) {
						cap = cap ? cap * 2 : 1024;
						e->locus = realloc(e->locus, cap * sizeof *e->locus);
						e->distinct = realloc(e->distinct, (size_t) cap * 2 * sizeof *e->distinct);
					}
					e->locus[e->n_sites] = strndup(f, n);
					for (int s = 0; s < 5; ++s) next_field(&q, end, &f, &n);      /* :51-56 skip to the sixth column */
					const char *nf; size_t nn;
					if (load_pair(&q, end, f, n, &e->distinct[2 * e->n_sites], &e->distinct[2 * e->n_sites + 1], &nf, &nn)) { free(buf); ntsm_eval_oracle_free(e); return NULL; }
					e->n_sites++;
				}
				p = end + 1;
			}
			free(buf);
		}
	}
	e->n_slots = 16;
	while (e->n_slots < 2 * e->n_sites + 2) e->n_slots *= 2;
	e->slot = malloc(e->n_slots * sizeof *e->slot);
	memset(e->slot, 0xFF, e->n_slots * sizeof *e->slot);
	for (unsigned s = 0; s < e->n_sites; ++s) {                       /* m_locusIDToIndex[locusID] = index: a repeated id keeps the LAST index */
		uint64_t h = str_hash(e->locus[s], strlen(e->locus[s]));
		for (;; ++h) {
			unsigned *sl = &e->slot[h & (e->n_slots - 1)];
			if (*sl == 0xFFFFFFFFu || strcmp(e->locus[*sl], e->locus[s]) == 0) { *sl = s; break; }
		}
	}
	const size_t cells = (size_t) n_files * e->n_sites * 2;
	e->counts = calloc(cells ? cells : 1, sizeof *e->counts);
	e->sums = calloc(cells ? cells : 1, sizeof *e->sums);
	e->raw_total = calloc(n_files, sizeof *e->raw_total);
	e->kmer_size = calloc(n_files, sizeof *e->kmer_size);
	e->total = calloc(n_files, sizeof *e->total);
	/* :68-113 every file: tags, counts and sums by locus id */
	for (unsigned i = 0; i < n_files; ++i) {
		size_t len = 0;
		char *buf = read_file(files[i], &len);
		if (!buf) continue;                                           /* fh.is_open() false: the sample stays all zero */
		const char *p = buf, *stop = buf + len;
		while (p < stop) {
			const char *nl = memchr(p, '\n', (size_t) (stop - p));
			const char *end = nl ? nl : stop;
			if (end > p) {
				const char *q = p, *f; size_t n;
				next_field(&q, end, &f, &n);
				if (p[0] == '#') {
					unsigned long long v;
					if (n == 4 && memcmp(f, "#@TK", 4) == 0) {
						next_field(&q, end, &f, &n);
						if (parse_ul(f, n, &v)) { free(buf); ntsm_eval_oracle_free(e); return NULL; }
						e->raw_total[i] = v;
					} else if (n == 4 && memcmp(f, "#@KS", 4) == 0) {
						next_field(&q, end, &f, &n);
						if (parse_ul(f, n, &v)) { free(buf); ntsm_eval_oracle_free(e); return NULL; }
						e->kmer_size[i] = (unsigned) v;
					}
				} else {
					const int s = locus_find(e, f, n);                /* .at(): an unknown locus throws */
					if (s < 0) { free(buf); ntsm_eval_oracle_free(e); return NULL; }
					unsigned *c = &e->counts[((size_t) i * e->n_sites + (unsigned) s) * 2];
					unsigned *m = &e->sums[((size_t) i * e->n_sites + (unsigned) s) * 2];
					const char *nf; size_t nn;
					next_field(&q, end, &f, &n);
					if (load_pair(&q, end, f, n, &c[0], &c[1], &nf, &nn)) { free(buf); ntsm_eval_oracle_free(e); return NULL; }
					e->total[i] += (uint64_t) c[0] + c[1];
					if (load_pair(&q, end, nf, nn, &m[0], &m[1], &nf, &nn)) { free(buf); ntsm_eval_oracle_free(e); return NULL; }
				}
			}
			p = end + 1;
		}
		free(buf);
	}
	return e;
}

void ntsm_eval_oracle_free(ntsm_eval_oracle *e)
{
	if (!e) return;
	for (unsigned i = 0; i < e->n_samples; ++i) free(e->names[i]);
	for (unsigned i = 0; i < e->n_sites; ++i) free(e->locus[i]);
	free(e->names); free(e->locus); free(e->distinct); free(e->counts); free(e->sums);
	free(e->raw_total); free(e->kmer_size); free(e->total); free(e->slot);
	free(e);
}

unsigned ntsm_eval_oracle_samples(const ntsm_eval_oracle *e) { return e->n_samples; }
unsigned ntsm_eval_oracle_sites(const ntsm_eval_oracle *e) { return e->n_sites; }
const unsigned *ntsm_eval_oracle_counts(const ntsm_eval_oracle *e) { return e->counts; }
const unsigned *ntsm_eval_oracle_sums(const ntsm_eval_oracle *e) { return e->sums; }
const unsigned *ntsm_eval_oracle_distinct(const ntsm_eval_oracle *e) { return e->distinct; }
uint64_t ntsm_eval_oracle_total(const ntsm_eval_oracle *e, unsigned i) { return e->total[i]; }
uint64_t ntsm_eval_oracle_raw_total(const ntsm_eval_oracle *e, unsigned i) { return e->raw_total[i]; }
unsigned ntsm_eval_oracle_kmer_size(const ntsm_eval_oracle *e, unsigned i) { return e->kmer_size[i]; }

/* src/CompareCounts.hpp:742-767 */
void ntsm_eval_oracle_genotype(const ntsm_eval_oracle *e, unsigned i, unsigned min_cov, unsigned out[3])
{
	unsigned hets = 0, homs = 0, miss = 0;
	const unsigned *c = &e->counts[(size_t) i * e->n_sites * 2];
	for (unsigned s = 0; s < e->n_sites; ++s) {
		if (c[2 * s] > min_cov) {
			if (c[2 * s + 1] > min_cov) ++hets; else ++homs;
		} else if (c[2 * s + 1] > min_cov) ++homs;
		else ++miss;
	}
	out[0] = hets; out[1] = homs; out[2] = miss;
}

/* src/CompareCounts.hpp:1198-1216 */
double ntsm_eval_oracle_error_rate(const ntsm_eval_oracle *e, unsigned i, uint64_t genome_size)
{
	if (e->raw_total[i] > 0 && e->kmer_size[i] > 0) {
		uint64_t sum = 0, distinct = 0;
		const unsigned *m = &e->sums[(size_t) i * e->n_sites * 2];
		for (unsigned s = 0; s < e->n_sites; ++s) {
			sum += m[2 * s] + m[2 * s + 1];                              /* unsigned + unsigned, then widened (as in the reference) */
			distinct += e->distinct[2 * s] + e->distinct[2 * s + 1];
		}
		const double expected = (double) e->raw_total[i] * (double) distinct / (double) genome_size;
		return 1.0 - pow((double) sum / expected, 1.0 / (double) e->kmer_size[i]);
	}
	return -1.0;
}

/* one pair: gatherValidEntries (:1057-1078), computeSumLogPJoint (:1013-1033), computeSumLogPSingle (:968-989) for
 * both samples, calcRelatedness (:1144-1196).  The three sums run over the valid sites in index order. */
void ntsm_eval_oracle_pair(const ntsm_eval_oracle *e, unsigned i1, unsigned i2, unsigned min_cov, ntsm_eval_pair *r)
{
	const unsigned *a = &e->counts[(size_t) i1 * e->n_sites * 2], *b = &e->counts[(size_t) i2 * e->n_sites * 2];
	memset(r, 0, sizeof *r);
	double joint = 0, s1 = 0, s2 = 0;
	for (unsigned s = 0; s < e->n_sites; ++s) {
		const unsigned a0 = a[2 * s], a1 = a[2 * s + 1], b0 = b[2 * s], b1 = b[2 * s + 1];
		if ((a0 <= min_cov && a1 <= min_cov) || (b0 <= min_cov && b1 <= min_cov)) continue;
		r->n_valid++;
		{   /* joint, :1018-1031 */
			double fAT = 0, fCG = 0;
			const unsigned cAT = a0 + b0, cCG = a1 + b1;
			if (cAT > min_cov) fAT = (double) cAT / (double) (cAT + cCG);
			if (cCG > min_cov) fCG = (double) cCG / (double) (cAT + cCG);
			joint += cAT * fAT + cCG * fCG;
		}
		{   /* single, sample 1, :971-987 */
			double fAT = 0, fCG = 0;
			if (a0 > min_cov) fAT = (double) a0 / (double) (a0 + a1);
			if (a1 > min_cov) fCG = (double) a1 / (double) (a0 + a1);
			s1 += a0 * fAT + a1 * fCG;
		}
		{
			double fAT = 0, fCG = 0;
			if (b0 > min_cov) fAT = (double) b0 / (double) (b0 + b1);
			if (b1 > min_cov) fCG = (double) b1 / (double) (b0 + b1);
			s2 += b0 * fAT + b1 * fCG;
		}
		/* :1151-1188 */
		enum { HET, HOM_AT, HOM_CG, UNKNOWN } t1 = UNKNOWN, t2 = UNKNOWN;
		if (a0 > min_cov) { if (a1 > min_cov) { t1 = HET; r->hets1++; } else { t1 = HOM_AT; r->homs1++; } }
		else if (a1 > min_cov) { t1 = HOM_CG; r->homs1++; }
		if (b0 > min_cov) { if (b1 > min_cov) { t2 = HET; r->hets2++; } else { t2 = HOM_AT; r->homs2++; } }
		else if (b1 > min_cov) { t2 = HOM_CG; r->homs2++; }
		if (t1 == HET && t2 == HET) { r->shared_hets++; r->ibs2++; }
		else if ((t1 == HOM_AT && t2 == HOM_AT) || (t1 == HOM_CG && t2 == HOM_CG)) { r->shared_homs++; r->ibs2++; }
		else if ((t1 == HOM_CG && t2 == HOM_AT) || (t1 == HOM_AT && t2 == HOM_CG)) r->ibs0++;
	}
	r->sum_joint = joint; r->sum_single1 = s1; r->sum_single2 = s2;
}

/* computeLogLikelihood (:1093-1099), skew (:1081-1083) and the division by the number of sites (:611-615) */
double ntsm_eval_oracle_score(const ntsm_eval_pair *r, double cov1, double cov2, double cov_skew)
{
	if (r->n_valid == 0) return DBL_MAX;
	double score = -2.0 * (r->sum_joint - (r->sum_single1 + r->sum_single2));
	score = score / pow(cov1 * cov2, cov_skew);
	return score / (double) r->n_valid;
}

/* std::to_string(double) = "%f" */
static void put_d(FILE *out, double v) { fprintf(out, "%f", v); }

/* computeScoreSingle (:541-585) for one file, computeScore (:591-624) otherwise; one thread */
int ntsm_eval_oracle_print(const ntsm_eval_oracle *e, FILE *out, unsigned min_cov, double score_thresh, int all, double cov_skew, uint64_t genome_size)
{
	const unsigned n = e->n_samples;
	unsigned (*geno)[3] = malloc((size_t) n * sizeof *geno);
	double *err = malloc(n * sizeof *err), *cov = malloc(n * sizeof *cov);
	for (unsigned i = 0; i < n; ++i) {
		ntsm_eval_oracle_genotype(e, i, min_cov, geno[i]);
		err[i] = ntsm_eval_oracle_error_rate(e, i, genome_size);
		cov[i] = (double) e->total[i] / (double) e->n_sites;
	}
	if (n == 1) {
		fputs("sample\tcov\terrorRate\tmiss\thom\thet\n", out);
		fputs(e->names[0], out); fputc('\t', out);
		put_d(out, cov[0]); fputc('\t', out); put_d(out, err[0]);
		fprintf(out, "\t%u\t%u\t%u", geno[0][2], geno[0][1], geno[0][0]);       /* no newline after the last row (:579-582) */
	} else {
		fputs("sample1\tsample2\tscore\tsame\tdist\trelate\tibs0\tibs2\thomConcord\thet1\thet2\tsharedHet\thom1\thom2\tsharedHom\tn"
		      "\tcov1\tcov2\terrorRate1\terrorRate2\tmiss1\tmiss2\tallHom1\tallHom2\tallHet1\tallHet2", out);
		fputc('\n', out);
		for (unsigned i = 0; i < n; ++i)
			for (unsigned j = i + 1; j < n; ++j) {
				ntsm_eval_pair r;
				ntsm_eval_oracle_pair(e, i, j, min_cov, &r);
				const double score = ntsm_eval_oracle_score(&r, cov[i], cov[j], cov_skew);
				if (!(all || score < score_thresh)) continue;
				const double hom_concord = ((double) r.shared_homs - 2.0 * (double) r.ibs0) / (double) (r.homs1 < r.homs2 ? r.homs1 : r.homs2);
				const double relate = ((double) r.shared_hets - 2.0 * (double) r.ibs0) / (double) (r.hets1 < r.hets2 ? r.hets1 : r.hets2);
				fputs(e->names[i], out); fputc('\t', out); fputs(e->names[j], out); fputc('\t', out);
				put_d(out, score);
				fputs(all ? (score < score_thresh ? "\t1\t" : "\t0\t") : "\t1\t", out);
				fputs("-1\t", out);
				put_d(out, relate);
				fprintf(out, "\t%u\t%u\t", r.ibs0, r.ibs2);
				put_d(out, hom_concord);
				fprintf(out, "\t%u\t%u\t%u\t%u\t%u\t%u\t%llu\t", r.hets1, r.hets2, r.shared_hets, r.homs1, r.homs2, r.shared_homs, (unsigned long long) r.n_valid);
				put_d(out, cov[i]); fputc('\t', out); put_d(out, cov[j]); fputc('\t', out);
				put_d(out, err[i]); fputc('\t', out); put_d(out, err[j]);
				fprintf(out, "\t%u\t%u\t%u\t%u\t%u\t%u\n", geno[i][2], geno[j][2], geno[i][1], geno[j][1], geno[i][0], geno[j][0]);
			}
	}
	free(geno); free(err); free(cov);
	return 0;
}

/* mergeCounts (:626-674): tags summed / taken from the first file, per locus the 32-bit sums of counts and sums over all
 * files, the first file's distinct columns.  Returns -1 where the reference's assert on unequal k fails. */
int ntsm_eval_oracle_merge(const ntsm_eval_oracle *e, FILE *out)
{
	for (unsigned i = 0; i < e->n_samples; ++i)
		for (unsigned j = i + 1; j < e->n_samples; ++j)
			if (e->kmer_size[i] != e->kmer_size[j]) return -1;
	uint64_t tk = 0;
	for (unsigned i = 0; i < e->n_samples; ++i) tk += e->raw_total[i];
	fprintf(out, "#@TK\t%llu\n#@KS\t%u\n#locusID\tcountAT\tcountCG\tsumAT\tsumCG\tdistinctAT\tdistinctCG\n", (unsigned long long) tk, e->kmer_size[0]);
	for (unsigned s = 0; s < e->n_sites; ++s) {
		unsigned cAT = 0, cCG = 0, sAT = 0, sCG = 0;
		for (unsigned j = 0; j < e->n_samples; ++j) {
			const size_t at = ((size_t) j * e->n_sites + s) * 2;
			cAT += e->counts[at]; cCG += e->counts[at + 1];
			sAT += e->sums[at]; sCG += e->sums[at + 1];
		}
		fprintf(out, "%s\t%u\t%u\t%u\t%u\t%u\t%u\n", e->locus[s], cAT, cCG, sAT, sCG, e->distinct[2 * s], e->distinct[2 * s + 1]);
	}
	return 0;
}
