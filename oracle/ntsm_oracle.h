/*
 * oracle/ntsm_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the reference's ntsmCount counting path
 * (JustinChu/ntsm v1.2.1).  It exists so that the HIP product can be checked bit-for-bit on a
 * GPU box where /root/reference does not exist.  Nothing in the product (ntsm_amd/, include/)
 * links, imports or executes this code; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may.
 *
 * PARITY PIN: this restatement is pinned against the UNMODIFIED reference compiled in place
 * (oracle/_ref/ref_ntsmCount, recipe in oracle/Makefile): tests/golden/ holds stdout/stderr the
 * reference produced here for every fixture (tests/golden/make_golden.py) and
 * tests/test_oracle_golden.py requires this code to reproduce them byte-for-byte.  The
 * reference ships no tests or known-answer vectors of its own.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root).
 */
#ifndef NTSM_ORACLE_H
#define NTSM_ORACLE_H
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- k-mer arithmetic: vendor/KseqHashIterator.hpp ------------------------------------------ */

/* s_seq_nt4_table, vendor/KseqHashIterator.hpp:114-127 */
int ntsm_oracle_nt4(unsigned char b);
/* m_mask, vendor/KseqHashIterator.hpp:29 ((1ULL << 2k) - 1; k = 32 mirrors the x86-64 result 0) */
uint64_t ntsm_oracle_mask(unsigned k);
/* hash64, vendor/KseqHashIterator.hpp:129-139 */
uint64_t ntsm_oracle_hash64(uint64_t key, uint64_t mask);

typedef struct {
	const unsigned char *seq;
	uint64_t len, mask, shift, pos;    /* pos = index of the next unread byte (getPos(), :62) */
	unsigned k, run;
	uint64_t fw, rv;
	uint64_t canon;                    /* min(fw, rv) of the current window          (:104)   */
	uint64_t hv;                       /* hash64(canon, mask) == *itr                (:104-105) */
} ntsm_oracle_iter;

/* ctor, vendor/KseqHashIterator.hpp:28-33 (without the implicit first next()) */
void ntsm_oracle_iter_init(ntsm_oracle_iter *it, const char *seq, uint64_t len, unsigned k);
/* next()/step(), vendor/KseqHashIterator.hpp:87-112.  Returns 1 and fills canon/hv/pos for the
 * next valid window, 0 at the end of the sequence. */
int ntsm_oracle_iter_next(ntsm_oracle_iter *it);

/* Convenience for tests: all windows of one sequence.  out_canon/out_hv/out_pos may be NULL;
 * returns the number of valid windows (writes at most cap entries). */
uint64_t ntsm_oracle_kmers(const char *seq, uint64_t len, unsigned k, uint64_t *out_canon,
		uint64_t *out_hv, uint64_t *out_pos, uint64_t cap);

/* ---- FASTA/FASTQ record reader: vendor/kseq.h:177-219 over gzread (:229, FingerPrint.hpp:27) - */

typedef struct ntsm_oracle_reader ntsm_oracle_reader;
ntsm_oracle_reader *ntsm_oracle_reader_open(const char *path);      /* gzopen + kseq_init       */
/* kseq_read: >=0 sequence length, -1 EOF, -2 truncated quality, -3 stream error */
int64_t ntsm_oracle_reader_next(ntsm_oracle_reader *r);
const char *ntsm_oracle_reader_seq(const ntsm_oracle_reader *r);    /* seq.s (not NUL-safe: use len) */
const char *ntsm_oracle_reader_name(const ntsm_oracle_reader *r);   /* name.s                   */
void ntsm_oracle_reader_close(ntsm_oracle_reader *r);               /* kseq_destroy + gzclose   */

/* ---- FingerPrint: src/FingerPrint.hpp -------------------------------------------------------- */

typedef struct ntsm_oracle_fp ntsm_oracle_fp;

/* FingerPrint() + initCountsHash(), src/FingerPrint.hpp:35-44, :490-564.
 * cov_thresh mirrors opt::covThresh (Options.h:32; DBL_MAX = never stop, 0 = disabled).
 * Collision warnings go to `err` (may be NULL).  Returns NULL if the file cannot be opened
 * (the reference exits 1, :493-499). */
ntsm_oracle_fp *ntsm_oracle_fp_create(const char *sites_path, unsigned k, double cov_thresh,
		int dupes, FILE *err);
void ntsm_oracle_fp_destroy(ntsm_oracle_fp *fp);

/* insertCount(), src/FingerPrint.hpp:89-103 (multiplier = 1) */
void ntsm_oracle_fp_insert_count(ntsm_oracle_fp *fp, const char *seq, uint64_t len);
/* the same with the reference's `unsigned multiplier` parameter (:89): used to reach counts >= 2^32 on small inputs */
void ntsm_oracle_fp_insert_count_mult(ntsm_oracle_fp *fp, const char *seq, uint64_t len, unsigned multiplier);
/* processSingleRead(), src/FingerPrint.hpp:473-488.  Returns 1 once the -m threshold tripped. */
int ntsm_oracle_fp_process_read(ntsm_oracle_fp *fp, const char *seq, uint64_t len);
/* computeCounts(), src/FingerPrint.hpp:46-87, serial in argv order (the single-thread schedule).
 * Returns 0, or 1 if a file cannot be opened (reference: exit(1), :51-57). */
int ntsm_oracle_fp_compute_counts(ntsm_oracle_fp *fp, const char *const *files, int n_files, FILE *err);

/* printOptionalHeader() + printCountsMax(), src/FingerPrint.hpp:261-311.
 * Returns 0, or -1 where the reference's m_counts.at() throws (duplicate k-mer erased without
 * -d, :282/:289 -> robin_hash.h:941-966): rows before the throwing site have been written. */
int ntsm_oracle_fp_print_counts(ntsm_oracle_fp *fp, FILE *out);
/* printInfoSummary(), src/FingerPrint.hpp:313-349: writes the 6-line summary into buf (and the
 * <75% warning to err); returns number of bytes written (excluding NUL). */
int ntsm_oracle_fp_info_summary(ntsm_oracle_fp *fp, char *buf, size_t cap, FILE *err);

/* State accessors (for parity tests against the HIP path) */
uint64_t ntsm_oracle_fp_total_kmers(const ntsm_oracle_fp *fp);   /* m_totalKmers  */
uint64_t ntsm_oracle_fp_total_hits(const ntsm_oracle_fp *fp);    /* m_totalCounts */
uint64_t ntsm_oracle_fp_total_bases(const ntsm_oracle_fp *fp);   /* m_totalBases  */
uint64_t ntsm_oracle_fp_max_hits(const ntsm_oracle_fp *fp);      /* m_maxCounts   */
int      ntsm_oracle_fp_early_term(const ntsm_oracle_fp *fp);    /* m_earlyTerm   */
uint64_t ntsm_oracle_fp_reads_processed(const ntsm_oracle_fp *fp); /* reads that went through processSingleRead */
uint64_t ntsm_oracle_fp_n_distinct(const ntsm_oracle_fp *fp);    /* m_counts.size() */
uint64_t ntsm_oracle_fp_n_sites(const ntsm_oracle_fp *fp);       /* m_alleleIDs.size() */
/* k-mers in first-seen order across the sites file (REF then VAR vectors, record order):
 * n = total length of all allele vectors.  Fills canonical codes, hash64 keys, and the current
 * count of each (0 if erased). Returns n (writes at most cap). */
uint64_t ntsm_oracle_fp_kmers(const ntsm_oracle_fp *fp, uint64_t *canon, uint64_t *hv,
		uint64_t *count, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* NTSM_ORACLE_H */
