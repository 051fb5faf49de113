/*
 * include/ntsm_synth.h -- C ABI of the synthetic-workload generator (bench/test tooling, not part
 * of the reference's interface; the reference ships no data generator and its
 * data/human_sites_n10.fa is absent from the checkout).
 * Definitions of the workloads: ntsm_amd/csrc/synth.h.
 */
#ifndef NTSM_SYNTH_API_H
#define NTSM_SYNTH_API_H
#include <stdint.h>
#include "../ntsm_amd/csrc/synth.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Generate `n_sites` duplicate-free SNP sites (window 31, k-mer size k <= 31).
 *   windows   : out, n_sites * 2 * 32 bytes of 2-bit codes ([site][allele][pos]), may be NULL
 *   fasta_path: if non-NULL, write the interleaved sites FASTA there (".gz" suffix => gzip)
 *   n_kmers   : out, number of distinct k-mers over all sites, may be NULL
 * Returns 0 or a negative error. */
int ntsm_synth_sites(uint64_t seed, uint32_t n_sites, unsigned k, uint8_t *windows,
		const char *fasta_path, uint64_t *n_kmers);
/* The same with the smallest number of k-mer start positions a site keeps (0 = the default 3; 13 at k = 19 keeps every
 * k-mer of every window: 96287 sites -> 2,503,462 site k-mers, the upper bound SURVEY.md section 8a gives for the real
 * human_sites_n10.fa -- bench.py's n10_full leg).  The draws of a site do not depend on this value beyond the count kept. */
int ntsm_synth_sites_keep(uint64_t seed, uint32_t n_sites, unsigned k, unsigned min_keep, uint8_t *windows,
		const char *fasta_path, uint64_t *n_kmers);

/* Fill parameter blocks from probabilities. */
void ntsm_synth_short_params(ntsm_synth_short *p, uint64_t seed, uint32_t read_len, uint32_t n_sites,
		double p_embed, double p_sub, double p_n);
void ntsm_synth_long_params(ntsm_synth_long *p, uint64_t seed, uint32_t n_sites, uint32_t spacing,
		double p_sub, double p_n);
/* 257-entry log-normal quantile table (mu, sigma of ln(length), clipped to [lo, hi]). */
void ntsm_synth_long_qtable(double mu, double sigma, uint32_t lo, uint32_t hi, uint32_t *qtable257);
/* read_end[i] for long reads r0 .. r0+n_reads-1 laid out from offset 0; returns total bytes. */
uint64_t ntsm_synth_long_layout(uint64_t seed, const uint32_t *qtable257, uint64_t r0, uint64_t n_reads,
		uint64_t *read_end);

/* Host fill of flat-stream bytes [g0, g0+n) of the short-read stream. */
void ntsm_synth_short_fill_host(const ntsm_synth_short *p, const uint8_t *windows, uint64_t g0,
		uint64_t n, uint8_t *out);
/* Host fill of the long-read stream for reads r0.. (read_end relative to out[0]). */
void ntsm_synth_long_fill_host(const ntsm_synth_long *p, const uint8_t *windows, const uint32_t *qtable257,
		uint64_t r0, uint64_t n_reads, const uint64_t *read_end, uint8_t *out);

/* Device fills (HIP).  d_* are device pointers; stream is a hipStream_t (NULL = default).
 * Return 0 or a negative hipError_t. */
int ntsm_synth_short_fill_device(const ntsm_synth_short *p, const void *d_windows, uint64_t g0,
		uint64_t n, void *d_out, void *stream);
int ntsm_synth_long_fill_device(const ntsm_synth_long *p, const void *d_windows, const void *d_qtable257,
		uint64_t r0, uint64_t n_reads, const void *d_read_end, uint64_t n_bytes, void *d_out, void *stream);

/* Write reads [r0, r0+n_reads) of the short-read stream as FASTQ (quality 'I'); ".gz" => gzip. */
int ntsm_synth_short_write_fastq(const ntsm_synth_short *p, const uint8_t *windows, uint64_t r0,
		uint64_t n_reads, const char *path);
/* The same file (plain output), written by n_threads threads with pwrite() at computed offsets. */
int ntsm_synth_short_write_fastq_mt(const ntsm_synth_short *p, const uint8_t *windows, uint64_t r0,
		uint64_t n_reads, const char *path, unsigned n_threads);
/* Both with a quality model (ntsm_synth_qual_char in synth.h: 0 = constant 'I' as above, 1 = Illumina-like -- position-dependent
 * decay, 20+ distinct scores, low scores in short runs; the text then compresses ~3.5:1 instead of 6:1).  The sequence lines,
 * hence the counts, do not depend on the model. */
int ntsm_synth_short_write_fastq_q(const ntsm_synth_short *p, const uint8_t *windows, uint64_t r0,
		uint64_t n_reads, const char *path, unsigned qual_model);
int ntsm_synth_short_write_fastq_mt_q(const ntsm_synth_short *p, const uint8_t *windows, uint64_t r0,
		uint64_t n_reads, const char *path, unsigned n_threads, unsigned qual_model);
int ntsm_synth_long_write_fastq(const ntsm_synth_long *p, const uint8_t *windows, const uint32_t *qtable257,
		uint64_t r0, uint64_t n_reads, const char *path);

#ifdef __cplusplus
}
#endif
#endif
