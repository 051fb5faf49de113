/*
 * include/ntsm_host.h -- C ABI over the host-side pieces of ntsmCount (site loading, read
 * flattening, report formatting) so that tests and bench.py can drive them without the CLI.
 * None of these functions touches a GPU.  Reference behaviour each one reproduces:
 *   ntsm_sites_*          FingerPrint::initCountsHash            src/FingerPrint.hpp:490-564
 *   ntsm_host_flatten     kseq_read loop of computeCounts        src/FingerPrint.hpp:64-69, vendor/kseq.h:177-219
 *   ntsm_host_format_*    printOptionalHeader/printCountsMax/    src/FingerPrint.hpp:261-349
 *                         printInfoSummary
 *   ntsm_host_max_hits    m_maxCounts                            src/FingerPrint.hpp:41-43
 */
#ifndef NTSM_HOST_H
#define NTSM_HOST_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ntsm_sites ntsm_sites;

/* Load a sites file.  Returns 0, or -1 if it cannot be opened.  Collision warnings go to stderr. */
int ntsm_sites_load(const char *path, unsigned k, int allow_dupes, ntsm_sites **out);
void ntsm_sites_free(ntsm_sites *s);
uint64_t ntsm_sites_n_keys(const ntsm_sites *s);          /* m_counts.size() */
const uint64_t *ntsm_sites_keys(const ntsm_sites *s);     /* canonical codes, first-seen order */
uint64_t ntsm_sites_n_sites(const ntsm_sites *s);         /* m_alleleIDs.size() */
uint64_t ntsm_sites_n_erased(const ntsm_sites *s);        /* duplicates removed (no -d) */

/* m_maxCounts for a k-mer set size and the -m value (0 = disabled / never) */
uint64_t ntsm_host_max_hits(uint64_t n_distinct, double cov_thresh);

/* Parse one FASTA/FASTQ(.gz) file into the flat stream layout of include/ntsm_hip.h.
 * Buffers are malloc'ed; release with ntsm_host_free.  *last_rc = the reader's terminating code
 * (-1 EOF, -2 truncated quality, -3 stream error).  Returns 0, or -1 if the file cannot be opened. */
int ntsm_host_flatten(const char *path, uint8_t **bases, uint64_t *n_bytes, uint64_t **read_end,
		uint64_t *n_reads, int *last_rc);
void ntsm_host_free(void *p);
/* Test hook for the packed producer lanes (ntsm_amd/csrc/host/pack2.hpp; include/ntsm_hip.h ntsm_lane_acquire_packed):
 * append one read at position pos (a multiple of 8) of a packed batch -- 2-bit codes + 1 validity bit per position, the
 * five classes of vendor/KseqHashIterator.hpp:114-127 -- and return where the next read starts.  force_scalar: 0 = the best
 * implementation the CPU has (AVX-512 VBMI, AVX2, portable), 1 = the portable one, 2 = at most AVX2. */
uint64_t ntsm_host_pack2_append(uint8_t *codes, uint8_t *valid, uint64_t pos, const uint8_t *seq, uint64_t len, int force_scalar);
const char *ntsm_host_pack2_impl(void);
/* Test hook for the gzip ingest (ntsm_amd/csrc/host/gz_stream.hpp): decode `path` with the decoder thread
 * (engine 0; engine n >= 2: with n decoder threads for BGZF input) or with zlib's gzread (engine 1), reading
 * `chunk` bytes per call.  out / len receive the bytes delivered
 * (free with ntsm_host_free); returns the last read's result: 0 = clean end, -1 = error, -2 = cannot open. */
int ntsm_host_gunzip(const char *path, int engine, unsigned chunk, uint8_t **out, uint64_t *len);
/* Test hook: compressed bytes per chunk of the parallel decoder for plain (non-BGZF) gzip input that engine n >= 2 uses
 * (0 = the default 1 MiB; files shorter than two chunks are decoded in order).  stats, if not NULL, receives what the last
 * ntsm_host_gunzip call with engine >= 2 did: [0] chunks spliced in, [1] chunks dropped (no block start found in their
 * range, start not confirmed by the in-order decoder, or decoding failed). */
void ntsm_host_gunzip_parallel_chunk(uint64_t bytes);
void ntsm_host_gunzip_parallel_stats(uint64_t stats[2]);
/* Test hook for the parallel gzip ingest (parallel_gz_fastq.hpp over gz_stream.hpp): n_decoders decoder threads inflate
 * `path`, n_parsers threads parse the pieces (sinks of sink_bytes each), the sequential reader finishes what the parallel
 * phase left.  Reads come back piece by piece; INSIDE a piece the records carried over from the previous piece may stand
 * before or between the piece's own (the counting path does not depend on read order).  Returns 0, 1 if the file is not
 * gzip, -1 if it cannot be opened.  *n_parallel = records committed by the parallel phase. */
int ntsm_host_flatten_parallel_gz(const char *path, unsigned n_decoders, unsigned n_parsers, uint64_t sink_bytes, uint8_t **bases,
		uint64_t *n_bytes, uint64_t **read_end, uint64_t *n_reads, uint64_t *n_pieces, uint64_t *n_parallel, int *final_status);
/* Test hook for the early ingest (ntsm_amd/csrc/host/early_ingest.hpp): parse `path` (plain FASTQ or gzip) into packed chunks
 * of chunk_positions positions, at most max_chunks at once, drained by n_consumers threads.  *text receives every position of
 * every chunk as 'A' 'C' 'G' 'T' (valid) or 'N' (invalid / terminator), chunks separated by 'N' -- so the maximal runs of
 * letters are the maximal runs of valid bases of the reads.  Returns 0, 1 if the file is not taken by this path. */
int ntsm_host_early_ingest(const char *path, unsigned n_parsers, unsigned n_decoders, uint64_t block_bytes, uint64_t chunk_positions, uint64_t max_chunks,
		unsigned n_consumers, uint8_t **text, uint64_t *n_text, uint64_t *n_reads, uint64_t *n_bases, uint64_t *n_parallel);
/* The same with the hand-over of a gzip stream (EarlyIngest::hand_over / release_stream): once `hand_over_after` chunks have
 * been drained the consumers ask for the stream; what it still holds is read by the sequential reader and appended to *text in
 * the same alphabet.  *n_rest = reads that came that way (0: the whole file went through the chunks, or a plain file). */
int ntsm_host_early_ingest_hand_over(const char *path, unsigned n_parsers, unsigned n_decoders, uint64_t block_bytes, uint64_t chunk_positions,
		uint64_t max_chunks, unsigned n_consumers, uint64_t hand_over_after, uint8_t **text, uint64_t *n_text, uint64_t *n_reads, uint64_t *n_bases,
		uint64_t *n_parallel, uint64_t *n_rest);
/* Thread counts of the ingest pipeline (ntsm_amd/csrc/host/host_shape.hpp): the CPUs this process is granted = min(affinity
 * mask, cgroup CPU quota), and the plan `ntsmCount -t threads_asked` follows on such a host (cpus = 0: this host's grant):
 * out = { cpus, feeders, decoders of one big .gz, decoders while the sites still load }.  feeders + decoders <= 2 x cpus. */
unsigned ntsm_host_granted_cpus(void);
void ntsm_host_ingest_plan(unsigned threads_asked, unsigned cpus, unsigned out[4]);
/* Test hook: the n-th chunk allocation of an early ingest from now on fails (0 = off).  The hooks above then return -3:
 * an allocation failure must surface as a failed run, never as a shorter one (the CLI exits 1 with a message and prints no
 * counts, like the reference for a file it cannot read: src/FingerPrint.hpp:51-57). */
void ntsm_host_debug_early_alloc_fail(long nth);
/* Test hook: the longest unparsed rest a piece of the piece-parallel gzip parse may carry into the next link (0 = the default
 * 256 MiB); a piece whose rest is longer commits what it parsed and ends the parallel phase there. */
void ntsm_host_debug_gz_max_tail(uint64_t bytes);
/* Block-parallel variant for plain 4-line FASTQ (ntsm_amd/csrc/host/parallel_fastq.hpp), for tests: the records
 * the parallel phase commits (in file order) followed by what the sequential reader yields from *resume on.
 * Returns 0, 1 if the file is not eligible (callers use ntsm_host_flatten), -1 if it cannot be opened.
 * *n_parallel = records parsed by the parallel phase; *resume = byte offset it stopped at (file size if complete). */
int ntsm_host_flatten_parallel(const char *path, unsigned n_threads, uint64_t block_bytes, uint8_t **bases,
		uint64_t *n_bytes, uint64_t **read_end, uint64_t *n_reads, uint64_t *n_blocks, uint64_t *n_parallel,
		uint64_t *resume);

/* counts.txt bytes for per-k-mer counts in key order.  Returns 0, or 1 if the reference would
 * abort while printing (then *out holds the rows written before the abort). */
int ntsm_host_format_counts(const ntsm_sites *s, const uint64_t *counts, uint64_t total_kmers,
		char **out, size_t *len);
/* six-line summary; *covered receives "Sites Covered by at least one k-mer". */
int ntsm_host_format_summary(const ntsm_sites *s, const uint64_t *counts, uint64_t total_bases,
		uint64_t total_kmers, uint64_t total_hits, char **out, size_t *len, uint64_t *covered);

#ifdef __cplusplus
}
#endif
#endif
