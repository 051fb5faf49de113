/*
 * include/ntsm_hip.h -- C ABI of the MI355X (gfx950) k-mer counting hot path of ntsmCount.
 *
 * The reference (JustinChu/ntsm v1.2.1) has no plugin/FFI layer; its seam is the FingerPrint
 * class that src/ntSeqMatchCount.cpp:177-181 drives.  This library replaces what happens
 * INSIDE that class between site loading and printing:
 *
 *   reference (CPU)                                            this ABI (HIP)
 *   ---------------------------------------------------------  --------------------------------
 *   tsl::robin_map<uint64_t,size_t> m_counts filled with       ntsm_create(keys, n_kmers, ...)
 *     m_counts[hv] = 0        (src/FingerPrint.hpp:528,:550)
 *   m_maxCounts = size*covThresh/2   (src/FingerPrint.hpp:41)  ntsm_create(..., max_hits)
 *   per read: insertCount(seq.s, seq.l)  -> KseqHashIterator   ntsm_submit / ntsm_submit_staged /
 *     + m_counts.find + atomic +=  (src/FingerPrint.hpp:89-103,  ntsm_count_resident   (batched)
 *     vendor/KseqHashIterator.hpp:87-139)
 *   processSingleRead's -m check (src/FingerPrint.hpp:473-488)  ntsm_sync -> ntsm_totals.early_stop
 *   m_totalKmers / m_totalCounts / m_totalBases                ntsm_sync -> ntsm_totals
 *   m_counts.at(hv) at print time (src/FingerPrint.hpp:282)    ntsm_counts (dense, key order)
 *   (none: single process)                                     ntsm_counts_device + RCCL SUM
 *
 * Flat read-stream layout consumed by submit/count: the reads of a batch are concatenated, each
 * read followed by exactly ONE terminator byte 'N' (an invalid base resets the k-mer window
 * exactly like vendor/KseqHashIterator.hpp:106, so no k-mer spans two reads).  read_end[i] is
 * the offset of read i's terminator; read i spans [i ? read_end[i-1]+1 : 0, read_end[i]);
 * n_bytes == read_end[n_reads-1] + 1.  Bases are the raw bytes of seq.s: the kernel applies
 * the reference's byte->code table itself (vendor/KseqHashIterator.hpp:114-127).
 *
 * Conventions: every function returns NTSM_OK (0) or a negative NTSM_ERR_* code
 * (ntsm_strerror); no C++ types or exceptions cross the boundary; the caller owns every
 * buffer it passes; a context owns its device memory and streams and must be driven by one
 * host thread at a time.  There is NO CPU fallback: without a usable HIP device ntsm_create
 * fails with NTSM_ERR_NO_DEVICE.
 */
#ifndef NTSM_HIP_H
#define NTSM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ntsm_ctx ntsm_ctx;

/* Cumulative totals of a context (the reference's m_totalKmers, m_totalCounts, m_totalBases). */
typedef struct {
	uint64_t total_kmers;      /* valid k-mer windows seen, hits or not  (src/FingerPrint.hpp:98-99) */
	uint64_t total_hits;       /* windows found in the site set          (src/FingerPrint.hpp:96-97) */
	uint64_t total_bases;      /* sum of read lengths, 'N' included      (src/FingerPrint.hpp:101-102) */
	uint64_t reads_consumed;   /* reads that contributed (all submitted reads unless early_stop) */
	int32_t  early_stop;       /* 1 once total_hits > max_hits after a whole read (src/FingerPrint.hpp:476-487) */
	int32_t  reserved;
} ntsm_totals;

enum {
	NTSM_OK = 0,
	NTSM_ERR_ARG = -1,         /* bad argument (NULL, k out of range, misaligned device pointer, bad layout) */
	NTSM_ERR_NO_DEVICE = -2,   /* no usable HIP device / device index out of range */
	NTSM_ERR_HIP = -3,         /* a HIP runtime call failed (ntsm_last_hip_error) */
	NTSM_ERR_DUP_KEY = -4,     /* duplicate key in the k-mer set */
	NTSM_ERR_NOMEM = -5,
	NTSM_ERR_STATE = -6,       /* call not valid in the current state (e.g. staged slot not acquired) */
	NTSM_ERR_RCCL = -7
};

enum {
	NTSM_KEYS_CANONICAL = 0,   /* keys are canonical 2-bit codes min(fw, rc) (KseqHashIterator.hpp:102-104) */
	NTSM_KEYS_HASH64 = 1       /* keys are hash64(canonical, mask): exactly m_counts' keys (KseqHashIterator.hpp:129-139) */
};

/* Build a context on HIP device `device`.
 *   k        : k-mer size, 1..32 (src/ntSeqMatchCount.cpp:147-150; k = 32 mirrors the reference's
 *              degenerate mask = 0)
 *   keys     : n_kmers distinct keys; the dense index of a key is its position in this array
 *   key_kind : NTSM_KEYS_CANONICAL or NTSM_KEYS_HASH64 (inverted on the host; hash64 is a bijection)
 *   max_hits : the reference's m_maxCounts; 0 disables the -m early stop */
int ntsm_create(ntsm_ctx **out, int device, int k, const uint64_t *keys, uint32_t n_kmers,
		int key_kind, uint64_t max_hits);
void ntsm_destroy(ntsm_ctx *ctx);

/* Count one batch of reads held in HOST memory (copied into pinned staging, then asynchronous
 * H2D + kernel on one of two internal streams).  The caller's buffers may be reused on return.
 * The staging copy runs on several threads (one thread's memcpy is about half of what a PCIe Gen5 x16 link
 * takes): ntsm_set_submit_threads sets how many, the submitting thread included (0 = automatic: min(6, CPUs of
 * the affinity mask); 1 = the submitting thread alone). */
int ntsm_submit(ntsm_ctx *ctx, const uint8_t *bases, uint64_t n_bytes, const uint64_t *read_end,
		uint32_t n_reads);
int ntsm_set_submit_threads(ntsm_ctx *ctx, int n_threads);
/* Zero-copy variant for a caller whose reads already lie in PINNED host memory (hipHostMalloc, hipHostRegister, or
 * ntsm_host_pin below; anything else is refused with NTSM_ERR_ARG): the H2D copy reads the caller's buffer directly, no
 * staging copy.  Asynchronous with two batches in flight: `bases` must stay untouched until the SECOND next
 * ntsm_submit_pinned call on this context has returned, or until ntsm_sync (an armed context, max_hits != 0, is
 * synchronous: free on return).  read_end is ordinary memory and free on return.  This is the form of the per-read call
 * src/FingerPrint.hpp:66-69 that needs no host-side copy at all when the parser writes into pinned memory it owns. */
int ntsm_submit_pinned(ntsm_ctx *ctx, const uint8_t *bases, uint64_t n_bytes, const uint64_t *read_end,
		uint32_t n_reads);
/* hipHostRegister / hipHostUnregister for callers that do not link the HIP runtime themselves: pin an existing host
 * allocation (about 0.16 ms per MiB, serialised by the driver: do it once per buffer, not per batch). */
int ntsm_host_pin(void *p, uint64_t bytes);
int ntsm_host_unpin(void *p);

/* Zero-copy variant: parse straight into pinned staging.  acquire blocks until the slot's previous
 * batch has left the host buffer; fill at most *cap_bytes bases / *cap_reads offsets, then submit. */
int ntsm_staging_acquire(ntsm_ctx *ctx, uint8_t **bases, uint64_t *cap_bytes, uint64_t **read_end,
		uint64_t *cap_reads);
int ntsm_submit_staged(ntsm_ctx *ctx, uint64_t n_bytes, uint32_t n_reads);
/* Resize the two staging slots (default 64 MiB of bases, 1 Mi reads each). */
int ntsm_set_batch_capacity(ntsm_ctx *ctx, uint64_t cap_bytes, uint64_t cap_reads);

/* Producer lanes: SEVERAL host threads feeding ONE context -- the reference's `omp parallel for` over files with a
 * shared m_counts and `#pragma omp atomic` increments (src/FingerPrint.hpp:47, :94-99).  A lane is one thread's
 * private pair of pinned staging slots (cap_bytes of bases / cap_reads offsets each; 0 = the context's defaults);
 * the lanes of a context share its two lane streams and every lane counts into the context's tables.  Calls on different lanes may run
 * concurrently; one lane is driven by one thread.  acquire/submit behave like ntsm_staging_acquire /
 * ntsm_submit_staged.  close drains the lane and folds its totals into the context; ntsm_sync, ntsm_counts*,
 * ntsm_reset and ntsm_set_tuning return NTSM_ERR_STATE while a lane is open.  Lanes are refused
 * (NTSM_ERR_STATE) on a context with max_hits != 0: the -m stop is defined on one ordered stream of reads. */
typedef struct ntsm_lane ntsm_lane;
int ntsm_lane_open(ntsm_ctx *ctx, uint64_t cap_bytes, uint64_t cap_reads, ntsm_lane **out);
int ntsm_lane_acquire(ntsm_lane *lane, uint8_t **bases, uint64_t *cap_bytes, uint64_t **read_end,
		uint64_t *cap_reads);
int ntsm_lane_submit(ntsm_lane *lane, uint64_t n_bytes, uint32_t n_reads);
int ntsm_lane_close(ntsm_lane *lane);
/* Packed batches on a lane: what crosses PCIe is 2 bits of code + 1 validity bit per stream position -- the five classes
 * the reference's byte table knows (vendor/KseqHashIterator.hpp:114-127: A a 0x00 | C c 0x01 | G g 0x02 | T t U u 0x03 |
 * everything else) -- 3/8 byte per position instead of 1; the library unpacks it on the device into the flat stream
 * (codes as raw bytes 0..3, invalid positions as 'N') and counts that.  Layout of a batch of P positions:
 *     codes[P / 4]   position p -> bits 2(p & 3).. of codes[p >> 2]  (little endian inside the byte; 0 where invalid)
 *     valid[P / 8]   position p -> bit p & 7 of valid[p >> 3]
 * Reads need not be separated by exactly one invalid position: any number >= 1 will do (a window restarts at every
 * invalid position, KseqHashIterator.hpp:106); ntsm_amd/csrc/host/pack2.hpp starts every read at a multiple of 8.
 * acquire hands out the two planes inside the lane's pinned slot (room for *cap_positions, a multiple of 32; buffers may
 * be written up to that position whatever the batch ends up holding); submit takes n_positions (a multiple of 8), the
 * number of reads and the sum of their lengths (the reference's m_totalBases, which the packed form no longer shows).
 * Both kinds of batches may be mixed on one lane.  A lane opened with ntsm_lane_open_packed takes packed batches only
 * (ntsm_lane_acquire answers NTSM_ERR_STATE) and pins 3/8 byte per position instead of 1: less pinned memory to allocate
 * at start-up (pinning costs ~0.16 ms/MiB and serialises with every other HIP call of the process). */
int ntsm_lane_open_packed(ntsm_ctx *ctx, uint64_t cap_positions, ntsm_lane **out);
int ntsm_lane_acquire_packed(ntsm_lane *lane, uint8_t **codes, uint8_t **valid, uint64_t *cap_positions);
int ntsm_lane_submit_packed(ntsm_lane *lane, uint64_t n_positions, uint32_t n_reads, uint64_t n_bases);

/* Initialise the HIP runtime and the device context of `device` and put `n_streams` ready-made streams into the
 * library's per-device stream pool (a context takes 4 -- two staging slots, resident batches, and the ONE copy stream every
 * host-to-device batch copy of the context is issued on --, plus 2 once it has lanes; streams go back to the pool when
 * the context is destroyed).  Thread-safe.  Creating a stream costs ~14 ms on this runtime, so a host calls this on a side thread
 * while it loads the sites file (src/FingerPrint.hpp:489-572), before ntsm_create.  Optional: everything is
 * created on demand otherwise. */
int ntsm_warmup(int device, int n_streams);

/* Reserve a process-wide pool of `bytes` of pinned host memory that contexts and lanes carve their staging slots
 * from (they fall back to individual allocations when it is exhausted).  Pinning costs ~0.4 ms/MiB and the driver
 * serialises it, so a host calls this once, early and off its critical path (ntsmCount: on a side thread while the
 * sites file is parsed).  A second call succeeds if the pool is already at least that large.  bytes = 0 releases
 * the pool (NTSM_ERR_STATE while slots are still taken from it); otherwise it lives until the process ends. */
int ntsm_staging_pool(uint64_t bytes);

/* Count a batch already RESIDENT in device memory (d_bases 16-byte aligned).  d_read_end may be
 * NULL when the context has no early stop armed (max_hits == 0).  sign = +1 counts, -1 removes
 * the batch's contribution again (exact: integer adds).  Asynchronous on the context's stream.
 * The kernels read the stream in aligned 16-byte granules: when n_bytes is not a multiple of 16 the last
 * granule is read whole, up to 15 bytes past n_bytes.  Those bytes are ignored and the access cannot fault
 * (an aligned granule never leaves the page of its first, valid byte), but tools that track allocations
 * byte-exactly will want the buffer padded to a multiple of 16. */
int ntsm_count_resident(ntsm_ctx *ctx, const void *d_bases, uint64_t n_bytes, const void *d_read_end,
		uint64_t n_reads, int sign);

/* Drain all streams; fill cumulative totals. */
int ntsm_sync(ntsm_ctx *ctx, ntsm_totals *totals);
/* Per-k-mer counts (dense, in key order), 64-bit like m_counts' mapped size_t.  Implies a sync. */
int ntsm_counts(ntsm_ctx *ctx, uint64_t *out);
/* Device pointer to the dense uint64 count vector followed by 4 uint64 totals
 * {kmers, hits, bases, reads}: n_kmers + 4 words, refreshed by this call (implies a sync), for a
 * caller-run RCCL SUM (torch.distributed all_reduce); ntsm_import_reduced loads the reduced
 * vector back so that ntsm_counts/ntsm_sync report job-wide values. */
int ntsm_counts_device(ntsm_ctx *ctx, void **d_vec, uint64_t *n_words);
int ntsm_import_reduced(ntsm_ctx *ctx);
/* Single-process multi-GPU merge: RCCL SUM over the n contexts' count vectors + totals (xGMI); afterwards every context
 * reports the job-wide result.  Contexts that share a device are summed there first and RCCL runs between one context per
 * distinct device -- with a single distinct device no collective runs at all (`ntsmCount -g 0,0`). */
int ntsm_allreduce(ntsm_ctx *const *ctxs, int n);
/* Bind RCCL now (dlopen of librccl.so.1 + the five entry points ntsm_allreduce calls: ncclCommInitAll, ncclGroupStart,
 * ncclGroupEnd, ncclAllReduce, ncclCommDestroy) and say whether it worked: NTSM_OK or NTSM_ERR_RCCL.  Touches no GPU.
 * A multi-device host calls it BEFORE counting, so that a missing RCCL shows up front rather than after the work. */
int ntsm_rccl_probe(void);
/* Re-arm or disarm the -m stop of a context.  The threshold is compared (strict '>', after every whole read) with
 * the context's OWN cumulative total_hits, so a caller that orders reads across several contexts passes
 * total_hits_of_this_context + (global threshold - hits counted globally before the next batch); see
 * ntsm_amd/dist.py OrderedEarlyStop for the multi-GPU protocol built on it (the reference's stop,
 * src/FingerPrint.hpp:473-488, is single-process).  armed = 0: batches are counted without the check (max_hits is
 * ignored); armed = 1 with max_hits = 0 stops after the first read that has a hit.  Does not clear early_stop. */
int ntsm_set_max_hits(ntsm_ctx *ctx, uint64_t max_hits, int armed);
/* Forget all counts and totals (table stays). */
int ntsm_reset(ntsm_ctx *ctx);

/* Kernel timing for bench.py: when on, every count launch is bracketed by hipEvents on the stream
 * it runs on; get returns the number of launches and the sum of their durations since `on`. */
int ntsm_set_timing(ntsm_ctx *ctx, int on);
int ntsm_get_timing(ntsm_ctx *ctx, uint64_t *n_launches, double *total_ms);
/* Tuning knobs (0 = automatic): log2 of filter bits (rebuilds the tables: counts and totals restart from zero; 100 + v =
 * 3 * 2^v bits; 2000000 + w = w KiB (with ntsm_set_kernel 5: the run form's filter; 4000000 + v: memory kind of filter / key table, an experiment that measured no effect); 200 + v / 250 + v / 1000000 + w = size of the two-level path's minimizer Bloom, 2^v /
 * 3 * 2^v bits / w KiB), grid blocks.
 * For profiling experiments. */
int ntsm_set_tuning(ntsm_ctx *ctx, int filter_log2_bits, int grid_blocks);
/* An armed (-m) batch is walked in chunks of about chunk_bytes of stream (at most 2^20 reads, at least 1024) so that the work
 * is proportional to what is consumed before the stop; 0 = the default (256 MiB).  Any value gives the same result. */
int ntsm_set_armed_chunk(ntsm_ctx *ctx, uint64_t chunk_bytes);
/* Kernel choice.  All choices give identical results.
 *   0  automatic: 13 <= k <= 31 the minimizer-blocked kernel, other k the generic kernel; 15 <= k <= 31 with a site set
 *      whose blocked filter is well out of the L2 (more than ~3.1 M k-mers) takes the two-level form of the kernel: 14-mer
 *      minimizers and a Bloom word over the distinct site minimizers in front of the block;
 *      k = 19 with 1.8 M <= site k-mers < 8 M takes the run-anchored kernel (5) for unarmed batches
 *   1  always the generic kernel
 *   2  the minimizer-blocked kernel, one level, whatever the size of the set
 *   4  15 <= k <= 31 only: the two-level form, whatever the size of the set
 *   5  k = 19 only: the run-anchored kernel (ntsm_amd/csrc/kernels_run.hip) -- one filter test per
 *      minimizer run on the run's anchored 16-mers instead of one per k-mer -- whatever the size of the set; armed (-m)
 *      batches of such a context still attribute hits per read with the minimizer-blocked kernel, whose one-level tables
 *      are kept beside the run form's filter
 *   3  the tabulated k = 19 kernel, a measured negative result (7 % slower) that only exists in
 *      -DNTSM_WITH_TAB builds (`make tab`: ntsm_amd/libntsm_hip_tab.so); the default library answers NTSM_ERR_ARG.
 * One-level, two-level and run-form filters are different tables: a call that changes the form rebuilds them, and counts and
 * totals restart from zero (like ntsm_set_tuning with a filter size); variant 3 always counts with the one-level tables (its
 * exotic tiles go to the one-level k = 19 kernel), so it rebuilds them on a context that had chosen two levels.
 * A rebuild (here or in ntsm_set_tuning) that fails half way -- out of device memory -- leaves no consistent set of tables:
 * the context is marked failed and every later call on it answers NTSM_ERR_STATE; ntsm_destroy is the only way out. */
int ntsm_set_kernel(ntsm_ctx *ctx, int variant);
/* Introspection for tests and profiles (implies a sync): out[0] = 64 KiB tiles the tabulated kernel handed to the
 * exact kernel because they hold bytes outside ACGTUNacgtun, out[1..3] = count launches by kernel
 * (tabulated, minimizer-blocked, generic), out[4] = windows that passed the tabulated kernel's first-level filter and
 * were queued for the look-up kernel (since creation or the last ntsm_reset), out[5] = 1 when the tables are the
 * two-level ones, 2 when the run-anchored kernel counts the unarmed batches, out[6] = words of the minimizer Bloom, out[7] = distinct site minimizers (two-level tables). */
int ntsm_debug_stats(ntsm_ctx *ctx, uint64_t out[8]);
/* Fault injection for tests of the failure paths (compiled into every build, armed ONLY through this call -- the library
 * reads no environment variable).  kind 1: device allocations, 2: host-to-device copies, 3: pinned host allocations.
 * nth > 0: the nth call of that kind from now on (process-wide, any context) does not reach the HIP runtime and reports
 * hipErrorOutOfMemory (allocations) / hipErrorUnknown (copies); only that one call fails.  nth = 0 disarms.  Returns the
 * number of calls of that kind seen since the previous arming (so a test can first count the calls an operation makes and
 * then fail each in turn), -1 for an unknown kind.  What a failure must look like to the caller: ntsm_create returns an
 * error and leaves nothing allocated; a failed table rebuild (ntsm_set_kernel / ntsm_set_tuning) or a lost lane batch marks
 * the context failed -- every later counting / merging / reporting call answers NTSM_ERR_STATE, ntsm_lane_close repeats the
 * lane's first error -- so an incomplete count can never be printed (the reference: exit(1) with a message,
 * src/FingerPrint.hpp:51-57, :493-499). */
long long ntsm_debug_fail_after(int kind, long long nth);
/* Test hook, host code only (no device needed): the run-anchored kernel's filter (ntsm_set_kernel 5, k = 19) for `keys` (canonical
 * codes) as ntsm_create would build it, kib = its size in KiB (0 = automatic).  *n_blocks receives the number of 128-bit blocks;
 * blocks_out, if not NULL, the image (4 words per block; call once with NULL to learn the size). */
int ntsm_debug_run_filter(const uint64_t *keys, uint32_t n_kmers, uint32_t kib, uint32_t *blocks_out, uint64_t *n_blocks);
/* Host code only (no device needed): which kernel form ntsm_create would choose by itself for this key set -- *form = 0 the one-level
 * minimizer-blocked kernel, 1 its two-level form, 2 the run-anchored kernel, 3 the generic kernel (k < 13, k = 32).  The choice is a
 * function of the SET of keys (count + cluster structure estimated on a minimizer-residue sample), not of the order they are passed in:
 * site-file order, shuffled, or m_counts' iteration order (hash order, src/FingerPrint.hpp:466) give the same answer. */
int ntsm_debug_form_choice(const uint64_t *keys, uint32_t n_kmers, int k, int key_kind, int *form);
/* The HIP stream (hipStream_t) ntsm_count_resident launches on. */
void *ntsm_stream(ntsm_ctx *ctx);

/* Host helpers: the reference's hash64 and its inverse on 2k bits (KseqHashIterator.hpp:129-139). */
uint64_t ntsm_hash64(uint64_t key, int k);
uint64_t ntsm_hash64_inv(uint64_t hv, int k);

const char *ntsm_strerror(int code);
int ntsm_last_hip_error(void);
const char *ntsm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NTSM_HIP_H */
