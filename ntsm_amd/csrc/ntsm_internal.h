/*
 * ntsm_internal.h -- what the translation units of libntsm_hip.so share on the HOST side: the context and lane structures
 * behind the opaque handles of include/ntsm_hip.h, the process-wide pools, the table builder, the kernel launchers and the
 * fault-injection gate every device allocation and upload passes through.
 *
 *   kernels_generic.hip  generic-k count kernel + helper kernels, launch_generic / launch_gather / ...
 *   kernels_mz.hip       minimizer-blocked count kernels, launch_mz (+ the tabulated kernel in `make tab` builds)
 *   kernels_run.hip      run-anchored count kernel (k = 19, one filter test per minimizer run), launch_run
 *   tables.cpp           host-side construction of the cuckoo key table and the filters (no HIP call, no device code)
 *   runtime.cpp          pools, staging slots, table upload, launch_count, the exact -m early stop (armed_batch)
 *   rccl_bind.cpp        RCCL bound with dlopen on first use; the group all-reduce of ntsm_allreduce
 *   capi.cpp             the C ABI (include/ntsm_hip.h): argument checks and state, nothing else
 */
#ifndef NTSM_INTERNAL_H
#define NTSM_INTERNAL_H
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

/* the library is compiled with -fvisibility=hidden: the C ABI is what it exports, nothing of ntsm_rt:: */
#pragma GCC visibility push(default)
#include "../../include/ntsm_hip.h"
#pragma GCC visibility pop
#include "ntsm_device.h"
#include "ntsm_hooks.h"

namespace ntsm_rt {

void set_last_hip(int e);
int last_hip();

constexpr int kTileC = 128;                 /* generic kernel: stream bytes per thread and tile */
constexpr int kTimingPool = 256;
constexpr int kMaxDevices = 64;

/* ---- fault injection (tests only; compiled in, armed through ntsm_debug_fail_after, never by the environment) ----------
 * Every device allocation, every pinned allocation and every host-to-device copy of this library goes through one of the
 * gates below.  Armed with (kind, n), the n-th call of that kind from then on does not reach the runtime and reports
 * hipErrorOutOfMemory (allocations) or hipErrorUnknown (copies) instead -- what the reference's `exit(1)` paths
 * (src/FingerPrint.hpp:51-57, :493-499) correspond to on a device. */
hipError_t dev_malloc(void **p, size_t bytes, unsigned mem_kind = 0);   /* mem_kind: 0 ordinary, 1 fine-grained, 3 uncached (hipExtMallocWithFlags) */
template <class T> inline hipError_t dev_malloc(T **p, size_t bytes, unsigned mem_kind = 0) { return dev_malloc((void **) p, bytes, mem_kind); }
hipError_t pinned_malloc(void **p, size_t bytes);
hipError_t h2d(void *dst, const void *src, size_t bytes);
hipError_t h2d_async(void *dst, const void *src, size_t bytes, hipStream_t st);
long long fault_arm(int kind, long long nth);   /* ntsm_debug_fail_after */

} // namespace ntsm_rt

#define HIPCHK(call)                                                                      \
	do {                                                                                  \
		hipError_t e_ = (call);                                                           \
		if (e_ != hipSuccess) {                                                           \
			ntsm_rt::set_last_hip((int) e_);                                              \
			fprintf(stderr, "ntsm_hip: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
			return NTSM_ERR_HIP;                                                          \
		}                                                                                 \
	} while (0)

namespace ntsm_rt {

struct Slot {
	uint8_t *h_bases = nullptr, *d_bases = nullptr;
	uint8_t *d_packed = nullptr;               /* packed lanes: device copy of codes + validity bits (3/8 byte per position) */
	uint64_t *h_read_end = nullptr, *d_read_end = nullptr;
	uint64_t h_bases_bytes = 0, h_ends_bytes = 0;
	uint64_t d_bases_bytes = 0, d_packed_bytes = 0;   /* sizes of the device buffers (device_cache of the context) */
	bool ends_on_device = true;                /* false (lanes): read_end never leaves the host, plain malloc */
	hipStream_t stream = nullptr;
	hipEvent_t done = nullptr;                 /* last use of the host buffer finished */
	hipEvent_t copied = nullptr;               /* the batch's H2D copies (on the context's copy stream) finished: what the slot's kernels wait for */
	bool busy = false, acquired = false;
};

/* Helper threads of ntsm_submit's staging copy: the caller's (pageable) batch is copied into the pinned slot in `parts` pieces,
 * piece 0 by the submitting thread, the others by persistent helpers -- one thread's memcpy (~25-30 GB/s into pinned memory) is
 * half of what the link takes (PCIe Gen5 x16: 55-57 GB/s measured), three or four are not. */
struct CopyPool {
	std::vector<std::thread> helpers;
	std::mutex mu;
	std::condition_variable cv_work, cv_done;
	uint8_t *dst = nullptr;
	const uint8_t *src = nullptr;
	uint64_t n = 0;
	unsigned parts = 0, pending = 0;
	uint64_t generation = 0;
	bool quit = false;
};

#ifdef NTSM_WITH_TAB
/* tabulated k = 19 path (ntsm_tab_kernel.inc, `make tab`) */
struct TabState {
	NtsmTabEntry *d_tab = nullptr;
	uint4 *d_tblocks = nullptr;
	uint64_t n_tblocks = 0;
	NtsmTabMap tblk_map = { 0, 0, 0 };
	bool tab_ok = false;                       /* filter built and no site k-mer has the reserved minimizer key */
	/* per launch stream: d_ctl = { [0] fill of the exotic-tile list of the launch in flight, [1 .. 1+2*seg_cap) fills of the
	 * look-up queue slots (tile, wave), two halves, [1+2*seg_cap] exotic tiles seen so far, then the exotic list };
	 * d_queue = canonical codes handed from the tabulated kernel to the look-up kernel (one segment at a time) */
	struct StreamBuf {
		hipStream_t stream; uint32_t *d_ctl; uint64_t seg_cap, exotic_cap; unsigned long long *d_queue; uint64_t queue_cap;
		hipEvent_t ev_tab[2], ev_look[2];          /* two halves of queue + fills: segment s uses half s & 1 */
	};
	std::vector<StreamBuf> sbuf;
	hipStream_t lstream = nullptr;             /* the look-up kernels of all launch streams run here, beside the next segment's scan */
	std::vector<uint32_t> tblocks_host;        /* built by tables.cpp, uploaded and released by runtime.cpp */
};
#else
struct TabState {};
#endif

} // namespace ntsm_rt

struct ntsm_ctx {
	int device = 0, k = 0;
	uint32_t n_kmers = 0;
	uint64_t max_hits = 0;
	bool armed = false;                        /* the -m stop is active (max_hits != 0 at creation, or ntsm_set_max_hits) */
	uint64_t armed_chunk_bytes = 256ull << 20;  /* stream bytes per chunk of an armed batch (ntsm_set_armed_chunk) */
	uint64_t mask = 0;
	/* device tables */
	uint32_t *d_filter = nullptr, *d_slot_of = nullptr, *d_read_hits = nullptr;
	uint64_t read_hits_cap = 0;
	uint64_t *d_keys = nullptr;
	unsigned long long *d_totals = nullptr, *d_vec = nullptr;
	uint8_t *d_lut = nullptr;
	uint2 *d_lut64 = nullptr;
	uint32_t filter_log2 = 0, bucket_log2 = 0;
	uint64_t n_slots = 0;
	uint4 *d_blocks = nullptr;                 /* k = 19 fast path: minimizer-addressed 128-bit filter blocks */
	uint64_t n_blocks = 0;                     /* number of 128-bit filter blocks: mult * 2^e, mult in {1, 3} */
	uint32_t *d_bloom = nullptr;               /* two-level path: Bloom over the distinct site minimizers, in front of the blocks */
	uint32_t n_bloom_words = 0;
	bool two_level = false;                    /* k = 19 and the blocked filter would not fit the L2: 14-mer minimizers + d_bloom */
	uint32_t n_site_minimizers = 0;            /* distinct minimizers of the site k-mers (two-level path only) */
	uint32_t bloom_words_req = 0;              /* tuning: Bloom size in words (0 = automatic) */
	uint32_t blocks_kib_req = 0;               /* tuning: blocked filter size in KiB (0 = automatic / filter_log2_req) */
	uint32_t prefilter_log2_req = 0;           /* tuning: log2 of the drain Bloom's bits (0 = automatic) */
	bool prefilter_forced = false;             /* tuning (code 2): keep the drain's Bloom on the two-level path as well */
	int filter_log2_req = 0;                   /* tuning: what ntsm_set_tuning asked for (kept across rebuilds) */
	/* tuning (ntsm_set_tuning 4000000 + v): memory kind of the block filter (v & 3) and of the key table ((v >> 2) & 3):
	 * 0 ordinary device memory, 1 fine-grained, 3 uncached -- DESIGN.md section 4.2b "what a miss moves" */
	unsigned blocks_mem_kind = 0, keys_mem_kind = 0;
	uint32_t *d_prefilter = nullptr;           /* second-level Bloom used by the drain */
	uint32_t prefilter_log2 = 0;               /* log2(bits) */
	NtsmBlockMap blk_map = { 1 };
	uint4 *d_rblocks = nullptr;                /* run-anchored kernel (variant 5, k = 19): signatures of anchored 16-mers, 128-bit blocks by minimizer */
	uint64_t n_rblocks = 0, n_rentries = 0;
	bool run_form = false;                     /* the run-anchored kernel counts unarmed batches: forced (ntsm_set_kernel 5) or chosen by size (wants_run_form) */
	ntsm_rt::TabState tab;                     /* empty unless built with NTSM_WITH_TAB */
	int look_blocks = 0;
	uint64_t n_launch[3] = { 0, 0, 0 };         /* count launches by kernel: tabulated, minimizer-blocked, generic */
	int kernel_variant = 0;                    /* 0 auto (minimizer-blocked kernel for 13 <= k <= 31), 1 generic, 2 = 0, 3 tabulated k = 19 kernel (NTSM_WITH_TAB builds) */
	std::vector<uint64_t> canon;               /* host copy of the canonical keys */
	std::vector<uint32_t> slot_of;
	/* batching */
	ntsm_rt::Slot slot[2];
	int next_slot = 0;
	uint64_t cap_bytes = 64ull << 20, cap_reads = 1ull << 20;
	hipStream_t rstream = nullptr;             /* stream for resident batches */
	/* ALL host-to-device batch copies of a context (staging slots, ntsm_submit_pinned, every lane) are issued on this ONE stream, in
	 * submission order; a slot's kernels run on the slot's own stream behind the slot's `copied` event.  Copies on two streams share
	 * the link instead of following each other; on most boxes that costs nothing, on some it does (pinned hipMemcpyAsync of 64 MiB
	 * batches alternating on two streams 48 GB/s against 56.8 on one, and ntsm_submit 36 instead of 51: DESIGN.md section 5.1). */
	hipStream_t cstream = nullptr;
	ntsm_rt::CopyPool *copy_pool = nullptr;    /* ntsm_submit's staging copy on several threads (created on the first large batch) */
	int submit_threads = 0;                    /* threads of that copy, the submitting one included (0 = automatic: min(6, CPUs of the affinity mask)) */
	/* host-side totals */
	uint64_t total_bases = 0, reads_consumed = 0;
	bool early_stop = false, reduced = false;
	/* A table rebuild failed half way (ntsm_set_kernel / ntsm_set_tuning: the tables no longer describe one consistent filter
	 * organisation), or a batch of a producer lane was lost (its copy or launch failed: the counts no longer cover what the
	 * caller submitted).  Every later call that would count, merge or report answers NTSM_ERR_STATE; only ntsm_destroy helps. */
	bool failed = false;
	uint64_t red_totals[4] = { 0, 0, 0, 0 };
	/* tuning / timing */
	int grid_blocks = 0, n_cu = 256;
	bool timing = false;
	hipEvent_t ev_a[ntsm_rt::kTimingPool], ev_b[ntsm_rt::kTimingPool];
	bool ev_used[ntsm_rt::kTimingPool];
	int ev_next = 0;
	uint64_t t_launches = 0;
	double t_ms = 0;
	/* producer lanes (ntsm_lane_*): several host threads feeding this context */
	std::mutex mu;                             /* guards open_lanes, lane_stream, the lane totals fold-in, the timing pool, device_cache and `failed` when lanes are open */
	int open_lanes = 0;
	/* Device buffers of closed lanes, kept for the next lane of the same size (and for the process's end): hipFree waits for
	 * the device and costs 0.2 ms a call, 12 ms for the sixteen lanes of an `ntsmCount -t 16` run that is otherwise over.
	 * At most kDeviceCacheMax entries; the oldest is freed when one more comes in.  Released by ntsm_destroy. */
	std::vector<std::pair<void *, uint64_t>> device_cache;
	static constexpr size_t kDeviceCacheMax = 128;
	hipStream_t lane_stream[2] = { nullptr, nullptr };   /* shared by all lanes (round robin): a stream costs 14 ms to create */
	unsigned lanes_opened = 0;
};

/* One producer thread's private staging: two pinned slots + their device mirrors and streams.  All lanes of a
 * context count into the same tables (atomic adds), which is the reference's omp-over-files with a shared m_counts
 * and `#pragma omp atomic` (src/FingerPrint.hpp:47, :94-99). */
struct ntsm_lane {
	ntsm_ctx *c = nullptr;
	ntsm_rt::Slot slot[2];
	int next_slot = 0;
	uint64_t cap_bytes = 0, cap_reads = 0;
	bool packed_only = false;                  /* ntsm_lane_open_packed: pinned slots of 3/8 byte per position, no byte batches */
	uint64_t total_bases = 0, reads_consumed = 0;     /* folded into the context by ntsm_lane_close */
	int error = NTSM_OK;                       /* first failed submit: sticky, reported again by ntsm_lane_close */
};

namespace ntsm_rt {

/* ---- kernels_generic.hip / kernels_mz.hip: launchers (the kernels themselves live in those files' anonymous namespaces) */
hipError_t launch_generic(const NtsmCountParams &p, unsigned grid, hipStream_t st, bool per_read);
hipError_t launch_mz(const NtsmCountParams &p, unsigned grid, hipStream_t st, int mode, bool per_read, bool two_level);
int mz_tile_bytes();                         /* stream bytes per tile of the minimizer-blocked kernels */
hipError_t launch_run(const NtsmCountParams &p, unsigned grid, hipStream_t st);   /* kernels_run.hip: run-anchored kernel, k = 19 */
int run_tile_bytes();
hipError_t launch_gather(const uint64_t *table, const uint32_t *slot_of, uint32_t n, unsigned long long *dense, hipStream_t st);
hipError_t launch_unpack(const uint32_t *codes, const uint16_t *valid, void *out, unsigned long long n16, hipStream_t st);
hipError_t launch_table_init(uint64_t *table, unsigned long long n_buckets, hipStream_t st);
hipError_t launch_table_scatter(uint64_t *table, const uint32_t *slot_of, const uint64_t *canon, uint32_t n, hipStream_t st);
hipError_t launch_zero_counts(uint64_t *table, unsigned long long n_buckets, hipStream_t st);
#ifdef NTSM_WITH_TAB
/* the whole tabulated launch (queue / list buffers of this stream, segments, look-up kernels, the list walker); caller holds c->mu */
int launch_tab(ntsm_ctx *c, hipStream_t st, const NtsmCountParams &p, uint64_t hi);
#endif

/* ---- tables.cpp: the four (five) structures as host images; no HIP call */
struct TableImages {
	std::vector<uint32_t> filter, blocks /* 4 words per block */, prefilter, bloom, rblocks /* run-anchored kernel */;
};
bool wants_two_level(uint64_t n_keys);
bool wants_run_form(int k, uint64_t n_keys);  /* k = 19 and 1.8 M <= keys < 7 M: kernels_run.hip beats both forms of kernels_mz.hip there */
bool choose_run_form(const ntsm_ctx *c, int variant, int filter_log2_req);   /* forced (5), or automatic: size window + cluster structure */
int build_tables_host(ntsm_ctx *c, int filter_log2_req, TableImages &img);   /* NTSM_OK / NTSM_ERR_DUP_KEY; sets the geometry fields of *c */
uint64_t mask_for_k(int k);
void build_lut(uint8_t *lut);                /* vendor/KseqHashIterator.hpp:114-127 as data */

/* ---- runtime.cpp */
void *pool_alloc(uint64_t bytes);
bool pool_free(void *ptr, uint64_t bytes);
int staging_pool(uint64_t bytes);
hipStream_t stream_get(int device);
void stream_put(int device, hipStream_t s);
int build_tables(ntsm_ctx *c, int filter_log2_req, int (*before_upload)(ntsm_ctx *) = nullptr);
hipError_t device_take(ntsm_ctx *c, void **p, uint64_t bytes);
void device_give(ntsm_ctx *c, void *p, uint64_t bytes);
int alloc_slot(Slot &s, int device, uint64_t cap_bytes, uint64_t cap_reads, bool ends_on_device, bool packed_only = false, ntsm_ctx *cache = nullptr,
		bool host_bases = true);
int slot_add_host_bases(Slot &s);            /* pin the bases staging of a slot created without one (ntsm_submit_pinned came first) */
void staged_copy(ntsm_ctx *c, uint8_t *dst, const uint8_t *src, uint64_t n);   /* ntsm_submit: the batch into the pinned slot, on submit_threads threads */
void copy_pool_release(ntsm_ctx *c);
/* enqueue `n_copies` host-to-device copies of one batch on the context's copy stream and make the slot's stream wait for them */
hipError_t slot_copy(ntsm_ctx *c, Slot &s, void *const *dst, const void *const *src, const size_t *bytes, int n_copies);
void free_slot(Slot &s, ntsm_ctx *cache = nullptr);
int launch_count(ntsm_ctx *c, hipStream_t st, const uint8_t *d_bases, uint64_t lo, uint64_t hi,
		const uint64_t *d_read_end, uint64_t n_reads, bool per_read, int sign);
int read_device_totals(ntsm_ctx *c, uint64_t out[2]);
int armed_batch(ntsm_ctx *c, hipStream_t st, const uint8_t *d_bases, uint64_t n_bytes,
		const uint64_t *d_read_end, const uint64_t *h_read_end_or_null, uint64_t n_reads);
int check_layout(const uint64_t *read_end, uint32_t n_reads, uint64_t n_bytes);
int submit_slot(ntsm_ctx *c, Slot &s, uint64_t n_bytes, uint32_t n_reads);
int wait_slot(Slot &s);
void tab_release(ntsm_ctx *c);               /* no-op unless NTSM_WITH_TAB */
uint64_t tab_exotic_seen(ntsm_ctx *c, int *rc);

/* ---- rccl_bind.cpp */
bool rccl_available();
int rccl_group_allreduce(ntsm_ctx *const *ctxs, int n);     /* n >= 2 contexts on distinct devices: SUM of d_vec, all synchronised on return */

} // namespace ntsm_rt
#endif
