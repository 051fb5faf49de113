/*
 * runtime.cpp -- the host runtime around the kernels: process-wide pools (pinned staging, HIP streams), staging slots of
 * contexts and producer lanes, the upload of the table images, the count launch (launch_count) and the exact -m early stop
 * (armed_batch: src/FingerPrint.hpp:473-488 -- checked after each whole read, strict '>').  No device code; every HIP
 * return value is checked (the reference's failure contract is `exit(1)` with a message, src/FingerPrint.hpp:51-57).
 */
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include <sched.h>

#include "ntsm_internal.h"

namespace ntsm_rt {

static thread_local int g_last_hip = 0;
void set_last_hip(int e) { g_last_hip = e; }
int last_hip() { return g_last_hip; }

/* ---- fault injection gate (ntsm_debug_fail_after) ------------------------------------------------------------------------
 * kind 1: device allocations, 2: host-to-device copies, 3: pinned host allocations.  armed[k] = n: the n-th call of kind k
 * from now on fails (and only that one); 0 = disarmed.  seen[k] counts the calls since the last arming, so a test can first
 * learn how many such calls an operation makes and then fail each of them in turn. */
static std::atomic<long long> g_fault_armed[4] = { { 0 }, { 0 }, { 0 }, { 0 } };
static std::atomic<long long> g_fault_seen[4] = { { 0 }, { 0 }, { 0 }, { 0 } };

long long fault_arm(int kind, long long nth)
{
	if (kind < 1 || kind > 3) return -1;
	const long long before = g_fault_seen[kind].exchange(0);
	g_fault_armed[kind].store(nth > 0 ? nth : 0);
	return before;
}

static inline bool fault_fires(int kind)
{
	const long long s = g_fault_seen[kind].fetch_add(1) + 1;
	const long long a = g_fault_armed[kind].load(std::memory_order_relaxed);
	return a > 0 && s == a;
}

hipError_t dev_malloc(void **p, size_t bytes, unsigned mem_kind)
{
	if (fault_fires(1)) { *p = nullptr; return hipErrorOutOfMemory; }
	if (mem_kind == 1) return hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained);
	if (mem_kind == 3) return hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached);
	return hipMalloc(p, bytes);
}

hipError_t pinned_malloc(void **p, size_t bytes)
{
	if (fault_fires(3)) { *p = nullptr; return hipErrorOutOfMemory; }
	return hipHostMalloc(p, bytes, hipHostMallocPortable);
}

hipError_t h2d(void *dst, const void *src, size_t bytes)
{
	if (fault_fires(2)) return hipErrorUnknown;
	return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
}

hipError_t h2d_async(void *dst, const void *src, size_t bytes, hipStream_t st)
{
	if (fault_fires(2)) return hipErrorUnknown;
	return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
}

/* Process-wide pool of pinned host memory (ntsm_staging_pool): pinning costs ~0.4 ms/MiB and the driver serialises
 * it, so staging slots are carved out of one early allocation instead of being pinned one by one. */
struct PinnedPool {
	std::mutex mu;
	uint8_t *base = nullptr;
	uint64_t size = 0, bump = 0, outstanding = 0;
	std::vector<std::pair<uint64_t, uint64_t>> free_list;      /* (offset, bytes) of returned pieces, reused by exact size */
};
static PinnedPool g_pool;

void *pool_alloc(uint64_t bytes)
{
	bytes = (bytes + 4095) & ~4095ull;
	std::lock_guard<std::mutex> lk(g_pool.mu);
	if (!g_pool.base) return nullptr;
	for (size_t i = 0; i < g_pool.free_list.size(); ++i)
		if (g_pool.free_list[i].second == bytes) {
			void *p = g_pool.base + g_pool.free_list[i].first;
			g_pool.free_list.erase(g_pool.free_list.begin() + (long) i);
			g_pool.outstanding++;
			return p;
		}
	if (g_pool.bump + bytes > g_pool.size) return nullptr;
	void *p = g_pool.base + g_pool.bump;
	g_pool.bump += bytes;
	g_pool.outstanding++;
	return p;
}

bool pool_free(void *ptr, uint64_t bytes)
{
	bytes = (bytes + 4095) & ~4095ull;
	std::lock_guard<std::mutex> lk(g_pool.mu);
	uint8_t *p = (uint8_t *) ptr;
	if (!g_pool.base || p < g_pool.base || p >= g_pool.base + g_pool.size) return false;
	g_pool.free_list.emplace_back((uint64_t) (p - g_pool.base), bytes);
	if (--g_pool.outstanding == 0) {                            /* everything came back: start over with one free region */
		g_pool.free_list.clear();
		g_pool.bump = 0;
	}
	return true;
}

/* Process-wide pool of non-blocking HIP streams per device: creating a stream costs ~14 ms and destroying one
 * ~3 ms on this runtime (tools/api_cost.hip), far more than anything else a lane needs, so streams are recycled
 * and can be created ahead of time by ntsm_warmup. */
struct StreamPool {
	std::mutex mu;
	std::vector<hipStream_t> idle[kMaxDevices];
};
static StreamPool g_streams;

hipStream_t stream_get(int device)            /* the calling thread's current device must be `device` */
{
	if (device >= 0 && device < kMaxDevices) {
		std::lock_guard<std::mutex> lk(g_streams.mu);
		auto &v = g_streams.idle[device];
		if (!v.empty()) { hipStream_t s = v.back(); v.pop_back(); return s; }
	}
	hipStream_t s = nullptr;
	if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
	return s;
}

void stream_put(int device, hipStream_t s)    /* s must be idle (synchronised) */
{
	if (!s) return;
	if (device < 0 || device >= kMaxDevices) { (void) hipStreamDestroy(s); return; }
	std::lock_guard<std::mutex> lk(g_streams.mu);
	g_streams.idle[device].push_back(s);
}

int staging_pool(uint64_t bytes)
{
	std::lock_guard<std::mutex> lk(g_pool.mu);
	if (bytes == 0) {                                     /* release */
		if (g_pool.outstanding) return NTSM_ERR_STATE;
		if (g_pool.base) HIPCHK(hipHostFree(g_pool.base));
		g_pool.base = nullptr;
		g_pool.size = g_pool.bump = 0;
		g_pool.free_list.clear();
		return NTSM_OK;
	}
	if (g_pool.base) return g_pool.size >= bytes ? NTSM_OK : NTSM_ERR_STATE;
	bytes = (bytes + 4095) & ~4095ull;
	void *p = nullptr;
	HIPCHK(pinned_malloc(&p, bytes));
	g_pool.base = (uint8_t *) p;
	g_pool.size = bytes;
	g_pool.bump = 0;
	return NTSM_OK;
}

#ifdef NTSM_WITH_TAB
#include "ntsm_tab_runtime.inc"
#else
static inline int tab_upload(ntsm_ctx *) { return NTSM_OK; }
static inline bool tab_applies(const ntsm_ctx *) { return false; }
static inline int tab_launch(ntsm_ctx *, hipStream_t, const NtsmCountParams &, uint64_t) { return NTSM_ERR_STATE; }
void tab_release(ntsm_ctx *) {}
uint64_t tab_exotic_seen(ntsm_ctx *, int *rc) { if (rc) *rc = NTSM_OK; return 0; }
#endif

/* Host part first (tables.cpp: no HIP call -- ntsm_create runs it while another thread may still be bringing the runtime up,
 * 0.2 s on this stack, during which every HIP call of this thread would only wait), then `before_upload` (ntsm_create: device
 * checks and hipSetDevice), then the uploads.  A failure after the first hipFree below leaves the context without a
 * consistent set of tables: callers that rebuild an existing context mark it failed (NTSM_ERR_STATE from then on). */
int build_tables(ntsm_ctx *c, int filter_log2_req, int (*before_upload)(ntsm_ctx *))
{
	const uint32_t n = c->n_kmers;
	TableImages img;
	const int cuckoo_rc = build_tables_host(c, filter_log2_req, img);
	if (cuckoo_rc) return cuckoo_rc;
	if (before_upload) {
		const int rc0 = before_upload(c);
		if (rc0) return rc0;
	}
	/* upload */
	if (c->d_blocks) (void) hipFree(c->d_blocks);
	if (c->d_prefilter) (void) hipFree(c->d_prefilter);
	if (c->d_bloom) (void) hipFree(c->d_bloom);
	c->d_blocks = nullptr;
	c->d_prefilter = nullptr;
	c->d_bloom = nullptr;
	if (!c->two_level) c->n_bloom_words = 0;
	if (!img.bloom.empty()) {
		HIPCHK(dev_malloc(&c->d_bloom, img.bloom.size() * sizeof(uint32_t)));
		HIPCHK(h2d(c->d_bloom, img.bloom.data(), img.bloom.size() * sizeof(uint32_t)));
	}
	if (!img.prefilter.empty()) {
		HIPCHK(dev_malloc(&c->d_prefilter, img.prefilter.size() * sizeof(uint32_t)));
		HIPCHK(h2d(c->d_prefilter, img.prefilter.data(), img.prefilter.size() * sizeof(uint32_t)));
	}
	if (!img.blocks.empty()) {
		HIPCHK(dev_malloc(&c->d_blocks, img.blocks.size() * sizeof(uint32_t), c->blocks_mem_kind));
		HIPCHK(h2d(c->d_blocks, img.blocks.data(), img.blocks.size() * sizeof(uint32_t)));
	}
	if (c->d_rblocks) (void) hipFree(c->d_rblocks);
	c->d_rblocks = nullptr;
	if (!img.rblocks.empty()) {
		HIPCHK(dev_malloc(&c->d_rblocks, img.rblocks.size() * sizeof(uint32_t)));
		HIPCHK(h2d(c->d_rblocks, img.rblocks.data(), img.rblocks.size() * sizeof(uint32_t)));
	}
	{
		const int rct = tab_upload(c);                          /* no-op in the default build */
		if (rct) return rct;
	}
	if (c->d_filter) (void) hipFree(c->d_filter);
	if (c->d_keys) (void) hipFree(c->d_keys);
	if (c->d_slot_of) (void) hipFree(c->d_slot_of);
	c->d_filter = nullptr; c->d_keys = nullptr; c->d_slot_of = nullptr;
	HIPCHK(dev_malloc(&c->d_filter, img.filter.size() * sizeof(uint32_t)));
	HIPCHK(dev_malloc(&c->d_keys, 2 * c->n_slots * sizeof(uint64_t), c->keys_mem_kind));   /* { key0, key1, count0, count1 } per bucket */
	HIPCHK(dev_malloc(&c->d_slot_of, (n ? n : 1) * sizeof(uint32_t)));
	HIPCHK(h2d(c->d_filter, img.filter.data(), img.filter.size() * sizeof(uint32_t)));
	if (n) HIPCHK(h2d(c->d_slot_of, c->slot_of.data(), n * sizeof(uint32_t)));
	{
		/* The bucket image { key0, key1, count0, count1 } is laid out on the device: every bucket starts empty with zeroed
		 * counters, then the n keys are scattered to their slots -- 12 bytes per key cross PCIe instead of 32 bytes per
		 * bucket of a table that is 2/3 empty, and the host never builds the image (64 MiB for the human set, 512 MiB for
		 * 16 M keys). */
		HIPCHK(launch_table_init(c->d_keys, (unsigned long long) (c->n_slots / 2), 0));
		if (n) {
			uint64_t *d_canon = nullptr;
			HIPCHK(dev_malloc(&d_canon, (uint64_t) n * sizeof(uint64_t)));
			hipError_t e1 = h2d(d_canon, c->canon.data(), (uint64_t) n * sizeof(uint64_t));
			if (e1 == hipSuccess) {
				e1 = launch_table_scatter(c->d_keys, c->d_slot_of, d_canon, n, 0);
			}
			if (e1 == hipSuccess) e1 = hipDeviceSynchronize();
			(void) hipFree(d_canon);
			HIPCHK(e1);
		}
	}
	HIPCHK(hipDeviceSynchronize());                      /* tables and zeroed counters visible before any stream uses them */
	return NTSM_OK;
}

/* device memory of a lane slot: from the context's cache of closed lanes' buffers when one of that size is there */
hipError_t device_take(ntsm_ctx *c, void **p, uint64_t bytes)
{
	if (c) {
		std::lock_guard<std::mutex> lk(c->mu);
		for (size_t i = c->device_cache.size(); i-- > 0;)
			if (c->device_cache[i].second == bytes) {
				*p = c->device_cache[i].first;
				c->device_cache.erase(c->device_cache.begin() + (long) i);
				return hipSuccess;
			}
	}
	return dev_malloc(p, bytes);
}

void device_give(ntsm_ctx *c, void *p, uint64_t bytes)
{
	if (!p) return;
	void *evict = nullptr;
	if (c && bytes) {
		std::lock_guard<std::mutex> lk(c->mu);
		if (c->device_cache.size() >= ntsm_ctx::kDeviceCacheMax) { evict = c->device_cache.front().first; c->device_cache.erase(c->device_cache.begin()); }
		c->device_cache.emplace_back(p, bytes);
		p = nullptr;
	}
	if (evict) (void) hipFree(evict);
	if (p) (void) hipFree(p);
}

int slot_add_host_bases(Slot &s)
{
	if (s.h_bases) return NTSM_OK;
	s.h_bases = (uint8_t *) pool_alloc(s.h_bases_bytes);
	if (!s.h_bases) HIPCHK(pinned_malloc((void **) &s.h_bases, s.h_bases_bytes));
	return NTSM_OK;
}

int alloc_slot(Slot &s, int device, uint64_t cap_bytes, uint64_t cap_reads, bool ends_on_device, bool packed_only, ntsm_ctx *cache, bool host_bases)
{
	s.ends_on_device = ends_on_device;
	s.h_bases_bytes = (packed_only ? (cap_bytes & ~31ull) / 4 + (cap_bytes & ~31ull) / 8 : cap_bytes) + 64;   /* packed: 3/8 byte per position */
	s.h_ends_bytes = cap_reads * sizeof(uint64_t);
	if (host_bases) {                                       /* ntsm_submit_pinned reads the caller's own pinned memory: no staging for the bases */
		const int rcb = slot_add_host_bases(s);
		if (rcb) return rcb;
	}
	if (ends_on_device) {
		s.h_read_end = (uint64_t *) pool_alloc(s.h_ends_bytes);
		if (!s.h_read_end) HIPCHK(pinned_malloc((void **) &s.h_read_end, s.h_ends_bytes));
		HIPCHK(dev_malloc(&s.d_read_end, s.h_ends_bytes));
	} else {
		s.h_read_end = (uint64_t *) malloc(s.h_ends_bytes);
		if (!s.h_read_end) return NTSM_ERR_NOMEM;
	}
	s.d_bases_bytes = cap_bytes + 64;
	HIPCHK(device_take(cache, (void **) &s.d_bases, s.d_bases_bytes));
	if (!s.stream) {
		s.stream = stream_get(device);
		if (!s.stream) return NTSM_ERR_HIP;
	}
	/* (events with hipEventBlockingSync -- waiting threads sleep instead of spinning -- were measured on the 16-CPU pod: no gain for
	 * 16 packed lanes, 68.5 vs 67.3 Gbases/s, and ntsm_submit 25 % slower, 39.9 vs 53.4 GB/s: NOTEBOOK.md round 6) */
	if (!s.done) HIPCHK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
	if (!s.copied) HIPCHK(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
	s.busy = false;
	s.acquired = false;
	return NTSM_OK;
}

void free_slot(Slot &s, ntsm_ctx *cache)
{
	if (s.h_bases && !pool_free(s.h_bases, s.h_bases_bytes)) (void) hipHostFree(s.h_bases);
	if (s.h_read_end) {
		if (!s.ends_on_device) free(s.h_read_end);
		else if (!pool_free(s.h_read_end, s.h_ends_bytes)) (void) hipHostFree(s.h_read_end);
	}
	device_give(cache, s.d_bases, s.d_bases_bytes);
	device_give(cache, s.d_packed, s.d_packed_bytes);
	if (s.d_read_end) (void) hipFree(s.d_read_end);
	s.h_bases = s.d_bases = s.d_packed = nullptr;
	s.h_read_end = s.d_read_end = nullptr;
}

/* launch one count pass over stream bytes [lo, hi) of d_bases */
int launch_count(ntsm_ctx *c, hipStream_t st, const uint8_t *d_bases, uint64_t lo, uint64_t hi,
		const uint64_t *d_read_end, uint64_t n_reads, bool per_read, int sign)
{
	if (hi <= lo) return NTSM_OK;
	NtsmCountParams p;
	memset(&p, 0, sizeof p);
	p.base = d_bases;
	p.lo = (long long) lo;
	p.hi = (long long) hi;
	p.t0 = (long long) (lo & ~15ull);
	const uint64_t tile = 256ull * kTileC;                     /* 256-thread workgroups */
	p.n_tiles = (hi - (uint64_t) p.t0 + tile - 1) / tile;
	p.filter = c->d_filter;
	p.keys = c->d_keys;
	p.totals = c->d_totals;
	p.read_end = (const unsigned long long *) d_read_end;
	p.read_hits = c->d_read_hits;
	p.n_reads = n_reads;
	p.sign = sign >= 0 ? 1ull : ~0ull;
	p.mask = c->mask;
	p.k = (uint32_t) c->k;
	p.rv_shift = (uint32_t) (2 * (c->k - 1));
	p.kmask = c->k >= 32 ? 0xFFFFFFFFu : ((1u << c->k) - 1);
	p.fshift = 32 - c->filter_log2;
	p.bshift = 32 - c->bucket_log2;
	p.lut = c->d_lut;
	p.lut64 = c->d_lut64;
	p.blocks = c->d_blocks;
	p.blk_map = c->blk_map;
	p.prefilter = c->d_prefilter;
	NTSM_ABL_LAUNCH_PARAMS(p)
	p.pf_shift = 32 - (c->prefilter_log2 - 5);
	p.blk_bytes = (uint32_t) (c->n_blocks * 16);
	const NtsmFastPlan plan = ntsm_fast_plan((uint32_t) c->k, c->two_level);
	const bool fast = plan.mode >= 0 && c->d_blocks && c->kernel_variant != 1;
	p.bloom = c->d_bloom;
	p.bloom_words = c->n_bloom_words;
	p.fk_k = plan.k; p.fk_m2 = 2 * plan.m; p.fk_a2 = 2 * plan.a;
	const bool tab = fast && !per_read && tab_applies(c);       /* always false in the default build */
	const bool runk = fast && !per_read && !tab && c->run_form && c->d_rblocks;   /* run-anchored kernel (k = 19; armed batches keep the per-read kernels) */
	if (runk) {
		p.blocks = c->d_rblocks;
		p.blk_map.n_blocks = (uint32_t) c->n_rblocks;
		p.blk_bytes = (uint32_t) (c->n_rblocks * 16);
	}
	if (fast && !tab) {                                     /* the minimizer-blocked kernels cut the stream into their own tiles */
		const uint64_t ftile = (uint64_t) (runk ? run_tile_bytes() : mz_tile_bytes());
		p.n_tiles = (hi - (uint64_t) p.t0 + ftile - 1) / ftile;
	}
	/* Grid: many more workgroups than fit on the chip at once (4 per CU), each walking ~8+ tiles.  A grid of
	 * exactly the resident workgroups (static tile assignment) measured 11 % slower: the slowest CU sets the
	 * finish time; with 32k-128k workgroups the dispatcher balances the load (measured plateau), while fewer
	 * than ~4 tiles per workgroup pays the per-workgroup setup too often. */
	uint64_t grid = c->grid_blocks > 0 ? (uint64_t) c->grid_blocks : std::min<uint64_t>(65536, std::max<uint64_t>((uint64_t) c->n_cu * 4, p.n_tiles / 8));
	if (grid > p.n_tiles) grid = p.n_tiles;
	int ev = -1;
	/* The event pool is shared by all lanes; and the tabulated path is three enqueues on one stream (reset of the tile
	 * list, kernel, list walker) that must not interleave with another lane's three on the same stream. */
	std::unique_lock<std::mutex> timing_lock(c->mu, std::defer_lock);
	if (c->timing || tab) timing_lock.lock();
	if (c->timing) {
		ev = c->ev_next;
		c->ev_next = (c->ev_next + 1) % kTimingPool;
		if (c->ev_used[ev]) {                             /* recycle: fold the old measurement in */
			float ms = 0;
			HIPCHK(hipEventSynchronize(c->ev_b[ev]));
			HIPCHK(hipEventElapsedTime(&ms, c->ev_a[ev], c->ev_b[ev]));
			c->t_ms += ms;
			c->ev_used[ev] = false;
		}
		HIPCHK(hipEventRecord(c->ev_a[ev], st));
	}
	if (tab) {
		const int rct = tab_launch(c, st, p, hi);               /* queue / list buffers, segments, look-up kernels, the list walker */
		if (rct) return rct;
		c->n_launch[0]++;
	} else
	if (runk) {
		HIPCHK(launch_run(p, (unsigned) grid, st));
		c->n_launch[1]++;
	} else if (fast) {
		const hipError_t le = launch_mz(p, (unsigned) grid, st, plan.mode, per_read, c->two_level);
		if (le == hipErrorInvalidValue) return NTSM_ERR_STATE;  /* no kernel for this plan */
		HIPCHK(le);
		c->n_launch[1]++;
	}
	else {
		HIPCHK(launch_generic(p, (unsigned) grid, st, per_read));
		c->n_launch[2]++;
	}
	if (ev >= 0) {
		HIPCHK(hipEventRecord(c->ev_b[ev], st));
		c->ev_used[ev] = true;
		c->t_launches++;
	}
	return NTSM_OK;
}

int read_device_totals(ntsm_ctx *c, uint64_t out[2])
{
	HIPCHK(hipMemcpy(out, c->d_totals, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
	return NTSM_OK;
}

/* Early-stop ("-m") batch: count with per-read attribution, then, if the running hit total
 * crossed max_hits inside this batch, find the first read r* after which total_hits > max_hits
 * (src/FingerPrint.hpp:476-487: checked after each whole read, strict '>') and take the reads
 * after r* out again with a sign = -1 pass.  d_read_end must be on the device. */
int armed_batch(ntsm_ctx *c, hipStream_t st, const uint8_t *d_bases, uint64_t n_bytes,
		const uint64_t *d_read_end, const uint64_t *h_read_end_or_null, uint64_t n_reads)
{
	/* The batch is walked in chunks of reads -- about 256 MB of stream each, at most 2^20 reads -- so that the work done is
	 * proportional to what is consumed before the stop, not to the size of the batch. */
	const uint64_t avg_len = std::max<uint64_t>(1, n_bytes / std::max<uint64_t>(1, n_reads));
	const uint64_t chunk_bytes = c->armed_chunk_bytes;      /* 256 MiB unless ntsm_set_armed_chunk changed it */
	const uint64_t CH = std::min<uint64_t>(1ull << 20, std::max<uint64_t>(1024, chunk_bytes / avg_len));
	const uint64_t n_chunks = (n_reads + CH - 1) / CH;
	std::vector<uint64_t> bend(n_chunks);                 /* offset of the last terminator of every chunk */
	if (h_read_end_or_null) {
		for (uint64_t k = 0; k < n_chunks; ++k) bend[k] = h_read_end_or_null[std::min(n_reads, (k + 1) * CH) - 1];
	} else {
		if (n_chunks > 1)                                  /* one 8-byte element per CH reads: strided copy */
			HIPCHK(hipMemcpy2D(bend.data(), sizeof(uint64_t), d_read_end + (CH - 1), CH * sizeof(uint64_t),
					sizeof(uint64_t), n_chunks - 1, hipMemcpyDeviceToHost));
		HIPCHK(hipMemcpy(&bend[n_chunks - 1], d_read_end + (n_reads - 1), sizeof(uint64_t), hipMemcpyDeviceToHost));
	}
	const uint64_t hits_cap = std::min(n_reads, CH);
	if (c->d_read_hits && c->read_hits_cap < hits_cap) { (void) hipFree(c->d_read_hits); c->d_read_hits = nullptr; }
	if (!c->d_read_hits) {
		HIPCHK(dev_malloc(&c->d_read_hits, hits_cap * sizeof(uint32_t)));
		c->read_hits_cap = hits_cap;
	}
	HIPCHK(hipStreamSynchronize(st));
	uint64_t run[2];
	int rc = read_device_totals(c, run);
	if (rc) return rc;
	/* Optimistic spans.  The per-read kernel (hits attributed to reads, 3 waves per SIMD) is only needed in the one chunk
	 * where the threshold is crossed.  Everything before it is counted by the plain kernel in spans of whole chunks, sized
	 * from the hit rate seen so far to use about half of the remaining budget; a span that crosses after all is taken out
	 * again (sign -1, exact) and walked chunk by chunk, and the crossing chunk is taken out and counted per read. */
	double rate = -1.0;                                    /* hits per read in the spans accepted so far */
	bool single = false;                                   /* a span crossed: one chunk at a time from here on */
	for (uint64_t k = 0; k < n_chunks;) {
		uint64_t span = 1;
		if (!single && rate >= 0) {
			const double budget = (double) (c->max_hits - run[1]);
			const double reads_ok = rate > 0 ? budget / (2.0 * rate) : 1e18;
			span = reads_ok >= (double) (64 * CH) ? 64 : std::max<uint64_t>(1, (uint64_t) (reads_ok / (double) CH));
			span = std::min(span, n_chunks - k);
		}
		const uint64_t r0 = k * CH, r1 = std::min(n_reads, (k + span) * CH), nr = r1 - r0;
		const uint64_t lo = k ? bend[k - 1] + 1 : 0, hi = bend[k + span - 1] + 1;
		rc = launch_count(c, st, d_bases, lo, hi, nullptr, 0, false, +1);
		if (rc) return rc;
		HIPCHK(hipStreamSynchronize(st));
		uint64_t after[2];
		rc = read_device_totals(c, after);
		if (rc) return rc;
		if (after[1] <= c->max_hits) {                    /* no crossing in this span */
			c->total_bases += (hi - lo) - nr;
			c->reads_consumed += nr;
			rate = (double) (after[1] - run[1]) / (double) nr;
			run[1] = after[1];
			k += span;
			continue;
		}
		rc = launch_count(c, st, d_bases, lo, hi, nullptr, 0, false, -1);   /* take the span out again */
		if (rc) return rc;
		if (span > 1) { single = true; continue; }
		/* the crossing chunk, per read */
		HIPCHK(hipMemsetAsync(c->d_read_hits, 0, nr * sizeof(uint32_t), st));
		rc = launch_count(c, st, d_bases, lo, hi, d_read_end + r0, nr, true, +1);
		if (rc) return rc;
		HIPCHK(hipStreamSynchronize(st));
		/* first read r* (strict '>') after which the cumulative hit count exceeds max_hits */
		std::vector<uint32_t> hits(nr);
		HIPCHK(hipMemcpy(hits.data(), c->d_read_hits, nr * sizeof(uint32_t), hipMemcpyDeviceToHost));
		std::vector<uint64_t> re_local;
		const uint64_t *re = h_read_end_or_null ? h_read_end_or_null + r0 : nullptr;
		if (!re) {
			re_local.resize(nr);
			HIPCHK(hipMemcpy(re_local.data(), d_read_end + r0, nr * sizeof(uint64_t), hipMemcpyDeviceToHost));
			re = re_local.data();
		}
		uint64_t acc = run[1], rstar = nr - 1;
		for (uint64_t r = 0; r < nr; ++r) {
			acc += hits[r];
			if (acc > c->max_hits) { rstar = r; break; }
		}
		rc = launch_count(c, st, d_bases, re[rstar] + 1, hi, nullptr, 0, false, -1);   /* take the reads after r* out again */
		if (rc) return rc;
		HIPCHK(hipStreamSynchronize(st));
		c->total_bases += (re[rstar] + 1 - lo) - (rstar + 1);
		c->reads_consumed += rstar + 1;
		c->early_stop = true;
		break;
	}
	return NTSM_OK;
}

/* ---- ntsm_submit's staging copy on several threads ----------------------------------------------------------------------------
 * Pieces are cut at multiples of 4 KiB; the helpers sleep on a condition variable between batches (a batch is milliseconds of
 * work: the wake-up is noise).  One submitting thread per context (include/ntsm_hip.h), so one job at a time. */
static void copy_helper(CopyPool *cp, unsigned idx)
{
	uint64_t seen = 0;
	for (;;) {
		uint8_t *dst; const uint8_t *src; uint64_t n; unsigned parts;
		{
			std::unique_lock<std::mutex> lk(cp->mu);
			cp->cv_work.wait(lk, [&] { return cp->quit || cp->generation != seen; });
			if (cp->quit) return;
			seen = cp->generation;
			dst = cp->dst; src = cp->src; n = cp->n; parts = cp->parts;
		}
		if (idx < parts) {
			const uint64_t lo = (n * idx / parts) & ~4095ull, hi = idx + 1 == parts ? n : (n * (idx + 1) / parts) & ~4095ull;
			if (hi > lo) memcpy(dst + lo, src + lo, hi - lo);
			std::lock_guard<std::mutex> lk(cp->mu);
			if (--cp->pending == 0) cp->cv_done.notify_one();
		}
	}
}

static int auto_submit_threads()
{
	cpu_set_t set;
	int cpus = 1;
	if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = CPU_COUNT(&set);
	return std::max(1, std::min(6, cpus));                  /* 2 / 3 / 4 / 6 / 8 threads: 77 / 91 / 86 / 94 / 93 % of the pinned-copy ceiling (tools/feed_bench.cpp) */
}

void staged_copy(ntsm_ctx *c, uint8_t *dst, const uint8_t *src, uint64_t n)
{
	const int want = c->submit_threads > 0 ? c->submit_threads : auto_submit_threads();
	unsigned parts = (unsigned) std::max<uint64_t>(1, std::min<uint64_t>((uint64_t) want, n >> 22));   /* at least 4 MiB per thread */
	if (parts <= 1) { memcpy(dst, src, n); return; }
	if (!c->copy_pool) c->copy_pool = new (std::nothrow) CopyPool();
	CopyPool *cp = c->copy_pool;
	if (!cp) { memcpy(dst, src, n); return; }
	while (cp->helpers.size() + 1 < parts) {
		try { cp->helpers.emplace_back(copy_helper, cp, (unsigned) cp->helpers.size() + 1); }
		catch (...) { break; }                              /* no more threads to be had: fewer pieces */
	}
	parts = std::min<unsigned>(parts, (unsigned) cp->helpers.size() + 1);
	if (parts <= 1) { memcpy(dst, src, n); return; }
	{
		std::lock_guard<std::mutex> lk(cp->mu);
		cp->dst = dst; cp->src = src; cp->n = n; cp->parts = parts;
		cp->pending = parts - 1;
		cp->generation++;
	}
	cp->cv_work.notify_all();
	const uint64_t hi0 = (n / parts) & ~4095ull;
	if (hi0) memcpy(dst, src, hi0);                          /* piece 0 on the submitting thread */
	std::unique_lock<std::mutex> lk(cp->mu);
	cp->cv_done.wait(lk, [&] { return cp->pending == 0; });
}

void copy_pool_release(ntsm_ctx *c)
{
	CopyPool *cp = c->copy_pool;
	if (!cp) return;
	{
		std::lock_guard<std::mutex> lk(cp->mu);
		cp->quit = true;
	}
	cp->cv_work.notify_all();
	for (auto &t : cp->helpers) t.join();
	delete cp;
	c->copy_pool = nullptr;
}

int check_layout(const uint64_t *read_end, uint32_t n_reads, uint64_t n_bytes)
{
	if (n_reads == 0) return n_bytes == 0 ? NTSM_OK : NTSM_ERR_ARG;
	if (!read_end || read_end[n_reads - 1] + 1 != n_bytes) return NTSM_ERR_ARG;
	return NTSM_OK;
}

hipError_t slot_copy(ntsm_ctx *c, Slot &s, void *const *dst, const void *const *src, const size_t *bytes, int n_copies)
{
	for (int i = 0; i < n_copies; ++i) {
		const hipError_t e = h2d_async(dst[i], src[i], bytes[i], c->cstream);
		if (e != hipSuccess) return e;
	}
	hipError_t e = hipEventRecord(s.copied, c->cstream);     /* lanes enqueue concurrently: the event may also cover a neighbour's copy, never less than ours */
	if (e != hipSuccess) return e;
	return hipStreamWaitEvent(s.stream, s.copied, 0);
}

int submit_slot(ntsm_ctx *c, Slot &s, uint64_t n_bytes, uint32_t n_reads)
{
	if (n_reads == 0) return NTSM_OK;
	if (c->armed) {                                          /* synchronous anyway: everything on the slot's stream */
		HIPCHK(h2d_async(s.d_bases, s.h_bases, n_bytes, s.stream));
		HIPCHK(h2d_async(s.d_read_end, s.h_read_end, n_reads * sizeof(uint64_t), s.stream));
		return armed_batch(c, s.stream, s.d_bases, n_bytes, s.d_read_end, s.h_read_end, n_reads);
	}
	{
		void *const dst[1] = { s.d_bases };
		const void *const src[1] = { s.h_bases };
		const size_t bytes[1] = { (size_t) n_bytes };
		HIPCHK(slot_copy(c, s, dst, src, bytes, 1));
	}
	int rc = launch_count(c, s.stream, s.d_bases, 0, n_bytes, nullptr, 0, false, +1);
	if (rc) return rc;
	HIPCHK(hipEventRecord(s.done, s.stream));
	s.busy = true;
	c->total_bases += n_bytes - n_reads;
	c->reads_consumed += n_reads;
	return NTSM_OK;
}

int wait_slot(Slot &s)
{
	if (s.busy) {
		HIPCHK(hipEventSynchronize(s.done));
		s.busy = false;
	}
	return NTSM_OK;
}

} // namespace ntsm_rt
