/*
 * capi.cpp -- the C ABI of include/ntsm_hip.h: argument checks, context state and the calls into runtime.cpp.  Each entry
 * point cites the reference interface it replaces in the header; the boundary sits where FingerPrint::insertCount is called
 * per read (src/FingerPrint.hpp:89, :475), batched.  No device code, no CPU fallback: ntsm_create fails with
 * NTSM_ERR_NO_DEVICE without a GPU.
 */
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "ntsm_internal.h"

using namespace ntsm_rt;

namespace {

uint64_t inv_odd(uint64_t a)                 /* multiplicative inverse mod 2^64 */
{
	uint64_t x = a;
	for (int i = 0; i < 6; ++i) x *= 2 - a * x;
	return x;
}

uint64_t unxorshift(uint64_t y, int s)
{
	for (int sh = s; sh < 64; sh *= 2) y ^= y >> sh;
	return y;
}

} // namespace

extern "C" {

uint64_t ntsm_hash64(uint64_t key, int k)
{
	const uint64_t mask = mask_for_k(k);
	key = (~key + (key << 21)) & mask;
	key ^= key >> 24;
	key = (key + (key << 3) + (key << 8)) & mask;
	key ^= key >> 14;
	key = (key + (key << 2) + (key << 4)) & mask;
	key ^= key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

uint64_t ntsm_hash64_inv(uint64_t hv, int k)
{
	const uint64_t mask = mask_for_k(k);
	uint64_t x = hv & mask;
	x = (x * inv_odd((1ull << 31) + 1)) & mask;
	x = unxorshift(x, 28);
	x = (x * inv_odd(21)) & mask;
	x = unxorshift(x, 14);
	x = (x * inv_odd(265)) & mask;
	x = unxorshift(x, 24);
	x = ((x + 1) * inv_odd((1ull << 21) - 1)) & mask;
	return x;
}

const char *ntsm_strerror(int code)
{
	switch (code) {
	case NTSM_OK: return "ok";
	case NTSM_ERR_ARG: return "invalid argument";
	case NTSM_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
	case NTSM_ERR_HIP: return "HIP runtime error";
	case NTSM_ERR_DUP_KEY: return "duplicate k-mer key";
	case NTSM_ERR_NOMEM: return "out of memory";
	case NTSM_ERR_STATE: return "invalid state for this call";
	case NTSM_ERR_RCCL: return "RCCL error";
	default: return "unknown error";
	}
}

int ntsm_last_hip_error(void) { return last_hip(); }
const char *ntsm_version(void) { return "ntsm_hip 0.1 (gfx950)"; }

int ntsm_create(ntsm_ctx **out, int device, int k, const uint64_t *keys, uint32_t n_kmers,
		int key_kind, uint64_t max_hits)
{
	if (!out || k < 1 || k > 32 || (n_kmers && !keys)) return NTSM_ERR_ARG;
	if (key_kind != NTSM_KEYS_CANONICAL && key_kind != NTSM_KEYS_HASH64) return NTSM_ERR_ARG;
	*out = nullptr;
	if (device < 0) return NTSM_ERR_NO_DEVICE;
	ntsm_ctx *c = new (std::nothrow) ntsm_ctx();
	if (!c) return NTSM_ERR_NOMEM;
	c->device = device;
	c->k = k;
	c->n_kmers = n_kmers;
	c->max_hits = max_hits;
	c->armed = max_hits != 0;
	c->mask = mask_for_k(k);
	memset(c->ev_used, 0, sizeof c->ev_used);
	c->canon.resize(n_kmers);
	for (uint32_t i = 0; i < n_kmers; ++i) {
		uint64_t x = key_kind == NTSM_KEYS_HASH64 ? ntsm_hash64_inv(keys[i], k) : keys[i];
		if (x & ~c->mask && k < 32) { delete c; return NTSM_ERR_ARG; }
		c->canon[i] = x;
	}
	/* The tables are built on the host BEFORE the first HIP call of this function: a caller that warms the runtime up on a
	 * side thread (ntsm_warmup) gets the 50-60 ms of table construction for free while the runtime initialises. */
	int rc = build_tables(c, 0, [](ntsm_ctx *cc) -> int {
		int n_dev = 0;
		hipError_t e = hipGetDeviceCount(&n_dev);
		if (e != hipSuccess || n_dev <= 0 || cc->device >= n_dev) {
			set_last_hip((int) e);
			return NTSM_ERR_NO_DEVICE;
		}
		HIPCHK(hipSetDevice(cc->device));
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, cc->device) == hipSuccess && prop.multiProcessorCount > 0)
			cc->n_cu = prop.multiProcessorCount;
		return NTSM_OK;
	});
	if (rc == NTSM_ERR_NO_DEVICE || rc == NTSM_ERR_DUP_KEY || rc == NTSM_ERR_ARG) { delete c; return rc; }   /* nothing on the device yet */
	if (rc) { ntsm_destroy(c); return rc; }
	uint8_t lut[256];
	build_lut(lut);
	auto fail = [&](int code) { ntsm_destroy(c); return code; };
	if (dev_malloc(&c->d_lut, 256) != hipSuccess) return fail(NTSM_ERR_HIP);
	if (h2d(c->d_lut, lut, 256) != hipSuccess) return fail(NTSM_ERR_HIP);
	{
		uint2 lut64[256];
		for (int i = 0; i < 256; ++i)
			lut64[i] = lut[i] < 4 ? make_uint2((uint32_t) lut[i], (3u - lut[i]) | 0x10000u) : make_uint2(0u, 3u);   /* { code, complement | valid << 16 } */
		if (dev_malloc(&c->d_lut64, sizeof lut64) != hipSuccess) return fail(NTSM_ERR_HIP);
		if (h2d(c->d_lut64, lut64, sizeof lut64) != hipSuccess) return fail(NTSM_ERR_HIP);
	}
	if (dev_malloc(&c->d_totals, 4 * sizeof(uint64_t)) != hipSuccess) return fail(NTSM_ERR_HIP);
	if (hipMemset(c->d_totals, 0, 4 * sizeof(uint64_t)) != hipSuccess) return fail(NTSM_ERR_HIP);
	if (hipDeviceSynchronize() != hipSuccess) return fail(NTSM_ERR_HIP);
	if (dev_malloc(&c->d_vec, ((uint64_t) n_kmers + 4) * sizeof(uint64_t)) != hipSuccess) return fail(NTSM_ERR_HIP);
	if (!(c->rstream = stream_get(device))) return fail(NTSM_ERR_HIP);
	if (!(c->cstream = stream_get(device))) return fail(NTSM_ERR_HIP);
	for (int i = 0; i < kTimingPool; ++i) {
		if (hipEventCreate(&c->ev_a[i]) != hipSuccess || hipEventCreate(&c->ev_b[i]) != hipSuccess) return fail(NTSM_ERR_HIP);
	}
	*out = c;
	return NTSM_OK;
}

void ntsm_destroy(ntsm_ctx *c)
{
	if (!c) return;
	(void) hipSetDevice(c->device);
	(void) hipDeviceSynchronize();
	for (auto &s : c->slot) {
		free_slot(s);
		stream_put(c->device, s.stream);
		if (s.done) (void) hipEventDestroy(s.done);
		if (s.copied) (void) hipEventDestroy(s.copied);
	}
	stream_put(c->device, c->rstream);
	stream_put(c->device, c->cstream);
	for (hipStream_t st : c->lane_stream) stream_put(c->device, st);
	for (int i = 0; i < kTimingPool; ++i) {
		if (c->ev_a[i]) (void) hipEventDestroy(c->ev_a[i]);
		if (c->ev_b[i]) (void) hipEventDestroy(c->ev_b[i]);
	}
	tab_release(c);                                        /* no-op in the default build */
	copy_pool_release(c);
	for (auto &b : c->device_cache) (void) hipFree(b.first);
	c->device_cache.clear();
	void *ptrs[] = { c->d_rblocks, c->d_bloom, c->d_prefilter, c->d_lut64, c->d_blocks, c->d_filter, c->d_keys, c->d_slot_of, c->d_read_hits, c->d_totals, c->d_vec, c->d_lut };
	for (void *p : ptrs) if (p) (void) hipFree(p);
	delete c;
}

int ntsm_set_batch_capacity(ntsm_ctx *c, uint64_t cap_bytes, uint64_t cap_reads)
{
	if (!c || cap_bytes < 4096 || cap_reads < 16) return NTSM_ERR_ARG;
	if (c->failed) return NTSM_ERR_STATE;
	HIPCHK(hipSetDevice(c->device));
	for (auto &s : c->slot) {
		int rc = wait_slot(s);
		if (rc) return rc;
		free_slot(s);
	}
	c->cap_bytes = cap_bytes;
	c->cap_reads = cap_reads;
	return NTSM_OK;
}

int ntsm_staging_acquire(ntsm_ctx *c, uint8_t **bases, uint64_t *cap_bytes, uint64_t **read_end, uint64_t *cap_reads)
{
	if (!c || !bases || !cap_bytes || !read_end || !cap_reads) return NTSM_ERR_ARG;
	if (c->failed) return NTSM_ERR_STATE;
	HIPCHK(hipSetDevice(c->device));
	Slot &s = c->slot[c->next_slot];
	if (!s.d_bases) {
		int rc = alloc_slot(s, c->device, c->cap_bytes, c->cap_reads, true);
		if (rc) return rc;
	} else if (!s.h_bases) {                               /* the slot was made by ntsm_submit_pinned, without staging for the bases */
		int rc = slot_add_host_bases(s);
		if (rc) return rc;
	}
	int rc = wait_slot(s);
	if (rc) return rc;
	s.acquired = true;
	*bases = s.h_bases;
	*cap_bytes = c->cap_bytes;
	*read_end = s.h_read_end;
	*cap_reads = c->cap_reads;
	return NTSM_OK;
}

int ntsm_submit_staged(ntsm_ctx *c, uint64_t n_bytes, uint32_t n_reads)
{
	if (!c) return NTSM_ERR_ARG;
	if (c->failed) return NTSM_ERR_STATE;
	Slot &s = c->slot[c->next_slot];
	if (!s.acquired) return NTSM_ERR_STATE;
	s.acquired = false;
	if (n_bytes > c->cap_bytes || n_reads > c->cap_reads) return NTSM_ERR_ARG;
	int rc = check_layout(s.h_read_end, n_reads, n_bytes);
	if (rc) return rc;
	if (c->early_stop) return NTSM_OK;                    /* threshold already tripped: nothing more is counted */
	c->reduced = false;
	HIPCHK(hipSetDevice(c->device));
	rc = submit_slot(c, s, n_bytes, n_reads);
	c->next_slot ^= 1;
	return rc;
}

int ntsm_warmup(int device, int n_streams)
{
	int n_dev = 0;
	hipError_t e = hipGetDeviceCount(&n_dev);
	if (e != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev) {
		set_last_hip((int) e);
		return NTSM_ERR_NO_DEVICE;
	}
	HIPCHK(hipSetDevice(device));
	HIPCHK(hipFree(nullptr));                             /* forces runtime + device context initialisation */
	std::vector<hipStream_t> made;
	for (int i = 0; i < n_streams && i < 1024; ++i) {
		hipStream_t s = nullptr;
		HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
		made.push_back(s);
	}
	for (hipStream_t s : made) stream_put(device, s);
	return NTSM_OK;
}

int ntsm_staging_pool(uint64_t bytes) { return staging_pool(bytes); }

static int lane_open(ntsm_ctx *c, uint64_t cap_bytes, uint64_t cap_reads, bool packed_only, ntsm_lane **out);

int ntsm_lane_open(ntsm_ctx *c, uint64_t cap_bytes, uint64_t cap_reads, ntsm_lane **out) { return lane_open(c, cap_bytes, cap_reads, false, out); }
int ntsm_lane_open_packed(ntsm_ctx *c, uint64_t cap_positions, ntsm_lane **out) { return lane_open(c, cap_positions, 16, true, out); }

static int lane_open(ntsm_ctx *c, uint64_t cap_bytes, uint64_t cap_reads, bool packed_only, ntsm_lane **out)
{
	if (!c || !out) return NTSM_ERR_ARG;
	*out = nullptr;
	if (c->armed || c->failed) return NTSM_ERR_STATE;     /* -m is defined on ONE ordered stream of reads */
	if (cap_bytes == 0) cap_bytes = c->cap_bytes;
	if (cap_reads == 0) cap_reads = cap_bytes / 64 + 16;
	if (cap_bytes < 4096 || cap_reads < 16) return NTSM_ERR_ARG;
	HIPCHK(hipSetDevice(c->device));
	ntsm_lane *l = new (std::nothrow) ntsm_lane();
	if (!l) return NTSM_ERR_NOMEM;
	l->c = c;
	l->cap_bytes = cap_bytes;
	l->cap_reads = cap_reads;
	l->packed_only = packed_only;
	/* Lanes do not own streams (creating one costs 14 ms): all lanes of a context share its two lane streams, round
	 * robin.  Copies and kernels of different lanes interleave there in submission order; a lane only waits on the
	 * events of its own slots. */
	hipStream_t st = nullptr;
	{
		std::lock_guard<std::mutex> lk(c->mu);
		hipStream_t &slot_stream = c->lane_stream[c->lanes_opened++ & 1u];
		if (!slot_stream) slot_stream = stream_get(c->device);
		st = slot_stream;
	}
	if (!st) { delete l; return NTSM_ERR_HIP; }
	for (int i = 0; i < 2; ++i) {
		Slot &s = l->slot[i];
		s.stream = st;
		int rc = alloc_slot(s, c->device, cap_bytes, cap_reads, false, packed_only, c);
		if (rc) {
			for (auto &q : l->slot) {
				free_slot(q, c);
				if (q.done) (void) hipEventDestroy(q.done);
			}
			delete l;
			return rc;
		}
	}
	{
		std::lock_guard<std::mutex> lk(c->mu);
		c->open_lanes++;
		c->reduced = false;
	}
	*out = l;
	return NTSM_OK;
}

int ntsm_lane_acquire(ntsm_lane *l, uint8_t **bases, uint64_t *cap_bytes, uint64_t **read_end, uint64_t *cap_reads)
{
	if (!l || !bases || !cap_bytes || !read_end || !cap_reads) return NTSM_ERR_ARG;
	if (l->packed_only) return NTSM_ERR_STATE;            /* its pinned slots hold packed batches only */
	HIPCHK(hipSetDevice(l->c->device));
	Slot &s = l->slot[l->next_slot];
	int rc = wait_slot(s);
	if (rc) return rc;
	s.acquired = true;
	*bases = s.h_bases;
	*cap_bytes = l->cap_bytes;
	*read_end = s.h_read_end;
	*cap_reads = l->cap_reads;
	return NTSM_OK;
}

/* A batch of a lane that cannot be copied or launched is LOST: the context's counts no longer cover what its callers
 * submitted.  The lane remembers the first such error (every later submit and ntsm_lane_close report it again) and the context
 * is marked failed, so that no ntsm_sync / ntsm_counts can hand out the incomplete result (the reference's contract for an
 * input it cannot process is exit(1) with a message, src/FingerPrint.hpp:51-57). */
static int lane_lost_batch(ntsm_lane *l, int rc)
{
	if (l->error == NTSM_OK) l->error = rc;
	std::lock_guard<std::mutex> lk(l->c->mu);
	l->c->failed = true;
	return rc;
}

#define LANECHK(call)                                                                     \
	do {                                                                                  \
		hipError_t e_ = (call);                                                           \
		if (e_ != hipSuccess) {                                                           \
			set_last_hip((int) e_);                                                       \
			fprintf(stderr, "ntsm_hip: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
			return lane_lost_batch(l, NTSM_ERR_HIP);                                      \
		}                                                                                 \
	} while (0)

int ntsm_lane_submit(ntsm_lane *l, uint64_t n_bytes, uint32_t n_reads)
{
	if (!l) return NTSM_ERR_ARG;
	if (l->error) return l->error;
	Slot &s = l->slot[l->next_slot];
	if (!s.acquired) return NTSM_ERR_STATE;
	s.acquired = false;
	if (n_bytes > l->cap_bytes || n_reads > l->cap_reads) return NTSM_ERR_ARG;
	int rc = check_layout(s.h_read_end, n_reads, n_bytes);
	if (rc) return rc;
	if (n_reads == 0) return NTSM_OK;
	ntsm_ctx *c = l->c;
	LANECHK(hipSetDevice(c->device));
	{
		void *const dst[1] = { s.d_bases };
		const void *const src[1] = { s.h_bases };
		const size_t bytes[1] = { (size_t) n_bytes };
		LANECHK(slot_copy(c, s, dst, src, bytes, 1));
	}
	rc = launch_count(c, s.stream, s.d_bases, 0, n_bytes, nullptr, 0, false, +1);
	if (rc) return lane_lost_batch(l, rc);
	LANECHK(hipEventRecord(s.done, s.stream));
	s.busy = true;
	l->total_bases += n_bytes - n_reads;
	l->reads_consumed += n_reads;
	l->next_slot ^= 1;
	return NTSM_OK;
}

int ntsm_lane_acquire_packed(ntsm_lane *l, uint8_t **codes, uint8_t **valid, uint64_t *cap_positions)
{
	if (!l || !codes || !valid || !cap_positions) return NTSM_ERR_ARG;
	HIPCHK(hipSetDevice(l->c->device));
	Slot &s = l->slot[l->next_slot];
	int rc = wait_slot(s);
	if (rc) return rc;
	/* the slot's pinned buffer (cap_bytes + 64) holds both planes of up to cap_bytes positions: 3/8 of it */
	const uint64_t cap_pos = l->cap_bytes & ~31ull;
	if (!s.d_packed) {
		s.d_packed_bytes = cap_pos / 4 + cap_pos / 8 + 64;
		HIPCHK(device_take(l->c, (void **) &s.d_packed, s.d_packed_bytes));
	}
	s.acquired = true;
	*codes = s.h_bases;
	*valid = s.h_bases + cap_pos / 4;
	*cap_positions = cap_pos;
	return NTSM_OK;
}

int ntsm_lane_submit_packed(ntsm_lane *l, uint64_t n_positions, uint32_t n_reads, uint64_t n_bases)
{
	if (!l) return NTSM_ERR_ARG;
	if (l->error) return l->error;
	Slot &s = l->slot[l->next_slot];
	if (!s.acquired || !s.d_packed) return NTSM_ERR_STATE;
	s.acquired = false;
	const uint64_t cap_pos = l->cap_bytes & ~31ull;
	if ((n_positions & 7) || n_positions > cap_pos || n_bases + n_reads > n_positions) return NTSM_ERR_ARG;
	if (n_reads == 0 || n_positions == 0) return NTSM_OK;
	ntsm_ctx *c = l->c;
	LANECHK(hipSetDevice(c->device));
	/* whole groups of 32 positions cross the link and are unpacked: what lies between the end of the batch and the next
	 * multiple of 32 is marked invalid here (the caller may have left anything there) */
	const uint64_t n_out = (n_positions + 31) & ~31ull;
	uint8_t *h_valid = s.h_bases + cap_pos / 4;
	for (uint64_t p = n_positions; p < n_out; p += 8) h_valid[p >> 3] = 0;
	{
		/* The two planes lie at the same offsets in the pinned slot and in its device image, so a (nearly) full batch crosses
		 * as ONE copy -- the codes plane to its end, then the used part of the validity plane.  Every copy on the copy stream is
		 * followed by ~50 us before the next one starts (profiles/r06_feed/lanes_packed_16.summary.txt), which is about what
		 * a 3 MiB batch itself takes: the unused tail of the codes plane (at most 1 MiB here) is cheaper than a second copy. */
		const uint64_t unused_codes = cap_pos / 4 - n_out / 4;
		if (unused_codes <= (1ull << 20)) {
			void *const dst[1] = { s.d_packed };
			const void *const src[1] = { s.h_bases };
			const size_t bytes[1] = { (size_t) (cap_pos / 4 + n_out / 8) };
			LANECHK(slot_copy(c, s, dst, src, bytes, 1));
		} else {
			void *const dst[2] = { s.d_packed, s.d_packed + cap_pos / 4 };
			const void *const src[2] = { s.h_bases, h_valid };
			const size_t bytes[2] = { (size_t) (n_out / 4), (size_t) (n_out / 8) };
			LANECHK(slot_copy(c, s, dst, src, bytes, 2));
		}
	}
	const uint64_t n16 = n_out / 16;
	LANECHK(launch_unpack((const uint32_t *) s.d_packed, (const uint16_t *) (s.d_packed + cap_pos / 4), s.d_bases, (unsigned long long) n16, s.stream));
	int rc = launch_count(c, s.stream, s.d_bases, 0, n_out, nullptr, 0, false, +1);
	if (rc) return lane_lost_batch(l, rc);
	LANECHK(hipEventRecord(s.done, s.stream));
	s.busy = true;
	l->total_bases += n_bases;
	l->reads_consumed += n_reads;
	l->next_slot ^= 1;
	return NTSM_OK;
}

int ntsm_lane_close(ntsm_lane *l)
{
	if (!l) return NTSM_ERR_ARG;
	ntsm_ctx *c = l->c;
	int rc = l->error;                                     /* a batch this lane lost: the caller hears about it here at the latest */
	if (hipSetDevice(c->device) != hipSuccess && !rc) rc = NTSM_ERR_HIP;
	for (auto &s : l->slot) {                            /* the stream is shared: wait for this lane's own batches only */
		if (s.busy && hipEventSynchronize(s.done) != hipSuccess && !rc) rc = NTSM_ERR_HIP;
		s.busy = false;
		free_slot(s, c);                                  /* device buffers go to the context's cache (nothing of this lane is in flight any more) */
		if (s.done) (void) hipEventDestroy(s.done);
		if (s.copied) (void) hipEventDestroy(s.copied);
	}
	{
		std::lock_guard<std::mutex> lk(c->mu);
		c->total_bases += l->total_bases;
		c->reads_consumed += l->reads_consumed;
		c->open_lanes--;
		if (rc) c->failed = true;
	}
	delete l;
	return rc;
}

int ntsm_submit(ntsm_ctx *c, const uint8_t *bases, uint64_t n_bytes, const uint64_t *read_end, uint32_t n_reads)
{
	if (!c || (n_bytes && !bases)) return NTSM_ERR_ARG;
	if (c->failed) return NTSM_ERR_STATE;
	int rc = check_layout(read_end, n_reads, n_bytes);
	if (rc) return rc;
	if (n_reads == 0) return NTSM_OK;
	if (n_bytes > c->cap_bytes || n_reads > c->cap_reads) {
		uint64_t nb = std::max(c->cap_bytes, n_bytes), nr = std::max<uint64_t>(c->cap_reads, n_reads);
		rc = ntsm_set_batch_capacity(c, nb, nr);
		if (rc) return rc;
	}
	uint8_t *hb; uint64_t cb, *hr, cr;
	rc = ntsm_staging_acquire(c, &hb, &cb, &hr, &cr);
	if (rc) return rc;
	/* The batch goes into the pinned slot on several threads: one thread's memcpy is half of what the link takes
	 * (runtime.cpp: staged_copy).  The offsets are only needed where a read is attributed its hits: an armed (-m) context;
	 * otherwise the slot keeps just the last one, which is all check_layout looks at. */
	staged_copy(c, hb, bases, n_bytes);
	if (c->armed) memcpy(hr, read_end, (size_t) n_reads * sizeof(uint64_t));
	else hr[n_reads - 1] = read_end[n_reads - 1];
	return ntsm_submit_staged(c, n_bytes, n_reads);
}

int ntsm_host_pin(void *p, uint64_t bytes)
{
	if (!p || !bytes) return NTSM_ERR_ARG;
	HIPCHK(hipHostRegister(p, bytes, hipHostRegisterPortable));
	return NTSM_OK;
}

int ntsm_host_unpin(void *p)
{
	if (!p) return NTSM_ERR_ARG;
	HIPCHK(hipHostUnregister(p));
	return NTSM_OK;
}

int ntsm_set_submit_threads(ntsm_ctx *c, int n_threads)
{
	if (!c || n_threads < 0 || n_threads > 64) return NTSM_ERR_ARG;
	c->submit_threads = n_threads;
	return NTSM_OK;
}

/* Zero-copy submit: the H2D copy reads the caller's own pinned memory.  The slot supplies the device buffer, the stream and the
 * "done" event only; two batches may be in flight, the third call waits for the first. */
int ntsm_submit_pinned(ntsm_ctx *c, const uint8_t *bases, uint64_t n_bytes, const uint64_t *read_end, uint32_t n_reads)
{
	if (!c || (n_bytes && !bases)) return NTSM_ERR_ARG;
	if (c->failed) return NTSM_ERR_STATE;
	int rc = check_layout(read_end, n_reads, n_bytes);
	if (rc) return rc;
	if (n_reads == 0) return NTSM_OK;
	HIPCHK(hipSetDevice(c->device));
	{
		/* the whole batch must lie in memory the runtime knows as pinned host memory (hipHostMalloc / hipHostRegister): a DMA from
		 * anything else would be staged by the runtime page by page behind our back, or fault */
		hipPointerAttribute_t at_lo, at_hi;
		const hipError_t e0 = hipPointerGetAttributes(&at_lo, bases), e1 = hipPointerGetAttributes(&at_hi, bases + n_bytes - 1);
		if (e0 != hipSuccess || e1 != hipSuccess || at_lo.type != hipMemoryTypeHost || at_hi.type != hipMemoryTypeHost) {
			(void) hipGetLastError();                             /* "not a registered pointer" is an answer, not a sticky error */
			return NTSM_ERR_ARG;
		}
	}
	if (n_bytes > c->cap_bytes || n_reads > c->cap_reads) {
		uint64_t nb = std::max(c->cap_bytes, n_bytes), nr = std::max<uint64_t>(c->cap_reads, n_reads);
		rc = ntsm_set_batch_capacity(c, nb, nr);
		if (rc) return rc;
	}
	if (c->early_stop) return NTSM_OK;                    /* threshold already tripped: nothing more is counted */
	Slot &s = c->slot[c->next_slot];
	if (s.acquired) return NTSM_ERR_STATE;                /* a staged batch is being filled on this slot */
	if (!s.d_bases) {
		rc = alloc_slot(s, c->device, c->cap_bytes, c->cap_reads, true, false, nullptr, false);
		if (rc) return rc;
	}
	rc = wait_slot(s);
	if (rc) return rc;
	c->reduced = false;
	c->next_slot ^= 1;
	if (c->armed) {
		HIPCHK(h2d_async(s.d_bases, bases, n_bytes, s.stream));
		memcpy(s.h_read_end, read_end, (size_t) n_reads * sizeof(uint64_t));
		HIPCHK(h2d_async(s.d_read_end, s.h_read_end, n_reads * sizeof(uint64_t), s.stream));
		return armed_batch(c, s.stream, s.d_bases, n_bytes, s.d_read_end, s.h_read_end, n_reads);   /* synchronous: the buffer is free on return */
	}
	{
		void *const dst[1] = { s.d_bases };
		const void *const src[1] = { bases };
		const size_t nb[1] = { (size_t) n_bytes };
		HIPCHK(slot_copy(c, s, dst, src, nb, 1));
	}
	rc = launch_count(c, s.stream, s.d_bases, 0, n_bytes, nullptr, 0, false, +1);
	if (rc) return rc;
	HIPCHK(hipEventRecord(s.done, s.stream));
	s.busy = true;
	c->total_bases += n_bytes - n_reads;
	c->reads_consumed += n_reads;
	return NTSM_OK;
}

int ntsm_count_resident(ntsm_ctx *c, const void *d_bases, uint64_t n_bytes, const void *d_read_end, uint64_t n_reads, int sign)
{
	if (!c || (n_bytes && !d_bases) || (sign != 1 && sign != -1)) return NTSM_ERR_ARG;
	if (((uintptr_t) d_bases & 15) != 0) return NTSM_ERR_ARG;
	if (c->failed) return NTSM_ERR_STATE;
	if (n_reads == 0 || n_bytes == 0) return NTSM_OK;
	if (c->early_stop) return NTSM_OK;
	c->reduced = false;
	HIPCHK(hipSetDevice(c->device));
	if (c->armed && sign > 0) {
		if (!d_read_end) return NTSM_ERR_ARG;
		return armed_batch(c, c->rstream, (const uint8_t *) d_bases, n_bytes, (const uint64_t *) d_read_end, nullptr, n_reads);
	}
	int rc = launch_count(c, c->rstream, (const uint8_t *) d_bases, 0, n_bytes, nullptr, 0, false, sign);
	if (rc) return rc;
	if (sign > 0) { c->total_bases += n_bytes - n_reads; c->reads_consumed += n_reads; }
	else { c->total_bases -= n_bytes - n_reads; c->reads_consumed -= n_reads; }
	return NTSM_OK;
}

int ntsm_sync(ntsm_ctx *c, ntsm_totals *t)
{
	if (!c) return NTSM_ERR_ARG;
	if (c->failed) return NTSM_ERR_STATE;
	{
		std::lock_guard<std::mutex> lk(c->mu);
		if (c->open_lanes) return NTSM_ERR_STATE;             /* lanes hold batches this call cannot see: close them first */
	}
	HIPCHK(hipSetDevice(c->device));
	for (auto &s : c->slot) {
		if (s.stream) HIPCHK(hipStreamSynchronize(s.stream));
		s.busy = false;
	}
	HIPCHK(hipStreamSynchronize(c->rstream));
	HIPCHK(hipStreamSynchronize(c->cstream));                /* (every copy there is followed by a kernel that was just waited for) */
	if (t) {
		memset(t, 0, sizeof *t);
		if (c->reduced) {
			t->total_kmers = c->red_totals[0];
			t->total_hits = c->red_totals[1];
			t->total_bases = c->red_totals[2];
			t->reads_consumed = c->red_totals[3];
		} else {
			uint64_t dv[2];
			int rc = read_device_totals(c, dv);
			if (rc) return rc;
			t->total_kmers = dv[0];
			t->total_hits = dv[1];
			t->total_bases = c->total_bases;
			t->reads_consumed = c->reads_consumed;
		}
		t->early_stop = c->early_stop ? 1 : 0;
	}
	return NTSM_OK;
}

int ntsm_counts_device(ntsm_ctx *c, void **d_vec, uint64_t *n_words)
{
	if (!c) return NTSM_ERR_ARG;
	ntsm_totals t;
	int rc = ntsm_sync(c, &t);
	if (rc) return rc;
	if (!c->reduced) {
		if (c->n_kmers) {
			HIPCHK(launch_gather(c->d_keys, c->d_slot_of, c->n_kmers, c->d_vec, c->rstream));
		}
		const uint64_t tail[4] = { t.total_kmers, t.total_hits, t.total_bases, t.reads_consumed };
		HIPCHK(h2d_async(c->d_vec + c->n_kmers, tail, sizeof tail, c->rstream));
		HIPCHK(hipStreamSynchronize(c->rstream));
	}
	if (d_vec) *d_vec = c->d_vec;
	if (n_words) *n_words = (uint64_t) c->n_kmers + 4;
	return NTSM_OK;
}

int ntsm_import_reduced(ntsm_ctx *c)
{
	if (!c) return NTSM_ERR_ARG;
	if (c->failed) return NTSM_ERR_STATE;
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipDeviceSynchronize());
	HIPCHK(hipMemcpy(c->red_totals, c->d_vec + c->n_kmers, sizeof c->red_totals, hipMemcpyDeviceToHost));
	c->reduced = true;
	return NTSM_OK;
}

int ntsm_counts(ntsm_ctx *c, uint64_t *out)
{
	if (!c || (!out && c->n_kmers)) return NTSM_ERR_ARG;
	int rc = ntsm_counts_device(c, nullptr, nullptr);
	if (rc) return rc;
	if (c->n_kmers) HIPCHK(hipMemcpy(out, c->d_vec, (uint64_t) c->n_kmers * sizeof(uint64_t), hipMemcpyDeviceToHost));
	return NTSM_OK;
}

int ntsm_reset(ntsm_ctx *c)
{
	if (!c) return NTSM_ERR_ARG;
	int rc = ntsm_sync(c, nullptr);
	if (rc) return rc;
	/* on the context's own stream and waited for: a null-stream memset is not ordered against the
	 * non-blocking streams the count kernels run on */
	HIPCHK(launch_zero_counts(c->d_keys, (unsigned long long) (c->n_slots / 2), c->rstream));
	HIPCHK(hipMemsetAsync(c->d_totals, 0, 4 * sizeof(uint64_t), c->rstream));
	HIPCHK(hipStreamSynchronize(c->rstream));
	c->total_bases = c->reads_consumed = 0;
	c->early_stop = c->reduced = false;
	return NTSM_OK;
}

int ntsm_set_max_hits(ntsm_ctx *c, uint64_t max_hits, int armed)
{
	if (!c) return NTSM_ERR_ARG;
	int rc = ntsm_sync(c, nullptr);                       /* also refuses while lanes are open */
	if (rc) return rc;
	c->max_hits = max_hits;
	c->armed = armed != 0;
	c->reduced = false;                                    /* what follows is counted locally: ntsm_sync reports this context's own totals again */
	return NTSM_OK;
}

int ntsm_set_timing(ntsm_ctx *c, int on)
{
	if (!c) return NTSM_ERR_ARG;
	int rc = ntsm_sync(c, nullptr);
	if (rc) return rc;
	memset(c->ev_used, 0, sizeof c->ev_used);
	c->ev_next = 0;
	c->t_launches = 0;
	c->t_ms = 0;
	c->timing = on != 0;
	return NTSM_OK;
}

int ntsm_get_timing(ntsm_ctx *c, uint64_t *n_launches, double *total_ms)
{
	if (!c) return NTSM_ERR_ARG;
	int rc = ntsm_sync(c, nullptr);
	if (rc) return rc;
	for (int i = 0; i < kTimingPool; ++i)
		if (c->ev_used[i]) {
			float ms = 0;
			HIPCHK(hipEventElapsedTime(&ms, c->ev_a[i], c->ev_b[i]));
			c->t_ms += ms;
			c->ev_used[i] = false;
		}
	if (n_launches) *n_launches = c->t_launches;
	if (total_ms) *total_ms = c->t_ms;
	return NTSM_OK;
}

int ntsm_set_tuning(ntsm_ctx *c, int filter_log2_bits, int grid_blocks)
{
	if (!c) return NTSM_ERR_ARG;
	int rc = ntsm_sync(c, nullptr);
	if (rc) return rc;
	c->grid_blocks = grid_blocks;
	if (filter_log2_bits >= 4000000 && filter_log2_bits < 4000016) {   /* 4000000 + v: memory kinds of block filter (v & 3) and key table (v >> 2) */
		c->blocks_mem_kind = (unsigned) (filter_log2_bits - 4000000) & 3u;
		c->keys_mem_kind = ((unsigned) (filter_log2_bits - 4000000) >> 2) & 3u;
		filter_log2_bits = 1000000 + (int) (c->bloom_words_req / 256u);     /* falls into the rebuild below */
	}
	if (filter_log2_bits >= 3000000 && filter_log2_bits < 3000040) {   /* 3000000 + v: drain Bloom of 2^v bits (0: automatic again) */
		c->prefilter_log2_req = (uint32_t) (filter_log2_bits - 3000000);
		if (c->prefilter_log2_req && (c->prefilter_log2_req < 10 || c->prefilter_log2_req > 30)) { c->prefilter_log2_req = 0; return NTSM_ERR_ARG; }
		filter_log2_bits = 1000000 + (int) (c->bloom_words_req / 256u);     /* falls into the rebuild below */
	}
	if (filter_log2_bits >= 2000000 && filter_log2_bits < 3000000) {   /* 2000000 + w: blocked filter of w KiB (0: automatic again) */
		c->blocks_kib_req = (uint32_t) (filter_log2_bits - 2000000);
		filter_log2_bits = 1000000 + (int) (c->bloom_words_req / 256u);     /* falls into the rebuild below */
	}
	if (filter_log2_bits == 2 || filter_log2_bits == 3) {       /* 2 / 3: two-level path with / without the drain's Bloom (default: without) */
		c->prefilter_forced = filter_log2_bits == 2;
		filter_log2_bits = 1000000 + (int) (c->bloom_words_req / 256u);     /* falls into the rebuild below (0 words = automatic) */
	}
	if ((filter_log2_bits >= 200 && filter_log2_bits < 300) || filter_log2_bits >= 1000000) {
		/* 200 + v: two-level path, Bloom of 2^v bits; 250 + v: 3 * 2^v bits; 1000000 + w: w KiB */
		if (filter_log2_bits >= 1000000) {
			c->bloom_words_req = (uint32_t) (filter_log2_bits - 1000000) * 256u;
			if (c->bloom_words_req > (1u << 26)) return NTSM_ERR_ARG;
		} else {
			const int v = filter_log2_bits >= 250 ? filter_log2_bits - 250 : filter_log2_bits - 200;
			if (v < 10 || v > 28) return NTSM_ERR_ARG;
			c->bloom_words_req = (filter_log2_bits >= 250 ? 3u : 1u) << (v - 5);
		}
		filter_log2_bits = c->filter_log2_req;
		HIPCHK(hipSetDevice(c->device));
		rc = build_tables(c, filter_log2_bits);
		if (rc) { c->failed = true; return rc; }          /* old tables freed, new ones incomplete: the context is unusable */
		HIPCHK(hipMemset(c->d_totals, 0, 4 * sizeof(uint64_t)));
		HIPCHK(hipDeviceSynchronize());
		c->total_bases = c->reads_consumed = 0;
		c->early_stop = c->reduced = false;
		return NTSM_OK;
	}
	if (filter_log2_bits > 0) {
		c->filter_log2_req = filter_log2_bits;
		rc = build_tables(c, filter_log2_bits);              /* rebuilds filters and table: counts start from zero again */
		if (rc) { c->failed = true; return rc; }
		HIPCHK(hipMemset(c->d_totals, 0, 4 * sizeof(uint64_t)));
		HIPCHK(hipDeviceSynchronize());
		c->total_bases = c->reads_consumed = 0;
		c->early_stop = c->reduced = false;
	}
	return NTSM_OK;
}

int ntsm_set_armed_chunk(ntsm_ctx *c, uint64_t chunk_bytes)
{
	if (!c) return NTSM_ERR_ARG;
	c->armed_chunk_bytes = chunk_bytes ? chunk_bytes : (256ull << 20);
	return NTSM_OK;
}

void *ntsm_stream(ntsm_ctx *c) { return c ? (void *) c->rstream : nullptr; }

int ntsm_rccl_probe(void) { return rccl_available() ? NTSM_OK : NTSM_ERR_RCCL; }

int ntsm_set_kernel(ntsm_ctx *c, int variant)
{
	if (!c || variant < 0 || variant > 5) return NTSM_ERR_ARG;
	if (variant == 5 && c->k != NTSM_FAST_K) return NTSM_ERR_ARG;   /* the run-anchored kernel exists for k = 19 */
	if (variant == 3 && !kWithTab) return NTSM_ERR_ARG;    /* the tabulated kernel is not part of this build (make tab) */
	if (variant == 4 && ntsm_fast_plan((uint32_t) c->k, true).m != NTSM_TWO_M) return NTSM_ERR_ARG;   /* 15 <= k <= 31 */
	int rc = ntsm_sync(c, nullptr);
	if (rc) return rc;
	const int before = c->kernel_variant;
	c->kernel_variant = variant;
	/* one-level and two-level filters are different tables (12-mer / 14-mer minimizers): a change of level rebuilds them */
	/* (the tabulated kernel hands its exotic tiles to the ONE-level k = 19 kernel: variant 3 on a context that had chosen two
	 * levels by itself rebuilds the one-level tables, otherwise those tiles would probe 14-mer-addressed blocks with 12-mers) */
	const bool want_run = choose_run_form(c, variant, c->filter_log2_req);
	const bool want_two = ntsm_fast_plan((uint32_t) c->k, true).m == NTSM_TWO_M && variant != 1 && variant != 3 && !want_run &&
		(variant == 4 || (variant == 0 && c->filter_log2_req == 0 && wants_two_level(c->n_kmers)));
	(void) before;
	if (variant != 1 && (want_two != c->two_level || want_run != c->run_form)) {   /* the run form has a filter of its own */
		HIPCHK(hipSetDevice(c->device));
		rc = build_tables(c, c->filter_log2_req);
		if (rc) { c->failed = true; return rc; }            /* old tables freed, new ones incomplete: the context is unusable */
		HIPCHK(hipMemset(c->d_totals, 0, 4 * sizeof(uint64_t)));
		HIPCHK(hipDeviceSynchronize());
		c->total_bases = c->reads_consumed = 0;
		c->early_stop = c->reduced = false;
	}
	return NTSM_OK;
}

long long ntsm_debug_fail_after(int kind, long long nth) { return fault_arm(kind, nth); }

/* The run-anchored kernel's filter as tables.cpp builds it, WITHOUT a device (host code only): for the CPU test that walks every
 * site k-mer through the device's derivation (tests/test_host_cpu.py::test_run_form_filter_has_no_false_negatives). */
int ntsm_debug_run_filter(const uint64_t *keys, uint32_t n_kmers, uint32_t kib, uint32_t *blocks_out, uint64_t *n_blocks)
{
	if (!keys || !n_kmers || !n_blocks) return NTSM_ERR_ARG;
	ntsm_ctx *c = new (std::nothrow) ntsm_ctx();
	if (!c) return NTSM_ERR_NOMEM;
	c->k = NTSM_FAST_K;
	c->n_kmers = n_kmers;
	c->mask = mask_for_k(c->k);
	c->kernel_variant = 5;
	c->blocks_kib_req = kib;
	c->canon.assign(keys, keys + n_kmers);
	TableImages img;
	const int rc = build_tables_host(c, 0, img);
	if (rc == NTSM_OK) {
		*n_blocks = c->n_rblocks;
		if (blocks_out) memcpy(blocks_out, img.rblocks.data(), img.rblocks.size() * sizeof(uint32_t));
	}
	delete c;
	return rc;
}

/* The automatic form choice without a device: what build_tables_host would decide for a fresh context (variant 0, no tuning). */
int ntsm_debug_form_choice(const uint64_t *keys, uint32_t n_kmers, int k, int key_kind, int *form)
{
	if ((!keys && n_kmers) || !form || k < 1 || k > 32) return NTSM_ERR_ARG;
	if (key_kind != NTSM_KEYS_CANONICAL && key_kind != NTSM_KEYS_HASH64) return NTSM_ERR_ARG;
	ntsm_ctx *c = new (std::nothrow) ntsm_ctx();
	if (!c) return NTSM_ERR_NOMEM;
	c->k = k;
	c->n_kmers = n_kmers;
	c->mask = mask_for_k(k);
	c->canon.resize(n_kmers);
	for (uint32_t i = 0; i < n_kmers; ++i) c->canon[i] = key_kind == NTSM_KEYS_HASH64 ? ntsm_hash64_inv(keys[i], k) : keys[i];
	const bool run = choose_run_form(c, 0, 0);
	const bool two = !run && ntsm_fast_plan((uint32_t) k, true).m == NTSM_TWO_M && wants_two_level(n_kmers);
	*form = ntsm_fast_plan((uint32_t) k, two).mode < 0 ? 3 : run ? 2 : two ? 1 : 0;
	delete c;
	return NTSM_OK;
}

int ntsm_debug_stats(ntsm_ctx *c, uint64_t out[8])
{
	if (!c || !out) return NTSM_ERR_ARG;
	int rc = ntsm_sync(c, nullptr);
	if (rc) return rc;
	HIPCHK(hipDeviceSynchronize());
	const uint64_t exotic = tab_exotic_seen(c, &rc);         /* 0 in the default build */
	if (rc) return rc;
	out[0] = exotic;
	out[1] = c->n_launch[0];
	out[2] = c->n_launch[1];
	out[3] = c->n_launch[2];
	uint64_t dv[4] = { 0, 0, 0, 0 };
	HIPCHK(hipMemcpy(dv, c->d_totals, sizeof dv, hipMemcpyDeviceToHost));
	out[4] = dv[2];
	out[5] = c->run_form ? 2 : c->two_level ? 1 : 0;
	out[6] = c->n_bloom_words;
	out[7] = c->n_site_minimizers;
	return NTSM_OK;
}

/* One process driving n GPUs: RCCL SUM of every context's dense count vector + totals over xGMI.
 * SUM (not MAX): the per-site maxima are taken on the host from the summed per-k-mer counts,
 * which is what a single reference run computes (src/FingerPrint.hpp:281-294). */
int ntsm_allreduce(ntsm_ctx *const *ctxs, int n)
{
	if (!ctxs || n < 1) return NTSM_ERR_ARG;
	for (int i = 0; i < n; ++i) {
		if (!ctxs[i] || ctxs[i]->n_kmers != ctxs[0]->n_kmers) return NTSM_ERR_ARG;
		int rc = ntsm_counts_device(ctxs[i], nullptr, nullptr);
		if (rc) return rc;
	}
	if (n > 1) {
		const int rc = rccl_group_allreduce(ctxs, n);
		if (rc) return rc;
	}
	for (int i = 0; i < n; ++i) {
		int rc = ntsm_import_reduced(ctxs[i]);
		if (rc) return rc;
	}
	return NTSM_OK;
}

} // extern "C"
