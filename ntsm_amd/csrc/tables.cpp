/*
 * tables.cpp -- host-side construction of everything the count kernels look things up in: the 2-choice cuckoo key table
 * (what tsl::robin_map<uint64_t, size_t> m_counts is in the reference: src/FingerPrint.hpp:466, filled by initCountsHash
 * :490-564), the generic kernel's bit filter, the minimizer-blocked filter, the drain's second-level Bloom and the
 * two-level path's minimizer Bloom.  Pure host code: no HIP call, no device code -- ntsm_create runs it
 * while another thread may still be bringing the runtime up; runtime.cpp uploads the images.
 *
 * The hash functions are shared with the kernels (ntsm_device.h) so that both sides agree bit for bit.
 */
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "ntsm_internal.h"

namespace ntsm_rt {

#ifndef NTSM_TABLE_LOAD
#define NTSM_TABLE_LOAD 0.25                   /* cuckoo key table: slots >= keys / load, power of two.  A sparser table costs memory, not time
                                                * (the Infinity Cache holds 128 MiB as well as 64): what a fuller one costs is the second-bucket probe of
                                                * a key whose first bucket is full -- 903 / 889 / 856 / 833 Gbases/s at load <= 0.25 / 0.4 / 0.6 / 0.8 */
#endif

/* vendor/KseqHashIterator.hpp:114-127 restated as data for the kernel's LDS table */
void build_lut(uint8_t *lut)
{
	for (int i = 0; i < 256; ++i) lut[i] = 4;
	for (int i = 0; i < 4; ++i) lut[i] = (uint8_t) i;
	lut['A'] = lut['a'] = 0;
	lut['C'] = lut['c'] = 1;
	lut['G'] = lut['g'] = 2;
	lut['T'] = lut['t'] = lut['U'] = lut['u'] = 3;
}

bool wants_two_level(uint64_t n_keys) { return (12ull * n_keys + 127) / 128 > (9ull << 15); }   /* 12 bits per key > 4.5 MiB */

/* Run-anchored kernel or minimizer-blocked kernel?  Measured on 150 bp reads, site windows with all 13 k-mers kept (the clusters
 * the run form feeds on), best minimizer-blocked form (one / two levels) against the run form, Gbases/s, same box (round 5, after
 * the run kernel's queues were made to outlive their tile): 1.04 M keys 847 / 848, 1.3 M 847 / 857, 1.56 M 805 / 843, 2.0 M 715 / 831,
 * 2.5 M 656 / 832, 3.4 M 616 / 808, 4.2 M 605 / 777, 5.7 M 563 / 649, 8.3 M 521 / 562, 10.4 M 502 / 486, 13.5 M 470 / 372; the bench
 * set (1.54 M keys, 3 .. 13 k-mers kept per window) 862 / 866.  Below 1.8 M keys the two are level (the one-level minimizer-blocked
 * filter still fits the L2) and the minimizer-blocked kernel stays; from 8-9 M keys on the two-level form's minimizer Bloom is
 * the better L2 resident. */
/* A second synthetic family (round 6, tools/window_family.sh: 3 .. 13 k-mers kept per allele, the bench set's structure; one level / two
 * levels / run form): 1.54 M keys 890 / 679 / 857, 1.84 M 809 / 665 / 849, 2.16 M 751 / 657 / 847, 2.56 M 717 / 651 / 830, 3.36 M 492 / 637 / 793,
 * 4.33 M 394 / 608 / 684, 6.09 M 322 / 567 / 609, 8.97 M 279 / 525 / 451, 12.0 M 252 / 476 / 308: the lower edge holds, the upper edge lies between
 * 6.1 and 9.0 M keys there (8.3 - 10.4 M on the first family) -- 8 M is the compromise (at most 8 % lost on either family next to the edge). */
bool wants_run_form(int k, uint64_t n_keys) { return k == NTSM_FAST_K && n_keys >= 1800000ull && n_keys < 8000000ull; }

uint64_t mask_for_k(int k) { return k >= 32 ? 0ull : ((1ull << (2 * k)) - 1); }   /* k = 32: see include/ntsm_hip.h */

#ifdef NTSM_WITH_TAB
#include "ntsm_tab_tables.inc"
#else
static inline void build_tab_filter(ntsm_ctx *, int) {}
#endif

/* ---- run-anchored kernel (kernels_run.hip): what the host derives from one site k-mer --------------------------------------
 * hmin = smallest order key among its eight canonical 12-mers, where = the start offsets q at which it occurs (bit q). */
static inline void run_minimizer(uint64_t x, uint32_t *hmin, uint8_t *where)
{
	uint64_t rc = ~x;
	rc = ((rc >> 2) & 0x3333333333333333ull) | ((rc & 0x3333333333333333ull) << 2);
	rc = ((rc >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((rc & 0x0F0F0F0F0F0F0F0Full) << 4);
	rc = __builtin_bswap64(rc) >> (64 - 2 * NTSM_FAST_K);
	uint32_t best = 0xFFFFFFFFu;
	uint8_t at = 0;
	for (uint32_t q = 0; q < 8; ++q) {
		const uint32_t sub = (uint32_t) (x >> (2 * (7 - q))) & 0xFFFFFFu, rsub = (uint32_t) (rc >> (2 * q)) & 0xFFFFFFu;
		const uint32_t h = ntsm_run_order24_host(sub, rsub);
		if (h < best) { best = h; at = (uint8_t) (1u << q); }
		else if (h == best) at |= (uint8_t) (1u << q);
	}
	*hmin = best;
	*where = at;
}
/* strand-symmetric signature word of the anchored 16-mer of a k-mer whose minimizer starts at offset q: q <= 3: M + 4 bases
 * right = bases q .. q+15 (class R); q >= 4: 4 bases left + M = bases q-4 .. q+11 (class L) */
static inline uint32_t run_signature(uint64_t x, uint32_t q)
{
	const uint32_t E = q <= 3 ? (uint32_t) (x >> (2 * (3 - q))) : (uint32_t) (x >> (2 * (7 - q)));
	return E + ntsm_rc16_word(E);
}

/* Does the site set have the cluster structure the run form feeds on?  ntsm's site files hold the k-mers of 31-base windows, and
 * the k-mers of a window that share a minimizer share their anchored 16-mer: on such a set far fewer distinct (minimizer,
 * signature) pairs than k-mers exist (2.5 M keys of 13 per window: 0.55 per key; 3 .. 13 kept: 0.39).  A set of unrelated k-mers has
 * one pair per key: the filter then holds no fewer entries than the minimizer-blocked one while every run still pays two tests --
 * the automatic choice leaves such a set to the other kernels.
 * The estimate is a property of the SET, not of the order the caller hands the keys over in (the reference-side binding iterates
 * m_counts, a robin_map: hash order, src/FingerPrint.hpp:466 -- consecutive keys are unrelated there although the set is the same):
 * every key whose minimizer falls into a fixed 1/32 residue class of the block hash is taken -- all k-mers of a window that share a
 * minimizer are in or out together --, the (minimizer, signature) pairs are sorted and distinct pairs / sampled keys is compared
 * with 0.8.  Minimizers of all keys on four threads: 25 ms for 2.5 M keys, only inside the size window of wants_run_form. */
static bool run_form_pays(const ntsm_ctx *c)
{
	const uint32_t n = c->n_kmers;
	if (n < 64) return false;
	const uint32_t parts = n >= (1u << 16) ? 4u : 1u;
	const uint32_t sample_shift = n >= (1u << 16) ? 27u : 32u;          /* small sets: every key */
	std::vector<std::vector<uint64_t>> found(parts);
	auto scan = [&](uint32_t part) {
		const uint32_t lo = (uint32_t) ((uint64_t) n * part / parts), hi = (uint32_t) ((uint64_t) n * (part + 1) / parts);
		std::vector<uint64_t> &out = found[part];
		for (uint32_t i = lo; i < hi; ++i) {
			uint32_t h; uint8_t at;
			run_minimizer(c->canon[i], &h, &at);
			if (sample_shift < 32 && (ntsm_block_hash(h) >> sample_shift) != 0) continue;
			out.push_back((uint64_t) h << 32 | run_signature(c->canon[i], (uint32_t) __builtin_ctz(at)));
		}
	};
	{
		std::vector<std::thread> pool;
		for (uint32_t t = 1; t < parts; ++t) pool.emplace_back(scan, t);
		scan(0);
		for (auto &th : pool) th.join();
	}
	std::vector<uint64_t> pairs;
	for (auto &v : found) pairs.insert(pairs.end(), v.begin(), v.end());
	if (pairs.size() < 16) return false;
	const uint64_t keys = pairs.size();
	std::sort(pairs.begin(), pairs.end());
	const uint64_t distinct = (uint64_t) (std::unique(pairs.begin(), pairs.end()) - pairs.begin());
	return 10 * distinct <= 8 * keys;
}

bool choose_run_form(const ntsm_ctx *c, int variant, int filter_log2_req)
{
	if (c->k != NTSM_FAST_K) return false;
	if (variant == 5) return true;
	return variant == 0 && filter_log2_req == 0 && !c->blocks_kib_req && wants_run_form(c->k, c->n_kmers) && run_form_pays(c);
}

/* The structures are independent functions of the key set: built on four host threads (the cuckoo table dominates). */
int build_tables_host(ntsm_ctx *c, int filter_log2_req, TableImages &img)
{
	const uint32_t n = c->n_kmers;
	/* The four structures are independent functions of the key set: built on four host threads (the cuckoo table
	 * dominates), uploaded afterwards. */
	std::vector<uint64_t> keys;
	std::vector<uint32_t> &filter = img.filter, &blocks = img.blocks /* 4 words per block */, &prefilter = img.prefilter, &bloom = img.bloom;
	filter.clear(); blocks.clear(); prefilter.clear(); bloom.clear();
	int cuckoo_rc = NTSM_OK;
	auto build_cuckoo = [&]() {
		/* slots: power of two with load <= NTSM_TABLE_LOAD; at least 32 */
		uint64_t slots = 32;
		while ((double) n > NTSM_TABLE_LOAD * (double) slots) slots <<= 1;
		for (;; slots <<= 1) {
			const uint32_t blog = (uint32_t) __builtin_ctzll(slots >> 1);
			const uint32_t bshift = 32 - blog;
			keys.assign(slots, NTSM_EMPTY_KEY);
			c->slot_of.assign(n, 0);
			std::vector<uint32_t> owner(slots, 0);            /* dense index stored in each slot */
			bool ok = true;
			uint64_t rng = 0x9E3779B97F4A7C15ull;
			/* 1.5 M keys go into 64 MB of slots at random: every insertion is two cache misses unless its two buckets are asked
			 * for a few dozen keys ahead (0.055 s -> 0.015 s for the human set) */
			constexpr uint32_t kAhead = 24;
			auto buckets_of = [&](uint64_t key, uint64_t b[2]) {
				const uint32_t f = ntsm_fold(key);
				b[0] = 2ull * (ntsm_h1(f) >> bshift);
				b[1] = 2ull * (ntsm_h2(f) >> bshift);
			};
			for (uint32_t i = 0; i < n && i < kAhead; ++i) {
				uint64_t b[2];
				buckets_of(c->canon[i], b);
				__builtin_prefetch(&keys[b[0]], 1);
				__builtin_prefetch(&keys[b[1]], 1);
			}
			for (uint32_t i = 0; i < n && ok; ++i) {
				if (i + kAhead < n) {
					uint64_t b[2];
					buckets_of(c->canon[i + kAhead], b);
					__builtin_prefetch(&keys[b[0]], 1);
					__builtin_prefetch(&keys[b[1]], 1);
				}
				uint64_t key = c->canon[i];
				uint32_t idx = i;
				uint64_t b[2];
				buckets_of(key, b);
				/* duplicate check against both candidate buckets */
				for (int q = 0; q < 2; ++q)
					for (int s = 0; s < 2; ++s)
						if (keys[b[q] + s] == key) { cuckoo_rc = NTSM_ERR_DUP_KEY; return; }
				bool placed = false;
				for (int kick = 0; kick < 1000 && !placed; ++kick) {
					if (kick) buckets_of(key, b);
					for (int q = 0; q < 2 && !placed; ++q)
						for (int s = 0; s < 2 && !placed; ++s)
							if (keys[b[q] + s] == NTSM_EMPTY_KEY) {
								keys[b[q] + s] = key;
								owner[b[q] + s] = idx;
								c->slot_of[idx] = (uint32_t) (b[q] + s);
								placed = true;
							}
					if (placed) break;
					rng = rng * 6364136223846793005ull + 1442695040888963407ull;
					const uint64_t victim = b[(rng >> 33) & 1] + ((rng >> 34) & 1);
					std::swap(key, keys[victim]);
					std::swap(idx, owner[victim]);
					c->slot_of[owner[victim]] = (uint32_t) victim;   /* the key that moved in; the one that moved out is placed next */
				}
				if (!placed) ok = false;
			}
			if (!ok) continue;                                /* grow and retry */
			c->n_slots = slots;
			c->bucket_log2 = blog;
			break;
		}
	};
	auto build_filter = [&]() {
		/* filter: >= 8 bits per key, 2^16 .. 2^28 bits; F = 24 (2 MiB) for the 1.5 M-key human set */
		uint32_t flog = 16;
		while (flog < 28 && (1ull << flog) < 8ull * n) ++flog;
		if (filter_log2_req >= 10 && filter_log2_req <= 30) flog = (uint32_t) filter_log2_req;
		c->filter_log2 = flog;
		filter.assign((1ull << flog) / 32, 0);
		const uint32_t fshift = 32 - flog;
		for (uint32_t i = 0; i < n; ++i) {
			const uint32_t bit = ntsm_h1(ntsm_fold(c->canon[i])) >> fshift;
			filter[bit >> 5] |= 1u << (bit & 31);
		}
	};
	/* Two-level path (15 <= k <= 31): chosen when the blocked filter at 12 bits per key is far enough out of the L2 (more than
	 * 4.5 MiB: beyond ~3.1 M site k-mers.  Measured one level / two levels: 1.9 M keys 797 / 665, 2.6 M 715 / 650, 4.0 M
	 * 457 / 623, 8.0 M 292 / 530, 16 M 219 / 404 Gbases/s), or forced either way with ntsm_set_kernel (4 / 2). */
	c->run_form = choose_run_form(c, c->kernel_variant, filter_log2_req);
	c->two_level = ntsm_fast_plan((uint32_t) c->k, true).m == NTSM_TWO_M && c->kernel_variant != 1 && !c->run_form &&
		(c->kernel_variant == 4 || (c->kernel_variant == 0 && filter_log2_req == 0 && wants_two_level(n)));
	const NtsmFastPlan plan = ntsm_fast_plan((uint32_t) c->k, c->two_level);
	auto build_blocks = [&]() {
		/* minimizer-addressed blocked filter (k = 19 and the other k of ntsm_fast_plan).  Size = smallest of {2^e,
		 * 3 * 2^(e-2)} blocks with at least 12 bits per key: 3 MiB for the 1.54 M-key human set -- it must leave room in the
		 * 4 MiB per-XCD L2 for the read stream and the bucket lines, a full 4 MiB filter misses L2 on 18 % of its reads. */
		if (plan.mode >= 0) {
			uint32_t e = 6, mult = 1;
			if (filter_log2_req >= 100 && filter_log2_req <= 130) {          /* 100 + v: 3 * 2^v bits */
				mult = 3; e = (uint32_t) (filter_log2_req - 100) - 7;
			} else if (filter_log2_req >= 10 && filter_log2_req <= 30) {
				e = (uint32_t) filter_log2_req - 7;
			} else {
				/* blocks: 12 bits per key; 16 on the two-level path, where the filter lives in the Infinity Cache anyway and a
				 * false positive costs a bucket read from HBM (16 M keys: 32 MiB measured 400 Gbases/s against 391 at 24 MiB) */
				const uint64_t want = ((c->two_level ? 16ull : 12ull) * n + 127) / 128;
				while ((1ull << e) < want && e < 23) ++e;
				if (e > 8 && (3ull << (e - 2)) >= want) { mult = 3; e -= 2; }   /* 0.75 * 2^e is enough */
			}
			if (e > 22) e = 22;
			if (e < 4) e = 4;
			c->n_blocks = (uint64_t) mult << e;
			if (filter_log2_req == 0 && !c->two_level) {
				/* One level, automatic: the index is a multiply-high range reduction, so the size need not be 2^e or 3 * 2^e.
				 * 13.5 bits per key in steps of 64 KiB, at most 3 MiB: the filter shares the 4 MiB L2 with the stream and the
				 * look-ups' lines, and past ~2.75 MiB every further bit per key is paid for in L2 misses.  Measured on the bench
				 * set (1.54 M keys, 3e8 reads; 2 / 2.25 / 2.375 / 2.5 / 2.625 / 2.75 / 3 MiB): 862 / 883 / 888 / 891 / 888 / 885 /
				 * 863 Gbases/s; 2.08 M keys: 3 MiB 777, 3.25 770; 2.56 M keys: 3 / 3.25 / 3.5 / 3.75 / 4 MiB: 718 / 721 / 722 / 711 / 716.
				 * The cap only makes sense where a bigger set has somewhere else to go: k = 13, 14 have no 14-mer minimizers and a
				 * context forced to one level (ntsm_set_kernel 2) must not fall back on a saturated 3 MiB filter -- those keep the
				 * 2^e / 3 * 2^(e-2) ladder at >= 12 bits per key computed above (up to 64 MiB). */
				const bool has_two_level_form = ntsm_fast_plan((uint32_t) c->k, true).m == NTSM_TWO_M && c->kernel_variant != 2 && c->kernel_variant != 3;   /* 3 (tabulated, make tab) is forced to one level too */
				const uint64_t kib_want = std::max<uint64_t>(1, (27ull * n / 16 + 1023) / 1024);   /* 13.5 bits = 27/16 bytes per key */
				if (has_two_level_form || kib_want <= 3072) {
					const uint64_t kib = std::min<uint64_t>(3072, kib_want);
					c->n_blocks = std::max<uint64_t>(16, ((kib + 63) / 64 * 64) * 64);
					if (kib < 64) c->n_blocks = std::max<uint64_t>(16, kib * 64);
				}
			}
			if (c->blocks_kib_req) c->n_blocks = (uint64_t) c->blocks_kib_req * 64;   /* tuning: any size, the index is a multiply-high range reduction */
			c->blk_map.n_blocks = (uint32_t) c->n_blocks;
			blocks.assign(c->n_blocks * 4, 0u);
			std::vector<uint32_t> site_mz;                            /* two-level path: every site k-mer's minimizer */
			if (c->two_level) site_mz.resize(n);
			/* the minimizer of a key is eight hashes and a reverse complement (40 ns): with the table build down to 15 ms this
			 * loop is what the four structures wait for, so it is cut into ranges of keys; the bits are OR-ed atomically */
			auto fill_blocks = [&](uint32_t lo, uint32_t hi) {
			for (uint32_t i = lo; i < hi; ++i) {
				const uint64_t x = c->canon[i];
				/* reverse complement of the 2k-bit code: complement, then reverse the 2-bit groups of the 64-bit word */
				uint64_t rc = ~x;
				rc = ((rc >> 2) & 0x3333333333333333ull) | ((rc & 0x3333333333333333ull) << 2);
				rc = ((rc >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((rc & 0x0F0F0F0F0F0F0F0Full) << 4);
				rc = __builtin_bswap64(rc) >> (64 - 2 * plan.k);
				const uint32_t mmask = (1u << (2 * plan.m)) - 1u;
				uint32_t mz = 0xFFFFFFFFu;
				for (uint32_t j = plan.a; j < plan.a + plan.w; ++j) {     /* the candidate m-mer at offset j from the end, and its reverse complement */
					const uint32_t sub = (uint32_t) (x >> (2 * j)) & mmask;
					const uint32_t rsub = (uint32_t) (rc >> (2 * (plan.k - plan.m - j))) & mmask;
					mz = std::min(mz, ntsm_mmer_hash_m(std::min(sub, rsub), plan.m));
				}
				if (c->two_level) site_mz[i] = mz;
				const uint32_t u = ntsm_kmer_sum(ntsm_code_top(x, plan.k), ntsm_code_top(rc, plan.k)), um = ntsm_kmer_mix(u);
				uint32_t *blk = &blocks[(size_t) ntsm_block_idx(mz, c->blk_map) * 4];
				__atomic_fetch_or(&blk[0], 1u << NTSM_KBIT0(u), __ATOMIC_RELAXED);
				__atomic_fetch_or(&blk[1], 1u << NTSM_KBIT1(um), __ATOMIC_RELAXED);
				__atomic_fetch_or(&blk[2], 1u << NTSM_KBIT2(um), __ATOMIC_RELAXED);
				__atomic_fetch_or(&blk[3], 1u << NTSM_KBIT3(um), __ATOMIC_RELAXED);
			}
			};
			{
				const uint32_t parts = n >= (1u << 18) ? 4u : 1u;
				std::vector<std::thread> pool;
				for (uint32_t t = 1; t < parts; ++t) pool.emplace_back(fill_blocks, (uint32_t) ((uint64_t) n * t / parts), (uint32_t) ((uint64_t) n * (t + 1) / parts));
				fill_blocks(0, (uint32_t) ((uint64_t) n / parts));
				for (auto &th : pool) th.join();
			}
			if (c->two_level) {
				/* Bloom over the DISTINCT site minimizers: one 32-bit word per minimizer, two bits; 12 bits per distinct
				 * minimizer, at most 2.25 MiB: it has to stay in the 4 MiB L2 beside the lines the block and bucket reads
				 * pull through it (16 M keys, 1.75 / 2 / 2.25 / 2.5 / 3 MiB: 376 / 391 / 394 / 393 / 379 Gbases/s); any
				 * word count will do, the index is a multiply-high range reduction */
				std::sort(site_mz.begin(), site_mz.end());
				site_mz.erase(std::unique(site_mz.begin(), site_mz.end()), site_mz.end());
				c->n_site_minimizers = (uint32_t) site_mz.size();
				const uint64_t want_w = std::max<uint64_t>(1024, std::min<uint64_t>(2304ull * 256, (12ull * site_mz.size() + 31) / 32));
				const uint32_t nw = c->bloom_words_req ? c->bloom_words_req : (uint32_t) ((want_w + 31) & ~31ull);
				c->n_bloom_words = nw;
				bloom.assign(nw, 0u);
				for (uint32_t mz : site_mz) {
					const uint32_t h = ntsm_block_hash(mz);
					bloom[ntsm_range(h, nw)] |= (1u << NTSM_BLOOM_BIT0(h)) | (1u << NTSM_BLOOM_BIT1(h));
				}
			}
			NTSM_ABL_BLOCKS_BUILT(blocks)
		}
	};
	auto build_prefilter = [&]() {
		/* second-level filter of the fast path: plain Bloom, 2 bits per key in one 32-bit word, >= 5 bits per key
		 * (1 MiB for the human set: with the 3 MiB first level it still fits the 4 MiB per-XCD L2) */
		if (plan.mode >= 0) {
			uint32_t pl = 10;
			/* The drain's Bloom pays while it sits in the L2 beside the first level (1 MiB for the human set).  A set that takes
			 * the two-level path is too big for that: its Bloom (16 MiB at 16 M keys) would be one more Infinity-Cache
			 * request per positive in front of the bucket read it is meant to save -- so it is left out (4 KiB, every bit
			 * set: always an L2 hit, always passes). */
			const bool pass_all = c->two_level && !c->prefilter_forced;
			while (!pass_all && pl < 28 && (1ull << pl) < 5ull * n) ++pl;
			if (c->prefilter_log2_req && !pass_all) pl = c->prefilter_log2_req;
			NTSM_ABL_PREFILTER_LOG2(pl)
			if (pl < 10) pl = 10;
			if (pl > 30) pl = 30;
			c->prefilter_log2 = pl;
			prefilter.assign((1ull << pl) / 32, pass_all ? 0xFFFFFFFFu : 0u);
			const uint32_t pshift = 32 - (pl - 5);
			for (uint32_t i = 0; i < n && !pass_all; ++i) {
				const uint32_t f = ntsm_fold(c->canon[i]), g1 = ntsm_h1(f), g2 = ntsm_h2(f);
				prefilter[g1 >> pshift] |= (1u << (g2 & 31u)) | (1u << ((g2 >> 5) & 31u));
			}
			NTSM_ABL_PREFILTER_BUILT(prefilter)
		}
	};
	auto build_tblocks = [&]() { build_tab_filter(c, filter_log2_req); };   /* no-op in the default build */
	/* Run-anchored kernel (kernels_run.hip; ntsm_set_kernel 5, k = 19): per site k-mer the signature of its ANCHORED 16-mer in
	 * the block of its minimizer.  With q bases to the left of the minimizer M (12-mer with the smallest order key, start offset
	 * q = 0 .. 7) the k-mer holds M + 4 bases right (q <= 3: class R, bases q .. q+15) or 4 bases left + M (q >= 4: class L,
	 * bases q-4 .. q+11).  The kernel's sliding minimum decides between equal 12-mers by their position in the lane's chunk, which
	 * the host cannot know: the signature is set for EVERY offset at which the minimum key occurs (a 12-mer repeated inside 19
	 * bases), so the filter has no false negatives.  Signature = the four block bits of the other kernels, taken from the
	 * strand-symmetric sum u = E + rc(E) of the 16-mer's code and its reverse complement's. */
	img.rblocks.clear();
	c->n_rblocks = 0;
	c->n_rentries = 0;
	auto build_rblocks = [&]() {
		if (!c->run_form) return;
		std::vector<uint32_t> hmin(n);
		std::vector<uint8_t> where(n);                        /* bit q: the minimum key occurs at start offset q */
		uint64_t entries = 0;
		for (uint32_t i = 0; i < n; ++i) {
			run_minimizer(c->canon[i], &hmin[i], &where[i]);
			entries += (uint64_t) __builtin_popcount(where[i]);
		}
		c->n_rentries = entries;
		/* 16 bits per site k-mer (about 29 per DISTINCT signature on a set of whole windows: the k-mers of a window that share a
		 * minimizer set the same bits) in steps of 64 KiB, at most 3 MiB -- the filter lives in the 4 MiB L2 beside the stream -- and
		 * 4 MiB beyond 5 M entries, where every bit saved costs more false look-ups than the L2 misses it avoids.  Measured,
		 * 2 / 2.5 / 3 / 4 MiB, Gbases/s: 2.5 M keys 824 / - / 832 / 763; 3.4 M 787 / 809 / 807 / 750; 4.2 M 749 / 775 /
		 * 777 / 735; 5.7 M 541 / 612 / 641 / 649; 8.3 M 363 / 446 / 503 / 562.  ntsm_set_tuning(2000000 + KiB) with ntsm_set_kernel(5)
		 * overrides */
		uint64_t kib = std::max<uint64_t>(64, std::min<uint64_t>(entries < 5000000ull ? 3072 : 4096, (2 * entries / 1024 + 63) / 64 * 64));
		if (c->blocks_kib_req && c->kernel_variant == 5) kib = c->blocks_kib_req;
		c->n_rblocks = kib * 64;
		img.rblocks.assign(c->n_rblocks * 4, 0u);
		for (uint32_t i = 0; i < n; ++i) {
			const uint64_t x = c->canon[i];
			uint32_t *blk = &img.rblocks[(size_t) ntsm_range(ntsm_block_hash(hmin[i]), (uint32_t) c->n_rblocks) * 4];
			for (uint32_t q = 0; q < 8; ++q) {
				if (!((where[i] >> q) & 1u)) continue;
				const uint32_t u = run_signature(x, q), um = ntsm_kmer_mix(u);
				blk[0] |= 1u << NTSM_KBIT0(u);
				blk[1] |= 1u << NTSM_KBIT1(um);
				blk[2] |= 1u << NTSM_KBIT2(um);
				blk[3] |= 1u << NTSM_KBIT3(um);
			}
		}
	};
	{
		std::thread t1(build_filter), t2(build_blocks), t3(build_prefilter), t4(build_tblocks), t5(build_rblocks);
		build_cuckoo();
		t1.join(); t2.join(); t3.join(); t4.join(); t5.join();
	}
	return cuckoo_rc;
}

} // namespace ntsm_rt
