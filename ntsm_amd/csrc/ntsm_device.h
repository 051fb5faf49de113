/*
 * ntsm_device.h -- data layout and hash functions shared by the host table builder and the
 * gfx950 kernels (kernels_generic.hip, kernels_mz.hip).
 *
 * HBM layout of one context:
 *   filter      : 2^F bits (uint32 words); bit f(x) set for every site k-mer x.  Sized to stay in
 *                 the 4 MiB per-XCD L2 (default F = 24 -> 2 MiB): >90 % of read k-mers end here.
 *   keys        : 2-choice cuckoo table, 2^B buckets of two uint64 slots (16 B, one dwordx4 load);
 *                 a site k-mer lives in bucket b1(x) or b2(x); empty slot = ~0.
 *   counters    : uint64 per slot, stored next to their keys (32-byte buckets {key0,key1,count0,count1}), bumped
 *                 with one no-return 64-bit atomic per hit.
 *   slot_of     : uint32 per site k-mer (dense index -> slot) for the final gather.
 * Keys are the reference's canonical codes (vendor/KseqHashIterator.hpp:99-104); bucket hashes
 * are free to choose because the reference's hash64 is a bijection (results depend only on set
 * membership of the canonical code).
 */
#ifndef NTSM_DEVICE_H
#define NTSM_DEVICE_H
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NTSM_DHD __host__ __device__ __forceinline__
#else
#define NTSM_DHD static inline
#endif

#define NTSM_EMPTY_KEY 0xFFFFFFFFFFFFFFFFULL

/* 32-bit mixes of a canonical code (2k <= 64 bits). */
NTSM_DHD uint32_t ntsm_fold(uint64_t x)
{
	return (uint32_t) x ^ ((uint32_t) (x >> 32) * 0x85EBCA6Bu);
}
NTSM_DHD uint32_t ntsm_h1(uint32_t folded) { return folded * 0x9E3779B1u; }
NTSM_DHD uint32_t ntsm_h2(uint32_t folded) { return (folded ^ 0x5BD1E995u) * 0xC2B2AE35u + 0x27D4EB2Fu; }

/* ---- k = 19 fast path: minimizer-blocked filter -----------------------------------------------
 * Consecutive k-mers of a read share their minimizer (smallest hashed canonical 12-mer inside the
 * 19-mer, 8 candidates) for 4.5 positions on average, so the filter is addressed by the MINIMIZER
 * (one 64-bit block per minimizer hash) and a lane re-reads L2 only when its minimizer changes:
 * ~0.22 L2 requests per k-mer instead of 1.  Inside the block every site k-mer sets two bits chosen
 * by a strand-symmetric hash of (fw, rc).  All functions below are shared by the host builder and
 * the kernel so that both sides agree bit for bit. */
#define NTSM_FAST_K 19
#ifndef NTSM_FAST_M
#define NTSM_FAST_M 12                 /* minimizer length: 12 (window of 8 m-mers) or 11 (window of 9) */
#endif
#define NTSM_FAST_W (NTSM_FAST_K - NTSM_FAST_M + 1)
#define NTSM_MMER_MASK ((1u << (2 * NTSM_FAST_M)) - 1u)
/* Large site sets (the blocked filter no longer fits the 4 MiB L2): k = 19 switches to 14-mer
 * minimizers (6 candidates) and puts a one-word Bloom filter over the DISTINCT SITE MINIMIZERS in front of the blocked
 * filter: a minimizer run first asks the L2-resident Bloom word, and only runs that pass (site minimizers + false
 * positives: 30 % at 16 M keys) go on to their 128-bit block in the Infinity Cache.  12-mers cannot do this: with 16 M
 * site k-mers 80 % of all read minimizers ARE site minimizers (tools/sim_two_level.cpp). */
#ifndef NTSM_TWO_M
#define NTSM_TWO_M 14
#endif
#define NTSM_TWO_W (NTSM_FAST_K - NTSM_TWO_M + 1)

/* The same filter for other k (13 <= k <= 31, k != 19): 12-mers throughout; what varies is which of them are minimizer
 * candidates.  The candidate set must map onto itself under reverse complement, i.e. be symmetric in the k-mer:
 *   k >= 19   the 8 (k odd) or 9 (k even) innermost 12-mers: offsets a .. a + w - 1 from either end,
 *             a = (k - 12 - (w - 1)) / 2  (k = 19: all 8, a = 0)
 *   k <  19   all k - 11 of them (2 .. 7)
 * so the sliding minimum runs over 2 .. 9 order hashes, computed `a` positions behind the newest base. */
struct NtsmFastPlan { int mode; uint32_t k, m, w, a; };      /* mode: -1 no fast path, 0 the k = 19 kernel, else w (the kernel's KMODE) */
NTSM_DHD NtsmFastPlan ntsm_fast_plan(uint32_t k, bool two_level = false)
{
	NtsmFastPlan pl = { -1, k, 0u, 0u, 0u };
	if (two_level && k >= NTSM_TWO_M + 1u && k <= 31u) {
		/* two-level form: the same rules with 14-mers -- all k - 13 candidates (2 .. 7) up to k = 20, the 8 / 9 innermost beyond */
		pl.m = NTSM_TWO_M;
		if (k <= NTSM_TWO_M + 6u) pl.w = k - (NTSM_TWO_M - 1u);
		else { pl.w = (k & 1u) ? 8u : 9u; pl.a = (k - NTSM_TWO_M - (pl.w - 1u)) / 2u; }
		pl.mode = k == NTSM_FAST_K ? 0 : (int) pl.w;
		return pl;
	}
	if (k == NTSM_FAST_K) { pl.mode = 0; pl.m = NTSM_FAST_M; pl.w = NTSM_FAST_W; }
	else if (k >= 19 && k <= 31) { pl.m = 12; pl.w = (k & 1u) ? 8u : 9u; pl.a = (k - 12u - (pl.w - 1u)) / 2u; pl.mode = (int) pl.w; }
	else if (k >= 13 && k < 19) { pl.m = 12; pl.w = k - 11u; pl.mode = (int) pl.w; }
	return pl;
}
/* the two 32-bit words whose sum picks the filter bits: the k-mer's forward and reverse-complement codes, left-aligned
 * (k >= 16: their top 32 bits = first 16 bases of either strand; together they cover all k bases for k <= 32) */
NTSM_DHD uint32_t ntsm_code_top(unsigned long long code, uint32_t k)
{
	return k >= 16 ? (uint32_t) (code >> (2 * k - 32)) : (uint32_t) (code << (32 - 2 * k));
}

/* Order hash of a 12-mer: its canonical code (the smaller of the 24-bit forward and reverse-complement codes) times an
 * odd constant, low 32 bits of the 48-bit product (v_mul_u32_u24), compared as an integer.  Cheaper strand-symmetric
 * combiners were measured and lose (tools/sim_mmer_order.cpp, tools/filter_fp.cpp on the hs_n10_like set):
 *   fw + rc, fw ^ rc   depend only on the six differences of mirrored bases: 117,649 values, 3/4 of the filter blocks
 *                      stay empty, first-level pass rate 63 %;
 *   fw * rc            two instructions fewer, but minimizer density 0.229 instead of 0.225 and pass rate 1.29 %
 *                      instead of 1.09 %: 831 instead of 845 Gbases/s -- the kernel is bound by L2 requests, not VALU. */
NTSM_DHD uint32_t ntsm_mmer_hash(uint32_t canon)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return (uint32_t) __umul24(canon, 0x9E3779u);       /* __umul24 is declared int in HIP */
#else
	return (uint32_t) ((uint64_t) (canon & 0xFFFFFFu) * 0x9E3779u);
#endif
}
/* the same for m-mers of more than 12 bases (canonical code up to 32 bits): full 32-bit multiply, a bijection */
NTSM_DHD uint32_t ntsm_mmer_hash_wide(uint32_t canon) { return canon * 0x9E3779B1u; }
NTSM_DHD uint32_t ntsm_mmer_hash_m(uint32_t canon, uint32_t m) { return m <= 12u ? ntsm_mmer_hash(canon) : ntsm_mmer_hash_wide(canon); }
/* Index of the 128-bit filter block of a minimizer value: h = mz * odd constant (full 32-bit multiply: the top bits of
 * a minimum are small; the multiply carries every bit of it into the top bits), then the multiply-high range reduction of h onto n_blocks, power of two or
 * not.  Two instructions (v_mul_lo_u32, v_mul_hi_u32); the kernel hands the index to a buffer load whose descriptor
 * has a 16-byte stride (idxen), so the byte offset costs nothing. */
struct NtsmBlockMap { uint32_t n_blocks; };
NTSM_DHD uint32_t ntsm_block_hash(uint32_t mz) { return mz * 0x9E3779B1u; }
NTSM_DHD uint32_t ntsm_range(uint32_t h, uint32_t n)       /* multiply-high range reduction of a 32-bit hash onto [0, n) */
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __umulhi(h, n);
#else
	return (uint32_t) (((uint64_t) h * n) >> 32);
#endif
}
NTSM_DHD uint32_t ntsm_block_idx(uint32_t mz, NtsmBlockMap m) { return ntsm_range(ntsm_block_hash(mz), m.n_blocks); }
/* Two-level path: the Bloom word of a minimizer is ntsm_range(h, n_words) with the same h = ntsm_block_hash(mz) that picks
 * its block; the two bits it sets / tests in that word are 31 - (h & 31) and 31 - ((h >> 8) & 31) -- byte-aligned fields
 * of the LOW half of h (the word index uses the top bits), taken by the kernel with byte selects and tested in the sign
 * position like the block bits.  Measured on the 1 M-site set: 0.087 second-level requests per k-mer with a 3 MiB
 * Bloom, against 0.078 for bits from a second multiply and 0.229 without the Bloom (tools/sim_two_level.cpp). */
#define NTSM_BLOOM_BIT0(h) (31u - ((h) & 31u))
#define NTSM_BLOOM_BIT1(h) (31u - (((h) >> 8) & 31u))
/* Four filter bits per site k-mer, one in each 32-bit word of its 128-bit block.  u = fh + rh where fh / rh are the
 * top 32 bits of the 38-bit forward and reverse-complement codes (code >> 6): together they cover all 19 bases
 * and the sum is symmetric in the two strands (no canonical min in the hot loop); um = u * odd constant spreads
 * single-base differences over all bits.  The four 5-bit positions are byte-aligned fields (u byte 3, um bytes 3,
 * 2, 1) so that the shifts can take them with a byte select.  Measured false-positive rate on the hs_n10_like
 * set with a 3 MiB filter: 1.1 % (two bits in a 64-bit block: 2.6 %) -- read minimizers and site minimizers
 * favour the same m-mers, so the blocks that queries hit are the loaded ones and wide blocks pay off. */
NTSM_DHD uint32_t ntsm_kmer_sum(uint32_t fh, uint32_t rh) { return fh + rh; }
NTSM_DHD uint32_t ntsm_kmer_mix(uint32_t u) { return u * 0x9E3779B1u; }
/* bit index = 31 - field: the kernel shifts the word LEFT by the field (a byte select, the hardware takes its low five
 * bits) and reads the sign bit, so the four tests and the validity of the window AND together without a final mask */
#define NTSM_KBIT0(u) (31u - (((u) >> 24) & 31u))
#define NTSM_KBIT1(um) (31u - (((um) >> 24) & 31u))
#define NTSM_KBIT2(um) (31u - (((um) >> 16) & 31u))
#define NTSM_KBIT3(um) (31u - (((um) >> 8) & 31u))


/* ---- k = 19 run-anchored path (kernels_run.hip) -------------------------------------------------
 * Order key of a 12-mer: a 24-bit order hash of the 12-mer, the same for both strands, in bits 8..31, the 12-mer's position
 * mod 16 in the low bits (the kernel's sliding minimum thereby also says WHERE the minimizer sits; the low bits only decide
 * between 12-mers of equal hash inside one window).  Two forms of the hash, chosen at build time:
 *   NTSM_RUN_ORDER 0  (canonical code * odd) mod 2^24: a bijection of the 24-bit canonical code, different 12-mers never tie;
 *                     five vector instructions per position (shift, mask, min, multiply, shift-or);
 *   NTSM_RUN_ORDER 1  bits 8..31 of the 48-bit product of the two strands' codes: symmetric by commutativity, no canonical
 *                     minimum -- three instructions (shift, v_mul_u32_u24, v_and_or_b32).  Not a bijection: different
 *                     12-mers tie with probability 2^-24 per pair, which costs nothing -- the host sets the signature for
 *                     EVERY offset at which the smallest hash occurs in a site k-mer, whatever 12-mer sits there. */
#ifndef NTSM_RUN_ORDER
#define NTSM_RUN_ORDER 0               /* measured, same box, form 1 against form 0: 35.8 against 38.6 vector instructions per position and minimizer
                                        * density 0.2284 against 0.2250; but the product's small values spread less evenly over the filter blocks (mean
                                        * false-positive rate of a signature test 0.00156 / 0.0072 / 0.0197 against 0.00131 / 0.0059 / 0.0164 at 2.5 / 5.7 /
                                        * 8.3 M keys), so it wins 1 % up to 2.5 M keys (840 against 832 Gbases/s) and loses 2 - 8 % from 4 M on
                                        * (4.2 M 764 / 777, 5.7 M 622 / 649, 8.3 M 518 / 562): form 0 stays */
#endif
NTSM_DHD uint32_t ntsm_run_hash24(uint32_t canon)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return (uint32_t) __umul24(canon, 0x9E3779u);       /* the whole 32-bit product: the kernel's shift by 8 drops the top byte */
#else
	return (uint32_t) (((uint64_t) (canon & 0xFFFFFFu) * 0x9E3779u) & 0xFFFFFFu);
#endif
}
/* host side of both forms: the 24-bit order hash of the 12-mer with forward code f12 and reverse-complement code r12 */
static inline uint32_t ntsm_run_order24_host(uint32_t f12, uint32_t r12)
{
	f12 &= 0xFFFFFFu; r12 &= 0xFFFFFFu;
#if NTSM_RUN_ORDER == 1
	return (uint32_t) ((((uint64_t) f12 * r12) >> 8) & 0xFFFFFFu);
#else
	return (uint32_t) (((uint64_t) (f12 < r12 ? f12 : r12) * 0x9E3779u) & 0xFFFFFFu);
#endif
}
/* reverse complement of a 16-base word (first base in the top bits) */
NTSM_DHD uint32_t ntsm_rc16_word(uint32_t w)
{
	uint32_t y = ~w;
	y = ((y >> 2) & 0x33333333u) | ((y & 0x33333333u) << 2);
	y = ((y >> 4) & 0x0F0F0F0Fu) | ((y & 0x0F0F0F0Fu) << 4);
	y = ((y >> 8) & 0x00FF00FFu) | ((y & 0x00FF00FFu) << 8);
	return (y >> 16) | (y << 16);
}

/* ---- k = 19 tabulated path ("tab" kernel) ------------------------------------
 * Measured on gfx950 (profiles/r02_valu_rate.txt): every wave64 VALU instruction of a mixed stream occupies its
 * SIMD for ~4.2 cycles whatever the opcode and whatever the occupancy, so the count kernel's speed is its VALU
 * instruction count.  This path therefore computes nothing per base that a table can supply:
 *   - the stream is 2-bit packed while it is staged (16 bases per word, first base in the low bits);
 *   - a 256-entry LDS table indexed by the 4-mer ending at position p returns { U, V, M, - };
 *   - order key of the 12-mer ending at i:  K(i) = U[e4(i-8)] + M[e4(i-4)] + V[e4(i)]   (one v_add3_u32);
 *   - bit-selection hash of the 19-mer ending at i:  h(i) = V[e4(i)] + U[e4(i-15)]       (one v_add_u32).
 * Strand symmetry needs no reverse complement at run time: with V[y] = U[rc(y)] and M[y] = M[rc(y)] both sums are
 * invariant under (a, b, c) -> (rc c, rc b, rc a) because addition commutes (XOR would map every palindromic
 * 12-mer to 0: measured 456 site k-mers in one block instead of 64).  The minimizer is the smallest K among the 8
 * 12-mers of the 19-mer; its LOW bits (uniform, independent of the order statistic) address a 128-bit block; the
 * four bytes of h give one bit position in each 32-bit word.  False-positive rate on hs_n10_like at 3 MiB: 1.44 %
 * (tools/sim_tab_filter.py; ideal random hashes of the canonical codes: 1.18 %). */
struct NtsmTabEntry { uint32_t u, v, m, pad; };
#define NTSM_TAB_SEED 0x6A09E667F3BCC909ULL

NTSM_DHD uint64_t ntsm_splitmix(uint64_t *s)
{
	uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
/* reverse complement of a 4-mer index (first base in bits 0-1) */
NTSM_DHD uint32_t ntsm_rc4(uint32_t y)
{
	const uint32_t c = ~y & 0xFFu;
	return ((c & 3u) << 6) | ((c & 0xCu) << 2) | ((c >> 2) & 0xCu) | (c >> 6);
}
static inline void ntsm_tab_build(struct NtsmTabEntry *t /* [256] */)
{
	uint32_t a[256], c[256], b[256];
	uint64_t s = NTSM_TAB_SEED;
	for (int i = 0; i < 256; ++i) { a[i] = (uint32_t) ntsm_splitmix(&s); c[i] = (uint32_t) ntsm_splitmix(&s); b[i] = (uint32_t) ntsm_splitmix(&s); }
	for (uint32_t y = 0; y < 256; ++y) {
		const uint32_t r = ntsm_rc4(y);
		t[y].u = a[y] ^ c[r];
		t[y].v = c[y] ^ a[r];                 /* = u[rc y] */
		t[y].m = b[y] + b[r];                 /* = m[rc y] */
		t[y].pad = 0;
	}
}
/* order key of the minimizer and bit-selection hash of a 19-mer given as its canonical (or any strand's) big-endian
 * 2-bit code: what the kernel computes from the packed stream, restated on the code for the table builder */
static inline void ntsm_tab_kmer(const struct NtsmTabEntry *t, uint64_t code, uint32_t *mz_out, uint32_t *h_out)
{
	uint32_t e4[19];                          /* e4[p]: index of the 4-mer ending at base p (p >= 3) */
	for (int p = 3; p < 19; ++p) {
		uint32_t y = 0;
		for (int q = 0; q < 4; ++q) y |= (uint32_t) ((code >> (2 * (18 - (p - 3 + q)))) & 3u) << (2 * q);
		e4[p] = y;
	}
	uint32_t mz = 0xFFFFFFFFu;
	for (int p = 11; p < 19; ++p) {
		const uint32_t k = t[e4[p - 8]].u + t[e4[p - 4]].m + t[e4[p]].v;
		if (k < mz) mz = k;
	}
	*mz_out = mz;
	*h_out = t[e4[18]].v + t[e4[3]].u;
}
/* byte offset of the 128-bit block of a minimizer key.  n_blocks = mult * 2^e, mult in {1, 3}.  Only LOW bits of the
 * key are used: they are uniform and independent of the order statistic (the top bits of a minimum are not).
 *   mult = 1: bits 4 .. e+3 of the key in place                      (one v_and_b32; mask = (2^e - 1) << 4)
 *   mult = 3: ((low 24 bits * 3) >> (20 - e)) & ~15, e <= 20          (v_mul_u32_u24, v_lshrrev_b32, v_and_b32) */
struct NtsmTabMap { uint32_t mask, shift, mult3; };
NTSM_DHD uint32_t ntsm_tab_block_off(uint32_t mz, NtsmTabMap m)
{
	if (!m.mult3) return mz & m.mask;
#if defined(__HIP_DEVICE_COMPILE__)
	return ((uint32_t) __umul24(mz, 3u) >> m.shift) & ~15u;
#else
	return (((mz & 0xFFFFFFu) * 3u) >> m.shift) & ~15u;
#endif
}
/* bit tested in word w of the block: (31 - ((h >> 8w) & 31)); the kernel shifts the word LEFT by (h >> 8w) & 31 and
 * reads the sign bit */
NTSM_DHD uint32_t ntsm_tab_bit(uint32_t h, int w) { return 31u - ((h >> (8 * w)) & 31u); }

struct NtsmCountParams {
	const uint8_t *base;               /* 16-byte aligned start of the flat stream */
	long long lo, hi;                  /* count windows ending at byte offsets in [lo, hi) */
	long long t0;                      /* tile origin (multiple of 16, <= lo) */
	unsigned long long n_tiles;
	const uint32_t *filter;
	const uint64_t *keys;
	unsigned long long *totals;        /* [0] k-mers, [1] hits */
	const unsigned long long *read_end;/* per-read attribution (early-stop mode only) */
	uint32_t *read_hits;
	unsigned long long n_reads;
	unsigned long long sign;           /* +1 or 2^64-1 */
	unsigned long long mask;           /* (1 << 2k) - 1 */
	uint32_t k, rv_shift;              /* 2(k-1) */
	uint32_t kmask;                    /* low k bits set: window validity */
	uint32_t fshift, bshift;           /* bit index = h1 >> fshift ; bucket = h >> bshift */
	const uint8_t *lut;                /* 256-byte base table, vendor/KseqHashIterator.hpp:114-127 */
	const uint2 *lut64;                /* fast path: per byte { code, (3 - code) | valid << 16 } */
	const uint4 *blocks;               /* k = 19 fast path: minimizer-addressed 128-bit filter blocks */
	NtsmBlockMap blk_map;              /* minimizer -> filter block offset */
	const uint32_t *bloom;             /* two-level path: one-word Bloom over the distinct site minimizers (L2 resident) */
	uint32_t bloom_words;
	const uint32_t *prefilter;         /* fast path, drain only: plain 2-bit Bloom over canonical codes (L2 resident) */
	uint32_t pf_shift;                 /* word index = h1(fold) >> pf_shift; bits = h2(fold) & 31, (h2 >> 5) & 31 */
	uint32_t debug;                    /* ablation switches (NTSM_DEBUG_KERNEL): 1 = drain discards its queue, 2 = drain stops after the k-mer rebuild */
	uint32_t blk_bytes;                /* size of the filter in bytes (buffer descriptor range) */
	uint32_t fk_k, fk_m2, fk_a2;       /* general-k fast path: k, 2 * minimizer length, 2 * candidate offset (NtsmFastPlan) */
	/* tabulated k = 19 path */
	const NtsmTabEntry *tab;           /* 256 entries, copied to LDS by every workgroup */
	const uint4 *tblocks;              /* its minimizer-addressed 128-bit filter blocks */
	uint32_t tblk_bytes;
	NtsmTabMap tblk_map;
	uint32_t *exotic_list;             /* tab kernel: appends the 32 KiB tiles it skipped (bytes outside ACGTUNacgtun) ... */
	uint32_t *exotic_count;            /* ... and the list's fill; the k19 kernel launched behind it counts exactly those tiles */
	uint32_t exotic_cap;
	uint32_t *exotic_seen;             /* statistics: tiles handed over since the context was created */
	uint32_t use_list;                 /* k19 kernel: walk exotic_list instead of all tiles */
	/* tab kernel -> look-up kernel: canonical codes of the windows that passed the first-level filter */
	unsigned long long *pos_queue;     /* [pos_cap] */
	uint32_t *pos_count;               /* entries reserved in this segment's queue (may exceed pos_cap: the rest was looked up in line) */
	uint32_t pos_cap;
	unsigned long long tile_base;      /* tab kernel: index of this segment's first 64 KiB tile within the launch (exotic list entries are global) */
};

#endif
