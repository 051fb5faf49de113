/*
 * kernels_mz.hip -- the minimizer-blocked count kernels (13 <= k <= 31; k = 19 is the reference default and every
 * BASELINE configuration) and their launcher.  DESIGN.md sections 4.2 / 4.2b.
 *
 * Same job as kernels_generic.hip -- the loop of FingerPrint::insertCount (src/FingerPrint.hpp:89-103: rolling canonical
 * k-mer of vendor/KseqHashIterator.hpp:95-112, tsl::robin_map find, `+= 1`) over a flat stream -- with the membership test
 * organised around the k-mer's minimizer so that a lane asks the L2 for a filter block once per minimizer run instead of
 * once per k-mer.
 */
#include "kernels_common.h"
#include "ntsm_internal.h"

namespace {

/* --------------------------------------------------------------------------------------------
 * Minimizer-blocked fast path: k = 19 with every constant folded, and 13 <= k <= 31
 * with k as a run-time parameter (ntsm_fast_plan, ntsm_device.h).  Same tiling as the generic kernel; per
 * position a lane
 *   1. rolls the forward / reverse-complement words, the run of valid bases and the canonical m-mer order hash,
 *   2. keeps the sliding minimum over the 8 (9) candidate m-mers of the k-mer (block-decomposed: prefix minima
 *      of the current 8-block against suffix minima of the previous one),
 *   3. re-reads its 128-bit filter block from L2 only when the minimizer changed,
 *   4. tests four block bits chosen by a strand-symmetric hash; positives are queued as { forward word of the
 *      first 16 bases, reverse word of the last 16 } in a wave-private LDS queue.
 * Whenever 64 positives are queued the wave drains them with every lane busy: the canonical code is rebuilt
 * from the two words, tested against the second-level filter, looked up in the cuckoo table (bucket 2 only if
 * bucket 1 is full), and the slot counter bumped with one 64-bit atomic.
 * ------------------------------------------------------------------------------------------ */
#ifndef NTSM_FAST_WAVES
#define NTSM_FAST_WAVES 4                              /* waves per SIMD the register budget is held to */
#endif
/* forward word update in one v_lshl_or_b32 (hipcc emits shift + or for the C expression: +1 %) */
#define NTSM_F_UPDATE(c_) asm("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(F) : "v"(F), "v"(c_));
#ifndef NTSM_FAST_C
#define NTSM_FAST_C 128                                /* stream bytes per thread and tile of the minimizer-blocked kernels (128 or 96) */
#endif
#ifndef NTSM_STEP_POS
#define NTSM_STEP_POS 8                                /* positions between issuing the filter-block loads and testing them (8 or 4) */
#endif
#ifndef NTSM_TWO_STEP_POS
#define NTSM_TWO_STEP_POS 4                            /* the same for the two-level form (one more load level in flight per step) */
#endif
constexpr int kFastC = NTSM_FAST_C;
[[maybe_unused]] constexpr int kListC = 128;                            /* list mode (tiles handed over by the tabulated kernel, make tab): always 32 KiB tiles */
constexpr int kQueueCap = 128;                         /* < 64 left over + one position's burst of <= 64 */

/* LDS image of a tile: row r (C bytes) = stream bytes of thread r-1 (row 0 = the 32 bytes in front of the tile, in its
 * last two slots).  The 16-byte slots of a row are permuted per row so that the per-thread ds_read_b64 of "slot s of my
 * row" spreads over the banks without padding: C = 128: slot s sits at s ^ ((r >> 1) & 7) (conflict free); other C:
 * rotated by r >> 3 (two-way). */
template <int C>
__device__ __forceinline__ int ntsm_tile_addr(int row, int byte_in_row)
{
	if (C == 128) return row * C + ((((byte_in_row >> 4) ^ (row >> 1)) & 7) << 4) + (byte_in_row & 15);
	return row * C + (int) ((((uint32_t) (byte_in_row >> 4) + ((uint32_t) row >> 3)) % (uint32_t) (C / 16)) << 4) + (byte_in_row & 15);
}

/* KMODE 0: k = 19 with every constant folded (the reference default and all BASELINE configurations).
 * KMODE 2 .. 9: any other k of ntsm_fast_plan(), KMODE = number of minimizer candidates; k, the minimizer length and
 * the candidate offset are run-time parameters, the rolling words are 64 bits wide (two registers each).
 * TWO (large site sets, 15 <= k <= 31): 14-mer minimizers and a Bloom word over the distinct site minimizers in front of
 * the block -- phase B between A and C: a run's block is only requested when its Bloom word passes (ntsm_device.h). */
template <int KMODE, bool PER_READ, int C, bool TWO>
__global__ __launch_bounds__(kThreads, PER_READ ? 3 : NTSM_FAST_WAVES) void ntsm_count_mz_kernel(const NtsmCountParams p)
{
	constexpr int VPT = C / 16, NB = C / 8, HB = TWO ? NTSM_TWO_STEP_POS : NTSM_STEP_POS;
	constexpr bool GEN = KMODE != 0;
	constexpr int MM = TWO ? NTSM_TWO_M : NTSM_FAST_M;                     /* minimizer length of the k = 19 kernels */
	constexpr int W = KMODE == 0 ? NTSM_FAST_K - MM + 1 : KMODE;
	static_assert(W >= 2 && W <= 9, "sliding minimum: 2 .. 9 candidates");
	const uint32_t gk = GEN ? p.fk_k : (uint32_t) NTSM_FAST_K;            /* wave-uniform run-time k of the general kernels */
	const uint32_t g_a2 = p.fk_a2, g_mmask = (1u << p.fk_m2) - 1u, g_rsh = 64u - p.fk_m2 - p.fk_a2, g_fsh = 64u - 2u * gk;
	const uint32_t g_rmask = gk >= 16 ? 0xFFFFFFFFu : 0xFFFFFFFFu << (32u - 2u * gk);
	__shared__ __attribute__((aligned(16))) uint8_t tile[(kThreads + 1) * C];
	__shared__ uint2 lut64[256];
	__shared__ uint2 queue_all[kThreads / 64][kQueueCap];          /* positives: { first-16 forward word, last-16 reverse word } */
	__shared__ uint16_t qpos_all[PER_READ ? kThreads / 64 : 1][PER_READ ? kQueueCap : 1];   /* -m mode: their tile offsets */
	const int t = threadIdx.x;
	const int lane = t & 63;
	uint2 *queue = queue_all[t >> 6];
	uint16_t *qpos = qpos_all[PER_READ ? (t >> 6) : 0];
	lut64[t] = p.lut64[t];
	const uint32_t bshift = p.bshift;
	const NtsmBlockMap blk_map = p.blk_map;
	/* Buffer resource over the filter blocks, 16-byte stride: the load takes a block INDEX (idxen), the address
	 * arithmetic and the range check (index >= number of blocks: returns 0, no memory request) are the hardware's. */
	const unsigned long long blk_base = (unsigned long long) p.blocks;
	const ntsm_i32x4 blk_rsrc = { (int) (uint32_t) blk_base, (int) ((uint32_t) (blk_base >> 32) | (16u << 16)),
			(int) (p.blk_bytes >> 4), 0x00020000 };
	/* two-level path: the Bloom words, 4-byte stride, same addressing */
	const unsigned long long blm_base = (unsigned long long) p.bloom;
	const ntsm_i32x4 blm_rsrc = { (int) (uint32_t) blm_base, (int) ((uint32_t) (blm_base >> 32) | (4u << 16)),
			(int) p.bloom_words, 0x00020000 };
	const uint32_t bloom_words = p.bloom_words;
	uint32_t nk_s = 0, nh = 0;                           /* nk_s: wave-uniform (scalar) count of valid windows */

	/* list mode (kWithTab builds only: folded away otherwise): only the tiles the tabulated kernel handed over (tiles with
	 * bytes outside ACGTUNacgtun) */
	const unsigned long long n_iter = (kWithTab && p.use_list) ? (unsigned long long) min(*p.exotic_count, p.exotic_cap) : p.n_tiles;
	for (unsigned long long it = blockIdx.x; it < n_iter; it += gridDim.x) {
		const unsigned long long ti = (kWithTab && p.use_list) ? (unsigned long long) p.exotic_list[it] : it;
		if (kWithTab && ti >= p.n_tiles) continue;
		const long long ts = p.t0 + (long long) (ti * (unsigned long long) (kThreads * C));
		__syncthreads();
		if (ts >= p.lo && ts + kThreads * C <= p.hi) {
			/* interior tile: plain coalesced loads (the boundary logic of ntsm_load_vec costs ~100 VALU instructions per
			 * vector, 6.5 per base position -- a sixth of this kernel's instruction count when it ran for every tile) */
#ifdef NTSM_STREAM_AUX
			/* the tile through a buffer descriptor of its own (scalar base, 32-bit lane offsets, cache-policy bits in the
			 * instruction): nt keeps the read-once stream from displacing the filter in the L2 -- without it the kernel runs 8 %
			 * slower (830 against 904 Gbases/s), and the buffer form is 0.8 % faster than a non-temporal global load (911) */
			const __amdgpu_buffer_rsrc_t st_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.base + ts), (short) 0, kThreads * C, 0x00020000);
#endif
#pragma unroll
			for (int q = 0; q < VPT; ++q) {
				const int v = t + kThreads * q;
#ifdef NTSM_STREAM_AUX
				const ntsm_u32x4 nt = __builtin_amdgcn_raw_buffer_load_b128(st_rsrc, 16 * v, 0, NTSM_STREAM_AUX);
#elif NTSM_STREAM_NT
				const u32x4 nt = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p.base + ts + 16ll * v));
#else
				const u32x4 nt = *reinterpret_cast<const u32x4 *>(p.base + ts + 16ll * v);
#endif
				*reinterpret_cast<uint4 *>(tile + ntsm_tile_addr<C>(1 + v / VPT, (v % VPT) * 16)) = make_uint4(nt.x, nt.y, nt.z, nt.w);
			}
		} else {
#pragma unroll 1
			for (int q = 0; q < VPT; ++q) {
				const int v = t + kThreads * q;
				const uint4 r = ntsm_load_vec(p, ts + 16ll * v);
				*reinterpret_cast<uint4 *>(tile + ntsm_tile_addr<C>(1 + v / VPT, (v % VPT) * 16)) = r;
			}
		}
		if (t < 2) {
			const uint4 r = ntsm_load_vec(p, ts - 32 + 16 * t);
			*reinterpret_cast<uint4 *>(tile + ntsm_tile_addr<C>(0, C - 32 + 16 * t)) = r;
		}
		__syncthreads();

		/* Rolling state: F = 2-bit codes of the last 16 bases (newest lowest), R = reverse complement
		 * of the last 16 bases (complement of the newest base on top), run = 1 + number of valid bases
		 * since the last invalid one (run * valid + 1: the table's second word carries the complement code
		 * in its low bits and valid (0/1) in its high half, v_mad_u32_u16 takes that half).  One op each
		 * per base; the 19-mer's two strands are covered by F three positions ago (its first 16 bases) and
		 * the current R (its last 16, reversed); the window is valid when run > 19. */
		uint32_t F = 0, R = 0, run = 1;
		uint32_t Fh = 0, Ro = 0;                            /* general k: bases 17..32 back of the forward word, of the reverse word */
		uint32_t sprev[8];                                  /* W >= 8: suffix minima of the previous 8-block, [1..7] used */
		uint32_t gprev[8], m2prev[8];                       /* W < 8: order hashes of the previous 8-block ([2..7] used) and their pair minima ([4..7]) */
		uint32_t fc0 = 0, fc1 = 0, fc2 = 0;                 /* k = 19: F at the three positions before the current block */
		uint32_t qn = 0;                                    /* wave-uniform queue fill */
#define NTSM_STEP(e_)                                                                     \
		{                                                                                 \
			if (GEN) {                                                                    \
				Fh = __builtin_amdgcn_alignbit(Fh, F, 30);                                \
				Ro = __builtin_amdgcn_alignbit(R, Ro, 2);                                 \
			}                                                                             \
			NTSM_F_UPDATE((e_).x)                                                         \
			R = __builtin_amdgcn_alignbit((e_).y, R, 2);                                  \
			asm("v_mad_u32_u16 %0, %1, %2, 1 op_sel:[0,1,0,0]" : "=v"(run) : "v"(run), "v"((e_).y)); \
		}
		/* order hash of the newest candidate m-mer: the one ending `a` bases behind the newest base */
		auto mmer_g = [&]() -> uint32_t {
			if (!GEN) {
				const uint32_t cm = min(F & ((1u << (2 * MM)) - 1u), R >> (32 - 2 * MM));
				return ntsm_mmer_hash_m(cm, (uint32_t) MM);            /* folded: 12-mers one 24-bit multiply, 14-mers a full one */
			}
			const uint32_t fm = __builtin_amdgcn_alignbit(Fh, F, g_a2) & g_mmask;
			const uint32_t rm = (uint32_t) (((((unsigned long long) R) << 32) | Ro) >> g_rsh) & g_mmask;
			return ntsm_mmer_hash_m(min(fm, rm), (uint32_t) MM);       /* 14-mers / 12-mers */
		};
		/* forward word of the window's first 16 bases (k < 16: its code, left-aligned) */
		auto f_top = [&]() -> uint32_t { return (uint32_t) ((((((unsigned long long) Fh) << 32) | F) << g_fsh) >> 32); };
#define NTSM_MMER_G() mmer_g()
		{   /* warm-up: the k - 1 bytes in front of the chunk (general k: all 32 of the prefix row, the run counter takes care of
		     * what lies before a window); order hashes of the last W - 1 positions */
			const uint4 v0 = *reinterpret_cast<const uint4 *>(tile + ntsm_tile_addr<C>(t, C - 32));
			const uint4 v1 = *reinterpret_cast<const uint4 *>(tile + ntsm_tile_addr<C>(t, C - 16));
			const uint32_t w[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
			uint32_t gw[8], fh[32];
#pragma unroll
			for (int i = GEN ? 0 : 32 - (NTSM_FAST_K - 1); i < 32; ++i) {
				const uint2 e = lut64[(w[i >> 2] >> ((i & 3) * 8)) & 0xFFu];
				NTSM_STEP(e)
				fh[i] = F;
				if (i >= 24 + (W >= 8 ? 9 - W : 1)) gw[i - 24] = NTSM_MMER_G();   /* the last positions of the previous 8-block */
			}
			fc0 = fh[29]; fc1 = fh[30]; fc2 = fh[31];
			if (W >= 8) {
				sprev[7] = gw[7];
#pragma unroll
				for (int i = 6; i >= 9 - W; --i) sprev[i] = min(gw[i], sprev[i + 1]);
			} else {
#pragma unroll
				for (int i = 1; i < 8; ++i) gprev[i] = gw[i];
#pragma unroll
				for (int i = 2; i < 8; ++i) m2prev[i] = min(gw[i], gw[i - 1]);
			}
		}
		uint32_t mz_prev = 0;
		uint4 cur = make_uint4(0, 0, 0, 0);                  /* the lane's cached 128-bit filter block */
		unsigned long long bad_prev = ~0ull;                /* nothing cached at the start of a chunk */

		/* Drain: look up queued positives 64 at a time, as a three-stage pipeline spread over consecutive
		 * calls so that no load is consumed in the call that issued it (the wave goes back to the main
		 * loop while its second-level-filter word, then its key bucket, are on their way):
		 *   stage 1  pop 64 entries, rebuild the canonical code, issue the second-level filter load
		 *   stage 2  (next call) test the filter word, issue the bucket load for the survivors
		 *   stage 3  (call after) compare the bucket, rare second bucket inline, bump the counter
		 * `flush` pushes everything through at the end of a tile. */
		uint32_t s1_klo = 0, s1_khi = 0, s1_g1 = 0, s1_g2 = 0, s1_pw = 0, s1_pos = 0;
		uint32_t s2_klo = 0, s2_khi = 0, s2_g2 = 0, s2_pos = 0;
		unsigned long long s2_b1 = 0;
		uint4 s2_ba = make_uint4(0, 0, 0, 0);
		bool s1_v = false, s2_v = false;
		auto drain_step = [&](bool take) {
			/* stage 3 */
			long long slot_of_hit = -1;
			if (s2_v) {
				long long slot = -1;
				if (s2_ba.x == s2_klo && s2_ba.y == s2_khi) slot = (long long) s2_b1;
				else if (s2_ba.z == s2_klo && s2_ba.w == s2_khi) slot = (long long) s2_b1 + 1;
				else if ((s2_ba.x & s2_ba.y) != 0xFFFFFFFFu && (s2_ba.z & s2_ba.w) != 0xFFFFFFFFu) {
					/* bucket 1 full and no match: the key can only be in bucket 2 */
					const unsigned long long b2 = 2ull * (s2_g2 >> bshift);
					const uint4 bb = *reinterpret_cast<const uint4 *>(p.keys + 2ull * b2);
					if (bb.x == s2_klo && bb.y == s2_khi) slot = (long long) b2;
					else if (bb.z == s2_klo && bb.w == s2_khi) slot = (long long) b2 + 1;
				}
				if (slot >= 0) {
					++nh;
					if (PER_READ) atomicAdd(p.read_hits + ntsm_read_of(p, (unsigned long long) (ts + (long long) s2_pos)), 1u);
				}
				slot_of_hit = slot;
			}
			NTSM_ABL_UNLESS_NO_ATOMICS(p)
			ntsm_add_hits(p, slot_of_hit, lane);                   /* equal slots inside the wave are added up first */
			/* stage 2 */
			s2_v = s1_v && (((s1_pw >> (s1_g2 & 31u)) & (s1_pw >> ((s1_g2 >> 5) & 31u)) & 1u) != 0);
			NTSM_ABL_STAGE2(s2_v, p)
			if (s2_v) {
				s2_klo = s1_klo; s2_khi = s1_khi; s2_g2 = s1_g2; s2_pos = s1_pos;
				s2_b1 = 2ull * (s1_g1 >> bshift);
				s2_ba = *reinterpret_cast<const uint4 *>(p.keys + 2ull * s2_b1);
			}
			/* stage 1 */
			s1_v = false;
			if (take) {
				const uint32_t n = qn < 64 ? qn : 64;
				qn -= n;
				s1_v = (uint32_t) lane < n;
				NTSM_ABL_STAGE1(s1_v, p)
				if (s1_v) {
					/* Rebuild the window from the tile bytes still in LDS (19 byte reads + table reads per 64 positives:
					 * 0.05 instructions per stream position at the filter's pass rate): f3 = forward word of its first 16
					 * bases, r = reverse word of its last 16, exactly what the rolling registers held at that position.
					 * Then both 38-bit strands from the two words: the forward code is the first 16 bases followed by
					 * the last 3 (complement-reversed top 3 groups of the reverse word), the reverse-complement code is
					 * the reverse word followed by the complement-reversed first 3. */
					const uint2 q = queue[qn + lane];
					const uint32_t f3 = q.x, r = q.y;
					if (PER_READ) s1_pos = qpos[qn + lane];
					if (!GEN) {
						const uint32_t tf = f3 >> 26, tr = r >> 26;
						const uint32_t l3 = 63u ^ (((tr & 3u) << 4) | (tr & 0xCu) | (tr >> 4));
						const uint32_t r3 = 63u ^ (((tf & 3u) << 4) | (tf & 0xCu) | (tf >> 4));
						const uint32_t a_hi = tf, a_lo = (f3 << 6) | l3;
						const uint32_t b_hi = tr, b_lo = (r << 6) | r3;
						const bool lt = a_hi < b_hi || (a_hi == b_hi && a_lo < b_lo);
						s1_klo = lt ? a_lo : b_lo;
						s1_khi = lt ? a_hi : b_hi;
					} else {
						/* general k: the same from the left-aligned words.  k >= 16: each strand = its 16-base word followed
						 * by the k - 16 bases only the other word holds = the complement of that word's top k - 16 groups in
						 * reverse order (bit reversal + swap inside the pairs reverses the groups) */
						unsigned long long fw, rv;
						if (gk >= 16) {
							auto crev = [](uint32_t x) { const uint32_t y = __builtin_bitreverse32(~x); return ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1); };
							const uint32_t g2 = 2u * gk - 32u, lowmask = (1u << g2) - 1u;       /* g2 <= 30 */
							fw = ((unsigned long long) f3 << g2) | (crev(r) & lowmask);
							rv = ((unsigned long long) r << g2) | (crev(f3) & lowmask);
						} else {
							fw = f3 >> (32u - 2u * gk);
							rv = r >> (32u - 2u * gk);
						}
						const unsigned long long key = fw < rv ? fw : rv;
						s1_klo = (uint32_t) key;
						s1_khi = (uint32_t) (key >> 32);
					}
					const uint32_t fo = ntsm_fold(((unsigned long long) s1_khi << 32) | s1_klo);
					s1_g1 = ntsm_h1(fo);
					s1_g2 = ntsm_h2(fo);
					/* second-level filter (L2 resident, exact canonical code, well-mixed hash): most first-level
					 * false positives stop here instead of costing an Infinity-Cache access to the key table */
					s1_pw = p.prefilter[s1_g1 >> p.pf_shift];
				}
			}
		};
		auto drain = [&](bool all) {
			if (!all) { drain_step(true); return; }
			while (qn > 0) drain_step(true);
			drain_step(false);                                 /* stage 1 -> 2 */
			drain_step(false);                                 /* stage 2 -> 3 */
			drain_step(false);                                 /* stage 3 */
		};

		/* Phase A of one 8-position block: roll, 12-mer order hashes, sliding minimum, k-mer bit hash; decides per
		 * position whether the lane needs a new filter block and issues the 8 block loads.  The per-lane conditions
		 * live in scalar registers as 64-bit wave masks (ballot / inverse ballot): one vector compare each for "window
		 * has an invalid base" and "minimizer changed", the rest is scalar logic that runs beside the vector unit.
		 *   bad   window invalid
		 *   ld    valid, and the minimizer differs from the previous position's or the previous window was invalid
		 *         (then nothing is cached): the lane requests its block; every other lane sends an out-of-range
		 *         offset -- the buffer load returns 0 for it and makes no memory request
		 *   sel   ld | bad: the lane replaces its cached block by what came back, which for a bad lane is 0: an
		 *         invalid window then fails the bit test by itself and needs no mask of its own */
		struct BlockState { uint32_t u[HB], f3[HB], r[HB]; unsigned long long sel[HB]; uint4 bl[HB];
			uint32_t h[TWO ? HB : 1], bw[TWO ? HB : 1]; unsigned long long ld[TWO ? HB : 1]; };   /* one step: HB positions */
		auto lut_reads = [&](const uint2 v, const int j0, uint2 (&e)[HB]) {   /* the table reads of one step issue together */
			const uint32_t w[2] = { v.x, v.y };
#pragma unroll
			for (int jj = 0; jj < HB; ++jj) { const int j = j0 + jj; e[jj] = lut64[(w[j >> 2] >> ((j & 3) * 8)) & 0xFFu]; }
		};
		/* gg / fh / pm: order hashes, forward words and prefix minimum of the current 8-block (they outlive a 4-position step);
		 * m2: minima of adjacent pairs (W < 8) */
		auto phase_a = [&](const uint2 (&e)[HB], BlockState &B, const int j0, uint32_t (&gg)[8], uint32_t (&fh)[8], uint32_t &pm, uint32_t (&m2)[8]) {
#pragma unroll
			for (int jj = 0; jj < HB; ++jj) {
				const int j = j0 + jj;
				NTSM_STEP(e[jj])
				fh[j] = F;
				gg[j] = NTSM_MMER_G();
				uint32_t mz;
				if (W >= 8) {
					pm = min(pm, gg[j]);
					mz = j + 9 - W <= 7 ? min(sprev[j + 9 - W], pm) : pm;
				} else {
					/* fewer than 8 candidates: minimum of the last W order hashes by doubling -- pairs, fours, then what is
					 * left of W (negative indices: the previous 8-block) */
					auto G = [&](int q) { return q >= 0 ? gg[q] : gprev[q + 8]; };
					auto M2 = [&](int q) { return q >= 0 ? m2[q] : m2prev[q + 8]; };
					m2[j] = min(gg[j], G(j - 1));
					const uint32_t m4 = min(m2[j], M2(j - 2));
					mz = W == 2 ? m2[j] : W == 3 ? min(m2[j], G(j - 2)) : W == 4 ? m4 : W == 5 ? min(m4, G(j - 4)) : W == 6 ? min(m4, M2(j - 4)) : min(m4, min(M2(j - 4), G(j - 6)));
				}
				B.f3[jj] = GEN ? f_top() : (j >= 3 ? fh[j - 3] : (j == 0 ? fc0 : (j == 1 ? fc1 : fc2)));
				B.r[jj] = GEN ? (R & g_rmask) : R;
				B.u[jj] = ntsm_kmer_sum(B.f3[jj], B.r[jj]);
				const unsigned long long bad = __builtin_amdgcn_ballot_w64(run <= gk);
				const unsigned long long ld = ~bad & (__builtin_amdgcn_ballot_w64(mz != mz_prev) | bad_prev);
				B.sel[jj] = ld | bad;
				const uint32_t bh = ntsm_block_hash(mz);
				const uint32_t bi = TWO ? ntsm_range(bh, bloom_words) : ntsm_range(bh, blk_map.n_blocks);   /* TWO: the Bloom word first */
				const uint32_t idx = NTSM_ABL_BLOCK_INDEX(p, mz, ld, bi);   /* ld lanes: bi; every other lane: out of range */
				mz_prev = mz;
				bad_prev = bad;
				nk_s += (uint32_t) __popcll(~bad);
				if (TWO) {
					B.bw[jj] = ntsm_struct_buffer_load_b32(blm_rsrc, (int) idx, 0, 0, 0);
					B.h[jj] = bh;
					B.ld[jj] = ld;
				} else {   /* issue each block load as soon as its offset is known: earlier positions get the rest of the block as
				     * cover (+2 % over issuing the eight loads together at the end of the phase) */
					const ntsm_u32x4 bv = ntsm_struct_buffer_load_b128(blk_rsrc, (int) idx, 0, 0, 0);
					B.bl[jj] = make_uint4(bv.x, bv.y, bv.z, bv.w);
				}
			}
		};
		/* Phase B (two-level path): test the run-start lanes' Bloom words (two bits, byte-aligned fields of the block
		 * hash, sign-bit test like phase C) and request the 128-bit block only for the runs that pass.  Every other lane
		 * -- no run start, or Bloom says "not a site minimizer" -- sends an out-of-range index and gets 0 back; a
		 * run-start lane that failed thereby caches an all-zero block, and its windows fail phase C by themselves. */
		auto phase_b = [&](BlockState &B) {
#pragma unroll
			for (int j = 0; j < HB; ++j) {
				uint32_t s0, s1;
				asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(s0) : "v"(B.h[j]), "v"(B.bw[j]));
				asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(s1) : "v"(B.h[j]), "v"(B.bw[j]));
				const unsigned long long go = B.ld[j] & __builtin_amdgcn_ballot_w64((int32_t) (s0 & s1) < 0);
				const uint32_t idx = __builtin_amdgcn_inverse_ballot_w64(go) ? ntsm_range(B.h[j], blk_map.n_blocks) : 0xFFFFFFFFu;
				const ntsm_u32x4 bv = ntsm_struct_buffer_load_b128(blk_rsrc, (int) idx, 0, 0, 0);
				B.bl[j] = make_uint4(bv.x, bv.y, bv.z, bv.w);
			}
		};
		auto block_end = [&](const uint32_t (&gg)[8], const uint32_t (&fh)[8], const uint32_t (&m2)[8]) {
			fc0 = fh[5]; fc1 = fh[6]; fc2 = fh[7];
			if (W >= 8) {
				sprev[7] = gg[7];
#pragma unroll
				for (int j = 6; j >= 9 - W; --j) sprev[j] = min(gg[j], sprev[j + 1]);
			} else {
#pragma unroll
				for (int j = 2; j < 8; ++j) gprev[j] = gg[j];
#pragma unroll
				for (int j = 4; j < 8; ++j) m2prev[j] = m2[j];
			}
		};
		/* Phase C: four-bit test against the (possibly just fetched) block.  word << field (NTSM_KBITn: bit 31 - field)
		 * puts the tested bit in the sign position -- the shifter takes the low five bits of the selected byte, so the
		 * fields need no mask -- and the sign of the AND of the four is the verdict.  Positives go to the wave's queue
		 * as { forward word of the first 16 bases, reverse word of the last 16 }.  (Queueing tile offsets instead and
		 * rebuilding the window from the tile bytes at drain time was measured: 3 fewer instructions per position in
		 * this loop, 53.2 instead of 52.2 ms per 3e8 reads.) */
		auto phase_c = [&](const BlockState &B, const int pos0) {
#pragma unroll
			for (int j = 0; j < HB; ++j) {
				const bool sel = __builtin_amdgcn_inverse_ballot_w64(B.sel[j]);
				cur.x = sel ? B.bl[j].x : cur.x;
				cur.y = sel ? B.bl[j].y : cur.y;
				cur.z = sel ? B.bl[j].z : cur.z;
				cur.w = sel ? B.bl[j].w : cur.w;
				const uint32_t u = B.u[j], um = ntsm_kmer_mix(u);
				uint32_t s0, s1, s2, s3;
				asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(s0) : "v"(u), "v"(cur.x));
				asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(s1) : "v"(um), "v"(cur.y));
				asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(s2) : "v"(um), "v"(cur.z));
				asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(s3) : "v"(um), "v"(cur.w));
				const bool pass = (int32_t) (__builtin_amdgcn_bitop3_b32(s0, s1, s2, 0x80) & s3) < 0;
				const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
				if (m) {
					if (pass) {
						const uint32_t at = qn + __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
						queue[at] = make_uint2(B.f3[j], B.r[j]);
						if (PER_READ) qpos[at] = (uint16_t) (pos0 + j);
					}
					qn += (uint32_t) __popcll(m);
					if (qn >= 64) drain(false);
				}
			}
		};

		BlockState S;
		uint2 e[HB];
#pragma unroll 1
		for (int b = 0; b < NB; ++b) {
			const uint2 v = *reinterpret_cast<const uint2 *>(tile + ntsm_tile_addr<C>(t + 1, b * 8));
			uint32_t gg[8], fh[8], m2[8], pm = 0xFFFFFFFFu;
#pragma unroll
			for (int j0 = 0; j0 < 8; j0 += HB) {
				lut_reads(v, j0, e);
				/* Wave priority: up for phase C -- the bit tests that consume the block loads -- and through the next step's table reads,
				 * down again for phase A, which issues the next block loads: the four waves of a SIMD then leave phase C one after the
				 * other instead of sharing its issue slots.  1.54 M keys, interleaved A/B on three boxes at 2e8 and 1e9 reads: 911-917
				 * against 903-908 Gbases/s (+1.0 %; priority 1, 2 or 3 alike).  WHERE the priority drops matters: right after phase C
				 * (the table reads at low priority) is 1 % SLOWER than no priority at all; raised in phase A instead -0.5 %, over the
				 * whole loop -1.2 %; the two-level form (fabric-bound) does not move (NOTEBOOK R6.13).  -DNTSM_NO_PRIO: A/B builds. */
#ifndef NTSM_NO_PRIO
				__builtin_amdgcn_s_setprio(0);
#endif
				phase_a(e, S, j0, gg, fh, pm, m2);
				if (TWO) phase_b(S);
#ifndef NTSM_NO_PRIO
				__builtin_amdgcn_s_setprio(2);
#endif
				phase_c(S, t * C + b * 8 + j0);
			}
			block_end(gg, fh, m2);
		}
		drain(true);
#undef NTSM_STEP
#undef NTSM_MMER_G
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) nh += __shfl_down(nh, off, 64);
	if ((t & 63) == 0) {
		if (nk_s) atomicAdd(p.totals + 0, p.sign * (unsigned long long) nk_s);
		if (nh) atomicAdd(p.totals + 1, p.sign * (unsigned long long) nh);
	}
}

#ifdef NTSM_WITH_TAB
/* The tabulated k = 19 kernel + its look-up kernel: a measured negative result (7 % slower than the minimizer-blocked
 * kernel), kept out of the default library; `make tab` builds ntsm_amd/libntsm_hip_tab.so with it. */
#include "ntsm_tab_kernel.inc"
#endif

} // namespace

namespace ntsm_rt {

int mz_tile_bytes() { return kThreads * kFastC; }

hipError_t launch_mz(const NtsmCountParams &p, unsigned grid, hipStream_t st, int mode, bool per_read, bool two_level)
{
	const dim3 g(grid), b(kThreads);
#define NTSM_MZ_CASE(M_) \
	case 2 * M_: hipLaunchKernelGGL((ntsm_count_mz_kernel<M_, false, kFastC, false>), g, b, 0, st, p); break; \
	case 2 * M_ + 1: hipLaunchKernelGGL((ntsm_count_mz_kernel<M_, true, kFastC, false>), g, b, 0, st, p); break;
#define NTSM_MZ2_CASE(M_) \
	case 2 * M_: hipLaunchKernelGGL((ntsm_count_mz_kernel<M_, false, kFastC, true>), g, b, 0, st, p); break; \
	case 2 * M_ + 1: hipLaunchKernelGGL((ntsm_count_mz_kernel<M_, true, kFastC, true>), g, b, 0, st, p); break;
	if (two_level) {
		switch (mode * 2 + (per_read ? 1 : 0)) {
		NTSM_MZ2_CASE(0) NTSM_MZ2_CASE(2) NTSM_MZ2_CASE(3) NTSM_MZ2_CASE(4) NTSM_MZ2_CASE(5) NTSM_MZ2_CASE(6) NTSM_MZ2_CASE(7) NTSM_MZ2_CASE(8) NTSM_MZ2_CASE(9)
		default: return hipErrorInvalidValue;
		}
	} else
	switch (mode * 2 + (per_read ? 1 : 0)) {
	NTSM_MZ_CASE(0) NTSM_MZ_CASE(2) NTSM_MZ_CASE(3) NTSM_MZ_CASE(4) NTSM_MZ_CASE(5) NTSM_MZ_CASE(6) NTSM_MZ_CASE(7) NTSM_MZ_CASE(8) NTSM_MZ_CASE(9)
	default: return hipErrorInvalidValue;
	}
#undef NTSM_MZ_CASE
#undef NTSM_MZ2_CASE
	return hipGetLastError();
}

#ifdef NTSM_WITH_TAB
#include "ntsm_tab_launch.inc"
#endif

} // namespace ntsm_rt
