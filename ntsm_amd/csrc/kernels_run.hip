/*
 * kernels_run.hip -- the run-anchored count kernel (k = 19; ntsm_set_kernel 5).
 *
 * Same job as the other count kernels -- the loop of FingerPrint::insertCount (src/FingerPrint.hpp:89-103: rolling canonical
 * k-mer of vendor/KseqHashIterator.hpp:95-112, tsl::robin_map find, `+= 1`) over a flat stream -- but the membership
 * pre-test is made ONCE PER MINIMIZER RUN instead of once per k-mer:
 *
 *   consecutive 19-mers of a read share their minimizer M (smallest order key among their eight canonical 12-mers) for 4.4
 *   positions on average, and a site's k-mers come in the same runs.  A 19-mer with b bases to the right of M either holds
 *   at least four of them (b >= 4: class R) or at least four bases to the left of M (b <= 3: class L); so every 19-mer
 *   contains one of the two ANCHORED 16-MERS of its run, E_R = M + 4 bases right or E_L = 4 bases left + M, and which one is
 *   a function of the k-mer alone.  The filter holds, per site k-mer, the signature of ITS anchored 16-mer in the 128-bit
 *   block of its minimizer: the k-mers of one site window that share a minimizer (3.3 on average, up to 8) set at most two
 *   signatures instead of one each, so the same false-positive rate needs roughly half the bits, the drain's second-level
 *   Bloom is not needed, and a run is tested with at most two signature tests however many k-mers it holds.
 *
 * Structure: the main loop only rolls the words, keeps the sliding minimum (the order key carries the position of its
 * 12-mer in its low bits, so the minimum also says where M is) and, when a lane's run ends, pushes one 12-byte record
 * { last 16 bases, the 10 before them + first / last position, key of M } into the wave's LDS queue.  Whenever 64 records are queued the
 * wave processes them with every lane busy: block index from the key, one 16-byte block load (one L2 request per run, as in
 * kernels_mz.hip), and -- one call later, when the block has arrived -- the two signature tests.  Runs that pass (true site runs
 * and ~1 % false positives) go to a second queue and are expanded 64 at a time: every k-mer of the passing class is rebuilt from
 * the record, looked up in the cuckoo table, and its counter bumped.  The queues and the pipelines' registers outlive the tile: a
 * record carries nothing of its tile, so partial batches wait for the next tile's records and everything is drained once, after
 * the workgroup's last tile.  Exactness does not depend on the filter: it only decides
 * which k-mers are looked up, and it has no false negatives because a k-mer's minimizer, class and anchored 16-mer are functions
 * of the k-mer alone (the host sets the signature for every position at which the minimum order key occurs in the k-mer).
 * Not instantiated for -m mode (per-read attribution): armed batches use kernels_mz.hip's PER_READ kernels.
 */
#include "kernels_common.h"
#include "ntsm_internal.h"

namespace {

#ifndef NTSM_RUN_C
#define NTSM_RUN_C 80                                  /* stream bytes per thread and tile: 80 leaves room for full-size queues at four workgroups per CU (38.1 of
                                                        * 40 KB).  Measured at 2.5 M keys, same box, first version of the kernel: C 128 / 3 workgroups 722 Gbases/s; C 96 with the
                                                        * queues cut to fit four workgroups 777; C 80 789; C 64 724 (still four workgroups: the 22-base warm-up per chunk weighs
                                                        * more).  Final version: C 80 797, C 64 761, C 64 with the candidate / k-mer queues cut to 96 entries for FIVE waves per
                                                        * SIMD 731 */
#endif
#ifndef NTSM_RUN_WAVES
#define NTSM_RUN_WAVES 4                               /* waves per SIMD the register budget is held to (LDS: 38.1 KB per workgroup at C = 80); three instead of four costs 15 % */
#endif
constexpr int kRunC = NTSM_RUN_C;
#ifndef NTSM_RUN_CAND_AT
#define NTSM_RUN_CAND_AT 64                            /* passing runs are expanded once this many are queued */
#endif
#ifndef NTSM_RUN_ABL
#define NTSM_RUN_ABL 0                                 /* experiment builds only: 1 = records are dropped instead of processed, 2 = no expansion */
#endif
constexpr int kRunQueue = 128;                         /* run records per wave: < 64 left over + one position's burst of <= 64 */
constexpr int kCandAt = NTSM_RUN_CAND_AT;
constexpr int kCandQueue = (kCandAt - 1 + 64 + 15) / 16 * 16;   /* passing runs per wave: < kCandAt left over + one batch's <= 64 */
#ifndef NTSM_RUN_KMER_AT
#define NTSM_RUN_KMER_AT 64                            /* queued k-mers are looked up (64 at a time) once this many wait */
#endif
constexpr int kKmerAt = NTSM_RUN_KMER_AT;
constexpr int kKmerQueue = (kKmerAt - 1 + 64 + 15) / 16 * 16;   /* < kKmerAt left over + one expansion step's <= 64 */

template <int C>
__device__ __forceinline__ int ntsm_run_tile_addr(int row, int byte_in_row)
{
	if (C == 128) return row * C + ((((byte_in_row >> 4) ^ (row >> 1)) & 7) << 4) + (byte_in_row & 15);
	return row * C + (int) ((((uint32_t) (byte_in_row >> 4) + ((uint32_t) row >> 3)) % (uint32_t) (C / 16)) << 4) + (byte_in_row & 15);
}

/* reverse complement of a 16-base word (oldest base in the top bits): complement, reverse the bits, swap inside the pairs */
__device__ __forceinline__ uint32_t ntsm_rc16(uint32_t w)
{
	const uint32_t y = __builtin_bitreverse32(~w);
	return ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
}

/* Order key of the 12-mer that ends at the newest base (ntsm_device.h: NTSM_RUN_ORDER): hash << 8 | position mod 16.  F holds the
 * forward code in its low 24 bits, R the reverse-complement code in its top 24. */
__device__ __forceinline__ uint32_t ntsm_run_key(uint32_t F, uint32_t R, uint32_t pos16, uint32_t kmask)
{
#if NTSM_RUN_ORDER == 1
	uint32_t prod, key;                                     /* the multiplier reads the low 24 bits of its operands */
	asm("v_mul_u32_u24 %0, %1, %2" : "=v"(prod) : "v"(F), "v"(R >> 8));
	asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(prod), "v"(kmask), "s"(pos16));   /* kmask = 0xFFFFFF00 in a register: one scalar operand per instruction */
	return key;
#else
	/* The product is hidden from the optimiser: left alone it folds the shift into the constant (a 32-bit multiplier instead of
	 * v_mul_u32_u24 + v_lshl_or_b32: same count, measured 0.7 % slower) */
	uint32_t h = ntsm_run_hash24(min(F & 0xFFFFFFu, R >> 8));
	asm("" : "+v"(h));
	return (h << 8) | pos16;
#endif
}

template <int C>
__global__ __launch_bounds__(kThreads, NTSM_RUN_WAVES) void ntsm_count_run_kernel(const NtsmCountParams p)
{
	constexpr int VPT = C / 16, NB = C / 8;
	__shared__ __attribute__((aligned(16))) uint8_t tile[(kThreads + 1) * C];
	__shared__ uint2 lut64[256];
	/* A run record is 12 bytes: F (the last 16 bases as of the run's last position), W = { the 10 bases before them : 20 bits |
	 * classes that passed : 2 (passing runs only) | spare : 3 | first position mod 8 : 3 | last position mod 16 : 4 }, K = order key
	 * of the minimizer (hash << 8 | its position mod 16).  26 bases are all a run ever needs: its oldest window begins 18 + 7 bases
	 * before its last position. */
	__shared__ uint32_t rq_all[kThreads / 64][3][kRunQueue]; /* run records, one array per word (F, W, K): one address per push, dword stride */
	__shared__ uint32_t cq_all[kThreads / 64][3][kCandQueue];/* passing runs, same layout */
	__shared__ uint2 kq_all[kThreads / 64][kKmerQueue];      /* canonical codes of their k-mers, waiting for the look-up */
	const int t = threadIdx.x;
	const int lane = t & 63;
	uint32_t *rq = rq_all[t >> 6][0], *cq = cq_all[t >> 6][0];
	uint2 *kq = kq_all[t >> 6];
	lut64[t] = p.lut64[t];
	const uint32_t bshift = p.bshift, n_blocks = p.blk_map.n_blocks;
	const unsigned long long blk_base = (unsigned long long) p.blocks;
	const ntsm_i32x4 blk_rsrc = { (int) (uint32_t) blk_base, (int) ((uint32_t) (blk_base >> 32) | (16u << 16)), (int) (p.blk_bytes >> 4), 0x00020000 };
	unsigned long long nk_s = 0;                             /* wave-uniform, lives across all tiles of the workgroup: 64 bits (a forced small grid over a large stream passes 2^32) */
	uint32_t nh = 0;
	uint32_t kmask = 0xFFFFFF00u;
	asm volatile("" : "+v"(kmask));
	static_assert(C != 128, "the main loop steps through the additive row rotation");
	const uint32_t rot0 = (uint32_t) (ntsm_run_tile_addr<C>(t + 1, 0) - (t + 1) * C);

	/* The queues and the two pipelines below live across tiles: records are self-contained (bases, positions mod 16, key), so a
	 * tile's last partial batches wait for the next tile's records instead of being flushed with half-empty waves; everything
	 * is drained once, after the workgroup's last tile. */
	uint32_t qn = 0, cn = 0, kn = 0;                     /* wave-uniform queue fills */

	/* ---- look-up of the k-mers of passing runs, 64 at a time, two stages over consecutive calls so that no bucket load is
	 * consumed by the call that issued it: (A) pop 64 canonical codes, issue the first bucket's load; (B, next call) compare
	 * (bucket 2 only if bucket 1 is full -- the host inserts with that invariant), one counter update per hit ---- */
	uint32_t l_klo = 0, l_khi = 0, l_g2 = 0;
	unsigned long long l_b1 = 0;
	uint4 l_ba = make_uint4(0, 0, 0, 0);
	bool l_v = false;
	auto lookup = [&](bool take) {
		long long slot = -1;
		if (l_v) {
			if (l_ba.x == l_klo && l_ba.y == l_khi) slot = (long long) l_b1;
			else if (l_ba.z == l_klo && l_ba.w == l_khi) slot = (long long) l_b1 + 1;
			else if ((l_ba.x & l_ba.y) != 0xFFFFFFFFu && (l_ba.z & l_ba.w) != 0xFFFFFFFFu) {
				const unsigned long long b2 = 2ull * (l_g2 >> bshift);
				const uint4 bb = *reinterpret_cast<const uint4 *>(p.keys + 2ull * b2);
				if (bb.x == l_klo && bb.y == l_khi) slot = (long long) b2;
				else if (bb.z == l_klo && bb.w == l_khi) slot = (long long) b2 + 1;
			}
			if (slot >= 0) ++nh;
		}
		ntsm_add_hits(p, slot, lane);
		l_v = false;
		if (take) {
			const uint32_t n = kn < 64 ? kn : 64;
			kn -= n;
			l_v = (uint32_t) lane < n;
			if (l_v) {
				const uint2 q = kq[kn + lane];
				l_klo = q.x; l_khi = q.y;
				const uint32_t fo = ntsm_fold(((unsigned long long) q.y << 32) | q.x), g1 = ntsm_h1(fo);
				l_g2 = ntsm_h2(fo);
				l_b1 = 2ull * (g1 >> bshift);
				l_ba = *reinterpret_cast<const uint4 *>(p.keys + 2ull * l_b1);
			}
		}
	};

	/* ---- expansion of passing runs: every k-mer of the classes that passed is rebuilt (no memory access) and queued ---- */
	auto expand = [&]() {
		const uint32_t n = cn < 64 ? cn : 64;
		cn -= n;
		const bool have = (uint32_t) lane < n;
		uint2 r = make_uint2(0, 0);
		uint32_t rkey = 0;
		if (have) { const uint32_t *q = cq + cn + lane; r = make_uint2(q[0], q[kCandQueue]); rkey = q[2 * kCandQueue]; }
		const uint32_t i1 = r.y & 15u, len1 = (i1 - (r.y >> 4)) & 7u, cls = (r.y >> 10) & 3u, fh = r.y >> 12;
		const uint32_t o1 = (i1 - rkey) & 15u, o0 = o1 - len1;
#pragma unroll 1
		for (uint32_t step = 0; step < 8; ++step) {
			const uint32_t o = o0 + step;                    /* bases to the right of M in this window */
			const bool act = have && o <= o1 && ((o >= 4u ? cls & 1u : cls & 2u) != 0u);
			if (__builtin_amdgcn_ballot_w64(have && o <= o1) == 0ull) break;
			const unsigned long long am = __builtin_amdgcn_ballot_w64(act);
			if (am == 0ull) continue;
			if (act) {
				/* the window ends o1 - o bases before the record's newest base: its last 16 bases and the 3 before them; the reverse
				 * complement strand is the reversed last 16 followed by the reversed first 3 */
				const uint32_t sh = 2u * (o1 - o);
				const uint32_t lo = __builtin_amdgcn_alignbit(fh, r.x, sh), t3 = (fh >> sh) & 63u;
				const uint32_t rl = ntsm_rc16(lo), n3 = ~t3;
				const uint32_t r3 = ((n3 & 3u) << 4) | (n3 & 0xCu) | ((n3 >> 4) & 3u);
				const uint32_t a_hi = t3, a_lo = lo, b_hi = rl >> 26, b_lo = (rl << 6) | r3;
				const bool lt = a_hi < b_hi || (a_hi == b_hi && a_lo < b_lo);
				const uint32_t at = kn + __builtin_amdgcn_mbcnt_hi((uint32_t) (am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) am, 0u));
				kq[at] = make_uint2(lt ? a_lo : b_lo, lt ? a_hi : b_hi);
			}
			kn += (uint32_t) __popcll(am);
			if (kn >= (uint32_t) kKmerAt) lookup(true);
		}
	};

	/* ---- run processing, two stages over consecutive calls: (1) pop 64 records, request their blocks; (2) test ---- */
	uint2 s_rec = make_uint2(0, 0);
	uint32_t s_key = 0;
	uint4 s_blk = make_uint4(0, 0, 0, 0);
	bool s_v = false;
	auto process = [&](bool take) {
		/* Wave priority (as in kernels_mz.hip, NOTEBOOK R6.13): low for the stage that tests the fetched blocks and requests the next
		 * ones, raised again behind the main loop's next table reads.  2.5 M keys, interleaved on two boxes: 828.6-831.5 against
		 * 821.9-824.7 Gbases/s (+0.8 %); the other way round (raised here, dropped in the loop) +0.2 %, raised around this stage only
		 * -0.7 %.  -DNTSM_NO_PRIO: A/B builds. */
#ifndef NTSM_NO_PRIO
		__builtin_amdgcn_s_setprio(0);
#endif
		/* stage 2 */
		uint32_t cls = 0;
		if (s_v) {
			const uint32_t i1 = s_rec.y & 15u, len1 = (i1 - (s_rec.y >> 4)) & 7u, fh = s_rec.y >> 12;
			const uint32_t o1 = (i1 - s_key) & 15u, o0 = o1 - len1;
			/* E_R = M + 4 bases right ends o1 - 4 bases before the newest one; E_L = 4 bases left + M ends o1 bases before it */
			const uint32_t wR = __builtin_amdgcn_alignbit(fh, s_rec.x, (2u * o1 - 8u) & 31u);
			const uint32_t wL = __builtin_amdgcn_alignbit(fh, s_rec.x, 2u * o1);
			const uint32_t uR = wR + ntsm_rc16(wR), uL = wL + ntsm_rc16(wL);
			const uint32_t mR = ntsm_kmer_mix(uR), mL = ntsm_kmer_mix(uL);
			/* word << field puts the tested bit (31 - field, NTSM_KBITn) into the sign position; the shifter takes the low five bits
			 * of the selected byte, and the sign of the AND of the four is the verdict (as in kernels_mz.hip's phase C) */
			auto passes = [&](uint32_t u, uint32_t um) -> bool {
				uint32_t s0, s1, s2, s3;
				asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(s0) : "v"(u), "v"(s_blk.x));
				asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(s1) : "v"(um), "v"(s_blk.y));
				asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(s2) : "v"(um), "v"(s_blk.z));
				asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(s3) : "v"(um), "v"(s_blk.w));
				return (int32_t) (__builtin_amdgcn_bitop3_b32(s0, s1, s2, 0x80) & s3) < 0;
			};
			cls = (o1 >= 4u && passes(uR, mR) ? 1u : 0u) | (o0 <= 3u && passes(uL, mL) ? 2u : 0u);
		}
		const unsigned long long pm = __builtin_amdgcn_ballot_w64(cls != 0u);
		if (pm && NTSM_RUN_ABL != 2) {
			if (cls) {
				const uint32_t at = cn + __builtin_amdgcn_mbcnt_hi((uint32_t) (pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) pm, 0u));
				uint32_t *q = cq + at;
				q[0] = s_rec.x; q[kCandQueue] = s_rec.y | (cls << 10); q[2 * kCandQueue] = s_key;
			}
			cn += (uint32_t) __popcll(pm);
			if (cn >= (uint32_t) kCandAt) expand();
		}
		/* stage 1 */
		s_v = false;
		uint32_t idx = 0xFFFFFFFFu;
		if (take) {
			const uint32_t n = qn < 64 ? qn : 64;
			qn -= n;
			s_v = (uint32_t) lane < n;
			if (s_v) {
				const uint32_t *q = rq + qn + lane;
				s_rec = make_uint2(q[0], q[kRunQueue]);
				s_key = q[2 * kRunQueue];
				idx = ntsm_range(ntsm_block_hash(s_key >> 8), n_blocks);
			}
		}
		const ntsm_u32x4 bv = ntsm_struct_buffer_load_b128(blk_rsrc, (int) idx, 0, 0, 0);
		s_blk = make_uint4(bv.x, bv.y, bv.z, bv.w);
	};

	for (unsigned long long ti = blockIdx.x; ti < p.n_tiles; ti += gridDim.x) {
		const long long ts = p.t0 + (long long) (ti * (unsigned long long) (kThreads * C));
		__syncthreads();
		if (ts >= p.lo && ts + kThreads * C <= p.hi) {
			const __amdgpu_buffer_rsrc_t st_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.base + ts), (short) 0, kThreads * C, 0x00020000);
#pragma unroll
			for (int q = 0; q < VPT; ++q) {
				const int v = t + kThreads * q;
				const ntsm_u32x4 nt = __builtin_amdgcn_raw_buffer_load_b128(st_rsrc, 16 * v, 0, NTSM_STREAM_AUX);
				*reinterpret_cast<uint4 *>(tile + ntsm_run_tile_addr<C>(1 + v / VPT, (v % VPT) * 16)) = make_uint4(nt.x, nt.y, nt.z, nt.w);
			}
		} else {
#pragma unroll 1
			for (int q = 0; q < VPT; ++q) {
				const int v = t + kThreads * q;
				const uint4 r = ntsm_load_vec(p, ts + 16ll * v);
				*reinterpret_cast<uint4 *>(tile + ntsm_run_tile_addr<C>(1 + v / VPT, (v % VPT) * 16)) = r;
			}
		}
		if (t < 2) {
			const uint4 r = ntsm_load_vec(p, ts - 32 + 16 * t);
			*reinterpret_cast<uint4 *>(tile + ntsm_run_tile_addr<C>(0, C - 32 + 16 * t)) = r;
		}
		__syncthreads();

		/* F = codes of the last 16 bases (newest lowest), Fh = the 16 before them, R = reverse complement of the last 16
		 * (complement of the newest on top), run = 1 + valid bases since the last invalid one (window valid when run > 19) */
		uint32_t F = 0, Fh = 0, R = 0, run = 1;
		/* Sliding minimum over the eight 12-mers of a window with two v_min3_u32 per position: t3[j] = min(g[j], g[j-1], g[j-2]) and
		 * mz[j] = min(t3[j], t3[j-3], t3[j-5]) (the three triples cover g[j-7 .. j]; g[j-5] twice).  Carried from one group of eight
		 * positions to the next: the last two keys and the last five triples. */
		uint32_t gp6, gp7, t3p[8];
#define NTSM_RSTEP(e_)                                                                    \
		{                                                                                 \
			Fh = __builtin_amdgcn_alignbit(Fh, F, 30);                                    \
			asm("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(F) : "v"(F), "v"((e_).x));          \
			R = __builtin_amdgcn_alignbit((e_).y, R, 2);                                  \
			asm("v_mad_u32_u16 %0, %1, %2, 1 op_sel:[0,1,0,0]" : "=v"(run) : "v"(run), "v"((e_).y)); \
		}
		/* order key of the 12-mer that ends at the newest base: its 24-bit order hash on top, its position mod 16 below (says where
		 * the minimum sits; decides only between 12-mers of equal hash).  mod 16, not 8: a 12-mer and its reverse-complement twin 8
		 * positions on (palindromic site windows have them) would carry the same key, and the second would take over from the first
		 * without the key -- hence the run -- changing */
#define NTSM_RKEY(pos16_) ntsm_run_key(F, R, (uint32_t) (pos16_), kmask)
		{
			const uint4 v0 = *reinterpret_cast<const uint4 *>(tile + ntsm_run_tile_addr<C>(t, C - 32));
			const uint4 v1 = *reinterpret_cast<const uint4 *>(tile + ntsm_run_tile_addr<C>(t, C - 16));
			const uint32_t w[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
			uint32_t gw[8];
			/* Only the last 22 of the 32 bytes in front of the chunk matter: a run of this chunk begins at a position >= 0, its
			 * minimizer ends at most 7 bases before that and its left-anchored 16-mer 15 before the minimizer's end (22 back);
			 * window 0 itself reaches 18 back.  Bases further back never enter a record that is read. */
#pragma unroll
			for (int i = 10; i < 32; ++i) {
				const uint2 e = lut64[(w[i >> 2] >> ((i & 3) * 8)) & 0xFFu];
				NTSM_RSTEP(e)
				if (i >= 25) gw[i - 24] = NTSM_RKEY(i & 15);
			}
			gp6 = gw[6]; gp7 = gw[7];                            /* gw[8 - d] is the key d positions before the chunk */
#pragma unroll
			for (int i = 3; i <= 7; ++i) t3p[i] = min(min(gw[i], gw[i - 1]), gw[i - 2]);
		}
		uint32_t mz_prev = 0, i0 = 0;
		uint32_t rot = rot0;                                /* where this thread's next 16 bytes sit in its (rotated) tile row */
		unsigned long long bad_prev = ~0ull;
#pragma unroll 1
		for (int b = 0; b < NB; ++b) {
			const uint2 v = *reinterpret_cast<const uint2 *>(tile + (t + 1) * C + rot + ((b & 1) << 3));
			if (b & 1) { rot += 16; rot = rot == (uint32_t) C ? 0u : rot; }
			const uint32_t w[2] = { v.x, v.y };
			uint2 e[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) e[j] = lut64[(w[j >> 2] >> ((j & 3) * 8)) & 0xFFu];
#ifndef NTSM_NO_PRIO
			__builtin_amdgcn_s_setprio(2);
#endif
			uint32_t gg[8], t3[8];
			const uint32_t pcb = (uint32_t) __builtin_amdgcn_readfirstlane((b & 1) << 3);   /* scalar: position mod 16 = pcb | j */
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const uint32_t Fp = F, Fhp = Fh;                 /* the words as of the previous position: what a run that ends there is recorded with */
				NTSM_RSTEP(e[j])
				gg[j] = NTSM_RKEY(pcb | (uint32_t) j);
				t3[j] = min(min(gg[j], j >= 1 ? gg[j >= 1 ? j - 1 : 0] : gp7), j >= 2 ? gg[j >= 2 ? j - 2 : 0] : j == 1 ? gp7 : gp6);
				const uint32_t mz = min(min(t3[j], j >= 3 ? t3[j >= 3 ? j - 3 : 0] : t3p[j + 5]), j >= 5 ? t3[j >= 5 ? j - 5 : 0] : t3p[j + 3]);
				const unsigned long long bad = __builtin_amdgcn_ballot_w64(run <= (uint32_t) NTSM_FAST_K);
				const unsigned long long chg = __builtin_amdgcn_ballot_w64(mz != mz_prev);
				const unsigned long long endm = ~bad_prev & (chg | bad);     /* the lane's run ended with the previous position */
				const unsigned long long startm = ~bad & (chg | bad_prev);
				nk_s += (unsigned long long) __popcll(~bad);
				if (endm) {
					if (__builtin_amdgcn_inverse_ballot_w64(endm)) {
						const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t) (endm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) endm, qn));
						uint32_t *q = rq + at;
						uint32_t w;                                  /* the last position is the previous one: a scalar */
						asm("v_lshl_or_b32 %0, %1, 4, %2" : "=v"(w) : "v"(i0), "s"((pcb + (uint32_t) j - 1u) & 15u));
						q[0] = Fp; q[kRunQueue] = (Fhp << 12) | w;
						q[2 * kRunQueue] = mz_prev;
					}
					qn += (uint32_t) __popcll(endm);
					if (NTSM_RUN_ABL == 1) qn = 0;
					if (qn >= 64) process(true);
				}
				i0 = __builtin_amdgcn_inverse_ballot_w64(startm) ? (uint32_t) j : i0;   /* mod 8 is enough: a run is at most 8 positions long */
				mz_prev = mz;
				bad_prev = bad;
			}
			gp6 = gg[6]; gp7 = gg[7];
#pragma unroll
			for (int j = 3; j <= 7; ++j) t3p[j] = t3[j];
		}
		{   /* the chunk ends: runs that are still open are recorded with the words as they are */
			const unsigned long long endm = ~bad_prev;
			if (endm) {
				if (__builtin_amdgcn_inverse_ballot_w64(endm)) {
					const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t) (endm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) endm, qn));
					uint32_t *q = rq + at;
					q[0] = F; q[kRunQueue] = (Fh << 12) | ((i0 << 4) | (uint32_t) ((C - 1) & 15));
					q[2 * kRunQueue] = mz_prev;
				}
				qn += (uint32_t) __popcll(endm);
				if (qn >= 64) process(true);                    /* fewer than 64 wait when the next tile begins */
			}
		}
#undef NTSM_RSTEP
#undef NTSM_RKEY
	}
	while (qn > 0) process(true);
	process(false);                                         /* stage 2 of the last batch */
	while (cn > 0) expand();
	while (kn > 0) lookup(true);
	lookup(false);
	unsigned long long nh_w = nh;                            /* per lane 32 bits are plenty; the sum over the wave is taken in 64 */
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) nh_w += __shfl_down(nh_w, off, 64);
	if ((t & 63) == 0) {
		if (nk_s) atomicAdd(p.totals + 0, p.sign * nk_s);
		if (nh_w) atomicAdd(p.totals + 1, p.sign * nh_w);
	}
}

} // namespace

namespace ntsm_rt {

int run_tile_bytes() { return kThreads * kRunC; }

hipError_t launch_run(const NtsmCountParams &p, unsigned grid, hipStream_t st)
{
	hipLaunchKernelGGL((ntsm_count_run_kernel<kRunC>), dim3(grid), dim3(kThreads), 0, st, p);
	return hipGetLastError();
}

} // namespace ntsm_rt
