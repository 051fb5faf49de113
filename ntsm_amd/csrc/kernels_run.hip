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
 * 12-mer in its low bits, so the minimum also says where M is) and, when a lane's run ends, pushes one 16-byte record
 * { last 16 bases, 16 before, key of M, first / last position } into the wave's LDS queue.  Whenever 64 records are queued the
 * wave processes them with every lane busy: block index from the key, one 16-byte block load (one L2 request per run, as in
 * kernels_mz.hip), and -- one call later, when the block has arrived -- the two signature tests.  Runs that pass (true site runs
 * and ~1 % false positives) go to a second queue and are expanded 64 at a time: every k-mer of the passing class is rebuilt from
 * the record, looked up in the cuckoo table, and its counter bumped.  Exactness does not depend on the filter: it only decides
 * which k-mers are looked up, and it has no false negatives because a k-mer's minimizer, class and anchored 16-mer are functions
 * of the k-mer alone (the host sets the signature for every position at which the minimum order key occurs in the k-mer).
 * Not instantiated for -m mode (per-read attribution): armed batches use kernels_mz.hip's PER_READ kernels.
 */
#include "kernels_common.h"
#include "ntsm_internal.h"

namespace {

#ifndef NTSM_RUN_C
#define NTSM_RUN_C 128                                 /* stream bytes per thread and tile */
#endif
#ifndef NTSM_RUN_WAVES
#define NTSM_RUN_WAVES 3                               /* waves per SIMD the register budget is held to (LDS: 51 KB per workgroup) */
#endif
constexpr int kRunC = NTSM_RUN_C;
constexpr int kRunQueue = 128;                         /* run records per wave: < 64 left over + one position's burst of <= 64 */
constexpr int kCandQueue = 128;                        /* passing runs per wave, same rule */

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

template <int C>
__global__ __launch_bounds__(kThreads, NTSM_RUN_WAVES) void ntsm_count_run_kernel(const NtsmCountParams p)
{
	constexpr int VPT = C / 16, NB = C / 8;
	__shared__ __attribute__((aligned(16))) uint8_t tile[(kThreads + 1) * C];
	__shared__ uint2 lut64[256];
	__shared__ uint4 rq_all[kThreads / 64][kRunQueue];       /* run records: { F, Fh, key of M, first << 8 | last position } */
	__shared__ uint4 cq_all[kThreads / 64][kCandQueue];      /* passing runs: the record, classes that passed in meta bits 24 / 25 */
	const int t = threadIdx.x;
	const int lane = t & 63;
	uint4 *rq = rq_all[t >> 6], *cq = cq_all[t >> 6];
	lut64[t] = p.lut64[t];
	const uint32_t bshift = p.bshift, n_blocks = p.blk_map.n_blocks;
	const unsigned long long blk_base = (unsigned long long) p.blocks;
	const ntsm_i32x4 blk_rsrc = { (int) (uint32_t) blk_base, (int) ((uint32_t) (blk_base >> 32) | (16u << 16)), (int) (p.blk_bytes >> 4), 0x00020000 };
	uint32_t nk_s = 0, nh = 0;

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
		uint32_t sprev[8];
#define NTSM_RSTEP(e_)                                                                    \
		{                                                                                 \
			Fh = __builtin_amdgcn_alignbit(Fh, F, 30);                                    \
			asm("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(F) : "v"(F), "v"((e_).x));          \
			R = __builtin_amdgcn_alignbit((e_).y, R, 2);                                  \
			asm("v_mad_u32_u16 %0, %1, %2, 1 op_sel:[0,1,0,0]" : "=v"(run) : "v"(run), "v"((e_).y)); \
		}
		/* order key of the 12-mer that ends at the newest base: bijective 24-bit hash of its canonical code on top (no ties
		 * between different 12-mers), its position mod 16 below (says where the minimum sits; decides only between equal 12-mers).
		 * mod 16, not 8: a 12-mer and its reverse-complement twin 8 positions on (palindromic site windows have them) would carry
		 * the same key, and the second would take over from the first without the key -- hence the run -- changing */
#define NTSM_RKEY(pos16_) ((ntsm_run_hash24(min(F & 0xFFFFFFu, R >> 8)) << 8) | (uint32_t) (pos16_))
		{
			const uint4 v0 = *reinterpret_cast<const uint4 *>(tile + ntsm_run_tile_addr<C>(t, C - 32));
			const uint4 v1 = *reinterpret_cast<const uint4 *>(tile + ntsm_run_tile_addr<C>(t, C - 16));
			const uint32_t w[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
			uint32_t gw[8];
#pragma unroll
			for (int i = 0; i < 32; ++i) {
				const uint2 e = lut64[(w[i >> 2] >> ((i & 3) * 8)) & 0xFFu];
				NTSM_RSTEP(e)
				if (i >= 25) gw[i - 24] = NTSM_RKEY(i & 15);
			}
			sprev[7] = gw[7];
#pragma unroll
			for (int i = 6; i >= 1; --i) sprev[i] = min(gw[i], sprev[i + 1]);
		}
		uint32_t mz_prev = 0, i0 = 0;
		unsigned long long bad_prev = ~0ull;
		uint32_t qn = 0, cn = 0;                             /* wave-uniform queue fills */

		/* ---- expansion of passing runs: every k-mer of the classes that passed is rebuilt and looked up ---- */
		auto expand = [&]() {
			const uint32_t n = cn < 64 ? cn : 64;
			cn -= n;
			const bool have = (uint32_t) lane < n;
			uint4 r = make_uint4(0, 0, 0, 0);
			if (have) r = cq[cn + lane];
			const uint32_t pc = r.z & 15u, i1 = r.w & 0xFFu, ib = (r.w >> 8) & 0xFFu, cls = (r.w >> 24) & 3u;
			const uint32_t o1 = (i1 - pc) & 15u, o0 = o1 - (i1 - ib);
			const unsigned long long ctx = ((unsigned long long) r.y << 32) | r.x;
#pragma unroll 1
			for (uint32_t step = 0; step < 8; ++step) {
				const uint32_t o = o0 + step;                    /* bases to the right of M in this window */
				const bool act = have && o <= o1 && ((o >= 4u ? cls & 1u : cls & 2u) != 0u);
				if (__builtin_amdgcn_ballot_w64(have && o <= o1) == 0ull) break;
				long long slot = -1;
				if (act) {
					const unsigned long long fw = (ctx >> (2u * (o1 - o))) & 0x3FFFFFFFFFull;
					unsigned long long y = ~fw;                  /* reverse complement of the 38-bit code */
					y = ((y >> 2) & 0x3333333333333333ull) | ((y & 0x3333333333333333ull) << 2);
					y = ((y >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((y & 0x0F0F0F0F0F0F0F0Full) << 4);
					y = ((y >> 8) & 0x00FF00FF00FF00FFull) | ((y & 0x00FF00FF00FF00FFull) << 8);
					y = ((y >> 16) & 0x0000FFFF0000FFFFull) | ((y & 0x0000FFFF0000FFFFull) << 16);
					y = (y >> 32) | (y << 32);
					const unsigned long long rc = y >> (64 - 38);
					const unsigned long long key = fw < rc ? fw : rc;
					const uint32_t klo = (uint32_t) key, khi = (uint32_t) (key >> 32);
					const uint32_t fo = ntsm_fold(key), g1 = ntsm_h1(fo), g2 = ntsm_h2(fo);
					const unsigned long long b1 = 2ull * (g1 >> bshift);
					const uint4 ba = *reinterpret_cast<const uint4 *>(p.keys + 2ull * b1);
					if (ba.x == klo && ba.y == khi) slot = (long long) b1;
					else if (ba.z == klo && ba.w == khi) slot = (long long) b1 + 1;
					else if ((ba.x & ba.y) != 0xFFFFFFFFu && (ba.z & ba.w) != 0xFFFFFFFFu) {
						const unsigned long long b2 = 2ull * (g2 >> bshift);
						const uint4 bb = *reinterpret_cast<const uint4 *>(p.keys + 2ull * b2);
						if (bb.x == klo && bb.y == khi) slot = (long long) b2;
						else if (bb.z == klo && bb.w == khi) slot = (long long) b2 + 1;
					}
					if (slot >= 0) ++nh;
				}
				ntsm_add_hits(p, slot, lane);
			}
		};

		/* ---- run processing, two stages over consecutive calls: (1) pop 64 records, request their blocks; (2) test ---- */
		uint4 s_rec = make_uint4(0, 0, 0, 0), s_blk = make_uint4(0, 0, 0, 0);
		bool s_v = false;
		auto process = [&](bool take) {
			/* stage 2 */
			uint32_t cls = 0;
			if (s_v) {
				const uint32_t pc = s_rec.z & 15u, i1 = s_rec.w & 0xFFu, ib = (s_rec.w >> 8) & 0xFFu;
				const uint32_t o1 = (i1 - pc) & 15u, o0 = o1 - (i1 - ib);
				/* E_R = M + 4 bases right ends 4 - o1 ... i.e. (o1 - 4) bases before the newest; E_L = 4 left + M ends o1 bases before it */
				const uint32_t wR = __builtin_amdgcn_alignbit(s_rec.y, s_rec.x, (2u * (o1 - 4u)) & 31u);
				const uint32_t wL = __builtin_amdgcn_alignbit(s_rec.y, s_rec.x, 2u * o1);
				const uint32_t uR = wR + ntsm_rc16(wR), uL = wL + ntsm_rc16(wL);
				const uint32_t mR = ntsm_kmer_mix(uR), mL = ntsm_kmer_mix(uL);
				const uint32_t tR = (s_blk.x >> NTSM_KBIT0(uR)) & (s_blk.y >> NTSM_KBIT1(mR)) & (s_blk.z >> NTSM_KBIT2(mR)) & (s_blk.w >> NTSM_KBIT3(mR)) & 1u;
				const uint32_t tL = (s_blk.x >> NTSM_KBIT0(uL)) & (s_blk.y >> NTSM_KBIT1(mL)) & (s_blk.z >> NTSM_KBIT2(mL)) & (s_blk.w >> NTSM_KBIT3(mL)) & 1u;
				cls = (o1 >= 4u ? tR : 0u) | (o0 <= 3u ? tL << 1 : 0u);
			}
			const unsigned long long pm = __builtin_amdgcn_ballot_w64(cls != 0u);
			if (pm) {
				if (cls) {
					const uint32_t at = cn + __builtin_amdgcn_mbcnt_hi((uint32_t) (pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) pm, 0u));
					cq[at] = make_uint4(s_rec.x, s_rec.y, s_rec.z, s_rec.w | (cls << 24));
				}
				cn += (uint32_t) __popcll(pm);
				if (cn >= 64) expand();
			}
			/* stage 1 */
			s_v = false;
			uint32_t idx = 0xFFFFFFFFu;
			if (take) {
				const uint32_t n = qn < 64 ? qn : 64;
				qn -= n;
				s_v = (uint32_t) lane < n;
				if (s_v) {
					s_rec = rq[qn + lane];
					idx = ntsm_range(ntsm_block_hash(s_rec.z >> 8), n_blocks);
				}
			}
			const ntsm_u32x4 bv = ntsm_struct_buffer_load_b128(blk_rsrc, (int) idx, 0, 0, 0);
			s_blk = make_uint4(bv.x, bv.y, bv.z, bv.w);
		};

#pragma unroll 1
		for (int b = 0; b < NB; ++b) {
			const uint2 v = *reinterpret_cast<const uint2 *>(tile + ntsm_run_tile_addr<C>(t + 1, b * 8));
			const uint32_t w[2] = { v.x, v.y };
			uint2 e[8];
#pragma unroll
			for (int j = 0; j < 8; ++j) e[j] = lut64[(w[j >> 2] >> ((j & 3) * 8)) & 0xFFu];
			uint32_t gg[8], pmin = 0xFFFFFFFFu;
			const uint32_t pcb = (uint32_t) (b & 1) << 3;        /* wave-uniform: position mod 16 = pcb | j */
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const uint32_t Fp = F, Fhp = Fh;                 /* the words as of the previous position: what a run that ends there is recorded with */
				NTSM_RSTEP(e[j])
				gg[j] = NTSM_RKEY(pcb | (uint32_t) j);
				pmin = min(pmin, gg[j]);
				const uint32_t mz = j + 1 <= 7 ? min(sprev[j + 1], pmin) : pmin;
				const unsigned long long bad = __builtin_amdgcn_ballot_w64(run <= (uint32_t) NTSM_FAST_K);
				const unsigned long long chg = __builtin_amdgcn_ballot_w64(mz != mz_prev);
				const unsigned long long endm = ~bad_prev & (chg | bad);     /* the lane's run ended with the previous position */
				const unsigned long long startm = ~bad & (chg | bad_prev);
				nk_s += (uint32_t) __popcll(~bad);
				const uint32_t pos = (uint32_t) (b * 8 + j);
				if (endm) {
					if (__builtin_amdgcn_inverse_ballot_w64(endm)) {
						const uint32_t at = qn + __builtin_amdgcn_mbcnt_hi((uint32_t) (endm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) endm, 0u));
						rq[at] = make_uint4(Fp, Fhp, mz_prev, (i0 << 8) | ((pos - 1u) & 0xFFu));
					}
					qn += (uint32_t) __popcll(endm);
					if (qn >= 64) process(true);
				}
				i0 = __builtin_amdgcn_inverse_ballot_w64(startm) ? pos : i0;
				mz_prev = mz;
				bad_prev = bad;
			}
			sprev[7] = gg[7];
#pragma unroll
			for (int j = 6; j >= 1; --j) sprev[j] = min(gg[j], sprev[j + 1]);
		}
		{   /* the chunk ends: runs that are still open are recorded with the words as they are */
			const unsigned long long endm = ~bad_prev;
			if (endm) {
				if (__builtin_amdgcn_inverse_ballot_w64(endm)) {
					const uint32_t at = qn + __builtin_amdgcn_mbcnt_hi((uint32_t) (endm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) endm, 0u));
					rq[at] = make_uint4(F, Fh, mz_prev, (i0 << 8) | (uint32_t) (C - 1));
				}
				qn += (uint32_t) __popcll(endm);
			}
		}
		while (qn > 0) process(true);
		process(false);                                     /* stage 2 of the last batch */
		while (cn > 0) expand();
#undef NTSM_RSTEP
#undef NTSM_RKEY
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) nh += __shfl_down(nh, off, 64);
	if ((t & 63) == 0) {
		if (nk_s) atomicAdd(p.totals + 0, p.sign * (unsigned long long) nk_s);
		if (nh) atomicAdd(p.totals + 1, p.sign * (unsigned long long) nh);
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
