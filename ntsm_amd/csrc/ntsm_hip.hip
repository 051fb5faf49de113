/*
 * ntsm_hip.hip -- gfx950 kernels + the C ABI of include/ntsm_hip.h.
 *
 * Replaces the inner loop of the reference's FingerPrint::insertCount
 * (src/FingerPrint.hpp:89-103): KseqHashIterator's rolling 2-bit canonical k-mer
 * (vendor/KseqHashIterator.hpp:95-112), the tsl::robin_map lookup (src/FingerPrint.hpp:92) and
 * the `+= 1` (src/FingerPrint.hpp:94-99), batched over a flat stream of reads.
 *
 * Kernel structure (wave64, integer only, no MFMA -- DESIGN.md section 4):
 *   - a workgroup of 256 threads owns a tile of 256*C contiguous stream bytes; the tile is staged
 *     through LDS with coalesced 16-byte loads, rows padded by 16 B so that the per-thread
 *     ds_read_b128 of its own C-byte chunk is bank-conflict free;
 *   - each thread rolls fw / rc codes over its chunk (after warming up on the 32 bytes before it)
 *     and keeps a shift register of "invalid base" flags, so window validity is purely local and
 *     no k-mer can span the 'N' terminator between reads;
 *   - every valid window probes a 1-bit filter (L2 resident); the rare positives read their two
 *     16-byte cuckoo buckets and bump a 64-bit counter with one no-return atomic.
 * There is no CPU fallback anywhere in this file.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>                     /* types only: the library is bound with dlopen in ntsm_allreduce */

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ntsm_hip.h"
#include "ntsm_device.h"

/* ============================================================================================
 * Device code
 * ========================================================================================== */

typedef uint32_t ntsm_u32x4 __attribute__((ext_vector_type(4)));
typedef int ntsm_i32x4 __attribute__((ext_vector_type(4)));
/* buffer_load_dwordx4 ... idxen: clang has a builtin for the raw (byte offset) form only, so the LLVM intrinsic is
 * declared by name.  (descriptor, index, byte offset inside the element, scalar offset, cache policy) */
__device__ ntsm_u32x4 ntsm_struct_buffer_load_b128(ntsm_i32x4 rsrc, int vindex, int voffset, int soffset, int aux)
		__asm("llvm.amdgcn.struct.buffer.load.v4i32");
__device__ uint32_t ntsm_struct_buffer_load_b32(ntsm_i32x4 rsrc, int vindex, int voffset, int soffset, int aux)
		__asm("llvm.amdgcn.struct.buffer.load.i32");

namespace {

#ifndef NTSM_STREAM_NT
#define NTSM_STREAM_NT 1                                /* read stream: non-temporal loads (read once) */
#endif
#if NTSM_STREAM_NT && !defined(NTSM_STREAM_AUX)
#define NTSM_STREAM_AUX 2                               /* minimizer-blocked kernels, interior tiles: buffer loads with the nt bit */
#endif
constexpr int kThreads = 256;
constexpr uint32_t kN4 = 0x4E4E4E4Eu;      /* "NNNN" */
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 ntsm_load_vec(const NtsmCountParams &p, long long o)
{
	uint4 r = make_uint4(kN4, kN4, kN4, kN4);
	if (o + 16 > p.lo && o < p.hi) {
#if NTSM_STREAM_NT
		const u32x4 nt = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p.base + o));
#else
		const u32x4 nt = *reinterpret_cast<const u32x4 *>(p.base + o);
#endif
		r = make_uint4(nt.x, nt.y, nt.z, nt.w);
		if (o < p.lo || o + 16 > p.hi) {                    /* first / last vector of the range */
			uint32_t w[4] = { r.x, r.y, r.z, r.w };
			for (int b = 0; b < 16; ++b) {
				long long pos = o + b;
				if (pos < p.lo || pos >= p.hi)
					w[b >> 2] = (w[b >> 2] & ~(0xFFu << ((b & 3) * 8))) | (0x4Eu << ((b & 3) * 8));
			}
			r = make_uint4(w[0], w[1], w[2], w[3]);
		}
	}
	return r;
}

/* Key table layout: 32-byte buckets { key0, key1, count0, count1 } -- the counter of a slot sits in the cache line
 * its key was just read from, so the atomic of a hit finds the line in L2 instead of costing a second
 * Infinity-Cache access.  Slot s = 2 * bucket + position. */
__device__ __forceinline__ unsigned long long *ntsm_count_ptr(const uint64_t *table, long long slot)
{
	return const_cast<unsigned long long *>(reinterpret_cast<const unsigned long long *>(table)) + 4 * (slot >> 1) + 2 + (slot & 1);
}

/* first read whose terminator lies beyond byte offset pos */
__device__ __forceinline__ unsigned long long ntsm_read_of(const NtsmCountParams &p, unsigned long long pos)
{
	unsigned long long lo = 0, hi = p.n_reads;
	while (lo < hi) {
		unsigned long long mid = (lo + hi) >> 1;
		if (p.read_end[mid] > pos) hi = mid; else lo = mid + 1;
	}
	return lo;
}

/* Counter update of the lanes that found their k-mer (slot >= 0; every lane of the wave must call this together).
 * One 64-bit atomic per hit -- unless lanes of this wave hit the SAME counter (low-complexity input whose k-mer is a site
 * k-mer: every lane, every time): equal slots are added up inside the wave first.  Rounds: the lowest lane that still has
 * a hit broadcasts its slot, the lanes with that slot are counted by a ballot and leave, the lowest lane adds their number.
 * A round that finds a single lane ends the search (ordinary traffic: hits spread over 1.5 M counters, one round of ~8
 * scalar / vector instructions); whoever is left adds 1 by itself.  (A workgroup-wide LDS accumulator behind this was built and
 * measured out: inlined or as a call it pushed the k = 19 kernel from 122 VGPRs to 128 + scratch.)  src/FingerPrint.hpp:94-95 (`m_counts[*itr] += 1`
 * under `omp atomic`) with the same result: integer adds commute. */
__device__ __forceinline__ void ntsm_add_hits(const NtsmCountParams &p, long long slot, int lane)
{
	bool act = slot >= 0;
	unsigned long long am = __builtin_amdgcn_ballot_w64(act);
	while (am) {
		const int leader = __builtin_ctzll(am);
		const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) slot, leader);
		const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) ((unsigned long long) slot >> 32), leader);
		const bool same = act && (uint32_t) slot == lo && (uint32_t) ((unsigned long long) slot >> 32) == hi;
		const unsigned long long grp = __builtin_amdgcn_ballot_w64(same);
		const unsigned long long cnt = (unsigned long long) __popcll(grp);
		if (lane == leader)
			__hip_atomic_fetch_add(ntsm_count_ptr(p.keys, slot), p.sign * cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		act = act && !same;
		am &= ~grp;
		if (cnt == 1) break;
	}
	if (act) __hip_atomic_fetch_add(ntsm_count_ptr(p.keys, slot), p.sign, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int C, bool PER_READ>
__global__ __launch_bounds__(kThreads, PER_READ ? 3 : 4) void ntsm_count_kernel(const NtsmCountParams p)
{
	constexpr int ROW = C + 16;
	constexpr int VPT = C / 16;                              /* vectors per thread */
	__shared__ __attribute__((aligned(16))) uint8_t tile[(kThreads + 1) * ROW];
	__shared__ uint8_t lut[256];
	const int t = threadIdx.x;
	lut[t] = p.lut[t];

	const unsigned long long mask = p.mask;
	const uint32_t rv_shift = p.rv_shift, kmask = p.kmask, fshift = p.fshift, bshift = p.bshift;
	uint32_t nk = 0, nh = 0;

	for (unsigned long long ti = blockIdx.x; ti < p.n_tiles; ti += gridDim.x) {
		const long long ts = p.t0 + (long long) (ti * (unsigned long long) (kThreads * C));
		__syncthreads();                                     /* previous tile fully consumed */
		if (ts >= p.lo && ts + kThreads * C <= p.hi) {           /* interior tile: no boundary logic (~100 VALU instructions per vector) */
#pragma unroll
			for (int q = 0; q < VPT; ++q) {
				const int v = t + kThreads * q;
#if NTSM_STREAM_NT
				const u32x4 nt = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p.base + ts + 16ll * v));
#else
				const u32x4 nt = *reinterpret_cast<const u32x4 *>(p.base + ts + 16ll * v);
#endif
				const int row = 1 + (16 * v) / C, col = (16 * v) % C;
				*reinterpret_cast<uint4 *>(tile + row * ROW + col) = make_uint4(nt.x, nt.y, nt.z, nt.w);
			}
		} else {
#pragma unroll 1
			for (int q = 0; q < VPT; ++q) {
				const int v = t + kThreads * q;
				const uint4 r = ntsm_load_vec(p, ts + 16ll * v);
				const int row = 1 + (16 * v) / C, col = (16 * v) % C;
				*reinterpret_cast<uint4 *>(tile + row * ROW + col) = r;
			}
		}
		if (t < 2) {
			const uint4 r = ntsm_load_vec(p, ts - 32 + 16 * t);
			*reinterpret_cast<uint4 *>(tile + (C - 32) + 16 * t) = r;
		}
		__syncthreads();

		unsigned long long fw = 0, rv = 0;
		uint32_t inv = 0xFFFFFFFFu;
#define NTSM_ROLL(byte_)                                                                  \
		{                                                                                 \
			const uint32_t code_ = lut[(byte_)];                                          \
			const unsigned long long c_ = code_ & 3u;                                     \
			fw = ((fw << 2) | c_) & mask;                                                 \
			rv = (rv >> 2) | ((3ull - c_) << rv_shift);                                   \
			inv = (inv << 1) | (code_ >> 2);                                              \
		}
		{   /* warm-up on the 32 bytes in front of this thread's chunk (k - 1 <= 31 needed) */
			const uint8_t *prev = tile + t * ROW + (C - 32);
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const uint4 v = *reinterpret_cast<const uint4 *>(prev + 16 * h);
				const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
				for (int i = 0; i < 16; ++i) NTSM_ROLL((w[i >> 2] >> ((i & 3) * 8)) & 0xFFu)
			}
		}
		const uint8_t *own = tile + (t + 1) * ROW;
#pragma unroll 1
		for (int g = 0; g < 2 * VPT; ++g) {                  /* 8 positions per step: half the live registers of a 16-wide step */
			const uint2 v = *reinterpret_cast<const uint2 *>(own + 8 * g);
			const uint32_t w[2] = { v.x, v.y };
			unsigned long long cn[8];
			uint32_t hh[8], fwd[8];
			bool ok[8];
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				NTSM_ROLL((w[i >> 2] >> ((i & 3) * 8)) & 0xFFu)
				ok[i] = (inv & kmask) == 0;
				cn[i] = fw < rv ? fw : rv;
				hh[i] = ntsm_fold(cn[i]);
				const uint32_t bit = ntsm_h1(hh[i]) >> fshift;
				fwd[i] = p.filter[bit >> 5];                 /* always in range: unconditional, keeps 8 loads in flight */
			}
#pragma unroll
			for (int s = 0; s < 2; ++s) {
				uint4 ba[4], bb[4];
				bool pos[4];
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const int i = 4 * s + j;
					const uint32_t h1 = ntsm_h1(hh[i]);
					nk += ok[i] ? 1u : 0u;
					pos[j] = ok[i] && ((fwd[i] >> ((h1 >> fshift) & 31u)) & 1u);
					if (pos[j]) {
						ba[j] = *reinterpret_cast<const uint4 *>(p.keys + 4ull * (h1 >> bshift));
						bb[j] = *reinterpret_cast<const uint4 *>(p.keys + 4ull * (ntsm_h2(hh[i]) >> bshift));
					}
				}
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const int i = 4 * s + j;
					long long slot = -1;
					if (pos[j]) {
						const uint32_t klo = (uint32_t) cn[i], khi = (uint32_t) (cn[i] >> 32);
						const unsigned long long b1 = 2ull * (ntsm_h1(hh[i]) >> bshift);
						const unsigned long long b2 = 2ull * (ntsm_h2(hh[i]) >> bshift);
						if (ba[j].x == klo && ba[j].y == khi) slot = (long long) b1;
						else if (ba[j].z == klo && ba[j].w == khi) slot = (long long) b1 + 1;
						else if (bb[j].x == klo && bb[j].y == khi) slot = (long long) b2;
						else if (bb[j].z == klo && bb[j].w == khi) slot = (long long) b2 + 1;
						if (slot >= 0) {
							++nh;
							if (PER_READ) {
								const unsigned long long pb = (unsigned long long) (ts + (long long) t * C + 8 * g + i);
								atomicAdd(p.read_hits + ntsm_read_of(p, pb), 1u);
							}
						}
					}
					ntsm_add_hits(p, slot, t & 63);                   /* all lanes: equal slots inside the wave are added up first */
				}
			}
		}
#undef NTSM_ROLL
	}
	/* per-wave reduction, one 64-bit atomic per wave and counter */
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		nk += __shfl_down(nk, off, 64);
		nh += __shfl_down(nh, off, 64);
	}
	if ((t & 63) == 0) {
		if (nk) atomicAdd(p.totals + 0, p.sign * (unsigned long long) nk);
		if (nh) atomicAdd(p.totals + 1, p.sign * (unsigned long long) nh);
	}
}

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
#ifdef NTSM_WITH_TAB
constexpr int kListC = 128;                            /* list mode (tiles handed over by the tabulated kernel): always 32 KiB tiles */
#endif
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

#ifdef NTSM_WITH_TAB
	/* list mode: only the tiles the tabulated kernel handed over (tiles with bytes outside ACGTUNacgtun) */
	const unsigned long long n_iter = p.use_list ? (unsigned long long) min(*p.exotic_count, p.exotic_cap) : p.n_tiles;
#else
	const unsigned long long n_iter = p.n_tiles;
#endif
	for (unsigned long long it = blockIdx.x; it < n_iter; it += gridDim.x) {
#ifdef NTSM_WITH_TAB
		const unsigned long long ti = p.use_list ? (unsigned long long) p.exotic_list[it] : it;
		if (ti >= p.n_tiles) continue;
#else
		const unsigned long long ti = it;
#endif
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
				return TWO ? ntsm_mmer_hash_wide(cm) : ntsm_mmer_hash(cm);
			}
			const uint32_t fm = __builtin_amdgcn_alignbit(Fh, F, g_a2) & g_mmask;
			const uint32_t rm = (uint32_t) (((((unsigned long long) R) << 32) | Ro) >> g_rsh) & g_mmask;
			return TWO ? ntsm_mmer_hash_wide(min(fm, rm)) : ntsm_mmer_hash(min(fm, rm));   /* 14-mers / 12-mers */
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
#ifdef NTSM_ABLATION
			if (!(p.debug & 4u))
#endif
			ntsm_add_hits(p, slot_of_hit, lane);                   /* equal slots inside the wave are added up first */
			/* stage 2 */
			s2_v = s1_v && (((s1_pw >> (s1_g2 & 31u)) & (s1_pw >> ((s1_g2 >> 5) & 31u)) & 1u) != 0);
#ifdef NTSM_ABLATION
			s2_v = s2_v && !(p.debug & 2u);
#endif
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
#ifdef NTSM_ABLATION
				s1_v = s1_v && !(p.debug & 1u);
#endif
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
#ifdef NTSM_ABLATION
				/* 8: no lane requests a block (all out of range); 16: every lane requests block 0 (one request per load) */
				/* 32 / 64: only 9/16 (5/16) of the runs request their block -- what an on-chip minimizer-set test would leave */
				const bool keep = (p.debug & 32u) ? ((mz * 0x85EBCA6Bu) >> 28) < 9u : (p.debug & 64u) ? ((mz * 0x85EBCA6Bu) >> 28) < 5u : true;
				const uint32_t idx = (p.debug & 8u) ? 0xFFFFFFFFu : (p.debug & 16u) ? 0u : (__builtin_amdgcn_inverse_ballot_w64(ld) && keep) ? bi : 0xFFFFFFFFu;
#else
				const uint32_t idx = __builtin_amdgcn_inverse_ballot_w64(ld) ? bi : 0xFFFFFFFFu;
#endif
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
				phase_a(e, S, j0, gg, fh, pm, m2);
				if (TWO) phase_b(S);
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

/* dense[i] = count of slot_of[i]; tail = totals */
__global__ void ntsm_gather_kernel(const uint64_t *table, const uint32_t *slot_of, uint32_t n, unsigned long long *dense)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
		dense[i] = *ntsm_count_ptr(table, (long long) slot_of[i]);
}

/* Packed producer lanes (ntsm_lane_acquire_packed): 2-bit codes + 1 validity bit per stream position come over PCIe
 * (3/8 byte per position instead of 1); this kernel writes them out as the flat byte stream the count kernels read --
 * a code as the raw byte 0..3, which the reference's table accepts as such (vendor/KseqHashIterator.hpp:115), an invalid
 * position as 'N'.  One thread = 16 positions = one 16-byte store.  HBM-bound and tiny next to the link it relieves. */
__global__ void ntsm_unpack_kernel(const uint32_t *codes, const uint16_t *valid, uint4 *out, unsigned long long n16)
{
	for (unsigned long long i = blockIdx.x * (unsigned long long) blockDim.x + threadIdx.x; i < n16; i += (unsigned long long) gridDim.x * blockDim.x) {
		const uint32_t c = codes[i], v = valid[i];
		uint32_t w[4];
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			uint32_t x = 0;
#pragma unroll
			for (int b = 0; b < 4; ++b) {
				const int pos = 4 * q + b;
				const uint32_t byte = ((v >> pos) & 1u) ? ((c >> (2 * pos)) & 3u) : 0x4Eu;
				x |= byte << (8 * b);
			}
			w[q] = x;
		}
		out[i] = make_uint4(w[0], w[1], w[2], w[3]);
	}
}

/* key table image on the device: all buckets { empty, empty, 0, 0 }, then every key to its slot */
__global__ void ntsm_table_init_kernel(uint64_t *table, unsigned long long n_buckets)
{
	uint4 *t = reinterpret_cast<uint4 *>(table);
	for (unsigned long long b = blockIdx.x * (unsigned long long) blockDim.x + threadIdx.x; b < n_buckets; b += (unsigned long long) gridDim.x * blockDim.x) {
		t[2 * b] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
		t[2 * b + 1] = make_uint4(0u, 0u, 0u, 0u);
	}
}

__global__ void ntsm_table_scatter_kernel(uint64_t *table, const uint32_t *slot_of, const uint64_t *canon, uint32_t n)
{
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const unsigned long long slot = slot_of[i];
		table[4 * (slot >> 1) + (slot & 1)] = canon[i];
	}
}

__global__ void ntsm_zero_counts_kernel(uint64_t *table, unsigned long long n_buckets)
{
	for (unsigned long long b = blockIdx.x * (unsigned long long) blockDim.x + threadIdx.x; b < n_buckets; b += (unsigned long long) gridDim.x * blockDim.x) {
		table[4 * b + 2] = 0;
		table[4 * b + 3] = 0;
	}
}

} // namespace

/* ============================================================================================
 * Host code: context, table build, batching, C ABI
 * ========================================================================================== */

static thread_local int g_last_hip = 0;

#define HIPCHK(call)                                                                      \
	do {                                                                                  \
		hipError_t e_ = (call);                                                           \
		if (e_ != hipSuccess) {                                                           \
			g_last_hip = (int) e_;                                                        \
			fprintf(stderr, "ntsm_hip: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
			return NTSM_ERR_HIP;                                                          \
		}                                                                                 \
	} while (0)

namespace {

constexpr int kTileC = 128;                 /* stream bytes per thread and tile */
#ifndef NTSM_TABLE_LOAD
#define NTSM_TABLE_LOAD 0.25                   /* cuckoo key table: slots >= keys / load, power of two.  A sparser table costs memory, not time
                                                * (the Infinity Cache holds 128 MiB as well as 64): what a fuller one costs is the second-bucket probe of
                                                * a key whose first bucket is full -- 903 / 889 / 856 / 833 Gbases/s at load <= 0.25 / 0.4 / 0.6 / 0.8 */
#endif
#ifdef NTSM_WITH_TAB
#ifndef NTSM_TAB_SEG_TILES
#define NTSM_TAB_SEG_TILES 16384
#endif
#ifndef NTSM_LOOK_WGS
#define NTSM_LOOK_WGS 256                       /* look-up workgroups per launch */
#endif
#ifndef NTSM_TAB_TILES_PER_WG
#define NTSM_TAB_TILES_PER_WG 1
#endif
constexpr uint64_t kTabSegTiles = NTSM_TAB_SEG_TILES;    /* tabulated path: 64 KiB tiles per launch segment (1 GiB); its look-up kernel runs beside the next segment */
#endif
constexpr int kTimingPool = 256;

/* Process-wide pool of pinned host memory (ntsm_staging_pool): pinning costs ~0.4 ms/MiB and the driver serialises
 * it, so staging slots are carved out of one early allocation instead of being pinned one by one. */
struct PinnedPool {
	std::mutex mu;
	uint8_t *base = nullptr;
	uint64_t size = 0, bump = 0, outstanding = 0;
	std::vector<std::pair<uint64_t, uint64_t>> free_list;      /* (offset, bytes) of returned pieces, reused by exact size */
};
PinnedPool g_pool;

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
constexpr int kMaxDevices = 64;
struct StreamPool {
	std::mutex mu;
	std::vector<hipStream_t> idle[kMaxDevices];
};
StreamPool g_streams;

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

struct Slot {
	uint8_t *h_bases = nullptr, *d_bases = nullptr;
	uint8_t *d_packed = nullptr;               /* packed lanes: device copy of codes + validity bits (3/8 byte per position) */
	uint64_t *h_read_end = nullptr, *d_read_end = nullptr;
	uint64_t h_bases_bytes = 0, h_ends_bytes = 0;
	uint64_t d_bases_bytes = 0, d_packed_bytes = 0;   /* sizes of the device buffers (device_cache of the context) */
	bool ends_on_device = true;                /* false (lanes): read_end never leaves the host, plain malloc */
	hipStream_t stream = nullptr;
	hipEvent_t done = nullptr;                 /* last use of the host buffer finished */
	bool busy = false, acquired = false;
};

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

bool ntsm_wants_two_level(uint64_t n_keys) { return (12ull * n_keys + 127) / 128 > (9ull << 15); }   /* 12 bits per key > 4.5 MiB */

uint64_t mask_for_k(int k) { return k >= 32 ? 0ull : ((1ull << (2 * k)) - 1); }   /* k = 32: see header */

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
	uint32_t *d_prefilter = nullptr;           /* second-level Bloom used by the drain */
	uint32_t prefilter_log2 = 0;               /* log2(bits) */
	NtsmBlockMap blk_map = { 1 };
#ifdef NTSM_WITH_TAB
	/* tabulated k = 19 path (ntsm_tab_kernel.inc) */
	NtsmTabEntry *d_tab = nullptr;
	uint4 *d_tblocks = nullptr;
	uint64_t n_tblocks = 0;
	NtsmTabMap tblk_map = { 0, 0, 0 };
	bool tab_ok = false;                       /* filter built and no site k-mer has the reserved minimizer key */
	/* per launch stream, tabulated path: d_ctl = { [0] fill of the exotic-tile list of the launch in flight, [1 .. 1+2*seg_cap)
	 * fills of the look-up queue slots (tile, wave), two halves, [1+2*seg_cap] exotic tiles seen so far, then the exotic list };
	 * d_queue = canonical codes handed from the tabulated kernel to the look-up kernel (one segment at a time) */
	struct StreamBuf {
		hipStream_t stream; uint32_t *d_ctl; uint64_t seg_cap, exotic_cap; unsigned long long *d_queue; uint64_t queue_cap;
		hipEvent_t ev_tab[2], ev_look[2];          /* two halves of queue + fills: segment s uses half s & 1 */
	};
	std::vector<StreamBuf> sbuf;
	int look_blocks = 0;                       /* tuning: look-up workgroups per CU (0 = 1) */
	hipStream_t lstream = nullptr;             /* the look-up kernels of all launch streams run here, beside the next segment's scan */
#endif
	uint64_t n_launch[3] = { 0, 0, 0 };         /* count launches by kernel: tabulated, minimizer-blocked, generic */
	int kernel_variant = 0;                    /* 0 auto (minimizer-blocked kernel for 13 <= k <= 31), 1 generic, 2 = 0, 3 tabulated k = 19 kernel (NTSM_WITH_TAB builds) */
	std::vector<uint64_t> canon;               /* host copy of the canonical keys */
	std::vector<uint32_t> slot_of;
	/* batching */
	Slot slot[2];
	int next_slot = 0;
	uint64_t cap_bytes = 64ull << 20, cap_reads = 1ull << 20;
	hipStream_t rstream = nullptr;             /* stream for resident batches */
	/* host-side totals */
	uint64_t total_bases = 0, reads_consumed = 0;
	bool early_stop = false, reduced = false;
	bool failed = false;                       /* a table rebuild failed half way (ntsm_set_kernel / ntsm_set_tuning): the tables no longer
	                                            * describe one consistent filter organisation; every later call answers NTSM_ERR_STATE */
	uint64_t red_totals[4] = { 0, 0, 0, 0 };
	/* tuning / timing */
	int grid_blocks = 0, n_cu = 256;
	bool timing = false;
	hipEvent_t ev_a[kTimingPool], ev_b[kTimingPool];
	bool ev_used[kTimingPool];
	int ev_next = 0;
	uint64_t t_launches = 0;
	double t_ms = 0;
	/* producer lanes (ntsm_lane_*): several host threads feeding this context */
	std::mutex mu;                             /* guards open_lanes, lane_stream, the lane totals fold-in, the timing pool and device_cache */
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
	Slot slot[2];
	int next_slot = 0;
	uint64_t cap_bytes = 0, cap_reads = 0;
	bool packed_only = false;                  /* ntsm_lane_open_packed: pinned slots of 3/8 byte per position, no byte batches */
	uint64_t total_bases = 0, reads_consumed = 0;     /* folded into the context by ntsm_lane_close */
};

namespace {

/* Host part first (no HIP call: ntsm_create runs it while another thread may still be bringing the runtime up -- 0.2 s on
 * this stack, during which every HIP call of this thread would only wait), then `before_upload` (ntsm_create: device checks
 * and hipSetDevice), then the uploads. */
int build_tables(ntsm_ctx *c, int filter_log2_req, int (*before_upload)(ntsm_ctx *) = nullptr)
{
	const uint32_t n = c->n_kmers;
	/* The four structures are independent functions of the key set: built on four host threads (the cuckoo table
	 * dominates), uploaded afterwards. */
	std::vector<uint64_t> keys;
	std::vector<uint32_t> filter, blocks /* 4 words per block */, prefilter;
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
	c->two_level = ntsm_fast_plan((uint32_t) c->k, true).m == NTSM_TWO_M && c->kernel_variant != 1 &&
		(c->kernel_variant == 4 || (c->kernel_variant == 0 && filter_log2_req == 0 && ntsm_wants_two_level(n)));
	const NtsmFastPlan plan = ntsm_fast_plan((uint32_t) c->k, c->two_level);
	std::vector<uint32_t> bloom;
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
#ifdef NTSM_ABLATION                                              /* tools/ablate*.py builds only: wrong counts */
			if (getenv("NTSM_DEBUG_ZERO_FILTER")) std::fill(blocks.begin(), blocks.end(), 0u);
#endif
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
#ifdef NTSM_ABLATION
			if (const char *ev = getenv("NTSM_PREFILTER_LOG2")) pl = (uint32_t) atoi(ev);
#endif
			if (pl < 10) pl = 10;
			if (pl > 30) pl = 30;
			c->prefilter_log2 = pl;
			prefilter.assign((1ull << pl) / 32, pass_all ? 0xFFFFFFFFu : 0u);
			const uint32_t pshift = 32 - (pl - 5);
			for (uint32_t i = 0; i < n && !pass_all; ++i) {
				const uint32_t f = ntsm_fold(c->canon[i]), g1 = ntsm_h1(f), g2 = ntsm_h2(f);
				prefilter[g1 >> pshift] |= (1u << (g2 & 31u)) | (1u << ((g2 >> 5) & 31u));
			}
#ifdef NTSM_ABLATION
			if (getenv("NTSM_PREFILTER_OFF")) std::fill(prefilter.begin(), prefilter.end(), 0xFFFFFFFFu);   /* everything passes */
#endif
		}
	};
#ifdef NTSM_WITH_TAB
	std::vector<uint32_t> tblocks;
	auto build_tblocks = [&]() {
		/* k = 19, tabulated kernel: the same kind of filter (128-bit blocks, 4 bits per key, >= 12 bits per key, 3 MiB for
		 * the human set) addressed and filled with the table-driven hashes of ntsm_device.h */
		c->tab_ok = false;
		if (c->k != NTSM_FAST_K) return;
		NtsmTabEntry tab[256];
		ntsm_tab_build(tab);
		uint32_t e = 6, mult = 1;
		if (filter_log2_req >= 100 && filter_log2_req <= 130) {
			mult = 3; e = (uint32_t) (filter_log2_req - 100) - 7;
		} else if (filter_log2_req >= 10 && filter_log2_req <= 30) {
			e = (uint32_t) filter_log2_req - 7;
		} else {
			const uint64_t want = (12ull * n + 127) / 128;
			while ((1ull << e) < want && e < 24) ++e;
			if (e > 8 && e - 2 <= 20 && (3ull << (e - 2)) >= want) { mult = 3; e -= 2; }
		}
		if (e < 4) e = 4;
		if (mult == 3 && e > 20) e = 20;
		if (e > 24) e = 24;
		c->n_tblocks = (uint64_t) mult << e;
		c->tblk_map.mult3 = mult == 3;
		c->tblk_map.mask = ((1u << e) - 1u) << 4;
		c->tblk_map.shift = mult == 3 ? 20 - e : 0;
		tblocks.assign(c->n_tblocks * 4, 0u);
		bool ok = true;
		for (uint32_t i = 0; i < n; ++i) {
			uint32_t mz, h;
			ntsm_tab_kmer(tab, c->canon[i], &mz, &h);
			if (mz == 0xFFFFFFFFu) ok = false;                    /* the key the kernel reserves for "window invalid" */
			uint32_t *blk = &tblocks[(size_t) (ntsm_tab_block_off(mz, c->tblk_map) >> 4) * 4];
			for (int w = 0; w < 4; ++w) blk[w] |= 1u << ntsm_tab_bit(h, w);
		}
		c->tab_ok = ok;
	};
#else
	auto build_tblocks = []() {};
#endif
	{
		std::thread t1(build_filter), t2(build_blocks), t3(build_prefilter), t4(build_tblocks);
		build_cuckoo();
		t1.join(); t2.join(); t3.join(); t4.join();
	}
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
	if (!bloom.empty()) {
		HIPCHK(hipMalloc(&c->d_bloom, bloom.size() * sizeof(uint32_t)));
		HIPCHK(hipMemcpy(c->d_bloom, bloom.data(), bloom.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	}
	if (!prefilter.empty()) {
		HIPCHK(hipMalloc(&c->d_prefilter, prefilter.size() * sizeof(uint32_t)));
		HIPCHK(hipMemcpy(c->d_prefilter, prefilter.data(), prefilter.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	}
	if (!blocks.empty()) {
		HIPCHK(hipMalloc(&c->d_blocks, blocks.size() * sizeof(uint32_t)));
		HIPCHK(hipMemcpy(c->d_blocks, blocks.data(), blocks.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	}
#ifdef NTSM_WITH_TAB
	if (c->d_tblocks) (void) hipFree(c->d_tblocks);
	c->d_tblocks = nullptr;
	if (!tblocks.empty()) {
		HIPCHK(hipMalloc(&c->d_tblocks, tblocks.size() * sizeof(uint32_t)));
		HIPCHK(hipMemcpy(c->d_tblocks, tblocks.data(), tblocks.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
		if (!c->d_tab) {
			NtsmTabEntry tab[256];
			ntsm_tab_build(tab);
			HIPCHK(hipMalloc(&c->d_tab, sizeof tab));
			HIPCHK(hipMemcpy(c->d_tab, tab, sizeof tab, hipMemcpyHostToDevice));
		}
	}
#endif
	if (c->d_filter) (void) hipFree(c->d_filter);
	if (c->d_keys) (void) hipFree(c->d_keys);
	if (c->d_slot_of) (void) hipFree(c->d_slot_of);
	c->d_filter = nullptr; c->d_keys = nullptr; c->d_slot_of = nullptr;
	HIPCHK(hipMalloc(&c->d_filter, filter.size() * sizeof(uint32_t)));
	HIPCHK(hipMalloc(&c->d_keys, 2 * c->n_slots * sizeof(uint64_t)));   /* { key0, key1, count0, count1 } per bucket */
	HIPCHK(hipMalloc(&c->d_slot_of, (n ? n : 1) * sizeof(uint32_t)));
	HIPCHK(hipMemcpy(c->d_filter, filter.data(), filter.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	if (n) HIPCHK(hipMemcpy(c->d_slot_of, c->slot_of.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
	{
		/* The bucket image { key0, key1, count0, count1 } is laid out on the device: every bucket starts empty with zeroed
		 * counters, then the n keys are scattered to their slots -- 12 bytes per key cross PCIe instead of 32 bytes per
		 * bucket of a table that is 2/3 empty, and the host never builds the image (64 MiB for the human set, 512 MiB for
		 * 16 M keys). */
		hipLaunchKernelGGL(ntsm_table_init_kernel, dim3(2048), dim3(256), 0, 0, c->d_keys, (unsigned long long) (c->n_slots / 2));
		HIPCHK(hipGetLastError());
		if (n) {
			uint64_t *d_canon = nullptr;
			HIPCHK(hipMalloc(&d_canon, (uint64_t) n * sizeof(uint64_t)));
			hipError_t e1 = hipMemcpy(d_canon, c->canon.data(), (uint64_t) n * sizeof(uint64_t), hipMemcpyHostToDevice);
			if (e1 == hipSuccess) {
				hipLaunchKernelGGL(ntsm_table_scatter_kernel, dim3(1024), dim3(256), 0, 0, c->d_keys, c->d_slot_of, d_canon, n);
				e1 = hipGetLastError();
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
	return hipMalloc(p, bytes);
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

int alloc_slot(Slot &s, int device, uint64_t cap_bytes, uint64_t cap_reads, bool ends_on_device, bool packed_only = false, ntsm_ctx *cache = nullptr)
{
	s.ends_on_device = ends_on_device;
	s.h_bases_bytes = (packed_only ? (cap_bytes & ~31ull) / 4 + (cap_bytes & ~31ull) / 8 : cap_bytes) + 64;   /* packed: 3/8 byte per position */
	s.h_ends_bytes = cap_reads * sizeof(uint64_t);
	s.h_bases = (uint8_t *) pool_alloc(s.h_bases_bytes);
	if (!s.h_bases) HIPCHK(hipHostMalloc(&s.h_bases, s.h_bases_bytes, hipHostMallocPortable));
	if (ends_on_device) {
		s.h_read_end = (uint64_t *) pool_alloc(s.h_ends_bytes);
		if (!s.h_read_end) HIPCHK(hipHostMalloc(&s.h_read_end, s.h_ends_bytes, hipHostMallocPortable));
		HIPCHK(hipMalloc(&s.d_read_end, s.h_ends_bytes));
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
	if (!s.done) HIPCHK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
	s.busy = false;
	s.acquired = false;
	return NTSM_OK;
}

void free_slot(Slot &s, ntsm_ctx *cache = nullptr)
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
	const uint64_t tile = (uint64_t) kThreads * kTileC;
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
#ifdef NTSM_ABLATION
	{
		static const char *dv = getenv("NTSM_DEBUG_KERNEL");
		if (dv) p.debug = (uint32_t) atoi(dv);
	}
#endif
	p.pf_shift = 32 - (c->prefilter_log2 - 5);
	p.blk_bytes = (uint32_t) (c->n_blocks * 16);
	const NtsmFastPlan plan = ntsm_fast_plan((uint32_t) c->k, c->two_level);
	const bool fast = plan.mode >= 0 && c->d_blocks && c->kernel_variant != 1;
	p.bloom = c->d_bloom;
	p.bloom_words = c->n_bloom_words;
	p.fk_k = plan.k; p.fk_m2 = 2 * plan.m; p.fk_a2 = 2 * plan.a;
#ifdef NTSM_WITH_TAB
	const bool tab = fast && !per_read && c->tab_ok && c->d_tblocks && c->kernel_variant == 3;
#else
	constexpr bool tab = false;
#endif
	if (fast && !tab) {                                     /* the minimizer-blocked kernels cut the stream into their own tiles */
		const uint64_t ftile = (uint64_t) kThreads * kFastC;
		p.n_tiles = (hi - (uint64_t) p.t0 + ftile - 1) / ftile;
	}
#ifdef NTSM_WITH_TAB
	NtsmCountParams pt = p;                                 /* the tabulated kernel's view: 64 KiB tiles, segments of kTabSegTiles */
	uint64_t tab_tiles = 0, tab_segs = 0;
	ntsm_ctx::StreamBuf tab_sb = { nullptr, nullptr, 0, 0, nullptr, 0, { nullptr, nullptr }, { nullptr, nullptr } };
	if (tab) {
		tab_tiles = (hi - (uint64_t) p.t0 + (uint64_t) kTabTile - 1) / (uint64_t) kTabTile;
		tab_segs = (tab_tiles + kTabSegTiles - 1) / kTabSegTiles;
		pt.tab = c->d_tab;
		pt.tblocks = c->d_tblocks;
		pt.tblk_bytes = (uint32_t) (c->n_tblocks * 16);
		pt.tblk_map = c->tblk_map;
		const uint64_t need_exotic = 2 * tab_tiles + 2;
		/* queue: kTabQueue codes per (tile, wave) of the largest segment -- one window in 16; a wave with more positives
		 * than that looks them up in line */
		const uint64_t seg_slots = std::min<uint64_t>(tab_tiles, kTabSegTiles) * (uint64_t) (kThreads / 64);
		const uint64_t need_queue = seg_slots * (uint64_t) kTabQueue;
		{
			std::lock_guard<std::mutex> lk(c->mu);
			ntsm_ctx::StreamBuf *sb = nullptr;
			for (auto &b : c->sbuf) if (b.stream == st) sb = &b;
			if (!sb) {
				ntsm_ctx::StreamBuf nb = { st, nullptr, 0, 0, nullptr, 0, { nullptr, nullptr }, { nullptr, nullptr } };
				for (int q = 0; q < 2; ++q) {
					HIPCHK(hipEventCreateWithFlags(&nb.ev_tab[q], hipEventDisableTiming));
					HIPCHK(hipEventCreateWithFlags(&nb.ev_look[q], hipEventDisableTiming));
				}
				c->sbuf.push_back(nb);
				sb = &c->sbuf.back();
			}
			if (!c->lstream && !(c->lstream = stream_get(c->device))) return NTSM_ERR_HIP;
			if (sb->exotic_cap < need_exotic || sb->seg_cap < seg_slots) {   /* first launch on this stream, or a bigger batch than ever */
				uint32_t seen = 0;
				if (sb->d_ctl) {
					HIPCHK(hipStreamSynchronize(st));
					HIPCHK(hipMemcpy(&seen, sb->d_ctl + 1 + 2 * sb->seg_cap, sizeof seen, hipMemcpyDeviceToHost));
					HIPCHK(hipFree(sb->d_ctl));
					sb->d_ctl = nullptr;
				}
				const uint64_t ecap = std::max<uint64_t>(need_exotic, 1024), scap = std::max<uint64_t>(seg_slots, 16);
				HIPCHK(hipMalloc(&sb->d_ctl, (2 + 2 * scap + ecap) * sizeof(uint32_t)));
				HIPCHK(hipMemcpy(sb->d_ctl + 1 + 2 * scap, &seen, sizeof seen, hipMemcpyHostToDevice));
				sb->exotic_cap = ecap;
				sb->seg_cap = scap;
			}
			if (sb->queue_cap < need_queue) {
				if (sb->d_queue) {
					HIPCHK(hipStreamSynchronize(st));
					HIPCHK(hipFree(sb->d_queue));
					sb->d_queue = nullptr;
				}
				HIPCHK(hipMalloc(&sb->d_queue, 2 * need_queue * sizeof(unsigned long long)));
				sb->queue_cap = need_queue;
			}
			pt.exotic_count = sb->d_ctl;
			pt.pos_count = sb->d_ctl + 1;
			pt.exotic_seen = sb->d_ctl + 1 + 2 * sb->seg_cap;
			pt.exotic_list = sb->d_ctl + 2 + 2 * sb->seg_cap;
			tab_sb = *sb;
			pt.exotic_cap = (uint32_t) std::min<uint64_t>(sb->exotic_cap, 0xFFFFFFFFu);
			pt.pos_queue = sb->d_queue;
			pt.pos_cap = 0;
		}
		p.exotic_count = pt.exotic_count;
		p.exotic_list = pt.exotic_list;
		p.exotic_cap = pt.exotic_cap;
		p.use_list = 1;
	}
#endif
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
#ifdef NTSM_WITH_TAB
	if (tab) {
		/* the tabulated kernel counts every tile whose bytes are all ACGTUNacgtun and lists the others; the exact
		 * minimizer-blocked kernel then walks that list (usually empty: it exits at once) */
		HIPCHK(hipMemsetAsync(pt.exotic_count, 0, sizeof(uint32_t), st));   /* fill of the exotic-tile list */
		for (uint64_t sg = 0; sg < tab_segs; ++sg) {
			const int hb = (int) (sg & 1);                        /* half of queue + fills this segment uses */
			NtsmCountParams ps = pt;
			ps.tile_base = sg * kTabSegTiles;
			ps.t0 = pt.t0 + (long long) (ps.tile_base * (uint64_t) kTabTile);
			ps.n_tiles = std::min<uint64_t>(kTabSegTiles, tab_tiles - ps.tile_base);
			ps.pos_count = pt.pos_count + (uint64_t) hb * tab_sb.seg_cap;
			ps.pos_queue = pt.pos_queue + (uint64_t) hb * tab_sb.queue_cap;
			if (sg >= 2) HIPCHK(hipStreamWaitEvent(st, tab_sb.ev_look[hb], 0));   /* the half's previous look-up has finished */
			const uint64_t tgrid = c->grid_blocks > 0 ? std::min<uint64_t>((uint64_t) c->grid_blocks, ps.n_tiles) : std::max<uint64_t>(1, ps.n_tiles / NTSM_TAB_TILES_PER_WG);
			if (ps.tblk_map.mult3) hipLaunchKernelGGL((ntsm_count_tab19_kernel<true>), dim3((unsigned) tgrid), dim3(kThreads), 0, st, ps);
			else hipLaunchKernelGGL((ntsm_count_tab19_kernel<false>), dim3((unsigned) tgrid), dim3(kThreads), 0, st, ps);
			HIPCHK(hipGetLastError());
			HIPCHK(hipEventRecord(tab_sb.ev_tab[hb], st));
			HIPCHK(hipStreamWaitEvent(c->lstream, tab_sb.ev_tab[hb], 0));
			/* one look-up workgroup per CU: its four waves fit into the registers the scan kernel's three waves per SIMD leave */
			const uint64_t lgrid = std::min<uint64_t>((uint64_t) NTSM_LOOK_WGS, std::max<uint64_t>(1, ps.n_tiles));
			hipLaunchKernelGGL(ntsm_lookup_kernel, dim3((unsigned) lgrid), dim3(kThreads), 0, c->lstream, ps);
			HIPCHK(hipGetLastError());
			HIPCHK(hipEventRecord(tab_sb.ev_look[hb], c->lstream));
		}
		/* whatever follows on the launch stream (the list walker, the timing event, the caller's sync) sees the look-ups done */
		HIPCHK(hipStreamWaitEvent(st, tab_sb.ev_look[(tab_segs - 1) & 1], 0));
		if (tab_segs > 1) HIPCHK(hipStreamWaitEvent(st, tab_sb.ev_look[tab_segs & 1], 0));
		hipLaunchKernelGGL((ntsm_count_mz_kernel<0, false, kListC, false>), dim3((unsigned) std::min<uint64_t>((uint64_t) c->n_cu * 4, p.n_tiles)), dim3(kThreads), 0, st, p);
		c->n_launch[0]++;
	} else
#endif
	if (fast) {
		const dim3 g((unsigned) grid), b(kThreads);
#define NTSM_MZ_CASE(M_) \
		case 2 * M_: hipLaunchKernelGGL((ntsm_count_mz_kernel<M_, false, kFastC, false>), g, b, 0, st, p); break; \
		case 2 * M_ + 1: hipLaunchKernelGGL((ntsm_count_mz_kernel<M_, true, kFastC, false>), g, b, 0, st, p); break;
#define NTSM_MZ2_CASE(M_) \
		case 2 * M_: hipLaunchKernelGGL((ntsm_count_mz_kernel<M_, false, kFastC, true>), g, b, 0, st, p); break; \
		case 2 * M_ + 1: hipLaunchKernelGGL((ntsm_count_mz_kernel<M_, true, kFastC, true>), g, b, 0, st, p); break;
		if (c->two_level) {
			switch (plan.mode * 2 + (per_read ? 1 : 0)) {
			NTSM_MZ2_CASE(0) NTSM_MZ2_CASE(2) NTSM_MZ2_CASE(3) NTSM_MZ2_CASE(4) NTSM_MZ2_CASE(5) NTSM_MZ2_CASE(6) NTSM_MZ2_CASE(7) NTSM_MZ2_CASE(8) NTSM_MZ2_CASE(9)
			default: return NTSM_ERR_STATE;
			}
		} else
		switch (plan.mode * 2 + (per_read ? 1 : 0)) {
		NTSM_MZ_CASE(0) NTSM_MZ_CASE(2) NTSM_MZ_CASE(3) NTSM_MZ_CASE(4) NTSM_MZ_CASE(5) NTSM_MZ_CASE(6) NTSM_MZ_CASE(7) NTSM_MZ_CASE(8) NTSM_MZ_CASE(9)
		default: return NTSM_ERR_STATE;
		}
#undef NTSM_MZ_CASE
#undef NTSM_MZ2_CASE
		c->n_launch[1]++;
	}
	else if (per_read) {
		hipLaunchKernelGGL((ntsm_count_kernel<kTileC, true>), dim3((unsigned) grid), dim3(kThreads), 0, st, p);
		c->n_launch[2]++;
	} else {
		hipLaunchKernelGGL((ntsm_count_kernel<kTileC, false>), dim3((unsigned) grid), dim3(kThreads), 0, st, p);
		c->n_launch[2]++;
	}
	HIPCHK(hipGetLastError());
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
		HIPCHK(hipMalloc(&c->d_read_hits, hits_cap * sizeof(uint32_t)));
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

int check_layout(const uint64_t *read_end, uint32_t n_reads, uint64_t n_bytes)
{
	if (n_reads == 0) return n_bytes == 0 ? NTSM_OK : NTSM_ERR_ARG;
	if (!read_end || read_end[n_reads - 1] + 1 != n_bytes) return NTSM_ERR_ARG;
	return NTSM_OK;
}

int submit_slot(ntsm_ctx *c, Slot &s, uint64_t n_bytes, uint32_t n_reads)
{
	if (n_reads == 0) return NTSM_OK;
	HIPCHK(hipMemcpyAsync(s.d_bases, s.h_bases, n_bytes, hipMemcpyHostToDevice, s.stream));
	if (c->armed) {
		HIPCHK(hipMemcpyAsync(s.d_read_end, s.h_read_end, n_reads * sizeof(uint64_t), hipMemcpyHostToDevice, s.stream));
		return armed_batch(c, s.stream, s.d_bases, n_bytes, s.d_read_end, s.h_read_end, n_reads);
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

/* RCCL is bound on first use: the library is 570 MB of code objects that the HIP runtime would otherwise map and
 * register in every process that links it (~0.1 s and 1.4 GB of RSS for a single-GPU ntsmCount). */
struct Rccl {
	decltype(&ncclCommInitAll) CommInitAll = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	bool ok = false;
	Rccl()
	{
		void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
		if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
		if (!h) return;
		CommInitAll = (decltype(CommInitAll)) dlsym(h, "ncclCommInitAll");
		GroupStart = (decltype(GroupStart)) dlsym(h, "ncclGroupStart");
		GroupEnd = (decltype(GroupEnd)) dlsym(h, "ncclGroupEnd");
		AllReduce = (decltype(AllReduce)) dlsym(h, "ncclAllReduce");
		CommDestroy = (decltype(CommDestroy)) dlsym(h, "ncclCommDestroy");
		ok = CommInitAll && GroupStart && GroupEnd && AllReduce && CommDestroy;
	}
};
const Rccl &rccl_bind()
{
	static Rccl rccl;                                     /* thread-safe one-time binding */
	return rccl;
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

int ntsm_last_hip_error(void) { return g_last_hip; }
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
			g_last_hip = (int) e;
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
	if (hipMalloc(&c->d_lut, 256) != hipSuccess) return fail(NTSM_ERR_HIP);
	if (hipMemcpy(c->d_lut, lut, 256, hipMemcpyHostToDevice) != hipSuccess) return fail(NTSM_ERR_HIP);
	{
		uint2 lut64[256];
		for (int i = 0; i < 256; ++i)
			lut64[i] = lut[i] < 4 ? make_uint2((uint32_t) lut[i], (3u - lut[i]) | 0x10000u) : make_uint2(0u, 3u);   /* { code, complement | valid << 16 } */
		if (hipMalloc(&c->d_lut64, sizeof lut64) != hipSuccess) return fail(NTSM_ERR_HIP);
		if (hipMemcpy(c->d_lut64, lut64, sizeof lut64, hipMemcpyHostToDevice) != hipSuccess) return fail(NTSM_ERR_HIP);
	}
	if (hipMalloc(&c->d_totals, 4 * sizeof(uint64_t)) != hipSuccess) return fail(NTSM_ERR_HIP);
	if (hipMemset(c->d_totals, 0, 4 * sizeof(uint64_t)) != hipSuccess) return fail(NTSM_ERR_HIP);
	if (hipDeviceSynchronize() != hipSuccess) return fail(NTSM_ERR_HIP);
	if (hipMalloc(&c->d_vec, ((uint64_t) n_kmers + 4) * sizeof(uint64_t)) != hipSuccess) return fail(NTSM_ERR_HIP);
	if (!(c->rstream = stream_get(device))) return fail(NTSM_ERR_HIP);
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
	}
	stream_put(c->device, c->rstream);
	for (hipStream_t st : c->lane_stream) stream_put(c->device, st);
	for (int i = 0; i < kTimingPool; ++i) {
		if (c->ev_a[i]) (void) hipEventDestroy(c->ev_a[i]);
		if (c->ev_b[i]) (void) hipEventDestroy(c->ev_b[i]);
	}
#ifdef NTSM_WITH_TAB
	for (auto &b : c->sbuf) {
		if (b.d_ctl) (void) hipFree(b.d_ctl);
		if (b.d_queue) (void) hipFree(b.d_queue);
		for (int q = 0; q < 2; ++q) {
			if (b.ev_tab[q]) (void) hipEventDestroy(b.ev_tab[q]);
			if (b.ev_look[q]) (void) hipEventDestroy(b.ev_look[q]);
		}
	}
	stream_put(c->device, c->lstream);
	if (c->d_tab) (void) hipFree(c->d_tab);
	if (c->d_tblocks) (void) hipFree(c->d_tblocks);
#endif
	for (auto &b : c->device_cache) (void) hipFree(b.first);
	c->device_cache.clear();
	void *ptrs[] = { c->d_bloom, c->d_prefilter, c->d_lut64, c->d_blocks, c->d_filter, c->d_keys, c->d_slot_of, c->d_read_hits, c->d_totals, c->d_vec, c->d_lut };
	for (void *p : ptrs) if (p) (void) hipFree(p);
	delete c;
}

int ntsm_set_batch_capacity(ntsm_ctx *c, uint64_t cap_bytes, uint64_t cap_reads)
{
	if (!c || cap_bytes < 4096 || cap_reads < 16) return NTSM_ERR_ARG;
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
	HIPCHK(hipSetDevice(c->device));
	Slot &s = c->slot[c->next_slot];
	if (!s.h_bases) {
		int rc = alloc_slot(s, c->device, c->cap_bytes, c->cap_reads, true);
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
		g_last_hip = (int) e;
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

int ntsm_staging_pool(uint64_t bytes)
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
	HIPCHK(hipHostMalloc(&p, bytes, hipHostMallocPortable));
	g_pool.base = (uint8_t *) p;
	g_pool.size = bytes;
	g_pool.bump = 0;
	return NTSM_OK;
}

static int lane_open(ntsm_ctx *c, uint64_t cap_bytes, uint64_t cap_reads, bool packed_only, ntsm_lane **out);

int ntsm_lane_open(ntsm_ctx *c, uint64_t cap_bytes, uint64_t cap_reads, ntsm_lane **out) { return lane_open(c, cap_bytes, cap_reads, false, out); }
int ntsm_lane_open_packed(ntsm_ctx *c, uint64_t cap_positions, ntsm_lane **out) { return lane_open(c, cap_positions, 16, true, out); }

static int lane_open(ntsm_ctx *c, uint64_t cap_bytes, uint64_t cap_reads, bool packed_only, ntsm_lane **out)
{
	if (!c || !out) return NTSM_ERR_ARG;
	*out = nullptr;
	if (c->armed) return NTSM_ERR_STATE;                  /* -m is defined on ONE ordered stream of reads */
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

int ntsm_lane_submit(ntsm_lane *l, uint64_t n_bytes, uint32_t n_reads)
{
	if (!l) return NTSM_ERR_ARG;
	Slot &s = l->slot[l->next_slot];
	if (!s.acquired) return NTSM_ERR_STATE;
	s.acquired = false;
	if (n_bytes > l->cap_bytes || n_reads > l->cap_reads) return NTSM_ERR_ARG;
	int rc = check_layout(s.h_read_end, n_reads, n_bytes);
	if (rc) return rc;
	if (n_reads == 0) return NTSM_OK;
	ntsm_ctx *c = l->c;
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipMemcpyAsync(s.d_bases, s.h_bases, n_bytes, hipMemcpyHostToDevice, s.stream));
	rc = launch_count(c, s.stream, s.d_bases, 0, n_bytes, nullptr, 0, false, +1);
	if (rc) return rc;
	HIPCHK(hipEventRecord(s.done, s.stream));
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
	Slot &s = l->slot[l->next_slot];
	if (!s.acquired || !s.d_packed) return NTSM_ERR_STATE;
	s.acquired = false;
	const uint64_t cap_pos = l->cap_bytes & ~31ull;
	if ((n_positions & 7) || n_positions > cap_pos || n_bases + n_reads > n_positions) return NTSM_ERR_ARG;
	if (n_reads == 0 || n_positions == 0) return NTSM_OK;
	ntsm_ctx *c = l->c;
	HIPCHK(hipSetDevice(c->device));
	/* whole groups of 32 positions cross the link and are unpacked: what lies between the end of the batch and the next
	 * multiple of 32 is marked invalid here (the caller may have left anything there) */
	const uint64_t n_out = (n_positions + 31) & ~31ull;
	uint8_t *h_valid = s.h_bases + cap_pos / 4;
	for (uint64_t p = n_positions; p < n_out; p += 8) h_valid[p >> 3] = 0;
	HIPCHK(hipMemcpyAsync(s.d_packed, s.h_bases, n_out / 4, hipMemcpyHostToDevice, s.stream));
	HIPCHK(hipMemcpyAsync(s.d_packed + cap_pos / 4, h_valid, n_out / 8, hipMemcpyHostToDevice, s.stream));
	const uint64_t n16 = n_out / 16;
	hipLaunchKernelGGL(ntsm_unpack_kernel, dim3((unsigned) std::min<uint64_t>(4096, (n16 + 255) / 256)), dim3(256), 0, s.stream,
			(const uint32_t *) s.d_packed, (const uint16_t *) (s.d_packed + cap_pos / 4), (uint4 *) s.d_bases, (unsigned long long) n16);
	HIPCHK(hipGetLastError());
	int rc = launch_count(c, s.stream, s.d_bases, 0, n_out, nullptr, 0, false, +1);
	if (rc) return rc;
	HIPCHK(hipEventRecord(s.done, s.stream));
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
	int rc = NTSM_OK;
	if (hipSetDevice(c->device) != hipSuccess) rc = NTSM_ERR_HIP;
	for (auto &s : l->slot) {                            /* the stream is shared: wait for this lane's own batches only */
		if (s.busy && hipEventSynchronize(s.done) != hipSuccess) rc = NTSM_ERR_HIP;
		s.busy = false;
		free_slot(s, c);                                  /* device buffers go to the context's cache (nothing of this lane is in flight any more) */
		if (s.done) (void) hipEventDestroy(s.done);
	}
	{
		std::lock_guard<std::mutex> lk(c->mu);
		c->total_bases += l->total_bases;
		c->reads_consumed += l->reads_consumed;
		c->open_lanes--;
	}
	delete l;
	return rc;
}

int ntsm_submit(ntsm_ctx *c, const uint8_t *bases, uint64_t n_bytes, const uint64_t *read_end, uint32_t n_reads)
{
	if (!c || (n_bytes && !bases)) return NTSM_ERR_ARG;
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
	memcpy(hb, bases, n_bytes);
	memcpy(hr, read_end, (size_t) n_reads * sizeof(uint64_t));
	return ntsm_submit_staged(c, n_bytes, n_reads);
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
			hipLaunchKernelGGL(ntsm_gather_kernel, dim3(1024), dim3(256), 0, c->rstream, c->d_keys, c->d_slot_of, c->n_kmers, c->d_vec);
			HIPCHK(hipGetLastError());
		}
		const uint64_t tail[4] = { t.total_kmers, t.total_hits, t.total_bases, t.reads_consumed };
		HIPCHK(hipMemcpyAsync(c->d_vec + c->n_kmers, tail, sizeof tail, hipMemcpyHostToDevice, c->rstream));
		HIPCHK(hipStreamSynchronize(c->rstream));
	}
	if (d_vec) *d_vec = c->d_vec;
	if (n_words) *n_words = (uint64_t) c->n_kmers + 4;
	return NTSM_OK;
}

int ntsm_import_reduced(ntsm_ctx *c)
{
	if (!c) return NTSM_ERR_ARG;
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
	hipLaunchKernelGGL(ntsm_zero_counts_kernel, dim3(1024), dim3(256), 0, c->rstream, c->d_keys, (unsigned long long) (c->n_slots / 2));
	HIPCHK(hipGetLastError());
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

int ntsm_rccl_probe(void) { return rccl_bind().ok ? NTSM_OK : NTSM_ERR_RCCL; }

int ntsm_set_kernel(ntsm_ctx *c, int variant)
{
	if (!c || variant < 0 || variant > 4) return NTSM_ERR_ARG;
#ifndef NTSM_WITH_TAB
	if (variant == 3) return NTSM_ERR_ARG;                 /* the tabulated kernel is not part of this build (make tab) */
#endif
	if (variant == 4 && ntsm_fast_plan((uint32_t) c->k, true).m != NTSM_TWO_M) return NTSM_ERR_ARG;   /* 15 <= k <= 31 */
	int rc = ntsm_sync(c, nullptr);
	if (rc) return rc;
	const int before = c->kernel_variant;
	c->kernel_variant = variant;
	/* one-level and two-level filters are different tables (12-mer / 14-mer minimizers): a change of level rebuilds them */
	/* (the tabulated kernel hands its exotic tiles to the ONE-level k = 19 kernel: variant 3 on a context that had chosen two
	 * levels by itself rebuilds the one-level tables, otherwise those tiles would probe 14-mer-addressed blocks with 12-mers) */
	const bool want_two = ntsm_fast_plan((uint32_t) c->k, true).m == NTSM_TWO_M && variant != 1 && variant != 3 &&
		(variant == 4 || (variant == 0 && c->filter_log2_req == 0 && ntsm_wants_two_level(c->n_kmers)));
	(void) before;
	if (variant != 1 && want_two != c->two_level) {
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

int ntsm_debug_stats(ntsm_ctx *c, uint64_t out[8])
{
	if (!c || !out) return NTSM_ERR_ARG;
	int rc = ntsm_sync(c, nullptr);
	if (rc) return rc;
	HIPCHK(hipDeviceSynchronize());
	uint64_t exotic = 0;
#ifdef NTSM_WITH_TAB
	for (auto &b : c->sbuf) {
		uint32_t seen = 0;
		if (b.d_ctl) HIPCHK(hipMemcpy(&seen, b.d_ctl + 1 + 2 * b.seg_cap, sizeof seen, hipMemcpyDeviceToHost));
		exotic += seen;
	}
#endif
	out[0] = exotic;
	out[1] = c->n_launch[0];
	out[2] = c->n_launch[1];
	out[3] = c->n_launch[2];
	uint64_t dv[4] = { 0, 0, 0, 0 };
	HIPCHK(hipMemcpy(dv, c->d_totals, sizeof dv, hipMemcpyDeviceToHost));
	out[4] = dv[2];
	out[5] = c->two_level ? 1 : 0;
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
		const Rccl &rccl = rccl_bind();
		if (!rccl.ok) return NTSM_ERR_RCCL;
		std::vector<int> devs(n);
		for (int i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
		for (int i = 0; i < n; ++i)
			for (int j = 0; j < i; ++j)
				if (devs[i] == devs[j]) return NTSM_ERR_ARG;    /* one context per device */
		/* Communicators are kept per device list for the life of the process: ncclCommInitAll costs far more than the
		 * 12 MB reduction it serves (tens of milliseconds against well under one). */
		static std::mutex comm_mu;
		static std::vector<std::pair<std::vector<int>, std::vector<ncclComm_t>>> comm_cache;
		std::lock_guard<std::mutex> comm_lock(comm_mu);
		std::vector<ncclComm_t> *comms = nullptr;
		for (auto &e : comm_cache) if (e.first == devs) comms = &e.second;
		if (!comms) {
			std::vector<ncclComm_t> made(n);
			if (rccl.CommInitAll(made.data(), n, devs.data()) != ncclSuccess) return NTSM_ERR_RCCL;
			comm_cache.emplace_back(devs, made);
			comms = &comm_cache.back().second;
		}
		bool ok = rccl.GroupStart() == ncclSuccess;
		for (int i = 0; i < n && ok; ++i) {
			ok = hipSetDevice(devs[i]) == hipSuccess &&
				rccl.AllReduce(ctxs[i]->d_vec, ctxs[i]->d_vec, (size_t) ctxs[i]->n_kmers + 4, ncclUint64, ncclSum,
						(*comms)[i], ctxs[i]->rstream) == ncclSuccess;
		}
		ok = (rccl.GroupEnd() == ncclSuccess) && ok;
		for (int i = 0; i < n; ++i) {
			(void) hipSetDevice(devs[i]);
			if (hipStreamSynchronize(ctxs[i]->rstream) != hipSuccess) ok = false;
		}
		if (!ok) return NTSM_ERR_RCCL;
	}
	for (int i = 0; i < n; ++i) {
		int rc = ntsm_import_reduced(ctxs[i]);
		if (rc) return rc;
	}
	return NTSM_OK;
}

} // extern "C"
