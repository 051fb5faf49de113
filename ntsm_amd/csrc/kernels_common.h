/*
 * kernels_common.h -- what the gfx950 count kernels share: launch geometry, the boundary-safe 16-byte stream load, the
 * key-table slot arithmetic, per-read attribution and the in-wave sum of equal hits.  Device code only; included by
 * kernels_generic.hip and kernels_mz.hip inside their anonymous namespaces' translation units.
 *
 * Replaces nothing of the reference by itself: the kernels built on it replace the loop of FingerPrint::insertCount
 * (src/FingerPrint.hpp:89-103).
 */
#ifndef NTSM_KERNELS_COMMON_H
#define NTSM_KERNELS_COMMON_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ntsm_device.h"
#include "ntsm_hooks.h"

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

} // namespace
#endif
