/*
 * kernels_generic.hip -- the generic-k count kernel (k = 1 .. 12 and 32; any k on request) and the small helper kernels
 * (dense gather, unpack of packed lanes, key-table image, counter reset), with their launchers.
 *
 * ntsm_count_kernel replaces the inner loop of the reference's FingerPrint::insertCount (src/FingerPrint.hpp:89-103):
 * KseqHashIterator's rolling 2-bit canonical k-mer (vendor/KseqHashIterator.hpp:95-112), the tsl::robin_map lookup
 * (src/FingerPrint.hpp:92) and the `+= 1` (:94-99), batched over a flat stream of reads.
 *
 * Kernel structure (wave64, integer only, no MFMA -- DESIGN.md section 4.1):
 *   - a workgroup of 256 threads owns a tile of 256*C contiguous stream bytes; the tile is staged
 *     through LDS with coalesced 16-byte loads, rows padded by 16 B so that the per-thread
 *     ds_read_b128 of its own C-byte chunk is bank-conflict free;
 *   - each thread rolls fw / rc codes over its chunk (after warming up on the 32 bytes before it)
 *     and keeps a shift register of "invalid base" flags, so window validity is purely local and
 *     no k-mer can span the 'N' terminator between reads;
 *   - every valid window probes a 1-bit filter (L2 resident); the rare positives read their two
 *     16-byte cuckoo buckets and bump a 64-bit counter with one no-return atomic.
 * There is no CPU fallback anywhere in this library.
 */
#include "kernels_common.h"
#include "ntsm_internal.h"

namespace {

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

namespace ntsm_rt {

hipError_t launch_generic(const NtsmCountParams &p, unsigned grid, hipStream_t st, bool per_read)
{
	if (per_read) hipLaunchKernelGGL((ntsm_count_kernel<kTileC, true>), dim3(grid), dim3(kThreads), 0, st, p);
	else hipLaunchKernelGGL((ntsm_count_kernel<kTileC, false>), dim3(grid), dim3(kThreads), 0, st, p);
	return hipGetLastError();
}

hipError_t launch_gather(const uint64_t *table, const uint32_t *slot_of, uint32_t n, unsigned long long *dense, hipStream_t st)
{
	hipLaunchKernelGGL(ntsm_gather_kernel, dim3(1024), dim3(256), 0, st, table, slot_of, n, dense);
	return hipGetLastError();
}

hipError_t launch_unpack(const uint32_t *codes, const uint16_t *valid, void *out, unsigned long long n16, hipStream_t st)
{
	hipLaunchKernelGGL(ntsm_unpack_kernel, dim3((unsigned) std::min<uint64_t>(4096, (n16 + 255) / 256)), dim3(256), 0, st, codes, valid, (uint4 *) out, n16);
	return hipGetLastError();
}

hipError_t launch_table_init(uint64_t *table, unsigned long long n_buckets, hipStream_t st)
{
	hipLaunchKernelGGL(ntsm_table_init_kernel, dim3(2048), dim3(256), 0, st, table, n_buckets);
	return hipGetLastError();
}

hipError_t launch_table_scatter(uint64_t *table, const uint32_t *slot_of, const uint64_t *canon, uint32_t n, hipStream_t st)
{
	hipLaunchKernelGGL(ntsm_table_scatter_kernel, dim3(1024), dim3(256), 0, st, table, slot_of, canon, n);
	return hipGetLastError();
}

hipError_t launch_zero_counts(uint64_t *table, unsigned long long n_buckets, hipStream_t st)
{
	hipLaunchKernelGGL(ntsm_zero_counts_kernel, dim3(1024), dim3(256), 0, st, table, n_buckets);
	return hipGetLastError();
}

} // namespace ntsm_rt
