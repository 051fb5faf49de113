/*
 * tools/gather_bench.hip -- microbenchmarks that bound the count kernel's probe stage on gfx950:
 *   (1) random 4-byte gathers from tables of 256 KiB .. 256 MiB (L2 / Infinity Cache / HBM), all
 *       lanes active and ~1/4 of the lanes active (the "one probe per minimizer run" regime);
 *   (2) random 16-byte gathers (cuckoo buckets);
 *   (3) streaming 16-byte loads (HBM ceiling for the input stream);
 *   (4) integer VALU rate for a 40-op roll+hash body (compute ceiling).
 * Prints one line per experiment: name, table bytes, G accesses/s, GB/s.
 */
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int ACTIVE_SHIFT>
__global__ __launch_bounds__(256) void gather4(const uint32_t *tab, uint32_t mask, int iters, uint32_t *out)
{
	uint32_t x = blockIdx.x * 256 + threadIdx.x, acc = 0;
	const bool active = ACTIVE_SHIFT == 0 || ((threadIdx.x >> 0) & ((1 << ACTIVE_SHIFT) - 1)) == 0;
	for (int i = 0; i < iters; i += 8) {
		uint32_t v[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) { x = mix(x + 0x9E3779B9u); v[j] = active ? tab[x & mask] : 0; }
#pragma unroll
		for (int j = 0; j < 8; ++j) acc += v[j];
	}
	if (acc == 0x12345) out[0] = acc;
}

__global__ __launch_bounds__(256) void gather16(const uint4 *tab, uint32_t mask, int iters, uint32_t *out)
{
	uint32_t x = blockIdx.x * 256 + threadIdx.x, acc = 0;
	for (int i = 0; i < iters; i += 4) {
		uint4 v[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) { x = mix(x + 0x9E3779B9u); v[j] = tab[x & mask]; }
#pragma unroll
		for (int j = 0; j < 4; ++j) acc += v[j].x ^ v[j].w;
	}
	if (acc == 0x12345) out[0] = acc;
}

__global__ __launch_bounds__(256) void stream16(const uint4 *src, size_t n_vec, uint32_t *out)
{
	uint32_t acc = 0;
	for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n_vec; i += (size_t) gridDim.x * 256) {
		typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
		u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src) + i);
		acc += v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345) out[0] = acc;
}

__global__ __launch_bounds__(256) void valu40(int iters, uint32_t *out)
{
	uint32_t flo = threadIdx.x, fhi = 1, rlo = blockIdx.x, rhi = 3, inv = 0, acc = 0;
	for (int i = 0; i < iters; ++i) {
#pragma unroll
		for (int j = 0; j < 16; ++j) {
			uint32_t c = (flo >> 7) & 3;
			fhi = ((fhi << 2) | (flo >> 30)) & 0x3F; flo = (flo << 2) | c;
			rlo = (rlo >> 2) | (rhi << 30); rhi = (rhi >> 2) | ((3 - c) << 4);
			inv = (inv << 1) | (c >> 1);
			bool lt = fhi < rhi || (fhi == rhi && flo < rlo);
			uint32_t klo = lt ? flo : rlo, khi = lt ? fhi : rhi;
			uint32_t h = (klo ^ (khi * 0x85EBCA6Bu)) * 0x9E3779B1u;
			acc += ((inv & 0x7FFFF) == 0) ? (h >> 8) : 1;
		}
	}
	if (acc == 0x12345) out[0] = acc;
}

int main()
{
	uint32_t *out; CHK(hipMalloc(&out, 64));
	hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
	const size_t maxb = 256ull << 20;
	uint32_t *tab; CHK(hipMalloc(&tab, maxb)); CHK(hipMemset(tab, 1, maxb));
	const int grid = 256 * 8, iters = 4096;
	float ms;
	for (size_t bytes = 256 << 10; bytes <= maxb; bytes *= 2) {
		uint32_t mask = (uint32_t) (bytes / 4 - 1);
		for (int rep = 0; rep < 2; ++rep) {
			CHK(hipEventRecord(a)); hipLaunchKernelGGL(gather4<0>, dim3(grid), dim3(256), 0, 0, tab, mask, iters, out); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
		}
		CHK(hipEventElapsedTime(&ms, a, b));
		double g = (double) grid * 256 * iters / (ms * 1e-3) / 1e9;
		printf("gather4_all   table=%8zu KiB  %8.1f G/s\n", bytes >> 10, g);
		for (int rep = 0; rep < 2; ++rep) {
			CHK(hipEventRecord(a)); hipLaunchKernelGGL(gather4<2>, dim3(grid), dim3(256), 0, 0, tab, mask, iters, out); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
		}
		CHK(hipEventElapsedTime(&ms, a, b));
		g = (double) grid * 64 * iters / (ms * 1e-3) / 1e9;
		printf("gather4_1of4  table=%8zu KiB  %8.1f G/s (active lanes)\n", bytes >> 10, g);
		uint32_t mask16 = (uint32_t) (bytes / 16 - 1);
		for (int rep = 0; rep < 2; ++rep) {
			CHK(hipEventRecord(a)); hipLaunchKernelGGL(gather16, dim3(grid), dim3(256), 0, 0, (const uint4 *) tab, mask16, iters / 4, out); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
		}
		CHK(hipEventElapsedTime(&ms, a, b));
		g = (double) grid * 256 * (iters / 4) / (ms * 1e-3) / 1e9;
		printf("gather16_all  table=%8zu KiB  %8.1f G/s\n", bytes >> 10, g);
	}
	{
		const size_t sb = 8ull << 30;
		uint4 *src; CHK(hipMalloc(&src, sb)); CHK(hipMemset(src, 1, sb));
		for (int rep = 0; rep < 3; ++rep) {
			CHK(hipEventRecord(a)); hipLaunchKernelGGL(stream16, dim3(256 * 16), dim3(256), 0, 0, src, sb / 16, out); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
		}
		CHK(hipEventElapsedTime(&ms, a, b));
		printf("stream16      bytes=%zu MiB  %8.1f GB/s\n", sb >> 20, sb / (ms * 1e-3) / 1e9);
		CHK(hipFree(src));
	}
	{
		const int it = 2048;
		for (int rep = 0; rep < 2; ++rep) {
			CHK(hipEventRecord(a)); hipLaunchKernelGGL(valu40, dim3(256 * 16), dim3(256), 0, 0, it, out); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
		}
		CHK(hipEventElapsedTime(&ms, a, b));
		printf("valu_roll_hash  %8.1f G positions/s\n", (double) 256 * 16 * 256 * it * 16 / (ms * 1e-3) / 1e9);
	}
	return 0;
}
