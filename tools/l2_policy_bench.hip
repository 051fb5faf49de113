/*
 * tools/l2_policy_bench.hip -- random 16-byte gathers from an L2-sized table (the first-level filter's access
 * pattern) under every cache-policy bit combination of buffer_load (aux: 1 = sc0, 2 = nt, 16 = sc1) and several
 * lane-sharing patterns.  Question it answers: is the ~266 G requests/s cap of profiles/r01_gather_microbench.txt a
 * property of the L1 fill path (128-byte lines per miss), and does bypassing the L1 raise it?
 * Prints G lane-requests/s (active lanes only).  Not on the product path.
 */
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

/* SHARE: lanes l .. l+SHARE-1 use the same address.  ACTIVE_SHIFT: one lane in 2^ACTIVE_SHIFT issues a request, the
 * others pass an out-of-range offset (no memory request), as the count kernel does for lanes that keep their block. */
template <int AUX, int SHARE, int ACTIVE_SHIFT, int WIDTH>
__global__ __launch_bounds__(256) void gather(const void *tab, uint32_t bytes, int iters, uint32_t *out)
{
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(tab), 0, (int) bytes, 0x00020000);
	uint32_t x = (blockIdx.x * 256 + threadIdx.x) / SHARE * 2654435761u, acc = 0;
	const bool active = (threadIdx.x & ((1 << ACTIVE_SHIFT) - 1)) == 0;
	const uint32_t mask = bytes - 1;
	for (int i = 0; i < iters; i += 8) {
		uint32_t v[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			x = mix(x + 0x9E3779B9u);
			const uint32_t off = active ? (x & mask & ~15u) : 0xFFFFFFF0u;
			if (WIDTH == 16) {
				const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int) off, 0, AUX);
				v[j] = r.x ^ r.w;
			} else {
				v[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int) off, 0, AUX);
			}
		}
#pragma unroll
		for (int j = 0; j < 8; ++j) acc += v[j];
	}
	if (acc == 0x12345) out[0] = acc;
}

template <int AUX, int SHARE, int ACTIVE_SHIFT, int WIDTH>
int run(const char *label, const void *tab, uint32_t bytes, uint32_t *out)
{
	hipEvent_t a, b;
	CHK(hipEventCreate(&a));
	CHK(hipEventCreate(&b));
	const int grid = 256 * 8, iters = 2048;
	float ms = 0;
	for (int rep = 0; rep < 2; ++rep) {
		CHK(hipEventRecord(a));
		hipLaunchKernelGGL((gather<AUX, SHARE, ACTIVE_SHIFT, WIDTH>), dim3(grid), dim3(256), 0, 0, tab, bytes, iters, out);
		CHK(hipEventRecord(b));
		CHK(hipEventSynchronize(b));
	}
	CHK(hipEventElapsedTime(&ms, a, b));
	const double lanes = (double) grid * 256 / (1 << ACTIVE_SHIFT) * iters;
	printf("%-44s table=%5u KiB  %8.1f G lane-requests/s  %8.1f G distinct/s\n", label, bytes >> 10, lanes / (ms * 1e-3) / 1e9,
			lanes / SHARE / (ms * 1e-3) / 1e9);
	fflush(stdout);
	return 0;
}

int main()
{
	uint32_t *out;
	CHK(hipMalloc(&out, 64));
	void *tab;
	CHK(hipMalloc(&tab, 64u << 20));
	CHK(hipMemset(tab, 1, 64u << 20));
	for (uint32_t bytes : { 1u << 20, 2u << 20, 4u << 20, 32u << 20 }) {
#define R(AUX, SHARE, ACT, W, LABEL) if (run<AUX, SHARE, ACT, W>(LABEL, tab, bytes, out)) return 1;
		R(0, 1, 2, 16, "b128 aux=0        1of4 lanes")
		R(1, 1, 2, 16, "b128 aux=1 sc0    1of4 lanes")
		R(2, 1, 2, 16, "b128 aux=2 nt     1of4 lanes")
		R(3, 1, 2, 16, "b128 aux=3 sc0 nt 1of4 lanes")
		R(16, 1, 2, 16, "b128 aux=16 sc1   1of4 lanes")
		R(17, 1, 2, 16, "b128 aux=17 sc0 sc1 1of4 lanes")
		R(18, 1, 2, 16, "b128 aux=18 sc1 nt 1of4 lanes")
		R(19, 1, 2, 16, "b128 aux=19 all   1of4 lanes")
		R(0, 1, 0, 16, "b128 aux=0        all lanes")
		R(2, 1, 0, 16, "b128 aux=2 nt     all lanes")
		R(16, 1, 0, 16, "b128 aux=16 sc1   all lanes")
		R(0, 1, 2, 4, "b32  aux=0        1of4 lanes")
		R(2, 1, 2, 4, "b32  aux=2 nt     1of4 lanes")
		R(16, 1, 2, 4, "b32  aux=16 sc1   1of4 lanes")
		R(0, 4, 0, 16, "b128 aux=0  4 lanes share an address")
		R(0, 8, 0, 16, "b128 aux=0  8 lanes share an address")
		R(0, 64, 0, 16, "b128 aux=0 64 lanes share an address")
#undef R
	}
	return 0;
}
