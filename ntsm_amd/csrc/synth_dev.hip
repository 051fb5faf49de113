/*
 * synth_dev.hip -- on-device fills of the synthetic read streams (include/ntsm_synth.h), so that
 * bench.py can materialise the 1e9-read configuration (150 GB) directly in HBM.  Byte-identical
 * to the host twins in synth_host.cpp (same header-only generator, synth.h).
 */
#include <hip/hip_runtime.h>
#include "../../include/ntsm_synth.h"

namespace {

__global__ __launch_bounds__(256) void short_fill_kernel(ntsm_synth_short p, const unsigned char *windows,
		unsigned long long g0, unsigned long long n, unsigned char *out)
{
	const unsigned long long n_vec = (n + 15) / 16;
	for (unsigned long long v = blockIdx.x * 256ull + threadIdx.x; v < n_vec; v += (unsigned long long) gridDim.x * 256ull) {
		uint32_t w[4] = { 0, 0, 0, 0 };
		const unsigned long long base = v * 16;
		if (base + 16 <= n) {
#pragma unroll
			for (int b = 0; b < 16; ++b)
				w[b >> 2] |= (uint32_t) ntsm_synth_short_byte(&p, windows, g0 + base + b) << ((b & 3) * 8);
			*reinterpret_cast<uint4 *>(out + base) = make_uint4(w[0], w[1], w[2], w[3]);
		} else {
			for (unsigned long long b = base; b < n; ++b) out[b] = ntsm_synth_short_byte(&p, windows, g0 + b);
		}
	}
}

__global__ __launch_bounds__(256) void long_fill_kernel(ntsm_synth_long p, const unsigned char *windows,
		unsigned long long r0, unsigned long long n_reads, const unsigned long long *read_end,
		unsigned long long n_bytes, unsigned char *out)
{
	for (unsigned long long g = blockIdx.x * 256ull + threadIdx.x; g < n_bytes; g += (unsigned long long) gridDim.x * 256ull) {
		unsigned long long lo = 0, hi = n_reads;           /* first read with read_end >= g */
		while (lo < hi) {
			unsigned long long mid = (lo + hi) >> 1;
			if (read_end[mid] >= g) hi = mid; else lo = mid + 1;
		}
		if (lo >= n_reads) { out[g] = 'N'; continue; }
		const unsigned long long start = lo ? read_end[lo - 1] + 1 : 0;
		if (g == read_end[lo]) { out[g] = 'N'; continue; }
		out[g] = ntsm_synth_long_byte(&p, windows, r0 + lo, (uint32_t) (read_end[lo] - start), (uint32_t) (g - start));
	}
}

} // namespace

extern "C" {

int ntsm_synth_short_fill_device(const ntsm_synth_short *p, const void *d_windows, uint64_t g0, uint64_t n,
		void *d_out, void *stream)
{
	if (!p || !d_out || ((uintptr_t) d_out & 15)) return -1;
	if (n == 0) return 0;
	unsigned long long n_vec = (n + 15) / 16;
	unsigned grid = (unsigned) ((n_vec + 255) / 256 > 65536 ? 65536 : (n_vec + 255) / 256);
	hipLaunchKernelGGL(short_fill_kernel, dim3(grid), dim3(256), 0, (hipStream_t) stream, *p,
			(const unsigned char *) d_windows, g0, n, (unsigned char *) d_out);
	hipError_t e = hipGetLastError();
	return e == hipSuccess ? 0 : -(int) e;
}

int ntsm_synth_long_fill_device(const ntsm_synth_long *p, const void *d_windows, const void *d_qtable257,
		uint64_t r0, uint64_t n_reads, const void *d_read_end, uint64_t n_bytes, void *d_out, void *stream)
{
	(void) d_qtable257;
	if (!p || !d_out || !d_read_end) return -1;
	if (n_bytes == 0) return 0;
	unsigned grid = (unsigned) ((n_bytes + 255) / 256 > 65536 ? 65536 : (n_bytes + 255) / 256);
	hipLaunchKernelGGL(long_fill_kernel, dim3(grid), dim3(256), 0, (hipStream_t) stream, *p,
			(const unsigned char *) d_windows, r0, n_reads, (const unsigned long long *) d_read_end, n_bytes,
			(unsigned char *) d_out);
	hipError_t e = hipGetLastError();
	return e == hipSuccess ? 0 : -(int) e;
}

} // extern "C"
