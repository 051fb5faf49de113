#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
__device__ u32x4v llvm_struct_buffer_load_v4(i32x4v rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v4i32");
__global__ void k(const uint4 *a, unsigned *o, const unsigned *ix, unsigned nrec) {
	const unsigned long long b = (unsigned long long) a;
	i32x4v rv = { (int) (unsigned) b, (int) ((unsigned) (b >> 32) | (16u << 16)), (int) nrec, 0x00020000 };
	u32x4v v = llvm_struct_buffer_load_v4(rv, (int) ix[threadIdx.x], 0, 0, 0);
	unsigned c = ix[threadIdx.x + 64], e = ix[threadIdx.x + 128];
	asm("v_mad_u32_u16 %0, %1, %2, 1 op_sel:[0,1,0,0]" : "=v"(c) : "v"(c), "v"(e));
	o[threadIdx.x * 2] = v.x + v.y + v.z + v.w;
	o[threadIdx.x * 2 + 1] = c;
}
int main() {
	const unsigned nrec = 1000;
	std::vector<uint4> a(nrec + 64);
	for (unsigned i = 0; i < a.size(); ++i) a[i] = make_uint4(i, 0, 0, 0);
	std::vector<unsigned> ix(192);
	for (int i = 0; i < 64; ++i) { ix[i] = i * 17; ix[64 + i] = i + 1; ix[128 + i] = (i & 1) << 16 | 3; }
	ix[5] = 999; ix[6] = 1000; ix[7] = 0xFFFFFFFFu; ix[8] = 0x10000000u; ix[9] = 1001;
	uint4 *da; unsigned *dix, *dout;
	hipMalloc(&da, a.size() * 16); hipMalloc(&dix, 192 * 4); hipMalloc(&dout, 128 * 4);
	hipMemcpy(da, a.data(), a.size() * 16, hipMemcpyHostToDevice); hipMemcpy(dix, ix.data(), 192 * 4, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dout, dix, nrec);
	std::vector<unsigned> out(128);
	hipMemcpy(out.data(), dout, 128 * 4, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int i = 0; i < 64; ++i) {
		const unsigned want = ix[i] < nrec ? ix[i] : 0, wc = (i + 1) * (i & 1) + 1;
		if (out[2 * i] != want || out[2 * i + 1] != wc) { printf("lane %d idx %u: got %u want %u; c got %u want %u\n", i, ix[i], out[2 * i], want, out[2 * i + 1], wc); ++bad; }
	}
	printf("struct buffer load + mad_u32_u16: %s\n", bad ? "MISMATCH" : "ok");
	return bad != 0;
}
