// tools/ubench/op_rate.hip -- issue rate of the integer vector instructions the count kernels are made of, on gfx950: time per wave64
// instruction and SIMD with 4 waves per SIMD resident and 8 independent chains per wave (every one of them,
// the 32-bit multiplies included, issues at the full rate -- v_mul_lo_u32 5.4 against 4.3-5.1 "cycles" at the nominal clock).
//   hipcc --offload-arch=gfx950 -O3 -o build/op_rate tools/ubench/op_rate.hip && gpurun -- ./build/op_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP> __global__ void k(uint32_t *out, uint32_t c, int iters)
{
	uint32_t a[8];
	unsigned long long m = 0x5555555555555555ull ^ (unsigned long long) iters, mm[2] = { 0, 0 };
	for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
			if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
			if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
			if (OP == 3) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(c));
			if (OP == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(c));
			if (OP == 5) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
			if (OP == 6) asm volatile("v_alignbit_b32 %0, %0, %1, 2" : "+v"(a[i]) : "v"(c));
			if (OP == 7) asm volatile("v_bfrev_b32 %0, %0" : "+v"(a[i]));
			if (OP == 8) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(c));
			if (OP == 10) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(a[(i + 1) & 7]));
			if (OP == 11) asm volatile("v_cndmask_b32_e64 %0, %0, 7, %1" : "+v"(a[i]) : "s"(m));
			if (OP == 12) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "s"((uint32_t) m));
			if (OP == 13) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a[i]) : "v"(c));
			if (OP == 14) asm volatile("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(mm[i & 1]) : "v"(a[i]), "v"(c));
			if (OP == 15) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(c), "v"(a[(i + 1) & 7]));
			if (OP == 16) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(a[(i + 1) & 7]));
			if (OP == 9) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(unsigned long long *) &a[i & 6]) : "v"(c), "v"(a[7]) : "vcc");
		}
	}
	uint32_t s = 0;
	for (int i = 0; i < 8; ++i) s += a[i];
	s += (uint32_t) (mm[0] ^ mm[1]);
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name)
{
	uint32_t *d; (void) hipMalloc(&d, 256 * 4 * 256 * 4 * 4);
	const int iters = 20000, grid = 256 * 4;   // 4 workgroups of 256 threads per CU: 4 waves per SIMD
	hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
	k<OP><<<grid, 256>>>(d, 0x9E3779B1u, 100); (void) hipDeviceSynchronize();
	(void) hipEventRecord(e0); k<OP><<<grid, 256>>>(d, 0x9E3779B1u, iters); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
	float ms; (void) hipEventElapsedTime(&ms, e0, e1);
	int clk = 0; (void) hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
	// per SIMD: 4 waves x iters x 8 instrs
	double instr = 4.0 * iters * 8; double cycles = ms * 1e-3 * clk * 1e3;
	printf("%-16s %8.3f ms  %.2f cycles per wave instruction (clock %d kHz)\n", name, ms, cycles / instr, clk);
	(void) hipFree(d);
}
int main()
{
	run<0>("v_add_u32"); run<1>("v_mul_lo_u32"); run<2>("v_mul_hi_u32"); run<3>("v_mul_u32_u24"); run<4>("v_mad_u32_u24"); run<5>("v_min_u32");
	run<6>("v_alignbit_b32"); run<7>("v_bfrev_b32"); run<8>("v_lshl_or_b32"); run<9>("v_mad_u64_u32");
	run<10>("v_min3_u32"); run<11>("v_cndmask_e64"); run<12>("v_mbcnt_lo"); run<13>("v_lshlrev_sdwa"); run<14>("v_cmp_e64->sgpr"); run<15>("v_bfi_b32"); run<16>("v_or3_b32");
	return 0;
}
