/*
 * tools/valu_rate.hip -- issue-rate microbenchmark for the instructions the count kernel is built from (gfx950).
 *
 * For every instruction kind: W workgroups of 256 threads per CU (W = waves per SIMD, capped with dynamic LDS), each
 * wave runs a long stream of INDEPENDENT instructions of that kind (8 destination registers round robin) and stamps
 * s_memtime around it.  Printed: SIMD cycles per wave-instruction = (t1 - t0) / (instructions per wave * W), median
 * over waves, for W = 1, 2, 4, 8.  A full-rate wave64 VALU op on a SIMD-32 reads 2.0 at W >= 2 and 4.0 at W = 1.
 * Not on the product path; results are kept under profiles/.
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define I8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define X16(S) S S S S S S S S S S S S S S S S
constexpr int kPerIter = 128;                 /* instructions per loop iteration: I8 x 16 */

#define KERNEL(NAME, F)                                                                                     \
	__global__ __launch_bounds__(256) void NAME(unsigned long long *out, int iters, unsigned long long *rt, unsigned long long *hwid)                       \
	{                                                                                                       \
		extern __shared__ uint32_t lds[];                                                                   \
		uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 7, a4 = a0 * 5, a5 = a0 | 64,   \
				a6 = a0 + 9, a7 = ~a0;                                                                      \
		uint32_t b = (blockIdx.x * 2654435761u) | 1u, c = threadIdx.x * 40503u + 17u;                       \
		const unsigned long long m = __ballot((threadIdx.x * 7 + blockIdx.x) & 1);                          \
		for (int q = threadIdx.x; q < 2048; q += 256) lds[q] = c + q;                                                                             \
		__syncthreads();                                                                                    \
		const uint32_t la = (threadIdx.x * 16) & 4095;                                                      \
		const uint32_t sc = __builtin_amdgcn_readfirstlane(b);                                              \
		unsigned long long w2 = c;                                                                          \
		u32x4 w4 = { c, c, c, c };                                                                          \
		const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                     \
		const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
		for (int it = 0; it < iters; ++it)                                                                  \
			asm volatile(X16(I8(F))                                                                         \
					: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)        \
					: "v"(b), "v"(c), "s"(m), "v"(la), "s"(sc), "v"(w2), "v"(w4)                            \
					: "vcc", "memory");                                                                     \
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
		const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
		const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                     \
		if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;                  \
		if ((threadIdx.x & 63) == 0) {                                                                      \
			uint32_t hw, xcc;                                                                               \
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                \
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                              \
			hwid[(blockIdx.x * 256 + threadIdx.x) >> 6] = ((unsigned long long) xcc << 32) | hw;            \
		}                                                                                                   \
		if (threadIdx.x == 0 && blockIdx.x == 0) rt[0] = ((t1 - t0) << 32) | (r1 - r0);                      \
		if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t) w2 ^ w4.x) == 0x12345678u) out[0] = 1;                             \
	}

/* operands: %0..%7 destinations (also sources), %8 = b, %9 = c, %10 = s[..] lane mask, %11 = LDS byte address */
#define F_ADD(i)      "v_add_u32 %" #i ", %" #i ", %8\n"
#define F_XOR(i)      "v_xor_b32 %" #i ", %" #i ", %8\n"
#define F_LSHL(i)     "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define F_ALIGNC(i)   "v_alignbit_b32 %" #i ", %" #i ", %8, 6\n"
#define F_ALIGNV(i)   "v_alignbit_b32 %" #i ", %" #i ", %8, %9\n"
#define F_LSHLOR(i)   "v_lshl_or_b32 %" #i ", %" #i ", 2, %8\n"
#define F_ANDOR(i)    "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define F_OR3(i)      "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define F_ADD3(i)     "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define F_BFE(i)      "v_bfe_u32 %" #i ", %" #i ", 5, 3\n"
#define F_BFEI(i)     "v_bfe_i32 %" #i ", %" #i ", 5, 1\n"
#define F_BFI(i)      "v_bfi_b32 %" #i ", %8, %" #i ", %9\n"
#define F_PERM(i)     "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define F_MIN(i)      "v_min_u32 %" #i ", %" #i ", %8\n"
#define F_MIN3(i)     "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
#define F_MUL24(i)    "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define F_MAD24(i)    "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define F_MULLO(i)    "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define F_MULHI(i)    "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define F_MULHI24(i)  "v_mul_hi_u32_u24 %" #i ", %" #i ", %8\n"
#define F_CNDVCC(i)   "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define F_CNDS(i)     "v_cndmask_b32 %" #i ", %" #i ", %8, %10\n"
#define F_CMPVCC(i)   "v_cmp_ne_u32 vcc, %" #i ", %8\n"
#define F_CMPCND(i)   "v_cmp_lt_u32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define F_DOT4(i)     "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
#define F_SDWA(i)     "v_lshrrev_b32_sdwa %" #i ", %8, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n"
#define F_PKMIN(i)    "v_pk_min_u16 %" #i ", %" #i ", %8\n"
#define F_PKADD(i)    "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define F_PKLSHR(i)   "v_pk_lshrrev_b16 %" #i ", 3, %" #i "\n"
#define F_PKMUL(i)    "v_pk_mul_lo_u16 %" #i ", %" #i ", %8\n"
#define F_DPPMOV(i)   "v_mov_b32_dpp %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_DPPMIN(i)   "v_min_u32_dpp %" #i ", %8, %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_DPPWSHR(i)  "v_min_u32_dpp %" #i ", %8, %" #i " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_BCNT(i)     "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define F_MBCNT(i)    "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i "\n"
#define F_ADDC(i)     "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define F_SAD(i)      "v_sad_u8 %" #i ", %" #i ", %8, %9\n"
#define F_DSR32(i)    "ds_read_b32 %" #i ", %11 offset:" #i "*4\n"
#define F_DSR64(i)    "s_nop 0\n"
#define F_SALU(i)     "s_and_b64 vcc, vcc, %10\n"
#define F_SALU2(i)    "s_lshl_b32 vcc_lo, vcc_lo, 1\n"
#define F_MIXVS(i)    "v_add_u32 %" #i ", %" #i ", %8\ns_and_b64 vcc, vcc, %10\n"
#define F_MIXV4S(i)   "v_add_u32 %" #i ", %" #i ", %8\nv_xor_b32 %" #i ", %" #i ", %9\nv_add_u32 %" #i ", %" #i ", %8\nv_xor_b32 %" #i ", %" #i ", %9\ns_and_b64 vcc, vcc, %10\n"
#define F_AND(i)      "v_and_b32 %" #i ", %" #i ", %8\n"
#define F_OR(i)       "v_or_b32 %" #i ", %" #i ", %8\n"
#define F_SUB(i)      "v_sub_u32 %" #i ", %" #i ", %8\n"
#define F_SUBREV(i)   "v_subrev_u32 %" #i ", %" #i ", %8\n"
#define F_MOV(i)      "v_mov_b32 %" #i ", %8\n"
#define F_NOT(i)      "v_not_b32 %" #i ", %" #i "\n"
#define F_LSHR(i)     "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define F_LSHLV(i)    "v_lshlrev_b32 %" #i ", %8, %" #i "\n"
#define F_ASHR(i)     "v_ashrrev_i32 %" #i ", 3, %" #i "\n"
#define F_MAX(i)      "v_max_u32 %" #i ", %" #i ", %8\n"
#define F_ADDCO(i)    "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define F_ADDE64(i)   "v_add_u32_e64 %" #i ", %" #i ", %8\n"
#define F_ADDLIT(i)   "v_add_u32 %" #i ", 0x12345, %" #i "\n"
#define F_ADDINL(i)   "v_add_u32 %" #i ", 7, %" #i "\n"
#define F_ANDLIT(i)   "v_and_b32 %" #i ", 0x03030303, %" #i "\n"
#define F_XORS(i)     "v_xor_b32 %" #i ", %12, %" #i "\n"
#define F_LSHLADD(i)  "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define F_ADDLSHL(i)  "v_add_lshl_u32 %" #i ", %" #i ", %8, 2\n"
#define F_XAD(i)      "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define F_MULI24(i)   "v_mul_i32_i24 %" #i ", %" #i ", %8\n"
#define F_ADDF(i)     "v_add_f32 %" #i ", %" #i ", %8\n"
#define F_FMAF(i)     "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define F_XNOR(i)     "v_xnor_b32 %" #i ", %" #i ", %8\n"
#define F_CMPS(i)     "v_cmp_ne_u32_e64 %10, %" #i ", %8\n"
#define F_FFBL(i)     "v_ffbl_b32 %" #i ", %" #i "\n"
#define F_SDWAADD(i)  "v_add_u32_sdwa %" #i ", %8, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define F_DSR64B(i)   "ds_read_b64 %13, %11 offset:" #i "*8\n"
#define F_DSR128(i)   "ds_read_b128 %14, %11 offset:" #i "*16\n"
#define F_DSW32(i)    "ds_write_b32 %11, %" #i " offset:" #i "*4\n"
#define F_BPERM(i)    "ds_bpermute_b32 %" #i ", %11, %" #i "\n"
#define F_MIXAB(i)    "v_add_u32 %" #i ", %" #i ", %8\nv_alignbit_b32 %" #i ", %" #i ", %8, 6\n"
#define F_MIXA3B(i)   "v_add_u32 %" #i ", %" #i ", %8\nv_xor_b32 %" #i ", %" #i ", %9\nv_add_u32 %" #i ", %" #i ", %8\nv_alignbit_b32 %" #i ", %" #i ", %8, 6\n"
#define F_MIXBB(i)    "v_min_u32 %" #i ", %" #i ", %8\nv_alignbit_b32 %" #i ", %" #i ", %8, 6\n"
#define F_MIXBD(i)    "v_alignbit_b32 %" #i ", %" #i ", %8, 6\nds_read_b32 %" #i ", %11 offset:" #i "*4\n"
#define F_DEP(i)      "v_add_u32 %0, %0, %8\n"
#define F_DEPALIGN(i) "v_alignbit_b32 %0, %0, %8, 6\n"

KERNEL(k_add, F_ADD)
KERNEL(k_xor, F_XOR)
KERNEL(k_lshl, F_LSHL)
KERNEL(k_alignbit_const, F_ALIGNC)
KERNEL(k_alignbit_vgpr, F_ALIGNV)
KERNEL(k_lshl_or, F_LSHLOR)
KERNEL(k_and_or, F_ANDOR)
KERNEL(k_or3, F_OR3)
KERNEL(k_add3, F_ADD3)
KERNEL(k_bfe_u32, F_BFE)
KERNEL(k_bfe_i32, F_BFEI)
KERNEL(k_bfi, F_BFI)
KERNEL(k_perm, F_PERM)
KERNEL(k_min, F_MIN)
KERNEL(k_min3, F_MIN3)
KERNEL(k_mul_u32_u24, F_MUL24)
KERNEL(k_mad_u32_u24, F_MAD24)
KERNEL(k_mul_lo_u32, F_MULLO)
KERNEL(k_mul_hi_u32, F_MULHI)
KERNEL(k_mul_hi_u32_u24, F_MULHI24)
KERNEL(k_cndmask_vcc, F_CNDVCC)
KERNEL(k_cndmask_sgpr, F_CNDS)
KERNEL(k_cmp_vcc, F_CMPVCC)
KERNEL(k_cmp_then_cndmask, F_CMPCND)
KERNEL(k_dot4_u32_u8, F_DOT4)
KERNEL(k_lshr_sdwa_byte, F_SDWA)
KERNEL(k_pk_min_u16, F_PKMIN)
KERNEL(k_pk_add_u16, F_PKADD)
KERNEL(k_pk_lshrrev_b16, F_PKLSHR)
KERNEL(k_pk_mul_lo_u16, F_PKMUL)
KERNEL(k_mov_dpp_row_shr, F_DPPMOV)
KERNEL(k_min_dpp_row_shr, F_DPPMIN)
KERNEL(k_min_dpp_wave_shr, F_DPPWSHR)
KERNEL(k_bcnt, F_BCNT)
KERNEL(k_mbcnt, F_MBCNT)
KERNEL(k_addc, F_ADDC)
KERNEL(k_sad_u8, F_SAD)
KERNEL(k_ds_read_b32, F_DSR32)
KERNEL(k_s_nop, F_DSR64)
KERNEL(k_salu_and_b64, F_SALU)
KERNEL(k_salu_lshl_b32, F_SALU2)
KERNEL(k_mix_1valu_1salu, F_MIXVS)
KERNEL(k_mix_4valu_1salu, F_MIXV4S)
KERNEL(k_and, F_AND)
KERNEL(k_or, F_OR)
KERNEL(k_sub, F_SUB)
KERNEL(k_subrev, F_SUBREV)
KERNEL(k_mov, F_MOV)
KERNEL(k_not, F_NOT)
KERNEL(k_lshr, F_LSHR)
KERNEL(k_lshlv, F_LSHLV)
KERNEL(k_ashr, F_ASHR)
KERNEL(k_max, F_MAX)
KERNEL(k_addco, F_ADDCO)
KERNEL(k_adde64, F_ADDE64)
KERNEL(k_addlit, F_ADDLIT)
KERNEL(k_addinl, F_ADDINL)
KERNEL(k_andlit, F_ANDLIT)
KERNEL(k_xors, F_XORS)
KERNEL(k_lshladd, F_LSHLADD)
KERNEL(k_addlshl, F_ADDLSHL)
KERNEL(k_xad, F_XAD)
KERNEL(k_muli24, F_MULI24)
KERNEL(k_addf, F_ADDF)
KERNEL(k_fmaf, F_FMAF)
KERNEL(k_xnor, F_XNOR)
KERNEL(k_cmps, F_CMPS)
KERNEL(k_ffbl, F_FFBL)
KERNEL(k_sdwaadd, F_SDWAADD)
KERNEL(k_dsr64, F_DSR64B)
KERNEL(k_dsr128, F_DSR128)
KERNEL(k_dsw32, F_DSW32)
KERNEL(k_bperm, F_BPERM)
KERNEL(k_mixab, F_MIXAB)
KERNEL(k_mixa3b, F_MIXA3B)
KERNEL(k_mixbb, F_MIXBB)
KERNEL(k_mixbd, F_MIXBD)
KERNEL(k_dep_add, F_DEP)
KERNEL(k_dep_alignbit, F_DEPALIGN)

struct Entry { const char *name; void (*fn)(unsigned long long *, int, unsigned long long *, unsigned long long *); int per_macro; };

int main(int argc, char **argv)
{
	const Entry all[] = {
		{ "v_add_u32", k_add, 1 }, { "v_xor_b32", k_xor, 1 }, { "v_lshlrev_b32 const", k_lshl, 1 },
		{ "v_alignbit_b32 const shift", k_alignbit_const, 1 }, { "v_alignbit_b32 vgpr shift", k_alignbit_vgpr, 1 },
		{ "v_lshl_or_b32", k_lshl_or, 1 }, { "v_and_or_b32", k_and_or, 1 }, { "v_or3_b32", k_or3, 1 }, { "v_add3_u32", k_add3, 1 },
		{ "v_bfe_u32", k_bfe_u32, 1 }, { "v_bfe_i32", k_bfe_i32, 1 }, { "v_bfi_b32", k_bfi, 1 }, { "v_perm_b32", k_perm, 1 },
		{ "v_min_u32", k_min, 1 }, { "v_min3_u32", k_min3, 1 },
		{ "v_mul_u32_u24", k_mul_u32_u24, 1 }, { "v_mad_u32_u24", k_mad_u32_u24, 1 }, { "v_mul_lo_u32", k_mul_lo_u32, 1 },
		{ "v_mul_hi_u32", k_mul_hi_u32, 1 }, { "v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1 },
		{ "v_cndmask_b32 vcc", k_cndmask_vcc, 1 }, { "v_cndmask_b32 sgpr pair", k_cndmask_sgpr, 1 },
		{ "v_cmp_ne_u32 vcc", k_cmp_vcc, 1 }, { "v_cmp + v_cndmask (pair)", k_cmp_then_cndmask, 2 },
		{ "v_dot4_u32_u8", k_dot4_u32_u8, 1 }, { "v_lshrrev_b32_sdwa byte3", k_lshr_sdwa_byte, 1 },
		{ "v_pk_min_u16", k_pk_min_u16, 1 }, { "v_pk_add_u16", k_pk_add_u16, 1 }, { "v_pk_lshrrev_b16", k_pk_lshrrev_b16, 1 },
		{ "v_pk_mul_lo_u16", k_pk_mul_lo_u16, 1 },
		{ "v_mov_b32_dpp row_shr:1", k_mov_dpp_row_shr, 1 }, { "v_min_u32_dpp row_shr:1", k_min_dpp_row_shr, 1 },
		{ "v_min_u32_dpp wave_shr:1", k_min_dpp_wave_shr, 1 },
		{ "v_bcnt_u32_b32", k_bcnt, 1 }, { "v_mbcnt_lo", k_mbcnt, 1 }, { "v_addc_co_u32", k_addc, 1 }, { "v_sad_u8", k_sad_u8, 1 },
		{ "ds_read_b32", k_ds_read_b32, 1 }, { "s_nop 0", k_s_nop, 1 },
		{ "s_and_b64", k_salu_and_b64, 1 }, { "s_lshl_b32", k_salu_lshl_b32, 1 },
		{ "mix 1 VALU + 1 SALU (pair)", k_mix_1valu_1salu, 2 }, { "mix 4 VALU + 1 SALU (group of 5)", k_mix_4valu_1salu, 5 },
		{ "v_and_b32", k_and, 1 }, { "v_or_b32", k_or, 1 }, { "v_sub_u32", k_sub, 1 }, { "v_subrev_u32", k_subrev, 1 },
		{ "v_mov_b32", k_mov, 1 }, { "v_not_b32", k_not, 1 }, { "v_lshrrev_b32 const", k_lshr, 1 }, { "v_lshlrev_b32 vgpr", k_lshlv, 1 },
		{ "v_ashrrev_i32 const", k_ashr, 1 }, { "v_max_u32", k_max, 1 }, { "v_add_co_u32", k_addco, 1 },
		{ "v_add_u32_e64 (VOP3 encoding)", k_adde64, 1 }, { "v_add_u32 literal", k_addlit, 1 }, { "v_add_u32 inline const", k_addinl, 1 },
		{ "v_and_b32 literal", k_andlit, 1 }, { "v_xor_b32 sgpr operand", k_xors, 1 },
		{ "v_lshl_add_u32", k_lshladd, 1 }, { "v_add_lshl_u32", k_addlshl, 1 }, { "v_xad_u32", k_xad, 1 }, { "v_mul_i32_i24", k_muli24, 1 },
		{ "v_add_f32", k_addf, 1 }, { "v_fma_f32", k_fmaf, 1 }, { "v_xnor_b32", k_xnor, 1 }, { "v_cmp_ne_u32_e64 sgpr", k_cmps, 1 },
		{ "v_ffbl_b32", k_ffbl, 1 }, { "v_add_u32_sdwa byte1", k_sdwaadd, 1 },
		{ "ds_read_b64", k_dsr64, 1 }, { "ds_read_b128", k_dsr128, 1 }, { "ds_write_b32", k_dsw32, 1 }, { "ds_bpermute_b32", k_bperm, 1 },
		{ "mix add + alignbit (pair)", k_mixab, 2 }, { "mix add xor add alignbit (group of 4)", k_mixa3b, 4 },
		{ "mix min + alignbit (pair)", k_mixbb, 2 }, { "mix alignbit + ds_read_b32 (pair)", k_mixbd, 2 },
		{ "dependent v_add_u32 chain", k_dep_add, 1 }, { "dependent v_alignbit chain", k_dep_alignbit, 1 },
	};
	const char *only = argc > 1 ? argv[1] : nullptr;
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	const int n_cu = prop.multiProcessorCount;
	const int iters = 400;
	unsigned long long *d_out;
	const int max_waves = n_cu * 8 * 4 * 2;
	CHK(hipMalloc(&d_out, (size_t) max_waves * sizeof(unsigned long long)));
	printf("# %s, %d CUs; SIMD cycles per wave-instruction (median over waves) at W waves per SIMD\n", prop.name, n_cu);
	hipEvent_t ev0, ev1;
	CHK(hipEventCreate(&ev0));
	CHK(hipEventCreate(&ev1));
	unsigned long long *d_hw;
	CHK(hipMalloc(&d_hw, (size_t) max_waves * sizeof(unsigned long long)));
	unsigned long long *d_rt;
	CHK(hipMalloc(&d_rt, 8));
	printf("%-38s %7s %7s %7s %7s %7s %7s %7s   %s\n", "instruction", "W=1", "W=2", "W=3", "W=4", "W=5", "W=6", "W=8", "s_memtime MHz (W=8)");
	for (const Entry &e : all) {
		if (only && !strstr(e.name, only)) continue;
		printf("%-38s", e.name);
		double mhz = 0, wall[8];
		int wi = 0;
		for (int W : { 1, 2, 3, 4, 5, 6, 8 }) {
			const size_t lds = W == 1 ? 160 * 1024 - 64 : (size_t) (160 * 1024 / W) & ~255u;   /* at most W workgroups per CU */
			CHK(hipFuncSetAttribute((const void *) e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
			const int grid = n_cu * W;
			float ms = 0;
			for (int rep = 0; rep < 2; ++rep) {
				CHK(hipEventRecord(ev0));
				hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), lds, 0, d_out, iters, d_rt, d_hw);
				CHK(hipGetLastError());
				CHK(hipEventRecord(ev1));
				CHK(hipDeviceSynchronize());
				CHK(hipEventElapsedTime(&ms, ev0, ev1));
			}
			wall[wi++] = (double) ms * 1e-3 * 2.4e9 / ((double) iters * kPerIter * e.per_macro * W);   /* cycles at 2.4 GHz per wave-instruction per SIMD, launch overhead included */
			std::vector<unsigned long long> h((size_t) grid * 4);
			CHK(hipMemcpy(h.data(), d_out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
			if (only) {                                        /* placement census: waves per (xcc, se, sh, cu, simd) */
				std::vector<unsigned long long> hw((size_t) grid * 4);
				CHK(hipMemcpy(hw.data(), d_hw, hw.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
				std::vector<unsigned long long> keys;
				for (unsigned long long v : hw) {
					const uint32_t id = (uint32_t) v, xcc = (uint32_t) (v >> 32) & 0xF;
					keys.push_back(((unsigned long long) xcc << 20) | (((id >> 13) & 7) << 16) | (((id >> 12) & 1) << 12) | (((id >> 8) & 15) << 4) | ((id >> 4) & 3));
				}
				std::sort(keys.begin(), keys.end());
				int hist[40] = { 0 }, distinct = 0;
				for (size_t a = 0; a < keys.size();) {
					size_t b = a;
					while (b < keys.size() && keys[b] == keys[a]) ++b;
					hist[std::min<size_t>(b - a, 39)]++;
					++distinct;
					a = b;
				}
				printf("\n   W=%d: %d distinct SIMDs; waves-per-SIMD histogram:", W, distinct);
				for (int q = 0; q < 40; ++q) if (hist[q]) printf(" %d:%d", q, hist[q]);
				std::vector<unsigned long long> hs(h);
				std::sort(hs.begin(), hs.end());
				printf("  per-wave cycles min %llu med %llu max %llu\n", hs.front(), hs[hs.size() / 2], hs.back());
			}
			std::sort(h.begin(), h.end());
			const double cyc = (double) h[h.size() / 2] / ((double) iters * kPerIter * e.per_macro * W);
			printf(" %7.2f", cyc);
			unsigned long long rt = 0;
			CHK(hipMemcpy(&rt, d_rt, 8, hipMemcpyDeviceToHost));
			mhz = (double) (rt >> 32) / (double) (rt & 0xFFFFFFFFull) * 100.0;
		}
		printf("   %7.0f   wall:", mhz);
		for (int q = 0; q < wi; ++q) printf(" %5.2f", wall[q]);
		printf("\n");
		fflush(stdout);
	}
	return 0;
}
