#include "pack2.hpp"

#include <cstring>
#if defined(__x86_64__) || defined(__i386__)
#define NTSM_PACK2_X86 1
#include <immintrin.h>
#else
#define NTSM_PACK2_X86 0                 /* other hosts: the portable loop only */
#endif

namespace ntsm {

namespace {

/* class of a byte: 0..3 = code, 4 = invalid (vendor/KseqHashIterator.hpp:114-127) */
struct ByteTable {
	uint8_t t[256];
	ByteTable()
	{
		for (int i = 0; i < 256; ++i) t[i] = 4;
		for (int i = 0; i < 4; ++i) t[i] = (uint8_t) i;
		t['A'] = t['a'] = 0;
		t['C'] = t['c'] = 1;
		t['G'] = t['g'] = 2;
		t['T'] = t['t'] = t['U'] = t['u'] = 3;
	}
};
const ByteTable kTable;

/* one group of 32 positions from 32 bytes of which the first n (<= 32) are sequence, the rest count as invalid */
inline void group_scalar(uint8_t *codes8, uint8_t *valid4, const uint8_t *b, unsigned n)
{
	uint64_t c = 0;
	uint32_t v = 0;
	for (unsigned i = 0; i < n; ++i) {
		const uint8_t k = kTable.t[b[i]];
		if (k < 4) {
			c |= (uint64_t) k << (2 * i);
			v |= 1u << i;
		}
	}
	memcpy(codes8, &c, 8);
	memcpy(valid4, &v, 4);
}

#if NTSM_PACK2_X86
__attribute__((target("avx2"))) inline void group_avx2(uint8_t *codes8, uint8_t *valid4, const uint8_t *b, unsigned n)
{
	const __m256i x = _mm256_loadu_si256((const __m256i *) b);
	/* letters: fold the case bit, compare with the five accepted letters */
	const __m256i lo = _mm256_or_si256(x, _mm256_set1_epi8(0x20));
	__m256i ok = _mm256_cmpeq_epi8(lo, _mm256_set1_epi8('a'));
	ok = _mm256_or_si256(ok, _mm256_cmpeq_epi8(lo, _mm256_set1_epi8('c')));
	ok = _mm256_or_si256(ok, _mm256_cmpeq_epi8(lo, _mm256_set1_epi8('g')));
	ok = _mm256_or_si256(ok, _mm256_cmpeq_epi8(lo, _mm256_set1_epi8('t')));
	ok = _mm256_or_si256(ok, _mm256_cmpeq_epi8(lo, _mm256_set1_epi8('u')));
	/* (x | 0x20 == 'a' iff x is 'A' or 'a': no other byte folds onto a letter) */
	/* raw code bytes 0x00..0x03 are valid too (vendor/KseqHashIterator.hpp:115): rare, handled by the portable loop */
	const __m256i raw = _mm256_cmpeq_epi8(_mm256_and_si256(x, _mm256_set1_epi8((char) 0xFC)), _mm256_setzero_si256());
	uint32_t v = (uint32_t) _mm256_movemask_epi8(ok);
	if (n < 32) v &= (1u << n) - 1u;
	uint32_t rawm = (uint32_t) _mm256_movemask_epi8(raw);
	if (n < 32) rawm &= (1u << n) - 1u;
	if (rawm) { group_scalar(codes8, valid4, b, n); return; }
	/* code of a letter: y = (x >> 1) & 3 gives A 0, C 1, T/U 2, G 3; y ^ (y >> 1) swaps the last two */
	const __m256i y = _mm256_and_si256(_mm256_srli_epi16(x, 1), _mm256_set1_epi8(3));
	__m256i c = _mm256_xor_si256(y, _mm256_and_si256(_mm256_srli_epi16(y, 1), _mm256_set1_epi8(1)));
	c = _mm256_and_si256(c, ok);                                       /* invalid positions pack as 0 */
	/* four 2-bit codes per byte: (b0 + 4 b1) + 16 (b2 + 4 b3) */
	const __m256i p16 = _mm256_maddubs_epi16(c, _mm256_set1_epi16(0x0401));
	const __m256i p32 = _mm256_madd_epi16(p16, _mm256_set1_epi32(0x00100001));
	/* low byte of every 32-bit lane -> 8 contiguous bytes */
	const __m256i sh = _mm256_shuffle_epi8(p32, _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
	                                                           0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1));
	const uint32_t lo4 = (uint32_t) _mm256_extract_epi32(sh, 0), hi4 = (uint32_t) _mm256_extract_epi32(sh, 4);
	uint64_t packed = (uint64_t) lo4 | ((uint64_t) hi4 << 32);
	if (n < 32) packed &= (1ull << (2 * n)) - 1ull;
	memcpy(codes8, &packed, 8);
	memcpy(valid4, &v, 4);
}
#endif

#if NTSM_PACK2_X86
/* AVX-512 (F + BW + VL + VBMI: Zen 4 / Zen 5, Ice Lake and later): the byte table itself as a 128-entry vpermi2b look-up -- entry =
 * code 0..3, or 0x80 for an invalid byte; bit 7 of the byte OR its entry is "invalid" (bytes >= 0x80 are never valid).  Raw code
 * bytes 0x00..0x03 need no side path, and a 64-position group is ten instructions where the AVX2 form takes twenty for 32. */
struct alignas(64) ClassTable128 {
	uint8_t t[128];
	ClassTable128() { for (int i = 0; i < 128; ++i) t[i] = kTable.t[i] < 4 ? kTable.t[i] : 0x80; }
};
const ClassTable128 kClass128;

#define NTSM_AVX512_TARGET __attribute__((target("avx512f,avx512bw,avx512vl,avx512vbmi")))
/* 64 positions, all of them sequence bytes */
NTSM_AVX512_TARGET inline void group64_avx512(uint8_t *codes16, uint8_t *valid8, const uint8_t *b)
{
	const __m512i x = _mm512_loadu_si512((const void *) b);
	const __m512i cls = _mm512_permutex2var_epi8(_mm512_load_si512((const void *) kClass128.t), x, _mm512_load_si512((const void *) (kClass128.t + 64)));
	const __mmask64 v = ~_mm512_movepi8_mask(_mm512_or_si512(cls, x));
	const __m512i c = _mm512_maskz_mov_epi8(v, cls);                    /* invalid positions pack as 0 */
	const __m512i p16 = _mm512_maddubs_epi16(c, _mm512_set1_epi16(0x0401));
	const __m512i p32 = _mm512_madd_epi16(p16, _mm512_set1_epi32(0x00100001));
	_mm_storeu_si128((__m128i *) codes16, _mm512_cvtepi32_epi8(p32));
	const uint64_t vv = (uint64_t) v;
	memcpy(valid8, &vv, 8);
}
/* the last 0..63 sequence bytes of a read: a masked load reads exactly them (masked-out bytes are not accessed), positions
 * behind them come out invalid; one 32-position group is written for n < 32, two for 32 <= n < 64 -- the groups the other
 * forms write */
NTSM_AVX512_TARGET inline void tail_avx512(uint8_t *codes, uint8_t *valid, const uint8_t *b, unsigned n)
{
	const __mmask64 have = n ? ~0ull >> (64 - n) : 0;
	const __m512i x = _mm512_maskz_loadu_epi8(have, (const void *) b);
	const __m512i cls = _mm512_permutex2var_epi8(_mm512_load_si512((const void *) kClass128.t), x, _mm512_load_si512((const void *) (kClass128.t + 64)));
	const __mmask64 v = ~_mm512_movepi8_mask(_mm512_or_si512(cls, x)) & have;
	const __m512i c = _mm512_maskz_mov_epi8(v, cls);
	const __m512i p16 = _mm512_maddubs_epi16(c, _mm512_set1_epi16(0x0401));
	const __m512i p32 = _mm512_madd_epi16(p16, _mm512_set1_epi32(0x00100001));
	const __m128i packed = _mm512_cvtepi32_epi8(p32);
	const uint64_t vv = (uint64_t) v;
	if (n >= 32) {
		_mm_storeu_si128((__m128i *) codes, packed);
		memcpy(valid, &vv, 8);
	} else {
		_mm_storel_epi64((__m128i *) codes, packed);
		memcpy(valid, &vv, 4);
	}
}
#endif

int g_force_impl = 0;                                                   /* test hook: 0 = best available, 1 = scalar, 2 = at most AVX2 */
bool has_avx2()
{
#if NTSM_PACK2_X86
	static const bool yes = __builtin_cpu_supports("avx2");
	return yes;
#else
	return false;
#endif
}
bool has_avx512()
{
#if NTSM_PACK2_X86
	static const bool yes = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") &&
	                        __builtin_cpu_supports("avx512vbmi");
	return yes;
#else
	return false;
#endif
}

/* groups of 32 positions starting at pos (a multiple of 8: codes byte pos / 4, valid byte pos / 8); the last group holds the
 * rest of the read (possibly none) + the terminator positions and is built in a local buffer: nothing past seq[len - 1] is read */
#define NTSM_PACK2_BODY(GROUP)                                                                  \
	uint64_t i = 0;                                                                             \
	for (; i + 32 <= len; i += 32) GROUP(codes + ((pos + i) >> 2), valid + ((pos + i) >> 3), s + i, 32); \
	uint8_t tail[32];                                                                           \
	const unsigned n = (unsigned) (len - i);                                                    \
	if (n) memcpy(tail, s + i, n);                                                              \
	memset(tail + n, 'N', 32 - n);                                                              \
	GROUP(codes + ((pos + i) >> 2), valid + ((pos + i) >> 3), tail, n);                         \
	return (pos + len + 8) & ~7ull;

#if NTSM_PACK2_X86
__attribute__((target("avx2"))) uint64_t append_avx2(uint8_t *codes, uint8_t *valid, uint64_t pos, const uint8_t *s, uint64_t len)
{
	NTSM_PACK2_BODY(group_avx2)
}
/* 64 positions at a time while 64 sequence bytes are left, then the rest through one masked group: the bytes written and the
 * extent (pack2_extent) are those of the other forms */
NTSM_AVX512_TARGET uint64_t append_avx512(uint8_t *codes, uint8_t *valid, uint64_t pos, const uint8_t *s, uint64_t len)
{
	uint64_t i = 0;
	for (; i + 64 <= len; i += 64) group64_avx512(codes + ((pos + i) >> 2), valid + ((pos + i) >> 3), s + i);
	tail_avx512(codes + ((pos + i) >> 2), valid + ((pos + i) >> 3), s + i, (unsigned) (len - i));
	return (pos + len + 8) & ~7ull;
}
#endif

uint64_t append_scalar(uint8_t *codes, uint8_t *valid, uint64_t pos, const uint8_t *s, uint64_t len)
{
	NTSM_PACK2_BODY(group_scalar)
}
#undef NTSM_PACK2_BODY

} // namespace

uint64_t pack2_append(uint8_t *codes, uint8_t *valid, uint64_t pos, const char *seq, uint64_t len)
{
#if NTSM_PACK2_X86
	if (g_force_impl == 0 && has_avx512()) return append_avx512(codes, valid, pos, (const uint8_t *) seq, len);
	if (g_force_impl != 1 && has_avx2()) return append_avx2(codes, valid, pos, (const uint8_t *) seq, len);
#endif
	return append_scalar(codes, valid, pos, (const uint8_t *) seq, len);
}

const char *pack2_impl() { return g_force_impl == 0 && has_avx512() ? "avx512vbmi" : g_force_impl != 1 && has_avx2() ? "avx2" : "scalar"; }
void pack2_force_scalar(bool on) { g_force_impl = on ? 1 : 0; }
void pack2_force_impl(int impl) { g_force_impl = impl; }

} // namespace ntsm
