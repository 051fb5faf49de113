#include "crc32_fast.hpp"

#include <immintrin.h>
#include <zlib.h>

namespace ntsm {

namespace {

/* len >= 64 and a multiple of 16; crc is the raw shift-register value (zlib's value complemented) */
__attribute__((target("pclmul,sse4.1")))
uint32_t fold_pclmul(const uint8_t *buf, size_t len, uint32_t crc)
{
	alignas(16) static const uint64_t k1k2[2] = { 0x0154442bd4ull, 0x01c6e41596ull };   /* x^(4*128+32), x^(4*128-32) mod P */
	alignas(16) static const uint64_t k3k4[2] = { 0x01751997d0ull, 0x00ccaa009eull };   /* x^(128+32), x^(128-32) mod P */
	alignas(16) static const uint64_t k5k0[2] = { 0x0163cd6124ull, 0 };                 /* x^64 mod P */
	alignas(16) static const uint64_t poly[2] = { 0x01db710641ull, 0x01f7011641ull };   /* P', mu */
	__m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
	x1 = _mm_loadu_si128((const __m128i *) (buf + 0x00));
	x2 = _mm_loadu_si128((const __m128i *) (buf + 0x10));
	x3 = _mm_loadu_si128((const __m128i *) (buf + 0x20));
	x4 = _mm_loadu_si128((const __m128i *) (buf + 0x30));
	x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int) crc));
	x0 = _mm_load_si128((const __m128i *) k1k2);
	buf += 64;
	len -= 64;
	while (len >= 64) {                                       /* four independent 128-bit lanes, folded 512 bits ahead */
		x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
		x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
		x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
		x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
		x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
		x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
		x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
		x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
		y5 = _mm_loadu_si128((const __m128i *) (buf + 0x00));
		y6 = _mm_loadu_si128((const __m128i *) (buf + 0x10));
		y7 = _mm_loadu_si128((const __m128i *) (buf + 0x20));
		y8 = _mm_loadu_si128((const __m128i *) (buf + 0x30));
		x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
		x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
		x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
		x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
		buf += 64;
		len -= 64;
	}
	x0 = _mm_load_si128((const __m128i *) k3k4);              /* four lanes -> one */
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
	x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
	x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
	x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
	x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
	x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
	x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
	while (len >= 16) {                                       /* remaining whole 16-byte blocks */
		x2 = _mm_loadu_si128((const __m128i *) buf);
		x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
		x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
		x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
		buf += 16;
		len -= 16;
	}
	x2 = _mm_clmulepi64_si128(x1, x0, 0x10);                  /* 128 -> 64 bits */
	x3 = _mm_setr_epi32(~0, 0, ~0, 0);
	x1 = _mm_srli_si128(x1, 8);
	x1 = _mm_xor_si128(x1, x2);
	x0 = _mm_loadl_epi64((const __m128i *) k5k0);
	x2 = _mm_srli_si128(x1, 4);
	x1 = _mm_and_si128(x1, x3);
	x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
	x1 = _mm_xor_si128(x1, x2);
	x0 = _mm_load_si128((const __m128i *) poly);              /* Barrett reduction to 32 bits */
	x2 = _mm_and_si128(x1, x3);
	x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
	x2 = _mm_and_si128(x2, x3);
	x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
	x1 = _mm_xor_si128(x1, x2);
	return (uint32_t) _mm_extract_epi32(x1, 1);
}

} // namespace

uint32_t crc32_fast(uint32_t crc, const uint8_t *buf, size_t len)
{
	if (!buf) return (uint32_t) crc32(0L, Z_NULL, 0);
	static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
	if (have && len >= 64) {
		const size_t n = len & ~(size_t) 15;
		crc = ~fold_pclmul(buf, n, ~crc);
		buf += n;
		len -= n;
	}
	while (len) {                                             /* zlib takes a 32-bit length */
		const size_t n = len > (1u << 30) ? (1u << 30) : len;
		crc = (uint32_t) crc32(crc, buf, (uInt) n);
		buf += n;
		len -= n;
	}
	return crc;
}

} // namespace ntsm
