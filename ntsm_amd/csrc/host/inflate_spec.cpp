#include "inflate_spec.hpp"

#include "crc32_fast.hpp"

#include <algorithm>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace ntsm {

namespace {
inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline void store64(void *p, uint64_t v) { memcpy(p, &v, 8); }
inline void copy16(void *dst, const void *src) { unsigned char t[16]; memcpy(t, src, 16); memcpy(dst, t, 16); }
} // namespace

void SpecInflate::fill_markers(uint16_t *sym)
{
	for (size_t j = 0; j < kWindow; ++j) sym[j] = (uint16_t) (kMarker | j);
}

/* Candidate filter in registers, then the full header parse of Inflate::open_block (the same checks the in-order decoder
 * applies).  The filter: BFINAL = 0, BTYPE = 2, HLIT <= 29, HDIST <= 29, and a precode whose lengths satisfy Kraft's
 * equality (zlib rejects an incomplete precode, inftrees.c). */
uint64_t SpecInflate::find(const uint8_t *base, const uint8_t *end, uint64_t from_bit, uint64_t to_bit)
{
	const uint64_t size_bits = (uint64_t) (end - base) * 8u;
	if (size_bits < 20 * 8) return ~0ull;
	const uint64_t last = std::min<uint64_t>(to_bit, size_bits - 18 * 8);    /* two 64-bit loads stay inside the input */
	static const struct K4 {
		uint16_t v[4096];
		K4() { for (unsigned i = 0; i < 4096; ++i) { unsigned k = 0, u = 0; for (unsigned j = 0; j < 4; ++j) { const unsigned l = (i >> (3 * j)) & 7u; if (l) { k += 128u >> l; ++u; } } v[i] = (uint16_t) (k | (u << 10)); } }
	} k4_table;
	const uint16_t *const k4 = k4_table.v;
	/* The first 13 bits of a candidate for 44 bit offsets at a time, with word operations on w = the 64 bits from offset g:
	 * bit i of `cand` is set iff bits i .. i+2 read 0, 0, 1 (BFINAL = 0, BTYPE = 10b), bits i+4 .. i+7 are not all ones
	 * (HLIT <= 29) and bits i+9 .. i+12 are not all ones (HDIST <= 29).  One offset in nine survives on random data. */
	uint64_t g = from_bit, cand = 0;
	for (;;) {
		while (!cand) {
			if (g >= last) return ~0ull;
			const uint64_t w = load64(base + (g >> 3)) >> (g & 7u);            /* >= 57 valid bits: offsets g .. g+43 have their 13 */
			uint64_t m = ~w & ~(w >> 1) & (w >> 2);
			m &= ~((w >> 4) & (w >> 5) & (w >> 6) & (w >> 7));
			m &= ~((w >> 9) & (w >> 10) & (w >> 11) & (w >> 12));
			const uint64_t span = std::min<uint64_t>(44, last - g);
			cand = m & ((1ull << span) - 1u);
			if (!cand) g += span;
		}
		const unsigned t = (unsigned) __builtin_ctzll(cand);
		cand &= cand - 1;
		const uint64_t b = g + t;
		if (!cand) g += std::min<uint64_t>(44, last - g);                     /* the word is used up: the next one starts behind it */
		const uint8_t *p = base + (b >> 3);
		const unsigned sh = (unsigned) (b & 7u);
		const uint64_t x = load64(p) >> sh;                                   /* >= 57 valid bits */
		const unsigned ncode = (unsigned) ((x >> 13) & 15u) + 4u;
		/* precode lengths: 3 bits each from bit 17 of the candidate, up to 19 of them (bits 17 .. 73).  x holds the first 13
		 * (bits 17 .. 55): most false candidates are over-subscribed by then (a random length adds 16 of the 128 on average). */
		unsigned kraft = 0, used = 0;
		{
			/* four lengths (12 bits) per table look-up: Kraft sum in the low 10 bits, number of non-zero lengths above */
			const unsigned n13 = ncode < 13u ? ncode : 13u;
			const uint64_t z = (x >> 17) & ((1ull << (3u * n13)) - 1u);
			const unsigned q = (unsigned) k4[z & 4095u] + k4[(z >> 12) & 4095u] + k4[(z >> 24) & 4095u] + k4[z >> 36];
			kraft = q & 1023u;
			used = q >> 10;
			if (kraft > 128u) continue;
			if (ncode > 13u) {
				unsigned __int128 w = ((((unsigned __int128) load64(p + 8)) << 64) | load64(p)) >> (sh + 17 + 39);
				for (unsigned i = 13; i < ncode; ++i) {
					const unsigned l = (unsigned) w & 7u;
					w >>= 3;
					if (l) { kraft += 128u >> l; ++used; }
				}
			}
		}
		if (kraft != 128u || used < 2) continue;
		reset_at(p, sh, end, 0);
		if (open_block() != MORE || m_last || m_mode != HUFFMAN) continue;
		return b;
	}
	return ~0ull;
}

NTSM_INFLATE_CLONES Inflate::Status SpecInflate::run_huffman16(uint16_t *buf, size_t *out, size_t out_stop)
{
	const uint8_t *in = m_in;
	uint64_t bb = m_bb;
	unsigned bc = m_bc;
	uint16_t *op = buf + *out;
	uint16_t *const op0 = op, *const op_stop = buf + out_stop;
	const uint64_t total0 = m_total;
	Status st = MORE;
	constexpr uint32_t lmask = (1u << kLitBits) - 1, dmask = (1u << kDistBits) - 1;
#define SAVE() do { m_in = in; m_bc = bc; m_bb = bc >= 64 ? bb : (bb & ((1ull << bc) - 1)); \
		m_total = total0 + (uint64_t) (op - op0); *out = (size_t) (op - buf); } while (0)
	if (m_end - in >= 16 && op_stop - op > 280) {
		const uint8_t *const in_fast = m_end - 16;
		uint16_t *const op_fast = op_stop - 280;
		while (in < in_fast && op < op_fast) {
			bb |= load64(in) << bc;
			in += (63 - bc) >> 3;
			bc |= 56;
			/* One shift per symbol: a length / distance entry counts its extra bits in (e & 31), the value of the extra bits
			 * is cut out of the copy `sv` off the critical path (table lookup -> shift -> next lookup). */
			uint64_t sv;
			uint32_t e = m_lit[bb & lmask];
			if (e & F_SUB) { bb >>= kLitBits; bc -= kLitBits; e = m_lit[(e >> 16) + (uint32_t) (bb & ((1u << ((e >> 8) & 15u)) - 1))]; }
			sv = bb; bb >>= (e & 31u); bc -= (e & 31u);
			if (e & F_LIT) {
				*op++ = (uint16_t) (e >> 16);
				e = m_lit[bb & lmask];
				if (e & F_SUB) { bb >>= kLitBits; bc -= kLitBits; e = m_lit[(e >> 16) + (uint32_t) (bb & ((1u << ((e >> 8) & 15u)) - 1))]; }
				sv = bb; bb >>= (e & 31u); bc -= (e & 31u);
				if (e & F_LIT) {
					*op++ = (uint16_t) (e >> 16);
					e = m_lit[bb & lmask];
					if (e & F_SUB) { bb >>= kLitBits; bc -= kLitBits; e = m_lit[(e >> 16) + (uint32_t) (bb & ((1u << ((e >> 8) & 15u)) - 1))]; }
					sv = bb; bb >>= (e & 31u); bc -= (e & 31u);
					if (e & F_LIT) { *op++ = (uint16_t) (e >> 16); continue; }
				}
			}
			if (e & (F_EOB | F_ERR)) {
				SAVE();
				return (e & F_ERR) ? DATA_ERROR : STREAM_END;
			}
			const unsigned lx = (e >> 8) & 15u;
			const uint32_t len = (e >> 16) + (uint32_t) ((sv >> ((e & 31u) - lx)) & ((1u << lx) - 1));
			/* the refill at the top (>= 56 bits) covers two literals and a length (15 + 15 + 20 bits at most); a distance
			 * needs up to 15 + 13 more */
			if (bc < 28u) {
				bb |= load64(in) << bc;
				in += (63 - bc) >> 3;
				bc |= 56;
			}
			uint32_t d = m_dist[bb & dmask];
			if (d & F_SUB) { bb >>= kDistBits; bc -= kDistBits; d = m_dist[(d >> 16) + (uint32_t) (bb & ((1u << ((d >> 8) & 15u)) - 1))]; }
			sv = bb; bb >>= (d & 31u); bc -= (d & 31u);
			if (d & F_ERR) { SAVE(); return DATA_ERROR; }
			const unsigned dx = (d >> 8) & 15u;
			const uint32_t dist = (d >> 16) + (uint32_t) ((sv >> ((d & 31u) - dx)) & ((1u << dx) - 1));
			/* dist <= 32768 always lands inside the buffer: its first kWindow symbols are the markers */
			const uint16_t *src = op - dist;
			uint16_t *const end = op + len;
			if (dist >= 8) {
				/* 8 symbols at a time; DNA text is mostly matches of 6-9 symbols, which the first two take without a loop */
				copy16(op, src);
				copy16(op + 8, src + 8);                              /* unconditionally: one well-predicted branch for len > 16 instead of a coin toss at 8 */
				if (len > 16) {
					op += 16; src += 16;
					do { copy16(op, src); op += 8; src += 8; } while (op < end);
				}
			} else if (dist >= 4) {
				do { store64(op, load64((const uint8_t *) src)); op += 4; src += 4; } while (op < end);
			} else if (dist == 1) {
				const uint64_t v = 0x0001000100010001ull * *src;
				do { store64(op, v); op += 4; } while (op < end);
			} else {
				do { *op++ = *src++; } while (op < end);
			}
			op = end;
		}
	}
	while (op < op_stop) {
		while (bc <= 56 && in < m_end) { bb |= (uint64_t) *in++ << bc; bc += 8; }
		uint32_t e = m_lit[bb & lmask];
		if (e & F_SUB) {
			if (bc < (unsigned) kLitBits) { st = TRUNCATED; break; }
			bb >>= kLitBits; bc -= kLitBits;
			e = m_lit[(e >> 16) + (uint32_t) (bb & ((1u << ((e >> 8) & 15u)) - 1))];
		}
		const unsigned lx = (e >> 8) & 15u, nb = (e & 31u) - lx;     /* extra bits (0 unless a length), bits of the code itself */
		if (nb > bc || ((e & F_ERR) && in == m_end && bc < 15)) { st = TRUNCATED; break; }
		if (e & F_ERR) { st = DATA_ERROR; break; }
		bb >>= nb; bc -= nb;
		if (e & F_LIT) { *op++ = (uint16_t) (e >> 16); continue; }
		if (e & F_EOB) { st = STREAM_END; break; }
		if (bc < lx) { st = TRUNCATED; break; }
		const uint32_t len = (e >> 16) + (uint32_t) (bb & ((1u << lx) - 1));
		bb >>= lx; bc -= lx;
		while (bc <= 56 && in < m_end) { bb |= (uint64_t) *in++ << bc; bc += 8; }
		uint32_t d = m_dist[bb & dmask];
		if (d & F_SUB) {
			if (bc < (unsigned) kDistBits) { st = TRUNCATED; break; }
			bb >>= kDistBits; bc -= kDistBits;
			d = m_dist[(d >> 16) + (uint32_t) (bb & ((1u << ((d >> 8) & 15u)) - 1))];
		}
		const unsigned dx = (d >> 8) & 15u, db = (d & 31u) - dx;
		if (db > bc || ((d & F_ERR) && in == m_end && bc < 15)) { st = TRUNCATED; break; }
		if (d & F_ERR) { st = DATA_ERROR; break; }
		bb >>= db; bc -= db;
		if (bc < dx) { st = TRUNCATED; break; }
		const uint32_t dist = (d >> 16) + (uint32_t) (bb & ((1u << dx) - 1));
		bb >>= dx; bc -= dx;
		const uint16_t *src = op - dist;
		for (uint32_t i = 0; i < len; ++i) op[i] = src[i];
		op += len;
	}
	SAVE();
	return st;
#undef SAVE
}

Inflate::Status SpecInflate::run16(uint16_t *sym, size_t *out, size_t out_stop)
{
	for (;;) {
		switch (m_mode) {
		case DONE:
			return STREAM_END;
		case HEADER: {
			if (m_last) { m_mode = DONE; return STREAM_END; }
			if (m_base && bit_pos(m_base) >= m_stop_bit) return BLOCK_STOP;
			const Status st = open_block();
			if (st != MORE) return st;
			break;
		}
		case STORED: {
			while (m_stored && m_bc) {
				if (*out >= out_stop) return MORE;
				sym[(*out)++] = (uint16_t) (m_bb & 0xFFu);
				m_bb >>= 8; m_bc -= 8;
				--m_stored;
				++m_total;
			}
			if (m_stored) {
				if (*out >= out_stop) return MORE;
				size_t n = m_stored;
				if (n > (size_t) (m_end - m_in)) n = (size_t) (m_end - m_in);
				if (n > out_stop - *out) n = out_stop - *out;
				for (size_t i = 0; i < n; ++i) sym[*out + i] = m_in[i];
				m_in += n;
				*out += n;
				m_total += n;
				m_stored -= (uint32_t) n;
				if (m_stored) return m_in == m_end ? TRUNCATED : MORE;
			}
			m_mode = HEADER;
			break;
		}
		case HUFFMAN: {
			const Status st = run_huffman16(sym, out, out_stop);
			if (st == STREAM_END) { m_mode = HEADER; break; }
			if (st == DATA_ERROR) m_mode = DONE;
			return st;
		}
		}
	}
}

namespace {
/* out[i] = sym[i] < 256 ? sym[i] : window[sym[i] & 0x7FFF], through one 64 KiB table (literals map to themselves, markers to
 * their window byte): FASTQ chunks are full of markers (every quality line is a copy of a copy ... of the window), so a
 * branch per symbol would mispredict all the time */
inline void table_run(const uint8_t *lut, const uint16_t *sym, size_t n, uint8_t *out)
{
	size_t i = 0;
	const size_t n8 = n & ~(size_t) 7;
	for (; i < n8; i += 8) {
		const uint64_t v = (uint64_t) lut[sym[i]] | ((uint64_t) lut[sym[i + 1]] << 8) | ((uint64_t) lut[sym[i + 2]] << 16) | ((uint64_t) lut[sym[i + 3]] << 24) |
			((uint64_t) lut[sym[i + 4]] << 32) | ((uint64_t) lut[sym[i + 5]] << 40) | ((uint64_t) lut[sym[i + 6]] << 48) | ((uint64_t) lut[sym[i + 7]] << 56);
		memcpy(out + i, &v, 8);
	}
	for (; i < n; ++i) out[i] = lut[sym[i]];
}
#if defined(__x86_64__)
/* 32 symbols at a time where they have one of the two shapes that make up almost all of a FASTQ chunk: 32 literals (a pack),
 * or 32 CONSECUTIVE table indices s, s + 1, ... -- a stretch of one copy out of the window, i.e. 32 bytes of the window in
 * order (a quality or header line that is a copy of a copy ... of a line in front of the chunk): one 32-byte load from the
 * table at s.  Anything else takes 8 symbols through the table and tries again from there, so a shape that starts in the
 * middle of a group is picked up 8 symbols later at most.  lut must be readable up to 65536 + 31. */
__attribute__((target("avx2"))) void resolve_avx2(const uint8_t *lut, const uint16_t *sym, size_t n, uint8_t *out)
{
	const __m256i hi = _mm256_set1_epi16((short) 0xFF00);
	const __m256i iota_a = _mm256_setr_epi16(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
	const __m256i iota_b = _mm256_setr_epi16(16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31);
	size_t i = 0;
	while (i + 32 <= n) {
		const __m256i a = _mm256_loadu_si256((const __m256i *) (sym + i));
		const __m256i b = _mm256_loadu_si256((const __m256i *) (sym + i + 16));
		if (_mm256_testz_si256(_mm256_or_si256(a, b), hi)) {
			_mm256_storeu_si256((__m256i *) (out + i), _mm256_permute4x64_epi64(_mm256_packus_epi16(a, b), 0xD8));
			i += 32;
			continue;
		}
		const __m256i first = _mm256_set1_epi16((short) sym[i]);
		const __m256i da = _mm256_xor_si256(_mm256_sub_epi16(a, iota_a), first), db = _mm256_xor_si256(_mm256_sub_epi16(b, iota_b), first);
		/* (a stretch that runs over the top of the 16-bit range -- the window's last bytes followed by literals 0, 1, 2 ... --
		 * compares equal modulo 65536 and is not a stretch of the table: sym[i] <= 65535 - 31) */
		if (sym[i] <= 65535 - 31 && _mm256_testz_si256(_mm256_or_si256(da, db), _mm256_set1_epi16(-1))) {
			_mm256_storeu_si256((__m256i *) (out + i), _mm256_loadu_si256((const __m256i *) (lut + sym[i])));
			i += 32;
			continue;
		}
		/* the same two shapes on the first 16 symbols alone (a line ends inside the group), else 8 through the table */
		if (_mm256_testz_si256(a, hi)) {
			_mm_storeu_si128((__m128i *) (out + i), _mm_packus_epi16(_mm256_castsi256_si128(a), _mm256_extracti128_si256(a, 1)));
			i += 16;
			continue;
		}
		if (sym[i] <= 65535 - 15 && _mm256_testz_si256(da, _mm256_set1_epi16(-1))) {
			_mm_storeu_si128((__m128i *) (out + i), _mm_loadu_si128((const __m128i *) (lut + sym[i])));
			i += 16;
			continue;
		}
		table_run(lut, sym + i, 8, out + i);
		i += 8;
	}
	table_run(lut, sym + i, n - i, out + i);
}
#endif
} // namespace

bool SpecInflate::resolve(const uint16_t *sym, size_t n, const uint8_t *window, size_t valid, uint8_t *out, uint32_t *crc)
{
	uint8_t lut[65536 + 32];                                    /* on the caller's stack: 0 .. 255 and 32768 .. 65535 are the only entries ever used */
	for (int i = 0; i < 256 + 32; ++i) lut[i] = (uint8_t) i;      /* (+ 32: the 32-byte loads of resolve_avx2 stay inside initialised memory) */
	memcpy(lut + kMarker, window, kWindow);
	memset(lut + 65536, 0, 32);
#if defined(__x86_64__)
	static const bool avx2 = __builtin_cpu_supports("avx2");
#else
	constexpr bool avx2 = false;
#endif
	/* in pieces that stay in the cache between the two passes when the caller wants the CRC-32 of the bytes as well */
	const size_t step = crc ? (size_t) 32768 : n;
	uint32_t c = crc ? *crc : 0;
	for (size_t at = 0; at < n; at += step) {
		const size_t m = std::min(step, n - at);
#if defined(__x86_64__)
		if (avx2) resolve_avx2(lut, sym + at, m, out + at);
		else
#endif
			table_run(lut, sym + at, m, out + at);
		if (crc) c = crc32_fast(c, out + at, m);
	}
	if (crc) *crc = c;
	if (valid < kWindow) {                                      /* a marker in front of the member's first byte: invalid distance */
		const uint16_t lowest = (uint16_t) (kMarker | (kWindow - valid));
		for (size_t j = 0; j < n; ++j) if (sym[j] >= 256 && sym[j] < lowest) return false;
	}
	return true;
}

} // namespace ntsm
