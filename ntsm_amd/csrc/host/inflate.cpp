#include "inflate.hpp"

#include <cstring>

namespace ntsm {

namespace {

constexpr uint32_t F_LIT = Inflate::F_LIT, F_EOB = Inflate::F_EOB, F_SUB = Inflate::F_SUB, F_ERR = Inflate::F_ERR;
constexpr uint32_t kErr = F_ERR | 1u;

const uint16_t kLenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
const uint8_t kLenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
const uint16_t kDistBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
	8193, 12289, 16385, 24577 };
const uint8_t kDistExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
const uint8_t kPrecodeOrder[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };

enum Kind { LITLEN, DIST, PRECODE };

inline uint32_t symbol_entry(Kind kind, int sym, int nbits)
{
	if (kind == PRECODE) return ((uint32_t) sym << 16) | (uint32_t) nbits;
	if (kind == DIST) {
		if (sym >= 30) return F_ERR | (uint32_t) nbits;
		return ((uint32_t) kDistBase[sym] << 16) | ((uint32_t) kDistExtra[sym] << 8) | (uint32_t) (nbits + kDistExtra[sym]);
	}
	if (sym < 256) return ((uint32_t) sym << 16) | F_LIT | (uint32_t) nbits;
	if (sym == 256) return F_EOB | (uint32_t) nbits;
	if (sym >= 286) return F_ERR | (uint32_t) nbits;
	return ((uint32_t) kLenBase[sym - 257] << 16) | ((uint32_t) kLenExtra[sym - 257] << 8) | (uint32_t) (nbits + kLenExtra[sym - 257]);
}

inline uint32_t reverse_bits(uint32_t c, int len)
{
	uint32_t r = 0;
	for (int i = 0; i < len; ++i) { r = (r << 1) | (c & 1u); c >>= 1; }
	return r;
}

/* Canonical Huffman decode table (codes are read LSB first, so table indices are bit-reversed codes).  False for
 * the code sets zlib's inflate_table rejects: over-subscribed, or incomplete unless it is one single 1-bit code. */
bool build_table(uint32_t *table, int tb, int max_size, const uint8_t *lens, int n, Kind kind)
{
	int count[16] = { 0 };
	for (int s = 0; s < n; ++s) count[lens[s]]++;
	count[0] = 0;
	int max_len = 0;
	for (int l = 1; l <= 15; ++l) if (count[l]) max_len = l;
	const int primary = 1 << tb;
	for (int i = 0; i < primary; ++i) table[i] = kErr;
	if (max_len == 0) return kind != PRECODE;                  /* no codes at all: every lookup is an error */
	int left = 1;
	for (int l = 1; l <= 15; ++l) {
		left = (left << 1) - count[l];
		if (left < 0) return false;                             /* over-subscribed */
	}
	if (left > 0 && (kind == PRECODE || max_len != 1)) return false;   /* incomplete */
	uint32_t next[16];
	{
		uint32_t code = 0;
		for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t) (l > 1 ? count[l - 1] : 0)) << 1; next[l] = code; }
	}
	uint16_t rcode[320];
	uint8_t need[1 << 11];                                     /* sub-table bits per primary index (tb <= 11) */
	bool any_long = false;
	for (int s = 0; s < n; ++s) {
		const int l = lens[s];
		if (!l) continue;
		const uint32_t r = reverse_bits(next[l]++, l);
		rcode[s] = (uint16_t) r;
		if (l <= tb) {
			const uint32_t e = symbol_entry(kind, s, l);
			for (uint32_t i = r; i < (uint32_t) primary; i += 1u << l) table[i] = e;
		} else {
			if (!any_long) { memset(need, 0, (size_t) primary); any_long = true; }
			const uint32_t p = r & (uint32_t) (primary - 1);
			if (l - tb > need[p]) need[p] = (uint8_t) (l - tb);
		}
	}
	if (!any_long) return true;
	int free_at = primary;
	for (int p = 0; p < primary; ++p) {
		if (!need[p]) continue;
		const int size = 1 << need[p];
		if (free_at + size > max_size) return false;
		table[p] = ((uint32_t) free_at << 16) | F_SUB | ((uint32_t) need[p] << 8) | (uint32_t) tb;
		for (int i = 0; i < size; ++i) table[free_at + i] = kErr;
		free_at += size;
	}
	for (int s = 0; s < n; ++s) {
		const int l = lens[s];
		if (l <= tb) continue;
		const uint32_t r = rcode[s], sub = table[r & (uint32_t) (primary - 1)];
		const uint32_t start = sub >> 16, sb = (sub >> 8) & 15u;
		const uint32_t e = symbol_entry(kind, s, l - tb);
		for (uint32_t i = r >> tb; i < (1u << sb); i += 1u << (l - tb)) table[start + i] = e;
	}
	return true;
}

inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline void store64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }

} // namespace

void Inflate::reset(const uint8_t *in, const uint8_t *in_end)
{
	m_in = in;
	m_end = in_end;
	m_bb = 0;
	m_bc = 0;
	m_mode = HEADER;
	m_last = false;
	m_stored = 0;
	m_total = 0;
	m_base = nullptr;
	m_stop_bit = 0;
}

void Inflate::reset_at(const uint8_t *in, unsigned bit, const uint8_t *in_end, uint64_t total)
{
	reset(in, in_end);
	m_total = total;
	if (bit && in < in_end) {                                  /* the rest of the first byte goes into the bit buffer */
		m_bb = (uint64_t) *in >> bit;
		m_bc = 8 - bit;
		m_in = in + 1;
	}
}

const uint8_t *Inflate::in() const { return m_in - (m_bc >> 3); }

bool Inflate::build(uint32_t *table, int table_bits, int max_size, const uint8_t *lens, int n, bool is_dist)
{
	return build_table(table, table_bits, max_size, lens, n, is_dist ? DIST : LITLEN);
}

void Inflate::set_fixed()
{
	uint8_t lens[288];
	for (int i = 0; i < 144; ++i) lens[i] = 8;
	for (int i = 144; i < 256; ++i) lens[i] = 9;
	for (int i = 256; i < 280; ++i) lens[i] = 7;
	for (int i = 280; i < 288; ++i) lens[i] = 8;
	build(m_lit, kLitBits, kLitSize, lens, 288, false);
	uint8_t dl[32];
	for (int i = 0; i < 32; ++i) dl[i] = 5;
	build(m_dist, kDistBits, kDistSize, dl, 32, true);
}

#define NTSM_REFILL_SAFE() do { while (m_bc <= 56 && m_in < m_end) { m_bb |= (uint64_t) *m_in++ << m_bc; m_bc += 8; } } while (0)
#define NTSM_TAKE(n) do { m_bb >>= (n); m_bc -= (n); } while (0)

/* Dynamic block header (RFC 1951 3.2.7).  Returns false with m_mode = DONE on bad data; truncation is reported by
 * leaving m_mode at HEADER with m_stored = 0xFFFFFFFF (checked by the caller). */
bool Inflate::read_dynamic_header()
{
	auto bits = [&](unsigned n, uint32_t *v) -> bool {
		NTSM_REFILL_SAFE();
		if (m_bc < n) return false;
		*v = (uint32_t) (m_bb & ((1ull << n) - 1));
		NTSM_TAKE(n);
		return true;
	};
	uint32_t hlit, hdist, hclen;
	if (!bits(5, &hlit) || !bits(5, &hdist) || !bits(4, &hclen)) { m_stored = 0xFFFFFFFFu; return false; }
	const int nlen = (int) hlit + 257, ndist = (int) hdist + 1, ncode = (int) hclen + 4;
	if (nlen > 286 || ndist > 30) return false;                /* zlib: "too many length or distance symbols" */
	uint8_t pl[19] = { 0 };
	for (int i = 0; i < ncode; ++i) {
		uint32_t v;
		if (!bits(3, &v)) { m_stored = 0xFFFFFFFFu; return false; }
		pl[kPrecodeOrder[i]] = (uint8_t) v;
	}
	uint32_t pre[128];
	if (!build_table(pre, 7, 128, pl, 19, PRECODE)) return false;
	uint8_t lens[320];
	int i = 0;
	while (i < nlen + ndist) {
		NTSM_REFILL_SAFE();
		const uint32_t e = pre[m_bb & 127u];
		const unsigned nb = e & 31u;
		if (e & F_ERR) {
			if (m_in == m_end && m_bc < 7) { m_stored = 0xFFFFFFFFu; return false; }
			return false;
		}
		if (nb > m_bc) { m_stored = 0xFFFFFFFFu; return false; }
		NTSM_TAKE(nb);
		const uint32_t sym = e >> 16;
		if (sym < 16) { lens[i++] = (uint8_t) sym; continue; }
		uint32_t rep, val = 0;
		if (sym == 16) {
			if (i == 0) return false;                            /* "invalid bit length repeat" */
			val = lens[i - 1];
			if (!bits(2, &rep)) { m_stored = 0xFFFFFFFFu; return false; }
			rep += 3;
		} else if (sym == 17) {
			if (!bits(3, &rep)) { m_stored = 0xFFFFFFFFu; return false; }
			rep += 3;
		} else {
			if (!bits(7, &rep)) { m_stored = 0xFFFFFFFFu; return false; }
			rep += 11;
		}
		if (i + (int) rep > nlen + ndist) return false;
		while (rep--) lens[i++] = (uint8_t) val;
	}
	if (lens[256] == 0) return false;                          /* "missing end-of-block" */
	if (!build(m_lit, kLitBits, kLitSize, lens, nlen, false)) return false;
	if (!build(m_dist, kDistBits, kDistSize, lens + nlen, ndist, true)) return false;
	return true;
}

/* Symbols of one Huffman block.  Returns MORE (output limit), STREAM_END (here: end of BLOCK), TRUNCATED, DATA_ERROR. */
NTSM_INFLATE_CLONES Inflate::Status Inflate::run_huffman(uint8_t *buf, size_t *out, size_t out_stop)
{
	const uint8_t *in = m_in;
	uint64_t bb = m_bb;
	unsigned bc = m_bc;
	uint8_t *op = buf + *out;
	uint8_t *const op0 = op, *const op_stop = buf + out_stop;
	const uint64_t total0 = m_total;
	Status st = MORE;
	constexpr uint32_t lmask = (1u << kLitBits) - 1, dmask = (1u << kDistBits) - 1;

	/* the fast refill leaves true-but-unaccounted stream bits above bc: mask them off before the state is stored */
#define SAVE() do { m_in = in; m_bc = bc; m_bb = bc >= 64 ? bb : (bb & ((1ull << bc) - 1)); \
		m_total = total0 + (uint64_t) (op - op0); *out = (size_t) (op - buf); } while (0)

	/* ---- fast loop: at least 16 input bytes and 274 output bytes of room, no checks inside ---- */
	if (m_end - in >= 16 && op_stop - op > 274) {
		const uint8_t *const in_fast = m_end - 16;
		uint8_t *const op_fast = op_stop - 274;
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
				*op++ = (uint8_t) (e >> 16);
				e = m_lit[bb & lmask];
				if (e & F_SUB) { bb >>= kLitBits; bc -= kLitBits; e = m_lit[(e >> 16) + (uint32_t) (bb & ((1u << ((e >> 8) & 15u)) - 1))]; }
				sv = bb; bb >>= (e & 31u); bc -= (e & 31u);
				if (e & F_LIT) {
					*op++ = (uint8_t) (e >> 16);
					e = m_lit[bb & lmask];
					if (e & F_SUB) { bb >>= kLitBits; bc -= kLitBits; e = m_lit[(e >> 16) + (uint32_t) (bb & ((1u << ((e >> 8) & 15u)) - 1))]; }
					sv = bb; bb >>= (e & 31u); bc -= (e & 31u);
					if (e & F_LIT) { *op++ = (uint8_t) (e >> 16); continue; }
				}
			}
			if (e & (F_EOB | F_ERR)) {
				if (e & F_ERR) { SAVE(); return DATA_ERROR; }
				SAVE();
				return STREAM_END;
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
			if ((uint64_t) dist > total0 + (uint64_t) (op - op0)) { SAVE(); return DATA_ERROR; }   /* "invalid distance too far back" */
			const uint8_t *src = op - dist;
			uint8_t *const end = op + len;
			if (dist >= 8) {
				store64(op, load64(src));                             /* 16 bytes unconditionally: text is mostly matches of 6-9, a branch at 8 would be a coin toss */
				store64(op + 8, load64(src + 8));
				if (len > 16) {
					op += 16; src += 16;
					do { store64(op, load64(src)); op += 8; src += 8; } while (op < end);
				}
			} else if (dist == 1) {
				const uint64_t v = 0x0101010101010101ull * *src;
				do { store64(op, v); op += 8; } while (op < end);
			} else {
				do { *op++ = *src++; } while (op < end);
			}
			op = end;
		}
	}
	/* ---- careful loop: one symbol at a time, every bit accounted for (ends of the input and of the output window) ---- */
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
		if (e & F_LIT) { *op++ = (uint8_t) (e >> 16); continue; }
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
		if ((uint64_t) dist > total0 + (uint64_t) (op - op0)) { st = DATA_ERROR; break; }
		const uint8_t *src = op - dist;
		for (uint32_t i = 0; i < len; ++i) op[i] = src[i];
		op += len;
	}
	SAVE();
	return st;
#undef SAVE
}

Inflate::Status Inflate::open_block()
{
	NTSM_REFILL_SAFE();
	if (m_bc < 3) return TRUNCATED;
	m_last = (m_bb & 1u) != 0;
	const unsigned type = (unsigned) (m_bb >> 1) & 3u;
	NTSM_TAKE(3);
	if (type == 0) {
		NTSM_TAKE(m_bc & 7u);                             /* skip to the byte boundary */
		NTSM_REFILL_SAFE();
		if (m_bc < 32) return TRUNCATED;
		const uint32_t len = (uint32_t) (m_bb & 0xFFFFu), nlen = (uint32_t) ((m_bb >> 16) & 0xFFFFu);
		if ((len ^ 0xFFFFu) != nlen) { m_mode = DONE; return DATA_ERROR; }   /* "invalid stored block lengths" */
		NTSM_TAKE(32);
		m_stored = len;
		m_mode = STORED;
	} else if (type == 1) {
		set_fixed();
		m_mode = HUFFMAN;
	} else if (type == 2) {
		m_stored = 0;
		if (!read_dynamic_header()) {
			const bool trunc = m_stored == 0xFFFFFFFFu;
			m_stored = 0;
			m_mode = DONE;
			return trunc ? TRUNCATED : DATA_ERROR;
		}
		m_mode = HUFFMAN;
	} else {
		m_mode = DONE;
		return DATA_ERROR;                                /* "invalid block type" */
	}
	return MORE;
}

Inflate::Status Inflate::run(uint8_t *buf, size_t *out, size_t out_stop)
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
			while (m_stored && m_bc) {                           /* whole bytes still in the bit buffer */
				if (*out >= out_stop) return MORE;
				buf[(*out)++] = (uint8_t) (m_bb & 0xFFu);
				NTSM_TAKE(8);
				--m_stored;
				++m_total;
			}
			if (m_stored) {
				if (*out >= out_stop) return MORE;
				size_t n = m_stored;
				if (n > (size_t) (m_end - m_in)) n = (size_t) (m_end - m_in);
				if (n > out_stop - *out) n = out_stop - *out;
				memcpy(buf + *out, m_in, n);
				m_in += n;
				*out += n;
				m_total += n;
				m_stored -= (uint32_t) n;
				if (m_stored) {
					if (m_in == m_end) return TRUNCATED;
					return MORE;
				}
			}
			m_mode = HEADER;
			break;
		}
		case HUFFMAN: {
			const Status st = run_huffman(buf, out, out_stop);
			if (st == STREAM_END) { m_mode = HEADER; break; }     /* end of this block */
			if (st == DATA_ERROR) m_mode = DONE;
			return st;
		}
		}
	}
}

#undef NTSM_REFILL_SAFE
#undef NTSM_TAKE

} // namespace ntsm
