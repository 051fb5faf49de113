/* False-positive rate of first-level filter layouts on a real site set, host only:
 *   filter_fp sites.fa  ->  FP of (A) four words x one bit each  (B) one word x rotated bit patterns
 * g++ -O2 -std=c++17 -I ntsm_amd/csrc -I ntsm_amd/csrc/host tools/filter_fp.cpp ntsm_amd/csrc/host/site_set.cpp ... */
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <random>
#include <vector>

struct uint2 { uint32_t x, y; }; struct uint4 { uint32_t x, y, z, w; };   /* host build without HIP headers */
#include "ntsm_device.h"
#include "site_set.hpp"

static uint64_t revcomp(uint64_t x, int k) { uint64_t rc = 0; for (int b = 0; b < k; ++b) rc |= (3ull - ((x >> (2 * b)) & 3ull)) << (2 * (k - 1 - b)); return rc; }
static uint32_t minimizer(uint64_t x)
{
	uint32_t mz = 0xFFFFFFFFu;
	for (int j = 0; j < NTSM_FAST_W; ++j) {
		const uint32_t sub = (uint32_t) (x >> (2 * j)) & NTSM_MMER_MASK;
		uint32_t rsub = 0;
		for (int b = 0; b < NTSM_FAST_M; ++b) rsub |= (3u - ((sub >> (2 * b)) & 3u)) << (2 * (NTSM_FAST_M - 1 - b));
		mz = std::min(mz, ntsm_mmer_hash(std::min(sub, rsub)));
	}
	return mz;
}
static inline uint32_t rotr(uint32_t x, unsigned s) { s &= 31; return s ? (x >> s) | (x << (32 - s)) : x; }

struct Scheme { const char *name; int kind; uint32_t p1, p2; };

int main(int argc, char **argv)
{
	ntsm::SiteSet S;
	if (argc < 2 || !S.load(argv[1], 19, false, std::cerr)) return 1;
	const uint64_t n_blocks = argc > 2 ? (uint64_t) atoll(argv[2]) * 64 : 3ull << 16;   /* argv[2]: filter size in KiB (default 3 MiB = 3 * 2^16 blocks) */
	NtsmBlockMap map; map.n_blocks = (uint32_t) n_blocks;
	printf("%zu site k-mers, %llu blocks = %.2f MiB, %.2f bits per key\n", S.keys.size(), (unsigned long long) n_blocks, n_blocks / 65536.0, n_blocks * 128.0 / S.keys.size());
	std::vector<Scheme> schemes = {
		{ "A: 4 words x 1 bit (current)", 0, 0, 0 },
		{ "A-ideal: 4 words x 1 bit, four independent 64-bit-mixed hashes", 5, 0, 0 },
		{ "A-um0: 4 words x 1 bit, bit 0 from um byte 0 instead of raw u byte 3", 6, 0, 0 },
		{ "A6: 4 words, 1+1+2+2 bits (second bits from um2 = um * odd)", 7, 0, 0 },
		{ "A8: 4 words x 2 bits", 8, 0, 0 },
		{ "B: 1 word, rotr(0x00010001,s1)|rotr(0x00000021,s2)", 1, 0x00010001u, 0x00000021u },
		{ "B: 1 word, rotr(0x00000101,s1)|rotr(0x00002001,s2)", 1, 0x00000101u, 0x00002001u },
		{ "B: 1 word, rotr(0x00010001,s1)|rotr(0x00000801,s2) (3-4 bits)", 1, 0x00010001u, 0x00000801u },
		{ "B5: 1 word, rotr(0x00010001,s1)|rotr(0x00200421,s2) (5 bits)", 1, 0x00010001u, 0x00200421u },
		{ "B3: 1 word, rotr(0x00000001,s1)|rotr(0x00010001,s2) (3 bits)", 1, 0x00000001u, 0x00010001u },
		{ "B24: 1 word, 24-bit hash, rotr(0x00000101,s1)|rotr(0x00002001,s2)", 2, 0x00000101u, 0x00002001u },
		{ "E24: 2 words x 2 bits, 24-bit hash, rotr(0x00000101,s1) / rotr(0x00002001,s2)", 3, 0x00000101u, 0x00002001u },
		{ "E24: 2 words x 2 bits, rotr(0x00010001,s1) / rotr(0x00000801,s2)", 3, 0x00010001u, 0x00000801u },
		{ "E24b: 2 words, 3+2 bits, rotr(0x00010101,s1) / rotr(0x00000801,s2)", 3, 0x00010101u, 0x00000801u },
		{ "E24c: 2 words, 3+3 bits, rotr(0x00010101,s1) / rotr(0x00200801,s2)", 3, 0x00010101u, 0x00200801u },
		{ "E32: 2 words x 2 bits, 32-bit hash", 4, 0x00000101u, 0x00002001u },
	};
	for (const Scheme &sc : schemes) {
		std::vector<uint32_t> blk(n_blocks * 4, 0);
		auto bits = [&](uint64_t x, uint32_t *w, uint32_t m[4]) {
			const uint64_t rc = revcomp(x, 19);
			const uint32_t u = ntsm_kmer_sum((uint32_t) (x >> 6), (uint32_t) (rc >> 6));
			if (sc.kind == 5) {
				uint64_t z = (x ^ (x >> 19)) * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
				m[0] = 1u << (z & 31); m[1] = 1u << ((z >> 8) & 31); m[2] = 1u << ((z >> 16) & 31); m[3] = 1u << ((z >> 24) & 31);
				*w = 4;
			} else if (sc.kind == 6) {
				const uint32_t um = ntsm_kmer_mix(u);
				m[0] = 1u << (um & 31u); m[1] = 1u << NTSM_KBIT1(um); m[2] = 1u << NTSM_KBIT2(um); m[3] = 1u << NTSM_KBIT3(um);
				*w = 4;
			} else if (sc.kind == 7 || sc.kind == 8) {
				const uint32_t um = ntsm_kmer_mix(u), u2 = um * 0x85EBCA6Bu;
				m[0] = 1u << NTSM_KBIT0(u); m[1] = 1u << NTSM_KBIT1(um); m[2] = 1u << NTSM_KBIT2(um); m[3] = 1u << NTSM_KBIT3(um);
				m[2] |= 1u << (31u - ((u2 >> 24) & 31u)); m[3] |= 1u << (31u - ((u2 >> 16) & 31u));
				if (sc.kind == 8) { m[0] |= 1u << (31u - ((u2 >> 8) & 31u)); m[1] |= 1u << (31u - (u2 & 31u)); }
				*w = 4;
			} else if (sc.kind == 0) {
				const uint32_t um = ntsm_kmer_mix(u);
				m[0] = 1u << NTSM_KBIT0(u); m[1] = 1u << NTSM_KBIT1(um); m[2] = 1u << NTSM_KBIT2(um); m[3] = 1u << NTSM_KBIT3(um);
				*w = 4;
			} else if (sc.kind == 1) {
				const uint32_t h = (uint32_t) (((uint64_t) u * 0x9E3779B1ull) >> 32);
				*w = (h >> 10) & 3u;
				m[0] = rotr(sc.p1, h) | rotr(sc.p2, h >> 5);
			} else {
				const uint32_t h = sc.kind == 4 ? (uint32_t) (((uint64_t) u * 0x9E3779B1ull) >> 32)
				                                : (uint32_t) (((uint64_t) (u & 0xFFFFFFu) * 0x9E3779ull) >> 32);   /* v_mul_hi_u32_u24: 16 bits */
				*w = (h >> 10) & 3u;
				if (sc.kind == 2) m[0] = rotr(sc.p1, h) | rotr(sc.p2, h >> 5);
				else { *w |= 8u; m[0] = rotr(sc.p1, h); m[1] = rotr(sc.p2, h >> 5); }
			}
		};
		for (uint64_t x : S.keys) {
			uint32_t w, m[4];
			bits(x, &w, m);
			uint32_t *b = &blk[(size_t) ntsm_block_idx(minimizer(x), map) * 4];
			if (w == 4) { b[0] |= m[0]; b[1] |= m[1]; b[2] |= m[2]; b[3] |= m[3]; }
			else if (w & 8u) { b[w & 3u] |= m[0]; b[(w & 3u) ^ 2u] |= m[1]; }
			else b[w] |= m[0];
		}
		std::mt19937_64 rng(7);
		uint64_t fp = 0, n = 0;
		std::vector<uint64_t> sorted(S.keys);
		std::sort(sorted.begin(), sorted.end());
		for (int i = 0; i < 4000000; ++i) {
			uint64_t x = rng() & ((1ull << 38) - 1);
			const uint64_t rc = revcomp(x, 19);
			if (rc < x) x = rc;
			if (std::binary_search(sorted.begin(), sorted.end(), x)) continue;
			uint32_t w, m[4];
			bits(x, &w, m);
			const uint32_t *b = &blk[(size_t) ntsm_block_idx(minimizer(x), map) * 4];
			const bool hit = w == 4 ? ((b[0] & m[0]) == m[0] && (b[1] & m[1]) == m[1] && (b[2] & m[2]) == m[2] && (b[3] & m[3]) == m[3])
			               : (w & 8u) ? ((b[w & 3u] & m[0]) == m[0] && (b[(w & 3u) ^ 2u] & m[1]) == m[1]) : ((b[w] & m[0]) == m[0]);
			fp += hit;
			++n;
		}
		uint64_t set = 0;
		for (uint32_t v : blk) set += (uint64_t) __builtin_popcount(v);
		printf("%-70s FP %.3f %%   fill %.1f %%\n", sc.name, 100.0 * (double) fp / (double) n, 100.0 * (double) set / (double) (n_blocks * 128));
	}
	return 0;
}
