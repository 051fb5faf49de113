/* tools/deflate_stats.cpp FILE.gz [max output bytes] -- what a gzip member is made of: literals and matches per block type,
 * match lengths, code lengths (how many symbols sit behind the decoders' first-level tables: literal / length codes longer
 * than 11 bits, distance codes longer than 8).  A bit-at-a-time reference parser of RFC 1951 (plain 10-byte gzip header only),
 * independent of ntsm_amd/csrc/host/inflate.cpp.  The numbers behind DESIGN.md section 5 ("the synthetic FASTQ is match text").
 * g++ -O2 -o build/deflate_stats tools/deflate_stats.cpp */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Bits {
	const uint8_t *p, *e;
	uint64_t bb = 0;
	unsigned bc = 0;
	unsigned take(unsigned n)
	{
		while (bc < n) { bb |= (uint64_t) (p < e ? *p++ : 0) << bc; bc += 8; }
		const unsigned v = (unsigned) (bb & ((1ull << n) - 1));
		bb >>= n;
		bc -= n;
		return v;
	}
};

struct Code {                                                     /* canonical Huffman code, decoded one bit at a time */
	uint16_t count[16], symbol[320];
	void build(const uint8_t *len, int n)
	{
		memset(count, 0, sizeof count);
		for (int i = 0; i < n; ++i) count[len[i]]++;
		count[0] = 0;
		uint16_t offs[16];
		offs[1] = 0;
		for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t) (offs[l] + count[l]);
		for (int i = 0; i < n; ++i) if (len[i]) symbol[offs[len[i]]++] = (uint16_t) i;
	}
	int decode(Bits &br, int *nbits)
	{
		int code = 0, first = 0, index = 0;
		for (int l = 1; l <= 15; ++l) {
			code |= (int) br.take(1);
			const int c = count[l];
			if (code - c < first) { *nbits = l; return symbol[index + (code - first)]; }
			index += c;
			first += c;
			first <<= 1;
			code <<= 1;
		}
		return -1;
	}
};

int main(int argc, char **argv)
{
	if (argc < 2) return 1;
	FILE *f = fopen(argv[1], "rb");
	if (!f) return 2;
	std::vector<uint8_t> d;
	{ uint8_t buf[1 << 16]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n); }
	fclose(f);
	const uint64_t limit = argc > 2 ? strtoull(argv[2], nullptr, 10) : ~0ull;
	Bits br { d.data() + 10, d.data() + d.size() };
	static const uint16_t lbase[] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
	static const uint8_t lext[] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
	static const uint16_t dbase[] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
	static const uint8_t dext[] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
	uint64_t blocks[3] = { 0, 0, 0 }, lits = 0, matches = 0, mlen = 0, litbits = 0, lenbits = 0, distbits = 0, far = 0, out = 0, lit_long = 0, len_long = 0, dist_long = 0;
	uint64_t hist[11] = { 0 };
	int max_ll = 0, max_dl = 0;
	for (;;) {
		const unsigned last = br.take(1), type = br.take(2);
		if (type > 2) { printf("invalid block type\n"); return 3; }
		++blocks[type];
		if (type == 0) {
			br.bb = 0; br.bc = 0;
			const unsigned len = br.p[0] | (br.p[1] << 8);
			br.p += 4 + len;
			out += len;
		} else {
			Code L, D;
			uint8_t lens[320];
			if (type == 1) {
				for (int i = 0; i < 144; ++i) lens[i] = 8;
				for (int i = 144; i < 256; ++i) lens[i] = 9;
				for (int i = 256; i < 280; ++i) lens[i] = 7;
				for (int i = 280; i < 288; ++i) lens[i] = 8;
				L.build(lens, 288);
				for (int i = 0; i < 30; ++i) lens[i] = 5;
				D.build(lens, 30);
			} else {
				const int nl = (int) br.take(5) + 257, nd = (int) br.take(5) + 1, nc = (int) br.take(4) + 4;
				static const uint8_t ord[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
				uint8_t cl[19] = { 0 };
				for (int i = 0; i < nc; ++i) cl[ord[i]] = (uint8_t) br.take(3);
				Code C;
				C.build(cl, 19);
				int i = 0;
				while (i < nl + nd) {
					int nb;
					const int s = C.decode(br, &nb);
					if (s < 16) { lens[i++] = (uint8_t) s; continue; }
					int rep, v = 0;
					if (s == 16) { v = lens[i - 1]; rep = 3 + (int) br.take(2); }
					else if (s == 17) rep = 3 + (int) br.take(3);
					else rep = 11 + (int) br.take(7);
					while (rep--) lens[i++] = (uint8_t) v;
				}
				for (int j = 0; j < nl; ++j) if (lens[j] > max_ll) max_ll = lens[j];
				for (int j = 0; j < nd; ++j) if (lens[nl + j] > max_dl) max_dl = lens[nl + j];
				L.build(lens, nl);
				D.build(lens + nl, nd);
			}
			for (;;) {
				int nb;
				int s = L.decode(br, &nb);
				if (s < 0) { printf("bad code\n"); return 3; }
				if (s < 256) { ++lits; litbits += (uint64_t) nb; if (nb > 11) ++lit_long; ++out; continue; }
				if (s == 256) break;
				if (nb > 11) ++len_long;
				s -= 257;
				const unsigned len = lbase[s] + br.take(lext[s]);
				lenbits += (uint64_t) nb + lext[s];
				int nb2;
				const int ds = D.decode(br, &nb2);
				if (ds < 0 || ds > 29) { printf("bad distance code\n"); return 3; }
				if (nb2 > 8) ++dist_long;
				const unsigned dist = dbase[ds] + br.take(dext[ds]);
				distbits += (uint64_t) nb2 + dext[ds];
				++matches;
				mlen += len;
				out += len;
				if (dist > 16384) ++far;
				hist[len < 4 ? 0 : len < 6 ? 1 : len < 8 ? 2 : len < 10 ? 3 : len < 12 ? 4 : len < 16 ? 5 : len < 32 ? 6 : len < 64 ? 7 : len < 128 ? 8 : len < 258 ? 9 : 10]++;
			}
		}
		if (last || out > limit) break;
	}
	printf("%llu bytes out; blocks: %llu stored, %llu fixed, %llu dynamic; longest literal/length code %d bits, distance code %d bits\n", (unsigned long long) out,
	       (unsigned long long) blocks[0], (unsigned long long) blocks[1], (unsigned long long) blocks[2], max_ll, max_dl);
	printf("literals %llu (%.2f bits on average, %llu with codes > 11 bits); matches %llu (average length %.1f; length code + extra %.2f bits, %llu codes > 11 bits; distance code + extra %.2f bits, %llu codes > 8 bits, %.1f %% farther than 16 Ki)\n",
	       (unsigned long long) lits, lits ? (double) litbits / (double) lits : 0.0, (unsigned long long) lit_long, (unsigned long long) matches, matches ? (double) mlen / (double) matches : 0.0,
	       matches ? (double) lenbits / (double) matches : 0.0, (unsigned long long) len_long, matches ? (double) distbits / (double) matches : 0.0, (unsigned long long) dist_long, matches ? 100.0 * (double) far / (double) matches : 0.0);
	printf("per 315 bytes of output: %.1f literals + %.1f matches = %.2f bytes per token\n", 315.0 * (double) lits / (double) out, 315.0 * (double) matches / (double) out, (double) out / (double) (lits + matches));
	static const char *names[] = { "3", "4-5", "6-7", "8-9", "10-11", "12-15", "16-31", "32-63", "64-127", "128-257", "258" };
	printf("match lengths:");
	for (int i = 0; i < 11; ++i) printf("  %s: %.1f %%", names[i], matches ? 100.0 * (double) hist[i] / (double) matches : 0.0);
	printf("\n");
	return 0;
}
