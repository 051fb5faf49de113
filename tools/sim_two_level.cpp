/* Host-only simulation behind DESIGN.md's large-site-set design (configs[4]): for a site k-mer set (raw uint64 canonical
 * codes) and a flat read stream, per minimizer length m:
 *   runs/base        density of minimizer runs (one first-level request each)
 *   distinct         distinct site minimizers
 *   member           share of read runs whose minimizer is a site minimizer (what an exact on-chip set would pass)
 *   bloom(S, b)      share of read runs passing a one-word Bloom of S MiB with b bits per key (members included)
 * g++ -O2 -std=c++17 -o /tmp/sim_two_level tools/sim_two_level.cpp;  sim_two_level keys.u64 reads.bin [k] */
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unordered_set>
#include <vector>

static std::vector<uint8_t> slurp(const char *p) { FILE *f = fopen(p, "rb"); if (!f) { perror(p); exit(1); } fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); std::vector<uint8_t> v(n); if (fread(v.data(), 1, n, f) != (size_t) n) exit(1); fclose(f); return v; }
static uint64_t revcomp(uint64_t x, int k) { uint64_t rc = 0; for (int b = 0; b < k; ++b) rc |= (3ull - ((x >> (2 * b)) & 3ull)) << (2 * (k - 1 - b)); return rc; }
static inline uint32_t ohash(uint32_t canon) { return canon * 0x9E3779B1u; }
static inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

static uint32_t minimizer(uint64_t fw, uint64_t rc, int k, int m)
{
	const uint32_t mm = m == 16 ? 0xFFFFFFFFu : (1u << (2 * m)) - 1u;
	uint32_t mz = 0xFFFFFFFFu;
	for (int j = 0; j + m <= k; ++j) {
		const uint32_t a = (uint32_t) (fw >> (2 * j)) & mm, b = (uint32_t) (rc >> (2 * (k - m - j))) & mm;
		mz = std::min(mz, ohash(std::min(a, b)));
	}
	return mz;
}

int main(int argc, char **argv)
{
	if (argc < 3) return 1;
	const int k = argc > 3 ? atoi(argv[3]) : 19;
	std::vector<uint8_t> kb = slurp(argv[1]), reads = slurp(argv[2]);
	const uint64_t *keys = (const uint64_t *) kb.data();
	const size_t n = kb.size() / 8;
	const uint64_t mask = (1ull << (2 * k)) - 1;
	uint8_t lut[256]; for (int i = 0; i < 256; ++i) lut[i] = 4;
	lut['A'] = lut['a'] = 0; lut['C'] = lut['c'] = 1; lut['G'] = lut['g'] = 2; lut['T'] = lut['t'] = 3;
	printf("%zu keys, %zu stream bytes, k = %d\n", n, reads.size(), k);
	for (int m = 11; m <= 16; ++m) {
		std::unordered_set<uint32_t> site;
		site.reserve(n);
		for (size_t i = 0; i < n; ++i) site.insert(minimizer(keys[i], revcomp(keys[i], k), k, m));
		struct Bl { double mib; int bits; std::vector<uint32_t> w; };
		std::vector<Bl> bl;
		for (double mib : { 1.0, 2.0, 3.0 }) for (int bits : { 1, 2, 3 }) bl.push_back({ mib, bits, std::vector<uint32_t>((size_t) (mib * 262144), 0u) });
		auto word_bits = [](uint32_t mz, const Bl &b, size_t &wi) { const uint32_t h = mix(mz); wi = (size_t) (((uint64_t) h * b.w.size()) >> 32); const uint32_t g = mix(mz ^ 0x68E31DA4u); uint32_t v = 0; for (int q = 0; q < b.bits; ++q) v |= 1u << ((g >> (5 * q)) & 31u); return v; };
		for (uint32_t mz : site) for (Bl &b : bl) { size_t wi; const uint32_t v = word_bits(mz, b, wi); b.w[wi] |= v; }
		uint64_t fw = 0, rc = 0; int run = 0; uint64_t pos = 0, runs = 0, member = 0; uint32_t prev = 0; bool have = false;
		std::vector<uint64_t> pass(bl.size(), 0);
		for (uint8_t c : reads) {
			const uint8_t code = lut[c];
			if (code > 3) { run = 0; have = false; continue; }
			fw = ((fw << 2) | code) & mask; rc = (rc >> 2) | ((uint64_t) (3 - code) << (2 * (k - 1)));
			if (++run < k) continue;
			++pos;
			const uint32_t mz = minimizer(fw, rc, k, m);
			if (have && mz == prev) continue;
			have = true; prev = mz; ++runs;
			if (site.count(mz)) ++member;
			for (size_t q = 0; q < bl.size(); ++q) { size_t wi; const uint32_t v = word_bits(mz, bl[q], wi); if ((bl[q].w[wi] & v) == v) ++pass[q]; }
		}
		printf("m=%2d w=%d: runs/kmer %.4f  distinct site minimizers %zu  member %.4f  |", m, k - m + 1, (double) runs / pos, site.size(), (double) member / runs);
		for (size_t q = 0; q < bl.size(); ++q) printf(" %.0fMiB/%db %.3f", bl[q].mib, bl[q].bits, (double) pass[q] / runs);
		printf("\n");
	}
	return 0;
}
