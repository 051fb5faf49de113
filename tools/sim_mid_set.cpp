/* tools/sim_mid_set.cpp -- host-only model behind DESIGN.md section 4.2c's round-5 study (site sets of 2-3 M k-mers).
 *
 * For a site k-mer set (raw uint64 canonical 19-mer codes) and a flat read stream it builds, with the SHIPPED hash functions
 * (ntsm_device.h), the structures of a filter organisation
 *     [minimizer Bloom of W KiB] -> blocked filter of B KiB (128-bit blocks, 4 bits per key) -> [drain Bloom of 2^d bits] -> key table
 * walks the reads the way the kernel does (12-mer minimizer runs, one request per run) and reports per base:
 *     runs, Bloom-word requests, block requests, first-level positives, drain-Bloom requests, bucket requests, hits
 * and, from the per-line access counts, the L2 misses an LRU cache of one XCD (4 MiB, 128-byte lines) would take under the
 * independent-reference model (Che's approximation): every line j with access probability p_j per base is resident with
 * probability 1 - exp(-p_j T); lines that are never re-used (stream lines weighted by `alpha`, bucket lines) occupy the cache
 * for T each; T solves  sum_j (1 - exp(-p_j T)) + pollution * T = C.
 * `alpha` (how much of the nt-hinted stream really occupies the LRU stack) and C are calibrated on the two measured points
 * of profiles/r04_mid (1.54 M keys: 0.0181 misses per base; 2.50 M keys: 0.0632) -- see tools/mid_study_model.py.
 *
 * g++ -O2 -std=c++17 -I ntsm_amd/csrc -o /tmp/sim_mid_set tools/sim_mid_set.cpp
 * sim_mid_set keys.u64 reads.bin  bloom_KiB(0 = none) blocks_KiB drain_log2(0 = none)  [alpha=1] [cache_lines=32768] */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unordered_set>
#include <vector>

struct uint2 { uint32_t x, y; };                    /* ntsm_device.h names the HIP vector types in its parameter block */
struct uint4 { uint32_t x, y, z, w; };
#include "ntsm_device.h"

static std::vector<uint8_t> slurp(const char *p) { FILE *f = fopen(p, "rb"); if (!f) { perror(p); exit(1); } fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); std::vector<uint8_t> v(n); if (fread(v.data(), 1, n, f) != (size_t) n) exit(1); fclose(f); return v; }
static uint64_t revcomp(uint64_t x, int k) { uint64_t rc = 0; for (int b = 0; b < k; ++b) rc |= (3ull - ((x >> (2 * b)) & 3ull)) << (2 * (k - 1 - b)); return rc; }

static uint32_t minimizer(uint64_t fw, uint64_t rc, int k, int m)
{
	const uint32_t mm = (1u << (2 * m)) - 1u;
	uint32_t mz = 0xFFFFFFFFu;
	for (int j = 0; j + m <= k; ++j) {
		const uint32_t a = (uint32_t) (fw >> (2 * j)) & mm, b = (uint32_t) (rc >> (2 * (k - m - j))) & mm;
		mz = std::min(mz, ntsm_mmer_hash_m(std::min(a, b), (uint32_t) m));
	}
	return mz;
}

struct Lines {                                      /* access counts per 128-byte line of one structure */
	std::vector<uint32_t> n;
	explicit Lines(size_t bytes) : n((bytes + 127) / 128, 0) {}
	void touch(size_t byte) { ++n[byte >> 7]; }
};

int main(int argc, char **argv)
{
	if (argc < 6) { fprintf(stderr, "usage: sim_mid_set keys.u64 reads.bin bloom_KiB blocks_KiB drain_log2 [alpha] [cache_lines]\n"); return 1; }
	const int k = 19, m = 12;
	std::vector<uint8_t> kb = slurp(argv[1]), reads = slurp(argv[2]);
	const uint64_t *keys = (const uint64_t *) kb.data();
	const size_t n = kb.size() / 8;
	const uint32_t bloom_words = (uint32_t) atoi(argv[3]) * 256u, n_blocks = (uint32_t) atoi(argv[4]) * 64u;
	const int drain_log2 = atoi(argv[5]);
	const double alpha = argc > 6 ? atof(argv[6]) : 1.0;
	const double C = argc > 7 ? atof(argv[7]) : 32768.0;
	const uint64_t mask = (1ull << (2 * k)) - 1;

	std::vector<uint32_t> bloom(bloom_words ? bloom_words : 1, 0u), blocks((size_t) n_blocks * 4, 0u), drain(drain_log2 ? (1ull << drain_log2) / 32 : 1, 0u);
	std::unordered_set<uint64_t> keyset(keys, keys + n);
	std::unordered_set<uint32_t> site_mz;
	const uint32_t pshift = drain_log2 ? 32 - (drain_log2 - 5) : 0;
	for (size_t i = 0; i < n; ++i) {
		const uint64_t x = keys[i], rc = revcomp(x, k);
		const uint32_t mz = minimizer(x, rc, k, m);
		site_mz.insert(mz);
		const uint32_t u = ntsm_kmer_sum(ntsm_code_top(x, k), ntsm_code_top(rc, k)), um = ntsm_kmer_mix(u);
		uint32_t *blk = &blocks[(size_t) ntsm_range(ntsm_block_hash(mz), n_blocks) * 4];
		blk[0] |= 1u << NTSM_KBIT0(u); blk[1] |= 1u << NTSM_KBIT1(um); blk[2] |= 1u << NTSM_KBIT2(um); blk[3] |= 1u << NTSM_KBIT3(um);
		if (drain_log2) { const uint32_t f = ntsm_fold(x), g1 = ntsm_h1(f), g2 = ntsm_h2(f); drain[g1 >> pshift] |= (1u << (g2 & 31u)) | (1u << ((g2 >> 5) & 31u)); }
	}
	if (bloom_words) for (uint32_t mz : site_mz) { const uint32_t h = ntsm_block_hash(mz); bloom[ntsm_range(h, bloom_words)] |= (1u << NTSM_BLOOM_BIT0(h)) | (1u << NTSM_BLOOM_BIT1(h)); }

	uint8_t lut[256]; for (int i = 0; i < 256; ++i) lut[i] = 4;
	lut['A'] = lut['a'] = 0; lut['C'] = lut['c'] = 1; lut['G'] = lut['g'] = 2; lut['T'] = lut['t'] = lut['U'] = lut['u'] = 3;
	Lines l_bloom((size_t) bloom_words * 4), l_blocks((size_t) n_blocks * 16), l_drain(drain.size() * 4);
	uint64_t fw = 0, rc = 0, bases = 0, windows = 0, runs = 0, member = 0, bloom_pass = 0, positives = 0, drain_pass = 0, hits = 0;
	int run = 0; bool have = false; uint32_t prev = 0; bool cur_ok = false; const uint32_t *cur = nullptr;
	for (uint8_t c : reads) {
		const uint8_t code = lut[c];
		if (c != 'N' || true) ++bases;
		if (code > 3) { run = 0; have = false; continue; }
		fw = ((fw << 2) | code) & mask; rc = (rc >> 2) | ((uint64_t) (3 - code) << (2 * (k - 1)));
		if (++run < k) continue;
		++windows;
		const uint32_t mz = minimizer(fw, rc, k, m);
		if (!have || mz != prev) {                              /* a new run: what the lane requests */
			++runs; have = true; prev = mz;
			member += site_mz.count(mz);
			const uint32_t h = ntsm_block_hash(mz);
			cur_ok = true;
			if (bloom_words) {
				const uint32_t wi = ntsm_range(h, bloom_words);
				l_bloom.touch((size_t) wi * 4);
				cur_ok = ((bloom[wi] >> NTSM_BLOOM_BIT0(h)) & (bloom[wi] >> NTSM_BLOOM_BIT1(h)) & 1u) != 0;
			}
			if (cur_ok) {
				++bloom_pass;
				const uint32_t bi = ntsm_range(h, n_blocks);
				l_blocks.touch((size_t) bi * 16);
				cur = &blocks[(size_t) bi * 4];
			}
		}
		if (!cur_ok) continue;
		const uint64_t canon = fw < rc ? fw : rc;
		const uint32_t u = ntsm_kmer_sum(ntsm_code_top(fw, k), ntsm_code_top(rc, k)), um = ntsm_kmer_mix(u);
		if (!(((cur[0] >> NTSM_KBIT0(u)) & (cur[1] >> NTSM_KBIT1(um)) & (cur[2] >> NTSM_KBIT2(um)) & (cur[3] >> NTSM_KBIT3(um))) & 1u)) continue;
		++positives;
		bool go = true;
		if (drain_log2) {
			const uint32_t f = ntsm_fold(canon), g1 = ntsm_h1(f), g2 = ntsm_h2(f);
			l_drain.touch((size_t) (g1 >> pshift) * 4);
			go = ((drain[g1 >> pshift] >> (g2 & 31u)) & (drain[g1 >> pshift] >> ((g2 >> 5) & 31u)) & 1u) != 0;
		}
		if (!go) continue;
		++drain_pass;
		hits += keyset.count(canon);
	}
	const double B = (double) bases;
	/* Che's approximation.  p_j per base; never re-used lines: stream (alpha / 128 per base), bucket reads (+ the atomic's line is the same) */
	std::vector<double> p;
	for (const Lines *L : { &l_bloom, &l_blocks, &l_drain }) for (uint32_t c : L->n) if (c) p.push_back(c / B);
	const double pollution = alpha / 128.0 + drain_pass / B;
	auto occupancy = [&](double T) { double s = pollution * T; for (double x : p) s += 1.0 - std::exp(-x * T); return s; };
	double lo = 0, hi = 1e12;
	if (occupancy(hi) < C) lo = hi; else for (int it = 0; it < 200; ++it) { const double mid = 0.5 * (lo + hi); (occupancy(mid) < C ? lo : hi) = mid; }
	const double T = lo;
	auto misses_of = [&](const Lines &L) { double s = 0; for (uint32_t c : L.n) if (c) { const double x = c / B; s += x * std::exp(-x * T); } return s; };
	const double m_bloom = bloom_words ? misses_of(l_bloom) : 0, m_blocks = misses_of(l_blocks), m_drain = drain_log2 ? misses_of(l_drain) : 0;
	const double req = 1.0 / 128 + (bloom_words ? runs / B : 0) + bloom_pass / B + (drain_log2 ? positives / B : 0) + drain_pass / B + hits / B;
	const double miss = 1.0 / 128 + m_bloom + m_blocks + m_drain + drain_pass / B + hits / B;   /* bucket reads and hit atomics go to the fabric */
	printf("{\"keys\": %zu, \"bloom_KiB\": %u, \"blocks_KiB\": %u, \"drain_log2\": %d, \"alpha\": %.3f, \"cache_lines\": %.0f, \"footprint_MiB\": %.3f,\n"
	       " \"runs_per_base\": %.4f, \"runs_member\": %.4f, \"runs_passing_bloom\": %.4f, \"positives_per_base\": %.5f, \"bucket_reads_per_base\": %.5f, \"hits_per_base\": %.5f,\n"
	       " \"first_level_pass_rate_of_windows\": %.5f, \"l2_requests_per_base\": %.4f, \"pred_misses_per_base\": %.4f, \"pred_miss_bloom\": %.4f, \"pred_miss_blocks\": %.4f, \"pred_miss_drain\": %.4f, \"che_T_bases\": %.3g}\n",
	       n, bloom_words / 256u, n_blocks / 64u, drain_log2, alpha, C, (bloom_words * 4.0 + n_blocks * 16.0 + (drain_log2 ? (1ull << drain_log2) / 8.0 : 0)) / 1048576.0,
	       runs / B, (double) member / runs, (double) bloom_pass / runs, positives / B, drain_pass / B, hits / B, (double) positives / windows, req, miss, m_bloom, m_blocks, m_drain, T);
	return 0;
}
