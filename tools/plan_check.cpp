/* tools/plan_check.cpp -- host-only check of the functions the filter builder and the kernel share (ntsm_device.h), compiled
 * and run by tests/test_host_cpu.py.  For every k of ntsm_fast_plan(): the candidate offsets are symmetric in the k-mer, and
 * for random k-mers the minimizer, the filter-bit hash u and the block index are the same for a k-mer and its reverse
 * complement -- a filter addressed by anything else would miss one strand of a site k-mer. */
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>

struct uint2 { uint32_t x, y; }; struct uint4 { uint32_t x, y, z, w; };   /* host build without HIP headers */
#include "../ntsm_amd/csrc/ntsm_device.h"

static uint64_t revcomp(uint64_t x, unsigned k) { uint64_t rc = 0; for (unsigned b = 0; b < k; ++b) rc |= (3ull - ((x >> (2 * b)) & 3ull)) << (2 * (k - 1 - b)); return rc; }

static uint32_t minimizer(uint64_t x, uint64_t rc, const NtsmFastPlan &pl)
{
	const uint32_t mmask = (1u << (2 * pl.m)) - 1u;
	uint32_t mz = 0xFFFFFFFFu;
	for (uint32_t j = pl.a; j < pl.a + pl.w; ++j) {
		const uint32_t sub = (uint32_t) (x >> (2 * j)) & mmask, rsub = (uint32_t) (rc >> (2 * (pl.k - pl.m - j))) & mmask;
		mz = std::min(mz, ntsm_mmer_hash_m(std::min(sub, rsub), pl.m));
	}
	return mz;
}

int main()
{
	std::mt19937_64 rng(99);
	int bad = 0, planned = 0;
	for (int two = 0; two < 2; ++two)                  /* the one-level plans (12-mers), then the two-level ones (14-mers, 15 <= k <= 31) */
	for (uint32_t k = 1; k <= 32; ++k) {
		const NtsmFastPlan pl = ntsm_fast_plan(k, two != 0);
		if (two && (k < 15 || k > 31)) {               /* no two-level plan: the call must hand back the one-level one */
			const NtsmFastPlan one = ntsm_fast_plan(k);
			if (pl.mode != one.mode || pl.m != one.m || pl.w != one.w || pl.a != one.a) { printf("k=%u: two-level request changed the plan\n", k); ++bad; }
			continue;
		}
		if (pl.mode < 0) { if (k >= 13 && k <= 31) { printf("k=%u has no plan\n", k); ++bad; } continue; }
		++planned;
		if (pl.k != k || pl.m != (two ? 14u : 12u) || pl.w < 2 || pl.w > 9 || pl.a + pl.w + pl.m - 1 > k || 2 * pl.a + pl.w - 1 != k - pl.m) {
			printf("k=%u: plan m=%u w=%u a=%u is not symmetric inside the k-mer\n", k, pl.m, pl.w, pl.a); ++bad; continue;
		}
		if ((pl.mode == 0) != (k == NTSM_FAST_K) || (pl.mode != 0 && pl.mode != (int) pl.w)) { printf("k=%u: mode %d\n", k, pl.mode); ++bad; }
		const NtsmBlockMap map = { 3u << 16 };
		for (int t = 0; t < 20000; ++t) {
			const uint64_t x = rng() & ((k == 32) ? ~0ull : ((1ull << (2 * k)) - 1)), rc = revcomp(x, k);
			if (two) {                                 /* the Bloom word and its two bits come from the same strand-symmetric minimizer */
				const uint32_t h = ntsm_block_hash(minimizer(x, rc, pl)), hr = ntsm_block_hash(minimizer(rc, x, pl));
				if (h != hr || ntsm_range(h, 589824u) >= 589824u || NTSM_BLOOM_BIT0(h) > 31u || NTSM_BLOOM_BIT1(h) > 31u) { printf("k=%u: Bloom asymmetry\n", k); ++bad; break; }
			}
			const uint32_t mz = minimizer(x, rc, pl), mzr = minimizer(rc, x, pl);
			const uint32_t u = ntsm_kmer_sum(ntsm_code_top(x, k), ntsm_code_top(rc, k)), ur = ntsm_kmer_sum(ntsm_code_top(rc, k), ntsm_code_top(x, k));
			if (mz != mzr || u != ur || ntsm_block_idx(mz, map) != ntsm_block_idx(mzr, map) || ntsm_block_idx(mz, map) >= map.n_blocks) {
				printf("k=%u: strand asymmetry for %016llx\n", k, (unsigned long long) x); ++bad; break;
			}
		}
	}
	printf("plans: %d, problems: %d\n", planned, bad);
	return bad != 0 || planned != 19 + 17;
}
