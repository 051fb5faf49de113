/* Host-only: density (sampled positions per k-mer window step) of minimizer-like sampling schemes at the kernel's
 * parameters (19-mer windows, 12-mer anchors: w = 8 candidates), on a random sequence, FORWARD strand only (a strand-
 * symmetric version can only be denser).  Behind DESIGN.md section 4.2's "lower-density sampling" row.
 *   random minimizer          the kernel's scheme: smallest hash among the 8 12-mers
 *   mod-minimizer (t)         position x of the smallest t-mer among the 19 - t + 1 t-mers of the window, anchor = 12-mer at x mod 8
 *   open-closed order         12-mers ranked by (open syncmer, closed syncmer, other) with s-mers of length s, then by hash
 *   lower bound               ceil((w + k) / w) / (w + k) for forward schemes (Kille et al. 2024)
 * g++ -O2 -o /tmp/sim_density tools/sim_sampling_density.cpp */
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
static inline uint32_t hsh(uint32_t x, uint32_t seed) { x ^= seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
int main()
{
	const int K = 19, M = 12, W = K - M + 1, N = 4000000;
	std::mt19937_64 rng(7);
	std::vector<uint8_t> g(N + 64);
	for (auto &c : g) c = rng() & 3;
	auto code = [&](int p, int len) { uint32_t v = 0; for (int i = 0; i < len; ++i) v = (v << 2) | g[p + i]; return v; };
	auto density = [&](auto pick, const char *name) {
		long changes = 0; int prev = -1;
		for (int p = 0; p + K <= N; ++p) { const int a = p + pick(p); if (a != prev) ++changes; prev = a; }
		printf("%-44s density %.4f\n", name, (double) changes / (N - K + 1));
	};
	density([&](int p) { int best = 0; uint32_t bh = 0xFFFFFFFFu; for (int j = 0; j < W; ++j) { const uint32_t h = hsh(code(p + j, M), 1); if (h < bh) { bh = h; best = j; } } return best; }, "random minimizer (the kernel's)");
	for (int t = 3; t <= 8; ++t) {
		char nm[64]; snprintf(nm, sizeof nm, "mod-minimizer, t = %d", t);
		density([&](int p) { int best = 0; uint32_t bh = 0xFFFFFFFFu; for (int j = 0; j + t <= K; ++j) { const uint32_t h = hsh(code(p + j, t), 2); if (h < bh) { bh = h; best = j; } } return best % W; }, nm);
	}
	for (int s = 3; s <= 8; ++s) {
		char nm[64]; snprintf(nm, sizeof nm, "open-closed syncmer order, s = %d", s);
		density([&](int p) {
			int best = 0; uint64_t bk = ~0ull;
			for (int j = 0; j < W; ++j) {
				int arg = 0; uint32_t mh = 0xFFFFFFFFu;
				for (int q = 0; q + s <= M; ++q) { const uint32_t h = hsh(code(p + j + q, s), 3); if (h < mh) { mh = h; arg = q; } }
				const int n_s = M - s + 1;
				const int cls = arg == n_s / 2 ? 0 : (arg == 0 || arg == n_s - 1) ? 1 : 2;      /* open (middle), closed (ends), other */
				const uint64_t key = ((uint64_t) cls << 32) | hsh(code(p + j, M), 1);
				if (key < bk) { bk = key; best = j; }
			}
			return best; }, nm);
	}
	printf("%-44s density %.4f\n", "forward-scheme lower bound", (double) ((W + M + W - 1) / W) / (W + M));
	return 0;
}
