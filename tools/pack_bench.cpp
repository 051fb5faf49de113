/* tools/pack_bench.cpp -- one thread's rate of the producer lanes' packer (ntsm_amd/csrc/host/pack2.cpp) in each of its forms:
 * 150-base reads at a stride of 151 bytes, packed back to back into one batch, best of seven passes.
 *   g++ -O3 -std=c++17 -I ntsm_amd/csrc/host -o build/pack_bench tools/pack_bench.cpp ntsm_amd/csrc/host/pack2.cpp && build/pack_bench
 * Build container (Xeon, 2.1 GHz): portable 1.05, AVX2 3.3, AVX-512 VBMI 14.6 Gbases/s; the GPU box (EPYC 9575F): 2.6 / 7.3 / 24.2
 * (DESIGN.md section 5.1). */
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>

#include "pack2.hpp"

int main()
{
	const size_t n = 200000, L = 150, stride = 151;
	std::vector<char> s(n * stride);
	std::mt19937 rng(1);
	for (auto &c : s) c = "ACGT"[rng() & 3];
	std::vector<uint8_t> codes((n * 160 + 64) / 4), valid((n * 160 + 64) / 8);
	for (int impl : { 1, 2, 0 }) {                            /* portable, at most AVX2, best the CPU has */
		ntsm::pack2_force_impl(impl);
		double best = 1e9;
		for (int rep = 0; rep < 7; ++rep) {
			const auto t0 = std::chrono::steady_clock::now();
			uint64_t pos = 0;
			for (size_t r = 0; r < n; ++r) pos = ntsm::pack2_append(codes.data(), valid.data(), pos, s.data() + r * stride, L);
			const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			if (dt < best) best = dt;
		}
		printf("%-12s %6.2f ns/read  %6.2f Gbases/s\n", ntsm::pack2_impl(), best / n * 1e9, n * L / best / 1e9);
	}
	return 0;
}
