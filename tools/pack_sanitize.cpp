/* ASan + UBSan driver of the host packer (ntsm_amd/csrc/host/pack2.cpp), compiled and run by tests/test_host_cpu.py:
 * reads of every length 0..299 made of arbitrary bytes appended at arbitrary (multiple-of-8) positions into buffers of EXACTLY
 * pack2_extent() positions, with the sequence in a heap block of exactly its length -- any write past the promised extent or
 * read past seq[len - 1] aborts.  Every implementation the CPU has. */
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "pack2.hpp"

int main()
{
	std::mt19937 rng(5);
	for (int force = 0; force < 3; ++force) {                  /* best available (AVX-512 VBMI where the CPU has it), portable, at most AVX2 */
		ntsm::pack2_force_impl(force);
		for (int t = 0; t < 30000; ++t) {
			const size_t len = t < 300 ? (size_t) t : rng() % 300;
			std::vector<char> seq(len);
			for (auto &c : seq) c = (char) (rng() % 256);
			const uint64_t pos = (uint64_t) (rng() % 64) * 8, ext = ntsm::pack2_extent(pos, len);
			std::vector<uint8_t> codes((ext + 3) / 4, 0xAA), valid((ext + 7) / 8, 0x55);
			const uint64_t r = ntsm::pack2_append(codes.data(), valid.data(), pos, seq.data(), len);
			if (r != ((pos + len + 8) & ~7ull) || r > ext) { fprintf(stderr, "bad return %llu\n", (unsigned long long) r); return 1; }
			for (uint64_t p = 0; p < pos; ++p)                     /* nothing in front of the read is touched */
				if (((codes[p >> 2] >> (2 * (p & 3))) & 3) != 2 || ((valid[p >> 3] >> (p & 7)) & 1) != ((p & 1) ? 0u : 1u)) { fprintf(stderr, "clobbered %llu\n", (unsigned long long) p); return 1; }
			for (uint64_t p = pos + len; p < r; ++p)               /* the terminator positions are invalid */
				if ((valid[p >> 3] >> (p & 7)) & 1) { fprintf(stderr, "terminator valid\n"); return 1; }
		}
	}
	printf("pack2 sanitize ok (%s + avx2 + scalar)\n", (ntsm::pack2_force_impl(0), ntsm::pack2_impl()));
	return 0;
}
