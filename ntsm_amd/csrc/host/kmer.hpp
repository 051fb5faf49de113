/*
 * kmer.hpp -- host-side rolling canonical k-mer (site loading only; reads are k-merised on the GPU).
 * Follows vendor/KseqHashIterator.hpp:95-127: A/a=0 C/c=1 G/g=2 T/t/U/u=3, bytes 0..3 map to
 * themselves, anything else restarts the window; canonical = min(forward, reverse complement).
 */
#ifndef NTSM_KMER_HPP
#define NTSM_KMER_HPP
#include <cstdint>

namespace ntsm {

inline int base_code(unsigned char b)
{
	switch (b) {
	case 0: case 'A': case 'a': return 0;
	case 1: case 'C': case 'c': return 1;
	case 2: case 'G': case 'g': return 2;
	case 3: case 'T': case 't': case 'U': case 'u': return 3;
	default: return 4;
	}
}

inline uint64_t kmer_mask(unsigned k) { return k >= 32 ? 0ull : ((1ull << (2 * k)) - 1); }

/* Calls f(canonical_code, end_pos) for every window of k consecutive valid bases; end_pos is the
 * index one past the window's last base (KseqHashIterator::getPos, :62). */
template <class F>
inline void for_each_kmer(const char *s, uint64_t len, unsigned k, F f)
{
	const uint64_t mask = kmer_mask(k);
	const unsigned shift = (2 * (k - 1)) & 63;
	uint64_t fw = 0, rv = 0;
	unsigned run = 0;
	for (uint64_t i = 0; i < len; ++i) {
		const int c = base_code((unsigned char) s[i]);
		if (c > 3) { run = 0; fw = rv = 0; continue; }
		fw = ((fw << 2) | (uint64_t) c) & mask;
		rv = (rv >> 2) | ((uint64_t) (3 - c) << shift);
		if (++run >= k) f(fw < rv ? fw : rv, i + 1);
	}
}

} // namespace ntsm
#endif
