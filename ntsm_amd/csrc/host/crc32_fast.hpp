/*
 * crc32_fast.hpp -- CRC-32 (IEEE 802.3, the gzip check of RFC 1952 8.) with carry-less multiplication.
 * zlib 1.2.11's table-driven crc32() runs at ~0.9 GB/s, below the inflater; folding 64 bytes per step with
 * PCLMULQDQ (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel 2009;
 * constants for the reflected polynomial 0x1DB710641) runs at > 10 GB/s.  Falls back to zlib's crc32() when the CPU
 * lacks the instruction or for the unaligned tail.  Same calling convention as zlib: crc32_fast(0, NULL, 0) = 0.
 */
#ifndef NTSM_CRC32_FAST_HPP
#define NTSM_CRC32_FAST_HPP
#include <cstddef>
#include <cstdint>

namespace ntsm {
uint32_t crc32_fast(uint32_t crc, const uint8_t *buf, size_t len);
}
#endif
