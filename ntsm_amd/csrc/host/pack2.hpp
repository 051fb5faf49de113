/*
 * pack2.hpp -- host side of the packed producer lanes (include/ntsm_hip.h: ntsm_lane_acquire_packed): sequence bytes are
 * reduced to what the count kernel can tell apart before they cross PCIe.
 *
 * The reference's byte table (vendor/KseqHashIterator.hpp:114-127) maps every byte to one of five classes: codes 0..3
 * (A a 0x00 | C c 0x01 | G g 0x02 | T t U u 0x03) or "invalid", and the k-mer iterator sees nothing else of a byte.  A
 * stream position is therefore fully described by 2 bits of code + 1 bit "valid": 3/8 byte instead of 1, and no byte
 * value needs an exception list (an invalid byte is an 'N' as far as counting goes).  Layout of a packed batch of P
 * positions (P a multiple of 32):
 *     codes[P / 4]   position p -> bits 2(p & 3) .. 2(p & 3) + 1 of codes[p >> 2]   (0 for invalid positions)
 *     valid[P / 8]   position p -> bit p & 7 of valid[p >> 3]
 * Reads are appended at positions that are multiples of 8, so every read is followed by 1..8 invalid positions instead
 * of exactly one 'N' (more terminators change no k-mer: an invalid position only restarts the window,
 * vendor/KseqHashIterator.hpp:106).
 */
#ifndef NTSM_PACK2_HPP
#define NTSM_PACK2_HPP
#include <cstdint>

namespace ntsm {

/* Append `len` sequence bytes at position `pos` (a multiple of 8) of a packed batch; returns the position the next read
 * starts at: (pos + len + 8) & ~7.  Writes whole groups of 32 positions counted from pos: up to 31 positions past the
 * returned one are overwritten with "invalid", so the batch buffers need room for pack2_extent(pos, len) positions.
 * Reads nothing beyond seq[len - 1]. */
uint64_t pack2_append(uint8_t *codes, uint8_t *valid, uint64_t pos, const char *seq, uint64_t len);

/* positions pack2_append() may write for a read of `len` bytes appended at `pos` */
inline uint64_t pack2_extent(uint64_t pos, uint64_t len) { return pos + (len & ~31ull) + 32; }

/* the implementation in use: "avx512vbmi", "avx2" or "scalar" (chosen at run time from the CPU's feature bits) */
const char *pack2_impl();
/* test hooks: force the portable implementation / 0 = best available, 1 = portable, 2 = at most AVX2 */
void pack2_force_scalar(bool on);
void pack2_force_impl(int impl);

} // namespace ntsm
#endif
