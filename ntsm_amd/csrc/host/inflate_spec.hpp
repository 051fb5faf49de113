/*
 * inflate_spec.hpp -- speculative DEFLATE decoding from the middle of a stream (parallel inflate of ONE plain gzip
 * member; replaces the single gzread stream under vendor/kseq.h:229).
 *
 * A deflate stream can only be decoded in order: every block may copy from the 32 KiB of output in front of it.  The
 * way around it (the two-pass scheme of pugz / rapidgzip) has two parts, both here:
 *
 *   find()   Where does a block start?  Every bit offset from a given one is tried as the header of a non-final dynamic
 *            block: 3 header bits, HLIT / HDIST in range, a COMPLETE precode, code lengths that fill exactly
 *            HLIT + HDIST entries, an end-of-block code, complete literal/length and distance codes (the checks zlib
 *            applies, RFC 1951 3.2.7).  Random bits pass that about once in 10^10 positions; a false candidate is harmless
 *            anyway, because the caller only accepts a chunk whose start is the exact bit at which the decoding of the
 *            stream in front of it ended (gz_parallel.cpp) -- it costs time, never correctness.  Stored and fixed blocks
 *            are not searched for (a chunk then starts at the next dynamic block, the caller decodes the gap in order).
 *   run16()  Decoding without the window: the output is written as 16-bit symbols, a literal as its byte value, a byte
 *            that would have been copied out of the unknown window as MARKER | j (j = index into the 32 KiB before the
 *            chunk).  The buffer starts with the 32768 markers themselves, so copies out of the window are ordinary
 *            copies and markers propagate through later copies by themselves.  Once the real window is known
 *            (resolve()), every marker is replaced by window[j].
 */
#ifndef NTSM_INFLATE_SPEC_HPP
#define NTSM_INFLATE_SPEC_HPP
#include "inflate.hpp"

namespace ntsm {

class SpecInflate : public Inflate {
public:
	static constexpr uint16_t kMarker = 0x8000u;
	static constexpr size_t kWindow = 32768;

	/* First bit offset in [from_bit, to_bit) (offsets from `base`) at which a non-final dynamic block header parses; the
	 * decoder is left positioned AFTER that header (block open).  ~0ull: none. */
	uint64_t find(const uint8_t *base, const uint8_t *end, uint64_t from_bit, uint64_t to_bit);

	/* Decode 16-bit symbols into sym (sym[0, kWindow) must hold the markers, see fill_markers) from *out on, until *out >=
	 * out_stop, the stream's final block ends (STREAM_END) or a block ends at a bit offset >= the set_stop() position
	 * (BLOCK_STOP).  May write up to 280 symbols past out_stop. */
	Status run16(uint16_t *sym, size_t *out, size_t out_stop);

	static void fill_markers(uint16_t *sym);                 /* sym[j] = kMarker | j, j < kWindow */
	/* bytes of n symbols: literal -> itself, marker -> window[j] (window = the 32768 bytes in front of the chunk; only the
	 * last `valid` of them exist: a marker below kWindow - valid refers to data before the start of the member).  Returns
	 * false on such a marker (the chunk is then decoded again in order, which reports the error where zlib does).
	 * crc != nullptr: *crc is continued over the n bytes (crc32_fast.hpp), 32 KiB at a time right behind the bytes' making. */
	static bool resolve(const uint16_t *sym, size_t n, const uint8_t *window, size_t valid, uint8_t *out, uint32_t *crc = nullptr);

private:
	NTSM_INFLATE_CLONES Status run_huffman16(uint16_t *buf, size_t *out, size_t out_stop);
};

} // namespace ntsm
#endif
