/*
 * inflate.hpp -- DEFLATE (RFC 1951) decoder for the gzip ingest path.
 *
 * The reference reads every input through zlib's gzread (src/FingerPrint.hpp:27, vendor/kseq.h:75-85).  zlib
 * 1.2.11 inflates FASTQ text at 0.2-0.4 GB/s, an order of magnitude below the parser, so a gzipped input is bound
 * by inflate alone.  This decoder is written for that one job: the whole compressed file is in memory (mmap), so
 * there is no input refill logic; a 64-bit bit buffer is topped up with one unaligned load; literal/length symbols
 * resolve through an 11-bit table (distance: 8-bit) with sub-tables for longer codes; matches are copied 8 bytes
 * at a time.  It accepts exactly the streams zlib's inflate accepts and fails (DATA_ERROR) on the ones it rejects
 * (over-subscribed or incomplete code sets other than a single one-bit code, missing end-of-block code, too many
 * length/distance symbols, distances beyond the start of the member or the 32 KiB window, stored-length mismatch).
 * Test infrastructure compares it with zlib on every gzip input of the test-suite and on fuzzed streams
 * (tests/test_host_cpu.py).
 */
#ifndef NTSM_INFLATE_HPP
#define NTSM_INFLATE_HPP
#include <cstddef>
#include <cstdint>

/* Two builds of the hot loop, picked at load time (GNU ifunc): BMI2 (shrx / bzhi: +10-15 %) and baseline x86-64.
 * Sanitizer builds keep a single version: their runtimes are not up yet when ifunc resolvers run. */
#if defined(__x86_64__) && !defined(__SANITIZE_THREAD__) && !defined(__SANITIZE_ADDRESS__)
#define NTSM_INFLATE_CLONES __attribute__((target_clones("bmi2", "default")))
#else
#define NTSM_INFLATE_CLONES
#endif

namespace ntsm {

class Inflate {
public:
	enum Status {
		MORE,          /* stopped because the output position reached out_stop; call again */
		STREAM_END,    /* the final block ended; in() is the first byte after the deflate stream (byte aligned) */
		TRUNCATED,     /* the input ended inside the stream; everything decodable has been written */
		DATA_ERROR,    /* invalid stream */
		BLOCK_STOP     /* set_stop(): a block just ended at or beyond the stop position; the next header is unread */
	};

	/* Start a new deflate stream at [in, in_end).  `window` = number of bytes before the current output position
	 * that belong to this stream (0 at the start of a gzip member). */
	void reset(const uint8_t *in, const uint8_t *in_end);
	/* The same in the middle of a stream (parallel gzip decoding, gz_parallel.cpp): the next block header starts `bit`
	 * bits (0..7) into *in, and `total` bytes of this stream have been produced before it (the caller keeps the last
	 * 32 KiB of them in front of the output position). */
	void reset_at(const uint8_t *in, unsigned bit, const uint8_t *in_end, uint64_t total);
	/* Make run() return BLOCK_STOP at the first block boundary whose bit offset from `base` is >= stop_bit (checked
	 * before every block header, the first one included).  base = nullptr: never (the default after reset()). */
	void set_stop(const uint8_t *base, uint64_t stop_bit) { m_base = base; m_stop_bit = stop_bit; }
	/* bit offset of the next unread bit from `base`; valid between run() calls */
	uint64_t bit_pos(const uint8_t *base) const { return (uint64_t) (m_in - base) * 8u - m_bc; }
	bool last_block_seen() const { return m_last; }

	/* Decode into buf (the bytes [0, *out) already hold this stream's history as far as it exists) until *out >=
	 * out_stop or the stream ends.  A call may write up to 258 + 8 bytes past out_stop. */
	Status run(uint8_t *buf, size_t *out, size_t out_stop);

	const uint8_t *in() const;          /* next unread input byte (valid after STREAM_END: bit buffer returned) */
	uint64_t total_out() const { return m_total; }   /* bytes produced since reset() */

	/* decode-table entry: bits 0-4 = bits to consume -- the code AND, for a length or distance, its extra bits (one shift takes
	 * both; <= 11 + 13) --, 8-11 = how many of them are extra bits (pointer entries: sub-table bits), 12-15 = flags,
	 * 16-31 = value (literal, base length / distance, sub-table offset) */
	static constexpr uint32_t F_LIT = 0x8000u, F_EOB = 0x4000u, F_SUB = 0x2000u, F_ERR = 0x1000u;

protected:
	static constexpr int kLitBits = 11, kDistBits = 8;
	static constexpr int kLitSize = (1 << kLitBits) + 1024, kDistSize = (1 << kDistBits) + 512;
	enum Mode { HEADER, STORED, HUFFMAN, DONE };
	bool build(uint32_t *table, int table_bits, int max_size, const uint8_t *lens, int n, bool is_dist);
	bool read_dynamic_header();
	void set_fixed();
	/* one block header at the current position: MORE = a block is open (m_mode STORED / HUFFMAN), else the status to return */
	Status open_block();
	NTSM_INFLATE_CLONES Status run_huffman(uint8_t *buf, size_t *out, size_t out_stop);

	const uint8_t *m_in = nullptr, *m_end = nullptr;
	uint64_t m_bb = 0;
	unsigned m_bc = 0;
	Mode m_mode = HEADER;
	bool m_last = false;
	uint32_t m_stored = 0;
	uint64_t m_total = 0;
	const uint8_t *m_base = nullptr;
	uint64_t m_stop_bit = 0;
	uint32_t m_lit[kLitSize], m_dist[kDistSize];
};

} // namespace ntsm
#endif
