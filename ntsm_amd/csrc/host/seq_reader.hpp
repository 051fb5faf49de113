/*
 * seq_reader.hpp -- streaming FASTA/FASTQ record reader over zlib (plain or gzip input).
 *
 * Host-side replacement for the reference's kseq parser as ntsmCount uses it
 * (vendor/kseq.h:177-219 instantiated over gzread at src/FingerPrint.hpp:27).  Plain input goes through zlib's
 * transparent mode like the reference; gzip input is inflated by a decoder thread (gz_stream.hpp, inflate.hpp)
 * unless NTSM_ZLIB_ONLY is set, which forces gzread for everything.  Record boundaries
 * and sequence bytes must match it exactly because every byte of seq.s reaches the k-mer window
 * (invalid bytes reset it and still count in "Total Bases Considered").  Behaviours kept:
 *   - the first header is searched for anywhere ('>' or '@'), later ones only at a line start;
 *   - name = header up to the first isspace() byte; the rest of the line is ignored;
 *   - sequence lines are concatenated until a line starts with '>', '@' or '+'; empty lines are
 *     skipped; one trailing '\r' per line is dropped only when the sequence so far is longer than
 *     one byte; any other byte (spaces, digits, high bytes) is sequence;
 *   - after '+': quality lines are consumed until their total length reaches the sequence length;
 *     a length mismatch or a missing quality block ends the FILE (negative return), exactly like
 *     the reference's `while (l >= 0 ...)` loop (src/FingerPrint.hpp:66-69).
 */
#ifndef NTSM_SEQ_READER_HPP
#define NTSM_SEQ_READER_HPP
#include <cstdint>
#include <string>
#include <vector>
#include <zlib.h>

#include "gz_stream.hpp"

namespace ntsm {

class SeqReader {
public:
	SeqReader() = default;
	~SeqReader() { close(); }
	SeqReader(const SeqReader &) = delete;
	SeqReader &operator=(const SeqReader &) = delete;

	/* offset: start reading at this byte of the (uncompressed) stream; it must be a record boundary */
	bool open(const std::string &path, uint64_t offset = 0);
	/* continue on an open gzip stream (after the parallel phase of parallel_gz_fastq.hpp handed back what it did not parse:
	 * the stream is positioned at a record boundary) */
	bool open_stream(std::unique_ptr<GzStream> gz);
	void close();
	/* Next record: returns the sequence length (>= 0), -1 at end of file, -2 on a truncated
	 * quality block, -3 on a stream error.  seq()/name() are valid until the next call. */
	int64_t next();
	/* sequence bytes of the current record (length = next()'s return value); may point into the
	 * read buffer (fast path) or into an internal vector (general path) */
	const char *seq_data() const { return seq_ptr_; }
	const std::string &name() const { return name_; }

private:
	static constexpr int kBuf = 1 << 22;
	/* Fast path for the common record shapes (4-line FASTQ, 2-line FASTA, no CR, record inside the
	 * buffer): returns false when the record needs the general byte-by-byte path. */
	bool fast_record(int64_t *len);
	int get();                                           /* next byte, -1 EOF, -3 error */
	/* append bytes up to (not including) the next '\n' (line = true) or isspace byte to dst;
	 * returns <0 exactly when the reference's ks_getuntil2 would; *delim = byte that stopped it */
	int64_t until(bool line, std::vector<char> &dst, int *delim);
	bool refill();                                       /* false at EOF/error */

	int source_read(void *dst, unsigned len);            /* gzread semantics: > 0 bytes, 0 end, -1 error */
	gzFile f_ = nullptr;                                 /* plain files (zlib passes them through) */
	std::unique_ptr<GzStream> gz_;                       /* gzip files: decoder thread (gz_stream.hpp) */
	std::vector<unsigned char> buf_;
	int beg_ = 0, end_ = 0;
	bool eof_ = false;
	int pending_ = 0;                                    /* header byte already consumed ('>' / '@'), 0 = none */
	std::vector<char> seq_, qual_, scratch_;
	const char *seq_ptr_ = nullptr;
	std::string name_;
	uint64_t qual_len_ = 0;
};

} // namespace ntsm
#endif
