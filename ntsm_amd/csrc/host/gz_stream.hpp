/*
 * gz_stream.hpp -- gzip input for the sequence reader: a decoder thread inflates the (memory-mapped) file with
 * ntsm::Inflate and hands 1 MiB pieces of text to the parsing thread, which also checks each member's CRC-32 and
 * length.  Replaces gzread (src/FingerPrint.hpp:27 instantiates kseq over it) for files that start with the gzip
 * magic; the call semantics follow gzread as kseq uses it: read() returns the number of bytes (> 0), 0 at the end
 * of the data -- also when the file is truncated inside a member, like gzread, which reports Z_BUF_ERROR only
 * through gzerror() --, -1 on invalid data (bad header, corrupt deflate stream, CRC or length mismatch).
 * Concatenated members are decoded back to back; anything after the last member that does not start with the
 * gzip magic is ignored (zlib gz_look: "trailing garbage").
 *
 * Block-parallel inflate: a deflate stream is sequential, but BGZF files (bgzip, samtools, htslib: members of at
 * most 64 KiB whose header carries their compressed size in a "BC" extra field, SAM spec 4.1) can be cut without
 * decoding.  With more than one decoder thread the producer walks the member headers, hands groups of members to
 * worker threads (each member: own Inflate, CRC-32 and length checked by the worker) and passes the finished groups
 * on in file order; at the first member that is not BGZF, is incomplete or fails, the sequential decoder takes over
 * at that byte, so every input keeps the single-thread semantics.
 *
 * An ordinary gzip file (one long deflate stream) is decoded in parallel as well (gz_parallel.cpp, inflate_spec.hpp): the
 * compressed bytes are cut into chunks; worker threads find the first dynamic block that starts inside their chunk and
 * decode from there WITHOUT the 32 KiB window, writing 16-bit symbols in which bytes copied out of the unknown window are
 * markers; the producer decodes in order from the member's start and, whenever it arrives at a block boundary that is
 * exactly the start bit of a finished chunk, splices the chunk in: its markers are resolved against the now known window
 * (by the workers, in parallel, CRC-32 of the bytes included) and the in-order decoder continues where the chunk ended.
 * A chunk whose start does not coincide, that failed, or that saw no dynamic block is simply dropped -- the in-order
 * decoder goes through its range itself, which is also what reports truncation and corruption with zlib's bytes and
 * verdict.  Member CRCs are combined from the per-chunk CRCs (crc32_combine).
 */
#ifndef NTSM_GZ_STREAM_HPP
#define NTSM_GZ_STREAM_HPP
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace ntsm {

class GzStream {
public:
	GzStream() = default;
	~GzStream() { close(); }
	GzStream(const GzStream &) = delete;
	GzStream &operator=(const GzStream &) = delete;

	static bool is_gzip(const std::string &path);      /* regular file that starts with 1f 8b */
	/* decoder threads used for BGZF input by streams opened from now on (process-wide; default 1 = sequential) */
	static void set_decoder_threads(unsigned n);
	/* compressed bytes per chunk of the parallel decoder for plain gzip (process-wide; default 1 MiB; 0 = default) */
	static void set_parallel_chunk(size_t bytes);
	/* what the most recently finished parallel decode did (process-wide, for tests and -v): chunks spliced / dropped */
	static void last_parallel_stats(uint64_t out[2]);
	/* workers: the compressed bytes [lo, hi) of the mapping are decoded (drops their page-table entries, see gz_stream.cpp) */
	static void release_input(const uint8_t *lo, const uint8_t *hi);
	bool open(const std::string &path);
	int read(void *dst, unsigned len);
	void close();

	struct Piece {
		std::vector<uint8_t> data;
		size_t len = 0;
		bool member_end = false;                       /* after these bytes a member ends: check crc / isize */
		bool checked = false;                          /* BGZF group: the workers verified crc / isize already */
		bool have_crc = false;                         /* parallel plain gzip: `crc` is the CRC-32 of these len bytes (combined by the reader) */
		bool ready = true;                             /* false: queued in order, a worker is still filling it (guarded by m_mu) */
		uint32_t crc = 0, isize = 0;
		int status = 0;                                /* after these bytes: 0 = more, 1 = end of data, -1 = error */
	};
	/* Piece-wise consumption (parallel_gz_fastq.hpp: several parsing threads instead of one read() loop).  take() returns the
	 * next piece of text in stream order (len > 0), nullptr once the data has ended -- final_status() then says how: 1 = clean
	 * end (or truncated file: every decodable byte was delivered, like gzread), -1 = invalid data / CRC / length.  Member
	 * checks are made as the pieces pass, exactly as read() makes them.  One caller at a time (callers serialise).
	 * give_back() recycles a piece's buffer; unread() puts pieces back IN FRONT of everything not yet taken (also in front
	 * of what an earlier unread() left), the first one from byte `offset` on: the next read() or take() starts there (the
	 * sequential reader after a parallel phase; a second parallel phase after the early ingest handed the stream over). */
	std::unique_ptr<Piece> take();
	int final_status() const { return m_final; }
	void give_back(std::unique_ptr<Piece> p);
	void unread(std::deque<std::unique_ptr<Piece>> pieces, size_t offset);

private:
	void produce();
	struct Parallel;                                   /* gz_parallel.cpp: chunk workers of the parallel plain-gzip decoder */
	friend struct Parallel;
	const uint8_t *produce_bgzf(const uint8_t *p, unsigned n_threads);   /* returns where the sequential decoder continues, nullptr: reader gone */
	bool push(std::unique_ptr<Piece> p);               /* false: reader went away */
	std::unique_ptr<Piece> blank();

	const uint8_t *m_map = nullptr;
	size_t m_size = 0;
	int m_fd = -1;
	std::thread m_thread;
	std::mutex m_mu;
	std::condition_variable m_cv;
	std::deque<std::unique_ptr<Piece>> m_ready, m_free;
	bool m_stop = false;
	unsigned m_nThreads = 1;                           /* decoder threads / chunk size of this stream (the process-wide settings at open()) */
	size_t m_chunkBytes = 2u << 20;
	/* reader side */
	std::unique_ptr<Piece> pop();                      /* next piece of any kind with the member accounting applied; nullptr after the final one */
	std::deque<std::unique_ptr<Piece>> m_stash;        /* unread(): consumed before anything else */
	size_t m_stashOff = 0;
	std::unique_ptr<Piece> m_cur;
	size_t m_off = 0;
	uint32_t m_crc = 0;
	uint64_t m_len = 0;
	int m_final = 0;                                   /* 1 / -1 once the end / an error has been reached */
};

} // namespace ntsm
#endif
