/*
 * early_ingest.hpp -- parsing that starts before the GPU context exists (the reference's clock,
 * src/ntSeqMatchCount.cpp:175-182, covers the table build and the scan one after the other).
 *
 * A run spends its first 0.25-0.3 s loading the sites file and building / uploading the tables; nothing of that depends on
 * the reads.  With -t N (and no -m: the result then does not depend on the order of the reads) the first input file -- by
 * default only if it is gzip, see Options::early_kinds for the measurement -- is therefore opened at once and parsed by the same block-parallel machinery as later (parallel_fastq.hpp for a plain FASTQ,
 * parallel_gz_fastq.hpp over the decoder pool for gzip), only into ordinary memory instead of pinned lane slots: every
 * parsing thread packs its reads (pack2.hpp: 2-bit codes + validity bits, 3/8 byte per position) into chunks laid out
 * exactly like a packed lane slot.  Once the context is there, the feeders copy the finished chunks into their lanes and
 * submit them (Feeder::submitChunk) while the rest of the file is still being parsed.  What the parallel phase cannot take
 * (records that are not plain 4-line FASTQ) goes through the sequential reader into the same chunks, so the whole file is
 * covered either way; a file that cannot be opened is left to the ordinary path, which reports it like the reference.
 * The chunks in flight are bounded (a feeder that is not there yet makes the parsers wait).  Once the feeders exist a gzip
 * stream is handed over to them at a record boundary (hand_over / release_stream): its rest goes straight into the lanes.
 */
#ifndef NTSM_EARLY_INGEST_HPP
#define NTSM_EARLY_INGEST_HPP
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace ntsm {

class ParallelFastq;
class GzStream;

struct PackedChunk {                                   /* one packed batch in ordinary memory, laid out like a packed lane slot */
	uint8_t *codes = nullptr, *valid = nullptr;        /* cap / 4 and cap / 8 bytes, both inside `mem` */
	uint64_t cap = 0, pos = 0, n_bases = 0;            /* positions: capacity, used; sum of the read lengths */
	uint32_t n_reads = 0;
	/* 2 MiB-aligned and advised as huge pages: a file's worth of chunks is gigabytes of memory touched for the first time,
	 * and with 4 KiB pages the faults (580 k for a 12.6 GB FASTQ, taken by 16 threads through one mm lock) cost six times the
	 * parsing itself -- measured 0.76 s against 0.12 s for the same file through the (reused, pinned) lane slots */
	void *mem = nullptr;
	void reserve(uint64_t positions);
	PackedChunk() = default;
	~PackedChunk();
	PackedChunk(const PackedChunk &) = delete;
	PackedChunk &operator=(const PackedChunk &) = delete;
};

class EarlyIngest {
public:
	/* chunk_positions: capacity of a chunk (= of a feeder's packed lane slot); max_chunks: chunks that may exist at once */
	/* kinds: 1 = plain FASTQ, 2 = gzip, 3 = both */
	EarlyIngest(std::string path, unsigned n_parsers, unsigned n_decoders, uint64_t block_bytes, uint64_t gz_min_bytes, uint64_t chunk_positions, size_t max_chunks, int kinds = 3);
	~EarlyIngest();
	EarlyIngest(const EarlyIngest &) = delete;
	EarlyIngest &operator=(const EarlyIngest &) = delete;

	bool taken() const { return m_taken; }             /* false: the file is not for this path (not there, tiny, FIFO ...): use the ordinary one */
	/* next finished chunk; false once the file has been consumed entirely and every chunk has been handed out */
	bool next(std::unique_ptr<PackedChunk> *out);
	void recycle(std::unique_ptr<PackedChunk> c);      /* hand a drained chunk back (its memory is reused) */
	/* gzip input: the consumers are there now.  No further piece of the stream is parsed into chunks; what is in hand is
	 * finished (next() still delivers it) and the stream, positioned at a record boundary, can be collected with
	 * release_stream() once next() has returned false -- its rest is better parsed straight into the lanes (one copy and
	 * gigabytes of first-touched memory less).  No effect on a plain file. */
	void hand_over() { m_handOver.store(true); }
	std::unique_ptr<GzStream> release_stream();        /* null: the whole file went through the chunks */
	/* true once something went wrong that LOSES reads (a chunk could not be allocated, the rest of the file could not be
	 * reopened): valid after next() returned false / release_stream(); the caller must not print counts (the reference's
	 * contract for a file it cannot read is exit(1) with a message, src/FingerPrint.hpp:51-57) */
	bool failed() const { return m_failed.load(); }
	const std::string &error() const { return m_error; }
	/* test hook: the n-th chunk allocation from now on fails (0 = off); compiled in, armed only through the host C API */
	static void debug_fail_allocation(long nth);
	/* statistics for the phase line, valid after next() returned false */
	uint64_t records() const { return m_records; }
	uint64_t parallel_records() const { return m_parallelRecords; }
	const std::string &how() const { return m_how; }
	double parse_seconds() const { return m_parseSeconds; }

	/* Sink of the parallel parsers (parallel_fastq.hpp): one per parsing thread */
	class Sink {
	public:
		explicit Sink(EarlyIngest *owner) : m_owner(owner) {}
		bool has_room(uint64_t len) const;
		void feed(const char *seq, uint64_t len);
		void flush();
		void discard();
		void begin_block(size_t) {}
		uint64_t fed = 0;
	private:
		EarlyIngest *m_owner;
		std::unique_ptr<PackedChunk> m_cur;
	};

private:
	friend class Sink;
	void run();
	std::unique_ptr<PackedChunk> blank(uint64_t min_positions);   /* waits while max_chunks are out */
	void publish(std::unique_ptr<PackedChunk> c);

	const std::string m_path;
	const unsigned m_nParsers, m_nDecoders;
	const uint64_t m_blockBytes, m_chunkPositions;
	const size_t m_maxChunks;
	bool m_taken = false;
	std::unique_ptr<ParallelFastq> m_plain;            /* exactly one of the two when taken */
	std::unique_ptr<GzStream> m_gz;
	std::thread m_thread;
	std::mutex m_mu;
	std::condition_variable m_cv;
	std::deque<std::unique_ptr<PackedChunk>> m_ready, m_free;
	size_t m_out = 0;                                  /* chunks that exist outside m_free */
	bool m_done = false, m_abandon = false;
	std::atomic<bool> m_handOver { false };
	std::atomic<bool> m_failed { false };
	std::string m_error;                               /* written once, under m_mu, before m_failed is set */
	void fail(const std::string &what);                /* records the first error and abandons the run */
	std::unique_ptr<GzStream> m_rest;                  /* set by run() when the parallel phase ended on hand_over() */
	uint64_t m_records = 0, m_parallelRecords = 0;
	std::string m_how;
	double m_parseSeconds = 0;
};

} // namespace ntsm
#endif
