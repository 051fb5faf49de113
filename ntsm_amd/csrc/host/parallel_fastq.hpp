/*
 * parallel_fastq.hpp -- block-parallel, single-pass ingest of ONE plain (uncompressed) FASTQ file.
 *
 * kseq (vendor/kseq.h:177-219) is a sequential state machine.  A file can still be parsed by N threads as long as
 * its records have the plain shape that the sequential reader's fast path accepts:
 *     @header \n SEQ \n +... \n QUAL \n     |QUAL| == |SEQ| >= 1, SEQ not starting with @ > +, no CR
 * After such a record kseq is back in its initial state (last_char == 0, positioned right after the quality
 * line), so "a run of strict records, then whatever follows handed to a fresh sequential reader at that byte
 * offset" yields exactly kseq's records.  That gives a scheme with no second pass and no rollback:
 *
 *   - the file is cut into fixed-size blocks, claimed by the threads in increasing order;
 *   - a thread finds the first record start inside its block by pattern (a line starting with '@' that parses as
 *     a strict record followed by another '@' or EOF), then parses strict records until it crosses the block end,
 *     copying the sequences into its own staging (Sink);
 *   - ORDERED COMMIT: nothing of block b is submitted before block b-1 has published the offset where its last
 *     record ended and that offset equals block b's guessed start.  Because b-1 was claimed earlier, the wait is
 *     short; the first block's start is byte 0.  A wrong guess or a non-strict record stops the parallel phase:
 *     everything before that offset has been committed, nothing after it has, and run() reports the offset for
 *     the sequential reader (SeqReader::open(path, offset)) to continue from.
 *
 * gzip input, FASTA, or small files are not eligible (open() returns false): the caller uses the sequential
 * reader for the whole file.
 *
 * Sink concept (one per thread; ntsm::Feeder and the test collector in host_capi.cpp):
 *     bool has_room(uint64_t len)   feed(seq, len) would not have to submit first
 *     void feed(const char *seq, uint64_t len)
 *     void flush()                  submit what is staged (only called for validated content)
 *     void discard()                drop what is staged
 *     void begin_block(size_t b)    a new block starts (staging is empty here)
 */
#ifndef NTSM_PARALLEL_FASTQ_HPP
#define NTSM_PARALLEL_FASTQ_HPP
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace ntsm {

class ParallelFastq {
public:
	struct Result {
		bool complete = false;      /* the whole file was consumed by the parallel phase */
		uint64_t resume = 0;        /* !complete: byte offset (a record boundary) the sequential reader continues from */
		uint64_t records = 0;       /* records committed by the parallel phase */
	};
	ParallelFastq() = default;
	~ParallelFastq();
	ParallelFastq(const ParallelFastq &) = delete;
	ParallelFastq &operator=(const ParallelFastq &) = delete;

	/* Map the file.  False: not eligible (use the sequential reader for all of it). */
	bool open(const std::string &path, uint64_t block_bytes = 16ull << 20);
	size_t n_blocks() const { return m_nBlocks; }
	uint64_t size() const { return m_size; }

	/* Parse with one thread per sink. */
	template <class Sink> Result run(const std::vector<Sink *> &sinks)
	{
		m_done.assign(m_nBlocks, 0);
		m_end.assign(m_nBlocks, 0);
		m_failBlock = kNoFail;
		m_resume = 0;
		m_next = 0;
		std::atomic<uint64_t> records(0);
		std::vector<std::thread> pool;
		for (Sink *s : sinks)
			pool.emplace_back([this, s, &records]() {
				uint64_t n = 0;
				for (size_t b = m_next++; b < m_nBlocks; b = m_next++) {
					n += work<Sink>(*s, b);
					release(b);
				}
				records += n;
			});
		for (auto &t : pool) t.join();
		Result r;
		r.complete = m_failBlock == kNoFail;
		r.resume = r.complete ? m_size : m_resume;
		if (r.complete && m_nBlocks && m_end[m_nBlocks - 1] != m_size) {   /* invariant: complete <=> the last block ended at EOF */
			r.complete = false;
			r.resume = m_end[m_nBlocks - 1];
		}
		r.records = records;
		return r;
	}

	/* one strict record at p: returns one past its quality newline and the sequence, or nullptr (also used by
	 * parallel_gz_fastq.hpp) */
	static const char *strict_record(const char *p, const char *e, const char **seq, uint64_t *len);

private:
	static constexpr uint64_t kNone = ~0ull;
	/* offset of the first plausible record start in [lo, hi), kNone if there is none */
	uint64_t find_start(uint64_t lo, uint64_t hi) const;
	/* ordered commit: wait for block b-1, then true iff no earlier block failed and b-1 ended at `first` (first ==
	 * kNone: the block has no record start and only passes its predecessor's end on).  A mismatch fails the
	 * parallel phase at b-1's end. */
	bool wait_start(size_t b, uint64_t first, uint64_t *prev_end);
	void publish(size_t b, uint64_t end);
	void fail(size_t b, uint64_t resume);
	/* drop block b's page-table entries (the page cache keeps the data): spreads the teardown of a multi-GB
	 * mapping over the worker threads instead of paying it single-threaded in munmap */
	void release(size_t b) const;

	template <class Sink> uint64_t work(Sink &s, size_t b)
	{
		const uint64_t lo = (uint64_t) b * m_block, hi = std::min(m_size, lo + m_block);
		bool started = false;
		uint64_t prev_end = 0, n = 0;
		if (m_failBlock.load(std::memory_order_relaxed) < b) return 0;   /* the parallel phase already stopped before this block */
		const uint64_t first = b == 0 ? 0 : find_start(lo, hi);
		s.begin_block(b);
		if (first == kNone) {
			/* No strict record starts in this block.  Either a record longer than the block runs through it
			 * (b-1 ended at or beyond hi: pass that end on), or b-1 ended INSIDE this block, i.e. a record begins
			 * here that does not parse as strict (wrapped, CRLF, no newline at the end of the file ...): the
			 * parallel phase stops at that record and the sequential reader takes over. */
			if (wait_start(b, kNone, &prev_end)) {
				if (prev_end < hi) fail(b, prev_end);
				else publish(b, prev_end);
			}
			return 0;
		}
		const char *const e = m_data + m_size, *const lim = m_data + hi;
		const char *p = m_data + first;
		while (p < lim) {
			const char *seq;
			uint64_t len;
			const char *r = strict_record(p, e, &seq, &len);
			if (!r) break;
			if (m_failBlock.load(std::memory_order_relaxed) < b) { s.discard(); return 0; }   /* never validated: nothing of b has left */
			if (!s.has_room(len)) {                             /* a submit is due: only validated content may leave */
				if (!started) {
					if (!wait_start(b, first, &prev_end)) { s.discard(); return 0; }
					started = true;
				}
				s.flush();
			}
			s.feed(seq, len);
			p = r;
			++n;
		}
		if (!started && !wait_start(b, first, &prev_end)) {
			s.discard();
			return 0;
		}
		s.flush();
		if (p < lim) fail(b, (uint64_t) (p - m_data));          /* non-strict record at p: sequential from here */
		else publish(b, (uint64_t) (p - m_data));
		return n;
	}

	const char *m_data = nullptr;
	uint64_t m_size = 0, m_block = 0;
	size_t m_nBlocks = 0;
	int m_fd = -1;
	std::atomic<size_t> m_next { 0 };
	std::mutex m_mu;
	std::condition_variable m_cv;
	std::vector<char> m_done;          /* block published its end */
	std::vector<uint64_t> m_end;
	static constexpr size_t kNoFail = ~(size_t) 0;
	std::atomic<size_t> m_failBlock { kNoFail };   /* lowest block at which the parallel phase stopped (written under m_mu) */
	uint64_t m_resume = 0;
};

} // namespace ntsm
#endif
