/*
 * parallel_gz_fastq.hpp -- block-parallel parsing of the TEXT a gzip input inflates to (the
 * sequential counterpart is kseq over gzread, vendor/kseq.h:177-219 / :229).
 *
 * gz_stream.hpp delivers the inflated text as an ordered sequence of pieces (a spliced chunk of the parallel plain-gzip
 * decoder, a group of BGZF members, or what the in-order decoder produced); with N decoder threads the pieces arrive faster
 * than one thread can parse them.  Here N parsing threads take the pieces in order, one each, and apply the scheme of
 * parallel_fastq.hpp to them -- strict 4-line records only, ordered commit, no rollback:
 *
 *   - the thread that took piece b looks for the first record start inside it (a line starting with '@' that parses as a
 *     strict record followed by another '@'), parses strict records from there while they lie entirely inside the piece, and
 *     keeps the unparsed rest as the piece's TAIL;
 *   - LINK b-1 -> b: it then waits for the thread of piece b-1 to publish its tail; tail(b-1) + the bytes of piece b in front
 *     of its first record start must parse as a whole number of strict records -- that proves the guessed start.  A piece
 *     without any record start (smaller than a record, or inside a very long read) passes `tail(b-1) + all its bytes` on as
 *     its own tail;
 *   - nothing of piece b is submitted before link b-1 -> b holds and every earlier link held (a sink that runs full earlier
 *     waits for that before it flushes).  The first link that fails -- or a tail that is still there when the data ends --
 *     stops the parallel phase at a record boundary: everything before it has been committed, nothing after it has, and the
 *     bytes from there on (the failed tail + the pieces taken since) go back to the GzStream (unread) for the sequential
 *     reader, which reproduces kseq on them byte for byte (wrapped lines, CR, FASTA, a last record without newline, ...).
 *
 * Sink concept: the same as parallel_fastq.hpp (has_room / feed / flush / discard / begin_block).
 */
#ifndef NTSM_PARALLEL_GZ_FASTQ_HPP
#define NTSM_PARALLEL_GZ_FASTQ_HPP
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gz_stream.hpp"
#include "parallel_fastq.hpp"

namespace ntsm {

class ParallelGzFastq {
public:
	struct Result {
		bool complete = false;      /* every byte of the stream was consumed by the parallel phase (and the stream ended cleanly) */
		bool stopped = false;       /* the caller asked for the phase to end (stop_flag): the stream is positioned at a record boundary */
		uint64_t records = 0;       /* records committed by the parallel phase */
		uint64_t pieces = 0;
		int status = 0;             /* complete: the stream's final status (1 clean, -1 error after the last piece) */
	};
	/* gz: an open stream nobody has read from yet.  After run(), !complete: gz is positioned (unread) at the record boundary
	 * where the parallel phase stopped; read() continues from there. */
	/* stop_flag (optional): once it reads true no further piece is taken; the pieces in hand are finished and committed, the
	 * unparsed end of the last one goes back to the stream like after a failed link (early_ingest.cpp hands the stream over
	 * to the feeders that way) */
	explicit ParallelGzFastq(GzStream *gz, const std::atomic<bool> *stop_flag = nullptr) : m_gz(gz), m_stop(stop_flag) {}

	template <class Sink> Result run(const std::vector<Sink *> &sinks)
	{
		std::vector<std::thread> pool;
		for (Sink *s : sinks) pool.emplace_back([this, s]() { worker<Sink>(*s); });
		for (auto &t : pool) t.join();
		Result r;
		r.records = m_records;
		r.pieces = m_nextSeq;
		/* what is left for the sequential reader: the tail in front of the first piece that did not commit, then those pieces */
		std::deque<std::unique_ptr<GzStream::Piece>> rest;
		const uint64_t fail = m_failSeq.load();
		const uint64_t stop = fail < m_nextSeq ? fail : m_nextSeq;                /* pieces [0, stop) committed */
		const std::string &tail = stop == 0 ? m_empty : m_link[stop - 1].tail;
		if (!tail.empty()) {
			std::unique_ptr<GzStream::Piece> t(new GzStream::Piece());
			t->data.assign(tail.begin(), tail.end());
			t->len = tail.size();
			rest.push_back(std::move(t));
		}
		for (uint64_t b = stop; b < m_nextSeq; ++b)
			if (m_kept.count(b)) rest.push_back(std::move(m_kept[b]));
		r.stopped = m_stopped;
		r.complete = rest.empty() && !m_stopped;
		r.status = m_gz->final_status();
		if (!rest.empty()) m_gz->unread(std::move(rest), 0);
		return r;
	}

private:
	/* end: the piece committed what it parsed but the parallel phase ends behind it (its tail is too long to carry on) */
	struct Link { bool done = false, ok = false, end = false; std::string tail; };
	static constexpr uint64_t kNone = ~0ull;
	/* a bridge longer than this (one read of > 128 Mb?) goes to the sequential reader; tests shrink it (set_max_tail) */
	static std::atomic<size_t> &max_tail() { static std::atomic<size_t> v { 256u << 20 }; return v; }
public:
	static void set_max_tail(size_t bytes) { max_tail().store(bytes ? bytes : (size_t) 256u << 20); }
private:

	/* first record start in [p, e): a line start where a strict record parses and is followed by '@' (or ends the piece exactly) */
	static uint64_t find_start(const char *p, const char *e, bool at_line_start)
	{
		const char *q = p;
		if (!at_line_start) {
			const char *nl = (const char *) memchr(q, '\n', (size_t) (e - q));
			q = nl ? nl + 1 : e;
		}
		while (q < e) {
			if (*q == '@') {
				const char *seq;
				uint64_t len;
				const char *r = ParallelFastq::strict_record(q, e, &seq, &len);
				if (r && r < e && *r == '@') return (uint64_t) (q - p);
			}
			const char *nl = (const char *) memchr(q, '\n', (size_t) (e - q));
			q = nl ? nl + 1 : e;
		}
		return kNone;
	}

	/* wait for link b-1 -> b; true iff the parallel phase is still alive up to piece b-1 */
	bool wait_prev(uint64_t b, std::string *tail)
	{
		if (b == 0) { tail->clear(); return true; }
		std::unique_lock<std::mutex> lk(m_mu);
		m_cv.wait(lk, [&]() { return m_link[b - 1].done; });
		if (!m_link[b - 1].ok || m_link[b - 1].end) return false;
		*tail = m_link[b - 1].tail;
		return true;
	}
	/* end_here (with ok): piece b is committed up to its tail, and the phase ends there -- run() hands the tail and the later
	 * pieces back to the stream exactly as after a failed link b -> b+1 */
	void publish(uint64_t b, bool ok, std::string tail, std::unique_ptr<GzStream::Piece> piece, bool end_here = false)
	{
		{
			std::lock_guard<std::mutex> lk(m_mu);
			Link &l = m_link[b];
			l.ok = ok;
			l.end = end_here;
			l.tail = std::move(tail);
			l.done = true;
			if (!ok) {
				if (b < m_failSeq) m_failSeq = b;
				m_kept[b] = std::move(piece);                      /* goes back to the stream */
			} else if (end_here && b + 1 < m_failSeq) {
				m_failSeq = b + 1;
			}
		}
		m_cv.notify_all();
		if (piece) m_gz->give_back(std::move(piece));
	}

	template <class Sink> void worker(Sink &s)
	{
		uint64_t n_records = 0;
		for (;;) {
			std::unique_ptr<GzStream::Piece> pc;
			uint64_t b;
			{
				std::lock_guard<std::mutex> lk(m_takeMu);
				if (m_failSeq.load() != kNone || m_ended) break;     /* the parallel phase is over: the rest stays in the stream */
				if (m_stop && m_stop->load(std::memory_order_relaxed)) { m_stopped = true; break; }
				pc = m_gz->take();
				if (!pc) { m_ended = true; break; }
				b = m_nextSeq++;
				std::lock_guard<std::mutex> lk2(m_mu);
				m_link.emplace_back();
			}
			s.begin_block((size_t) b);
			const char *const p0 = (const char *) pc->data.data(), *const e = p0 + pc->len;
			/* piece 0 starts the file: its first byte must start a record (kseq would skip junk in front of the first header:
			 * that case parses nothing here and fails the first link); later pieces: first proven-looking line start */
			const uint64_t first = b == 0 ? 0 : find_start(p0, e, false);
			bool linked = false, alive = true;
			std::string prev_tail;
			uint64_t n_here = 0;
			/* Link b-1 -> b: tail(b-1) + this piece's bytes in front of `first` must be a whole number of strict records.  They
			 * are validated before any of them is fed, so that a flush in between never submits unproven content. */
			auto link_now = [&]() -> bool {
				if (!wait_prev(b, &prev_tail)) return false;
				if (first == kNone) return true;                     /* no start in this piece: everything joins the tail (below) */
				if (prev_tail.empty() && first == 0) return true;
				std::string bridge = std::move(prev_tail);
				prev_tail.clear();
				bridge.append(p0, (size_t) first);
				const char *const qe = bridge.data() + bridge.size();
				const char *seq;
				uint64_t len;
				for (const char *q = bridge.data(); q < qe;) {
					q = ParallelFastq::strict_record(q, qe, &seq, &len);
					if (!q) return false;                            /* the guess (or the input) is off */
				}
				for (const char *q = bridge.data(); q < qe;) {
					q = ParallelFastq::strict_record(q, qe, &seq, &len);
					if (!s.has_room(len)) s.flush();                 /* proven: the link holds, what is staged lies behind it */
					s.feed(seq, len);
					++n_here;
				}
				return true;
			};
			const char *p = first == kNone ? e : p0 + first;
			while (p < e) {
				const char *seq;
				uint64_t len;
				const char *r = ParallelFastq::strict_record(p, e, &seq, &len);
				if (!r) break;                                       /* incomplete (this piece's tail) or not strict: the next link decides */
				if (!s.has_room(len)) {                              /* a submit is due: only proven content may leave */
					if (!linked) {
						alive = link_now();
						linked = true;
						if (!alive) break;
					}
					s.flush();
				}
				s.feed(seq, len);
				++n_here;
				p = r;
			}
			if (alive && !linked) { alive = link_now(); linked = true; }
			std::string tail;
			bool end_here = false;
			if (alive) {
				if (first == kNone) {                               /* nothing parsed here: the previous tail + the whole piece */
					tail = std::move(prev_tail);
					tail.append(p0, pc->len);
				} else {
					tail.assign(p, (size_t) (e - p));
				}
				/* Too long to carry on.  The link INTO this piece held and records of it may have left with a flush already, so
				 * the piece cannot go back whole (the sequential reader would count them a second time): it commits what it
				 * parsed, and the phase ends at [p, e) like after a failed next link. */
				if (tail.size() > max_tail().load(std::memory_order_relaxed)) end_here = true;
			}
			if (!alive) {
				s.discard();
				publish(b, false, std::string(), std::move(pc));
				continue;                                           /* the take lock sees m_failSeq: every worker ends */
			}
			s.flush();
			n_records += n_here;
			publish(b, true, std::move(tail), std::move(pc), end_here);
			if (end_here) break;
		}
		m_records += n_records;
	}

	GzStream *m_gz;
	std::mutex m_takeMu, m_mu;
	std::condition_variable m_cv;
	std::deque<Link> m_link;                            /* by piece number (deque: references stay valid while it grows) */
	std::map<uint64_t, std::unique_ptr<GzStream::Piece>> m_kept;
	std::atomic<uint64_t> m_failSeq { kNone };
	uint64_t m_nextSeq = 0;
	bool m_ended = false, m_stopped = false;              /* (under m_takeMu) */
	const std::atomic<bool> *m_stop = nullptr;
	std::atomic<uint64_t> m_records { 0 };
	const std::string m_empty;
};

} // namespace ntsm
#endif
