#include "early_ingest.hpp"

#include <sys/mman.h>
#include <sys/stat.h>

#include <chrono>
#include <cstdlib>
#include <cstring>

#include "gz_stream.hpp"
#include "pack2.hpp"
#include "parallel_fastq.hpp"
#include "parallel_gz_fastq.hpp"
#include "seq_reader.hpp"

namespace ntsm {

PackedChunk::~PackedChunk() { free(mem); }

static std::atomic<long> g_failAlloc { 0 };                  /* test hook: countdown to a failing chunk allocation */
void EarlyIngest::debug_fail_allocation(long nth) { g_failAlloc.store(nth); }

void PackedChunk::reserve(uint64_t positions)
{
	if (cap >= positions) return;
	free(mem);
	const size_t huge = 2u << 20;
	const size_t bytes = ((size_t) (positions / 4 + positions / 8) + huge - 1) & ~(huge - 1);
	mem = (g_failAlloc.load() > 0 && g_failAlloc.fetch_sub(1) == 1) ? nullptr : aligned_alloc(huge, bytes);
	if (!mem) { cap = 0; codes = valid = nullptr; return; }
	(void) madvise(mem, bytes, MADV_HUGEPAGE);
	codes = (uint8_t *) mem;
	valid = codes + positions / 4;
	cap = positions;
}

bool EarlyIngest::Sink::has_room(uint64_t len) const
{
	return !m_cur || pack2_extent(m_cur->pos, len) <= m_cur->cap;
}

void EarlyIngest::Sink::feed(const char *seq, uint64_t len)
{
	if (!m_cur) {
		m_cur = m_owner->blank(pack2_extent(0, len));
		if (!m_cur) return;                                    /* the run is being abandoned (by its owner, or after fail(): then failed() says so) */
	}
	m_cur->pos = pack2_append(m_cur->codes, m_cur->valid, m_cur->pos, seq, len);
	m_cur->n_bases += len;
	++m_cur->n_reads;
	++fed;
}

void EarlyIngest::Sink::flush()
{
	if (m_cur && m_cur->n_reads) m_owner->publish(std::move(m_cur));
}

void EarlyIngest::Sink::discard()
{
	if (m_cur) {
		fed -= m_cur->n_reads;
		m_cur->pos = m_cur->n_bases = 0;
		m_cur->n_reads = 0;
	}
}

std::unique_ptr<PackedChunk> EarlyIngest::blank(uint64_t min_positions)
{
	std::unique_ptr<PackedChunk> c;
	{
		std::unique_lock<std::mutex> lk(m_mu);
		m_cv.wait(lk, [&]() { return m_abandon || m_out < m_maxChunks; });
		if (m_abandon) return nullptr;
		++m_out;
		if (!m_free.empty()) { c = std::move(m_free.front()); m_free.pop_front(); }
	}
	const uint64_t want = std::max<uint64_t>(m_chunkPositions, (min_positions + 31) & ~31ull);
	if (!c) c.reset(new PackedChunk());
	c->reserve(want);
	if (!c->cap) {                                             /* out of memory: not an abandon -- reads would be lost silently */
		{
			std::lock_guard<std::mutex> lk(m_mu);
			--m_out;
		}
		fail("cannot allocate a " + std::to_string((want / 4 + want / 8) >> 20) + " MiB chunk for the early ingest of " + m_path + ": out of memory");
		return nullptr;
	}
	c->pos = c->n_bases = 0;
	c->n_reads = 0;
	return c;
}

void EarlyIngest::fail(const std::string &what)
{
	{
		std::lock_guard<std::mutex> lk(m_mu);
		if (!m_failed.load()) { m_error = what; m_failed.store(true); }
		m_abandon = true;                                      /* every parser stops at its next chunk */
	}
	m_cv.notify_all();
}

void EarlyIngest::publish(std::unique_ptr<PackedChunk> c)
{
	{
		std::lock_guard<std::mutex> lk(m_mu);
		m_ready.push_back(std::move(c));
	}
	m_cv.notify_all();
}

void EarlyIngest::recycle(std::unique_ptr<PackedChunk> c)
{
	{
		std::lock_guard<std::mutex> lk(m_mu);
		if (m_free.size() < 64) m_free.push_back(std::move(c));
		--m_out;
	}
	m_cv.notify_all();
}

bool EarlyIngest::next(std::unique_ptr<PackedChunk> *out)
{
	std::unique_lock<std::mutex> lk(m_mu);
	m_cv.wait(lk, [&]() { return m_done || !m_ready.empty(); });
	if (m_ready.empty()) return false;
	*out = std::move(m_ready.front());
	m_ready.pop_front();
	return true;
}

EarlyIngest::EarlyIngest(std::string path, unsigned n_parsers, unsigned n_decoders, uint64_t block_bytes, uint64_t gz_min_bytes, uint64_t chunk_positions, size_t max_chunks, int kinds)
	: m_path(std::move(path)), m_nParsers(n_parsers ? n_parsers : 1), m_nDecoders(n_decoders ? n_decoders : 1), m_blockBytes(block_bytes),
	  m_chunkPositions(chunk_positions & ~31ull), m_maxChunks(max_chunks < 2 * (size_t) (n_parsers ? n_parsers : 1) ? 2 * (size_t) (n_parsers ? n_parsers : 1) : max_chunks)
{
	struct stat st;
	if (stat(m_path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) return;
	m_plain.reset(new ParallelFastq());
	if (m_plain->open(m_path, m_blockBytes)) {
		if (!(kinds & 1)) { m_plain.reset(); return; }
		m_how = "plain FASTQ, block-parallel";
	} else {
		m_plain.reset();
		if (!(kinds & 2) || getenv("NTSM_ZLIB_ONLY") || (uint64_t) st.st_size < gz_min_bytes || !GzStream::is_gzip(m_path)) return;
		GzStream::set_decoder_threads(m_nDecoders);
		m_gz.reset(new GzStream());
		const bool ok = m_gz->open(m_path);
		GzStream::set_decoder_threads(1);
		if (!ok) { m_gz.reset(); return; }
		m_how = "gzip, decoder pool + piece-parallel";
	}
	m_taken = true;
	m_thread = std::thread([this]() { run(); });
}

std::unique_ptr<GzStream> EarlyIngest::release_stream()
{
	std::unique_lock<std::mutex> lk(m_mu);
	m_cv.wait(lk, [&]() { return m_done; });
	return std::move(m_rest);
}

EarlyIngest::~EarlyIngest()
{
	{
		std::lock_guard<std::mutex> lk(m_mu);
		m_abandon = true;
	}
	m_cv.notify_all();
	if (m_thread.joinable()) m_thread.join();
}

void EarlyIngest::run()
{
	const auto t0 = std::chrono::steady_clock::now();
	std::vector<Sink> sinks;
	sinks.reserve(m_nParsers);
	for (unsigned i = 0; i < m_nParsers; ++i) sinks.emplace_back(this);
	std::vector<Sink *> ptrs;
	for (auto &s : sinks) ptrs.push_back(&s);
	auto sequential = [&](SeqReader &rd) {                     /* what the parallel phase left: kseq-exact, into the first sink */
		Sink &s = sinks[0];
		for (int64_t l = rd.next(); l >= 0; l = rd.next()) {
			if (!s.has_room((uint64_t) l)) s.flush();
			s.feed(rd.seq_data(), (uint64_t) l);
		}
		s.flush();
	};
	if (m_plain) {
		const ParallelFastq::Result r = m_plain->run(ptrs);
		m_parallelRecords = r.records;
		if (!r.complete && !m_failed.load()) {
			SeqReader rd;
			if (rd.open(m_path, r.resume)) sequential(rd);
			else fail("cannot reopen " + m_path + " for the records the block-parallel phase left");
		}
		m_plain.reset();
	} else {
		ParallelGzFastq pg(m_gz.get(), &m_handOver);
		const ParallelGzFastq::Result r = pg.run(ptrs);
		m_parallelRecords = r.records;
		if (r.stopped) {
			m_rest = std::move(m_gz);                            /* the consumers take it from here (release_stream) */
		} else if (!r.complete && !m_failed.load()) {
			SeqReader rd;
			if (rd.open_stream(std::move(m_gz))) sequential(rd);
			else fail("cannot continue reading " + m_path + " after the piece-parallel phase");
		}
		m_gz.reset();
	}
	uint64_t n = 0;
	for (auto &s : sinks) { s.flush(); n += s.fed; }
	m_records = n;
	m_parseSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	{
		std::lock_guard<std::mutex> lk(m_mu);
		m_done = true;
	}
	m_cv.notify_all();
}

} // namespace ntsm
