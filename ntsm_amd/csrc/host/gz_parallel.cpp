#include "gz_parallel.hpp"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "crc32_fast.hpp"
#include "inflate_spec.hpp"

namespace ntsm {

namespace {
constexpr size_t kWin = SpecInflate::kWindow;
constexpr size_t kSymSlack = 512;
/* NTSM_PGZ_PROF=1: where the workers' time goes (find / decode / resolve + crc / waiting), printed when a stream closes */
std::atomic<uint64_t> g_ns_find { 0 }, g_ns_decode { 0 }, g_ns_resolve { 0 }, g_ns_wait { 0 }, g_sym { 0 };
inline uint64_t now_ns() { return (uint64_t) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}

GzStream::Parallel::Parallel(GzStream *gz, unsigned n_threads, size_t chunk_bytes)
	: m_gz(gz), m_base(gz->m_map), m_end(gz->m_map + gz->m_size), m_chunkBytes(chunk_bytes)
{
	m_nChunks = (gz->m_size + chunk_bytes - 1) / chunk_bytes;
	m_chunks.resize(m_nChunks);
	m_depth = 2u * n_threads + 2u;
	{
		std::lock_guard<std::mutex> lk(m_mu);
		issue_locked();
	}
	for (unsigned i = 0; i < n_threads; ++i) m_pool.emplace_back([this]() { worker(); });
}

GzStream::Parallel::~Parallel()
{
	{
		std::lock_guard<std::mutex> lk(m_mu);
		m_quit = true;
	}
	m_cv.notify_all();
	for (auto &t : m_pool) t.join();
	if (getenv("NTSM_PGZ_PROF"))
		fprintf(stderr, "[pgz] workers: find %.3f s, decode %.3f s (%.1f M symbols), resolve+crc %.3f s, idle %.3f s; spliced %zu dropped %zu\n",
		        g_ns_find / 1e9, g_ns_decode / 1e9, g_sym / 1e6, g_ns_resolve / 1e9, g_ns_wait / 1e9, spliced, dropped);
}

/* keep chunks [m_next, m_next + m_depth) with the workers (m_mu held) */
void GzStream::Parallel::issue_locked()
{
	if (m_issue < m_next) m_issue = m_next;
	bool any = false;
	while (m_issue < m_nChunks && m_issue < m_next + m_depth) {
		std::unique_ptr<Chunk> c(new Chunk());
		c->index = m_issue;
		c->issued = true;
		if (!m_bufPool.empty()) { c->sym = std::move(m_bufPool.back()); m_bufPool.pop_back(); }
		m_todo.push_back(c.get());
		m_chunks[m_issue] = std::move(c);
		++m_issue;
		any = true;
	}
	if (any) m_cv.notify_all();
}

uint64_t GzStream::Parallel::target(uint64_t pos)
{
	std::unique_lock<std::mutex> lk(m_mu);
	for (;;) {
		if (m_next >= m_nChunks) return ~0ull;
		const uint64_t s_bit = (uint64_t) m_next * m_chunkBytes * 8u;
		if (s_bit > pos) { issue_locked(); return s_bit; }
		Chunk *c = m_chunks[m_next].get();
		if (!c) { ++m_next; continue; }
		m_cv.wait(lk, [&]() { return c->done; });
		if (!c->found || c->b_bit < pos) {                  /* no dynamic block seen, decoding failed, or a false start */
			if (c->sym.capacity()) m_bufPool.push_back(std::move(c->sym));
			m_chunks[m_next].reset();
			++m_next;
			++dropped;
			continue;
		}
		issue_locked();
		return c->b_bit;
	}
}

std::unique_ptr<GzStream::Parallel::Chunk> GzStream::Parallel::take()
{
	std::lock_guard<std::mutex> lk(m_mu);
	std::unique_ptr<Chunk> c = std::move(m_chunks[m_next]);
	++m_next;
	++spliced;
	issue_locked();
	return c;
}

void GzStream::Parallel::drop(std::unique_ptr<Chunk> c)
{
	if (!c) return;
	std::lock_guard<std::mutex> lk(m_mu);
	if (c->sym.capacity()) m_bufPool.push_back(std::move(c->sym));
	--spliced;
	++dropped;
}

void GzStream::Parallel::resolve_async(std::unique_ptr<Chunk> c, const uint8_t *window, Piece *piece)
{
	Resolve r;
	r.chunk = std::move(c);
	r.window.assign(window, window + kWin);
	r.piece = piece;
	{
		std::lock_guard<std::mutex> lk(m_mu);
		m_resolve.push_back(std::move(r));
	}
	m_cv.notify_all();
}

/* One chunk: first dynamic block that starts in [S, S + chunk) and everything after it up to the first block boundary at or
 * beyond the end of the range (or the end of the member).  A candidate whose decoding fails was not a block start (or the
 * data is corrupt -- the in-order decoder will find out): the search goes on behind it. */
void GzStream::Parallel::decode_chunk(Chunk &c, SpecInflate &sp)
{
	const uint64_t s_bit = (uint64_t) c.index * m_chunkBytes * 8u;
	const uint64_t limit = std::min<uint64_t>((uint64_t) (c.index + 1) * m_chunkBytes, (uint64_t) (m_end - m_base)) * 8u;
	if (c.sym.size() < kWin + (m_chunkBytes * 4u) + kSymSlack) {
		c.sym.resize(kWin + m_chunkBytes * 4u + kSymSlack);
		SpecInflate::fill_markers(c.sym.data());
	}
	uint64_t from = s_bit;
	for (int attempt = 0; attempt < 8; ++attempt) {
		const uint64_t t0 = now_ns();
		const uint64_t b = sp.find(m_base, m_end, from, limit);
		g_ns_find += now_ns() - t0;
		if (b == ~0ull) return;
		const uint64_t t1 = now_ns();
		sp.set_stop(m_base, limit);
		size_t out = kWin;
		Inflate::Status st;
		for (;;) {
			st = sp.run16(c.sym.data(), &out, c.sym.size() - kSymSlack);
			if (st != Inflate::MORE) break;
			/* a chunk that inflates to more than 64 times its size (text: 4-8 times) is left to the in-order decoder, which
			 * streams in 1 MiB pieces: 16 workers must not each hold gigabytes of symbols of a pathological member */
			if (c.sym.size() >= kWin + 64 * m_chunkBytes) { c.sym.resize(kWin + 4 * m_chunkBytes + kSymSlack); c.sym.shrink_to_fit(); SpecInflate::fill_markers(c.sym.data()); return; }
			c.sym.resize(c.sym.size() * 2);
		}
		g_ns_decode += now_ns() - t1;
		g_sym += out - kWin;
		if (st == Inflate::BLOCK_STOP || st == Inflate::STREAM_END) {
			c.found = true;
			c.hit_final = st == Inflate::STREAM_END;
			c.b_bit = b;
			c.e_bit = sp.bit_pos(m_base);
			c.n_sym = out - kWin;
			return;
		}
		from = b + 1;
	}
}

void GzStream::Parallel::worker()
{
	SpecInflate sp;
	for (;;) {
		Resolve r;
		r.piece = nullptr;
		Chunk *c = nullptr;
		{
			const uint64_t tw = now_ns();
			std::unique_lock<std::mutex> lk(m_mu);
			m_cv.wait(lk, [&]() { return m_quit || !m_resolve.empty() || !m_todo.empty(); });
			g_ns_wait += now_ns() - tw;
			/* pieces that are already queued for the reader are always filled, also on the way out */
			if (!m_resolve.empty()) { r = std::move(m_resolve.front()); m_resolve.pop_front(); }
			else if (m_quit) return;
			else { c = m_todo.front(); m_todo.pop_front(); }
		}
		if (c) {
			decode_chunk(*c, sp);
			{
				std::lock_guard<std::mutex> lk(m_mu);
				c->done = true;
			}
			m_cv.notify_all();
			continue;
		}
		/* bytes of a spliced chunk: markers -> window bytes, CRC-32 for the member check */
		Piece *pc = r.piece;
		const size_t n = r.chunk->n_sym;
		const uint64_t tr = now_ns();
		uint32_t crc = 0;
		const bool ok = SpecInflate::resolve(r.chunk->sym.data() + kWin, n, r.window.data(), kWin, pc->data.data(), &crc);
		pc->crc = crc;
		if (!ok) pc->status = -1;                             /* cannot happen: the splice requires a full window */
		g_ns_resolve += now_ns() - tr;
		GzStream::release_input(m_base + r.chunk->index * m_chunkBytes, m_base + std::min<size_t>((r.chunk->index + 1) * m_chunkBytes, (size_t) (m_end - m_base)));
		{
			std::lock_guard<std::mutex> lk(m_mu);
			m_bufPool.push_back(std::move(r.chunk->sym));
		}
		{
			std::lock_guard<std::mutex> lk(m_gz->m_mu);
			pc->ready = true;
		}
		m_gz->m_cv.notify_all();
	}
}

} // namespace ntsm
