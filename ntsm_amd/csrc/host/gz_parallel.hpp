/*
 * gz_parallel.hpp -- chunk workers of the parallel decoder for ONE plain gzip stream (gz_stream.hpp has the overview;
 * replaces the single gzread stream of vendor/kseq.h:229 for `reads.fq.gz`).  Internal to gz_stream.cpp / gz_parallel.cpp.
 */
#ifndef NTSM_GZ_PARALLEL_HPP
#define NTSM_GZ_PARALLEL_HPP
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "gz_stream.hpp"

namespace ntsm {

struct GzStream::Parallel {
	struct Chunk {
		size_t index = 0;
		bool issued = false, done = false, found = false, hit_final = false;
		uint64_t b_bit = 0, e_bit = 0;                 /* first block of the chunk, end of its last block (bit offsets in the file) */
		std::vector<uint16_t> sym;                     /* kWindow markers, then n_sym symbols */
		size_t n_sym = 0;
	};
	struct Resolve { std::unique_ptr<Chunk> chunk; std::vector<uint8_t> window; Piece *piece; };

	Parallel(GzStream *gz, unsigned n_threads, size_t chunk_bytes);
	~Parallel();
	/* Called by the in-order decoder at a block boundary (or at the start of a member) at bit `pos`: where it has to stop
	 * next.  == pos: the chunk returned by take() starts exactly here; > pos: decode in order up to the first boundary at or
	 * beyond that bit; ~0ull: no chunk left. */
	uint64_t target(uint64_t pos);
	std::unique_ptr<Chunk> take();                     /* the chunk target() just matched */
	/* a taken chunk that is not spliced in after all (the member has no full window yet): counted as dropped, not as
	 * spliced, and its symbol buffer goes back to the pool */
	void drop(std::unique_ptr<Chunk> c);
	/* hand a spliced chunk to the workers: piece (already queued in order, ready = false) gets its bytes and CRC */
	void resolve_async(std::unique_ptr<Chunk> c, const uint8_t *window, Piece *piece);
	size_t spliced = 0, dropped = 0;                   /* statistics */

private:
	void worker();
	void decode_chunk(Chunk &c, class SpecInflate &sp);
	void issue_locked();
	GzStream *m_gz;
	const uint8_t *m_base, *m_end;
	size_t m_chunkBytes, m_nChunks, m_next = 1, m_issue = 1, m_depth;
	std::vector<std::unique_ptr<Chunk>> m_chunks;      /* by index; null once taken or dropped */
	std::deque<Chunk *> m_todo;
	std::deque<Resolve> m_resolve;
	std::vector<std::vector<uint16_t>> m_bufPool;
	std::mutex m_mu;
	std::condition_variable m_cv;
	bool m_quit = false;
	std::vector<std::thread> m_pool;
};

} // namespace ntsm
#endif
