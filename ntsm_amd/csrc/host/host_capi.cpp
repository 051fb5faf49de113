#include "../../../include/ntsm_host.h"

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>

#include <algorithm>
#include <zlib.h>

#include "gz_stream.hpp"
#include "pack2.hpp"
#include "early_ingest.hpp"
#include "host_shape.hpp"
#include "parallel_fastq.hpp"
#include "parallel_gz_fastq.hpp"
#include "report.hpp"
#include "seq_reader.hpp"
#include "site_set.hpp"

struct ntsm_sites {
	ntsm::SiteSet set;
};

static char *dup_string(const std::string &s, size_t *len)
{
	char *p = (char *) malloc(s.size() + 1);
	memcpy(p, s.data(), s.size());
	p[s.size()] = 0;
	if (len) *len = s.size();
	return p;
}

extern "C" {

int ntsm_sites_load(const char *path, unsigned k, int allow_dupes, ntsm_sites **out)
{
	if (!path || !out) return -1;
	ntsm_sites *s = new ntsm_sites();
	if (!s->set.load(path, k, allow_dupes != 0, std::cerr)) { delete s; return -1; }
	*out = s;
	return 0;
}
void ntsm_sites_free(ntsm_sites *s) { delete s; }
uint64_t ntsm_sites_n_keys(const ntsm_sites *s) { return s->set.keys.size(); }
const uint64_t *ntsm_sites_keys(const ntsm_sites *s) { return s->set.keys.data(); }
uint64_t ntsm_sites_n_sites(const ntsm_sites *s) { return s->set.ids.size(); }
uint64_t ntsm_sites_n_erased(const ntsm_sites *s) { return s->set.n_erased; }

uint64_t ntsm_host_max_hits(uint64_t n_distinct, double cov)
{
	if (cov == 0) return 0;
	const double x = ((double) n_distinct * cov) / 2;
	if (!(x == x) || x >= 18446744073709551616.0) return 0;
	if (x < 0) return UINT64_MAX;
	return (uint64_t) x;
}

int ntsm_host_flatten(const char *path, uint8_t **bases, uint64_t *n_bytes, uint64_t **read_end,
		uint64_t *n_reads, int *last_rc)
{
	ntsm::SeqReader rd;
	if (!rd.open(path)) return -1;
	std::vector<uint8_t> b;
	std::vector<uint64_t> e;
	int64_t l;
	while ((l = rd.next()) >= 0) {
		b.insert(b.end(), rd.seq_data(), rd.seq_data() + l);
		e.push_back(b.size());
		b.push_back('N');
	}
	if (last_rc) *last_rc = (int) l;
	*bases = (uint8_t *) malloc(b.size() + 16);
	memcpy(*bases, b.data(), b.size());
	*n_bytes = b.size();
	*read_end = (uint64_t *) malloc((e.size() + 1) * sizeof(uint64_t));
	memcpy(*read_end, e.data(), e.size() * sizeof(uint64_t));
	*n_reads = e.size();
	return 0;
}

void ntsm_host_free(void *p) { free(p); }

uint64_t ntsm_host_pack2_append(uint8_t *codes, uint8_t *valid, uint64_t pos, const uint8_t *seq, uint64_t len, int force_scalar)
{
	ntsm::pack2_force_impl(force_scalar);                     /* 0 = best available, 1 = portable, 2 = at most AVX2 */
	const uint64_t r = ntsm::pack2_append(codes, valid, pos, (const char *) seq, len);
	ntsm::pack2_force_impl(0);
	return r;
}
const char *ntsm_host_pack2_impl(void) { return ntsm::pack2_impl(); }

void ntsm_host_gunzip_parallel_chunk(uint64_t bytes) { ntsm::GzStream::set_parallel_chunk((size_t) bytes); }
void ntsm_host_gunzip_parallel_stats(uint64_t stats[2]) { ntsm::GzStream::last_parallel_stats(stats); }

int ntsm_host_gunzip(const char *path, int engine, unsigned chunk, uint8_t **out, uint64_t *len)
{
	std::vector<uint8_t> all, buf(chunk ? chunk : 1);
	int r;
	if (engine != 1) {
		ntsm::GzStream::set_decoder_threads(engine >= 2 ? (unsigned) engine : 1u);
		ntsm::GzStream gz;
		if (!gz.open(path)) return -2;
		while ((r = gz.read(buf.data(), (unsigned) buf.size())) > 0) all.insert(all.end(), buf.begin(), buf.begin() + r);
		ntsm::GzStream::set_decoder_threads(1);
	} else {
		gzFile f = gzopen(path, "r");
		if (!f) return -2;
		while ((r = gzread(f, buf.data(), (unsigned) buf.size())) > 0) all.insert(all.end(), buf.begin(), buf.begin() + r);
		gzclose(f);
	}
	*out = (uint8_t *) malloc(all.size() + 1);
	memcpy(*out, all.data(), all.size());
	*len = all.size();
	return r;
}

namespace {
/* Test sink of the block-parallel ingest: a small "staging" buffer whose flushes are kept as (block, order) chunks */
struct CollectSink {
	struct Chunk { size_t block; std::vector<uint8_t> bases; std::vector<uint64_t> lens; };
	uint64_t cap = 0;
	size_t cur = 0;
	std::vector<uint8_t> bases;
	std::vector<uint64_t> lens;
	std::vector<Chunk> out;
	bool has_room(uint64_t len) const { return lens.empty() || bases.size() + len + 1 <= cap; }
	void feed(const char *seq, uint64_t len) { bases.insert(bases.end(), seq, seq + len); bases.push_back('N'); lens.push_back(len); }
	void flush() { if (!lens.empty()) { out.push_back(Chunk { cur, std::move(bases), std::move(lens) }); bases.clear(); lens.clear(); } }
	void discard() { bases.clear(); lens.clear(); }
	void begin_block(size_t b) { cur = b; }
};
} // namespace

int ntsm_host_flatten_parallel(const char *path, unsigned n_threads, uint64_t block_bytes, uint8_t **bases,
		uint64_t *n_bytes, uint64_t **read_end, uint64_t *n_reads, uint64_t *n_blocks, uint64_t *n_parallel,
		uint64_t *resume)
{
	FILE *probe = fopen(path, "rb");
	if (!probe) return -1;
	fclose(probe);
	ntsm::ParallelFastq pf;
	if (!pf.open(path, block_bytes)) return 1;
	if (n_threads == 0) n_threads = 1;
	std::vector<CollectSink> sinks(n_threads);
	std::vector<CollectSink *> ptrs;
	for (auto &s : sinks) { s.cap = block_bytes / 5 + 64; ptrs.push_back(&s); }     /* several flushes per block */
	const ntsm::ParallelFastq::Result r = pf.run(ptrs);
	std::vector<const CollectSink::Chunk *> order;
	for (auto &s : sinks) {
		if (!s.lens.empty()) return -2;                                             /* staged but never committed nor dropped */
		for (auto &c : s.out) order.push_back(&c);
	}
	/* chunks of one block come from one thread in feed order; stable sort by block restores file order */
	std::stable_sort(order.begin(), order.end(), [](const CollectSink::Chunk *a, const CollectSink::Chunk *b) { return a->block < b->block; });
	std::vector<uint8_t> b;
	std::vector<uint64_t> e;
	for (const CollectSink::Chunk *c : order) {
		uint64_t off = 0;
		for (uint64_t len : c->lens) {
			b.insert(b.end(), c->bases.begin() + (long) off, c->bases.begin() + (long) (off + len + 1));
			e.push_back(b.size() - 1);
			off += len + 1;
		}
	}
	if (n_parallel) *n_parallel = e.size();
	if (resume) *resume = r.resume;
	if (e.size() != r.records) return -2;
	if (!r.complete) {                                                              /* sequential reader takes over */
		ntsm::SeqReader rd;
		if (!rd.open(path, r.resume)) return -1;
		for (int64_t l = rd.next(); l >= 0; l = rd.next()) {
			b.insert(b.end(), rd.seq_data(), rd.seq_data() + l);
			e.push_back(b.size());
			b.push_back('N');
		}
	}
	*bases = (uint8_t *) malloc(b.size() + 16);
	memcpy(*bases, b.data(), b.size());
	*n_bytes = b.size();
	*read_end = (uint64_t *) malloc((e.size() + 1) * sizeof(uint64_t));
	memcpy(*read_end, e.data(), e.size() * sizeof(uint64_t));
	*n_reads = e.size();
	if (n_blocks) *n_blocks = pf.n_blocks();
	return 0;
}

int ntsm_host_flatten_parallel_gz(const char *path, unsigned n_decoders, unsigned n_parsers, uint64_t sink_bytes, uint8_t **bases,
		uint64_t *n_bytes, uint64_t **read_end, uint64_t *n_reads, uint64_t *n_pieces, uint64_t *n_parallel, int *final_status)
{
	if (!ntsm::GzStream::is_gzip(path)) return 1;
	ntsm::GzStream::set_decoder_threads(n_decoders ? n_decoders : 1);
	std::unique_ptr<ntsm::GzStream> gz(new ntsm::GzStream());
	const bool opened = gz->open(path);
	ntsm::GzStream::set_decoder_threads(1);
	if (!opened) return -1;
	if (n_parsers == 0) n_parsers = 1;
	std::vector<CollectSink> sinks(n_parsers);
	std::vector<CollectSink *> ptrs;
	for (auto &s : sinks) { s.cap = sink_bytes ? sink_bytes : (1u << 20); ptrs.push_back(&s); }
	ntsm::ParallelGzFastq pg(gz.get());
	const ntsm::ParallelGzFastq::Result r = pg.run(ptrs);
	std::vector<const CollectSink::Chunk *> order;
	for (auto &s : sinks) {
		if (!s.lens.empty()) return -2;                                             /* staged but never committed nor dropped */
		for (auto &c : s.out) order.push_back(&c);
	}
	/* the chunks of one piece come from one thread; the bridge records of a piece are fed before or between its own records,
	 * so inside a piece the order is not the file's -- callers compare as multisets per file or sort (counts do not depend on
	 * the order of the reads) */
	std::stable_sort(order.begin(), order.end(), [](const CollectSink::Chunk *a, const CollectSink::Chunk *b) { return a->block < b->block; });
	std::vector<uint8_t> b;
	std::vector<uint64_t> e;
	for (const CollectSink::Chunk *c : order) {
		uint64_t off = 0;
		for (uint64_t len : c->lens) {
			b.insert(b.end(), c->bases.begin() + (long) off, c->bases.begin() + (long) (off + len + 1));
			e.push_back(b.size() - 1);
			off += len + 1;
		}
	}
	if (n_parallel) *n_parallel = e.size();
	if (e.size() != r.records) return -2;
	if (!r.complete) {                                                              /* the sequential reader takes over on the same stream */
		ntsm::SeqReader rd;
		if (!rd.open_stream(std::move(gz))) return -1;
		for (int64_t l = rd.next(); l >= 0; l = rd.next()) {
			b.insert(b.end(), rd.seq_data(), rd.seq_data() + l);
			e.push_back(b.size());
			b.push_back('N');
		}
	}
	*bases = (uint8_t *) malloc(b.size() + 16);
	memcpy(*bases, b.data(), b.size());
	*n_bytes = b.size();
	*read_end = (uint64_t *) malloc((e.size() + 1) * sizeof(uint64_t));
	memcpy(*read_end, e.data(), e.size() * sizeof(uint64_t));
	*n_reads = e.size();
	if (n_pieces) *n_pieces = r.pieces;
	if (final_status) *final_status = r.status;
	return 0;
}

int ntsm_host_early_ingest_hand_over(const char *path, unsigned n_parsers, unsigned n_decoders, uint64_t block_bytes, uint64_t chunk_positions,
		uint64_t max_chunks, unsigned n_consumers, uint64_t hand_over_after, uint8_t **text, uint64_t *n_text, uint64_t *n_reads, uint64_t *n_bases,
		uint64_t *n_parallel, uint64_t *n_rest)
{
	ntsm::EarlyIngest ei(path, n_parsers, n_decoders, block_bytes, 1000, chunk_positions, (size_t) max_chunks);
	if (!ei.taken()) return 1;
	std::mutex mu;
	std::vector<uint8_t> all;
	uint64_t reads = 0, bases = 0;
	std::atomic<uint64_t> drained { 0 };
	if (hand_over_after == 0) ei.hand_over();
	std::vector<std::thread> pool;
	for (unsigned t = 0; t < (n_consumers ? n_consumers : 1); ++t)
		pool.emplace_back([&]() {
			std::unique_ptr<ntsm::PackedChunk> c;
			std::vector<uint8_t> mine;
			uint64_t r = 0, b = 0;
			while (ei.next(&c)) {
				for (uint64_t p = 0; p < c->pos; ++p) {
					const bool v = (c->valid[p >> 3] >> (p & 7)) & 1;
					mine.push_back(v ? (uint8_t) "ACGT"[(c->codes[p >> 2] >> (2 * (p & 3))) & 3] : (uint8_t) 'N');
				}
				mine.push_back('N');
				r += c->n_reads;
				b += c->n_bases;
				ei.recycle(std::move(c));
				if (++drained == hand_over_after) ei.hand_over();
			}
			std::lock_guard<std::mutex> lk(mu);
			all.insert(all.end(), mine.begin(), mine.end());
			reads += r;
			bases += b;
		});
	for (auto &th : pool) th.join();
	if (ei.failed()) return -3;                                 /* reads were lost: what FingerPrint::drainEarly turns into exit(1) */
	const uint64_t through_chunks = reads;
	uint64_t rest_reads = 0;
	if (std::unique_ptr<ntsm::GzStream> rest = ei.release_stream()) {
		/* what the stream still holds, taken the way FingerPrint::countGzStream takes it: a second piece-parallel phase on the
		 * same stream (it starts with what the first one handed back), then the sequential reader for what that leaves */
		std::vector<uint8_t> codes, valid;
		auto add_read = [&](const char *seq, uint64_t l) {
			const uint64_t ext = ntsm::pack2_extent(0, l);
			codes.assign(ext / 4 + 8, 0);
			valid.assign(ext / 8 + 8, 0);
			const uint64_t end = ntsm::pack2_append(codes.data(), valid.data(), 0, seq, l);
			for (uint64_t p = 0; p < end; ++p) {
				const bool v = (valid[p >> 3] >> (p & 7)) & 1;
				all.push_back(v ? (uint8_t) "ACGT"[(codes[p >> 2] >> (2 * (p & 3))) & 3] : (uint8_t) 'N');
			}
			all.push_back('N');
			++reads;
			bases += l;
			++rest_reads;
		};
		std::vector<CollectSink> sinks(n_parsers ? n_parsers : 1);
		std::vector<CollectSink *> ptrs;
		for (auto &s : sinks) { s.cap = 1u << 16; ptrs.push_back(&s); }
		ntsm::ParallelGzFastq pg(rest.get());
		const ntsm::ParallelGzFastq::Result r = pg.run(ptrs);
		uint64_t got = 0;
		for (auto &s : sinks) {
			if (!s.lens.empty()) return -2;
			for (auto &c : s.out) {
				uint64_t off = 0;
				for (uint64_t len : c.lens) { add_read((const char *) c.bases.data() + off, len); off += len + 1; ++got; }
			}
		}
		if (got != r.records) return -2;
		if (!r.complete) {
			ntsm::SeqReader rd;
			if (rd.open_stream(std::move(rest)))
				for (int64_t l = rd.next(); l >= 0; l = rd.next()) add_read(rd.seq_data(), (uint64_t) l);
		}
	}
	*text = (uint8_t *) malloc(all.size() + 1);
	memcpy(*text, all.data(), all.size());
	*n_text = all.size();
	*n_reads = reads;
	*n_bases = bases;
	if (n_parallel) *n_parallel = ei.parallel_records();
	if (n_rest) *n_rest = rest_reads;
	return through_chunks == ei.records() ? 0 : -2;
}

unsigned ntsm_host_granted_cpus(void) { return ntsm::granted_cpus(); }

void ntsm_host_ingest_plan(unsigned threads_asked, unsigned cpus, unsigned out[4])
{
	const ntsm::IngestPlan p = ntsm::ingest_plan(threads_asked, cpus ? cpus : ntsm::granted_cpus());
	out[0] = p.cpus; out[1] = p.feeders; out[2] = p.decoders; out[3] = p.early_decoders;
}

void ntsm_host_debug_gz_max_tail(uint64_t bytes) { ntsm::ParallelGzFastq::set_max_tail((size_t) bytes); }

void ntsm_host_debug_early_alloc_fail(long nth) { ntsm::EarlyIngest::debug_fail_allocation(nth); }

int ntsm_host_early_ingest(const char *path, unsigned n_parsers, unsigned n_decoders, uint64_t block_bytes, uint64_t chunk_positions, uint64_t max_chunks,
		unsigned n_consumers, uint8_t **text, uint64_t *n_text, uint64_t *n_reads, uint64_t *n_bases, uint64_t *n_parallel)
{
	return ntsm_host_early_ingest_hand_over(path, n_parsers, n_decoders, block_bytes, chunk_positions, max_chunks, n_consumers, ~0ull, text, n_text, n_reads,
			n_bases, n_parallel, nullptr);
}

int ntsm_host_format_counts(const ntsm_sites *s, const uint64_t *counts, uint64_t total_kmers, char **out, size_t *len)
{
	std::ostringstream os;
	std::vector<uint64_t> c(counts, counts + s->set.keys.size());
	ntsm::print_optional_header(os, total_kmers, s->set.k);
	const bool ok = ntsm::print_counts_max(os, s->set, c);
	*out = dup_string(os.str(), len);
	return ok ? 0 : 1;
}

int ntsm_host_format_summary(const ntsm_sites *s, const uint64_t *counts, uint64_t total_bases, uint64_t total_kmers,
		uint64_t total_hits, char **out, size_t *len, uint64_t *covered)
{
	std::vector<uint64_t> c(counts, counts + s->set.keys.size());
	*out = dup_string(ntsm::info_summary(s->set, c, total_bases, total_kmers, total_hits), len);
	if (covered) *covered = ntsm::sites_covered(s->set, c);
	return 0;
}

} // extern "C"
