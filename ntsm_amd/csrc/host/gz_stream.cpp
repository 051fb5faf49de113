#include "gz_stream.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>

#include "crc32_fast.hpp"
#include "gz_parallel.hpp"
#include "inflate.hpp"
#include "inflate_spec.hpp"

namespace ntsm {

namespace {
constexpr size_t kPiece = 1u << 20, kWindow = 32768, kSlack = 512, kQueue = 32;
inline uint32_t le32(const uint8_t *p) { return (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16) | ((uint32_t) p[3] << 24); }

/* gzip member header (RFC 1952 2.3): returns the offset of the deflate data, 0 on invalid data, SIZE_MAX when the
 * input ends inside the header */
size_t skip_header(const uint8_t *p, size_t n)
{
	if (n < 10) return SIZE_MAX;
	if (p[0] != 0x1f || p[1] != 0x8b) return 0;
	if (p[2] != 8) return 0;                               /* zlib: "unknown compression method" */
	const unsigned flg = p[3];
	if (flg & 0xE0) return 0;                              /* zlib: "unknown header flags set" */
	size_t o = 10;
	if (flg & 4) {                                         /* FEXTRA */
		if (n < o + 2) return SIZE_MAX;
		const size_t xlen = (size_t) p[o] | ((size_t) p[o + 1] << 8);
		o += 2;
		if (n < o + xlen) return SIZE_MAX;
		o += xlen;
	}
	for (unsigned bit = 8; bit <= 16; bit <<= 1)           /* FNAME, FCOMMENT: zero-terminated */
		if (flg & bit) {
			const void *z = memchr(p + o, 0, n - o);
			if (!z) return SIZE_MAX;
			o = (size_t) ((const uint8_t *) z - p) + 1;
		}
	if (flg & 2) {                                         /* FHCRC */
		if (n < o + 2) return SIZE_MAX;
		const uint32_t want = (uint32_t) p[o] | ((uint32_t) p[o + 1] << 8);
		if ((crc32(0L, p, (uInt) o) & 0xFFFFu) != want) return 0;   /* zlib: "header crc mismatch" */
		o += 2;
	}
	return o;
}
/* BGZF member at p: total size of the member (header + data + trailer) from its "BC" extra subfield, 0 if p is not a
 * complete BGZF member */
size_t bgzf_member_size(const uint8_t *p, size_t n)
{
	if (n < 28 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || p[3] != 4) return 0;       /* FLG = FEXTRA only */
	const size_t xlen = (size_t) p[10] | ((size_t) p[11] << 8);
	if (n < 12 + xlen) return 0;
	for (size_t o = 12; o + 4 <= 12 + xlen;) {
		const size_t slen = (size_t) p[o + 2] | ((size_t) p[o + 3] << 8);
		if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) {
			const size_t bsize = ((size_t) p[o + 4] | ((size_t) p[o + 5] << 8)) + 1;
			if (bsize < 12 + xlen + 8 || bsize > n) return 0;
			return bsize;
		}
		o += 4 + slen;
	}
	return 0;
}

std::atomic<unsigned> g_decoder_threads { 1 };
std::atomic<size_t> g_parallel_chunk { 1u << 20 };   /* measured: 0.5 / 1 / 2 / 4 MiB -> 0.71 / 0.51 / 0.57 / 0.89 s for a 1.05 GB .gz with 16 + 32 threads */
std::atomic<uint64_t> g_par_spliced { 0 }, g_par_dropped { 0 };
} // namespace

void GzStream::last_parallel_stats(uint64_t out[2]) { out[0] = g_par_spliced; out[1] = g_par_dropped; }

void GzStream::set_decoder_threads(unsigned n) { g_decoder_threads = n < 1 ? 1 : (n > 64 ? 64 : n); }
void GzStream::set_parallel_chunk(size_t bytes) { g_parallel_chunk = bytes ? std::max<size_t>(bytes, 1024) : (1u << 20); }

bool GzStream::is_gzip(const std::string &path)
{
	struct stat st;
	/* stat before open: opening and closing a FIFO or a /dev/fd pipe just to look at it can end its writer */
	if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 2) return false;
	const int fd = ::open(path.c_str(), O_RDONLY);
	if (fd < 0) return false;
	unsigned char magic[2] = { 0, 0 };
	const bool ok = pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
	::close(fd);
	return ok;
}

bool GzStream::open(const std::string &path)
{
	close();
	m_fd = ::open(path.c_str(), O_RDONLY);
	if (m_fd < 0) return false;
	struct stat st;
	if (fstat(m_fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 2) { ::close(m_fd); m_fd = -1; return false; }
	m_size = (size_t) st.st_size;
	void *m = mmap(nullptr, m_size, PROT_READ, MAP_PRIVATE, m_fd, 0);
	if (m == MAP_FAILED) { ::close(m_fd); m_fd = -1; return false; }
	madvise(m, m_size, MADV_SEQUENTIAL);
	m_map = (const uint8_t *) m;
	m_stop = false;
	m_final = 0;
	m_off = 0;
	m_crc = 0;
	m_len = 0;
	m_nThreads = g_decoder_threads;                        /* as set when the stream was opened */
	m_chunkBytes = g_parallel_chunk;
	m_thread = std::thread([this]() { produce(); });
	return true;
}

void GzStream::close()
{
	if (m_thread.joinable()) {
		{
			std::lock_guard<std::mutex> lk(m_mu);
			m_stop = true;
		}
		m_cv.notify_all();
		m_thread.join();
	}
	m_ready.clear();
	m_free.clear();
	m_cur.reset();
	if (m_map) munmap(const_cast<uint8_t *>(m_map), m_size);
	if (m_fd >= 0) ::close(m_fd);
	m_map = nullptr;
	m_fd = -1;
}

std::unique_ptr<GzStream::Piece> GzStream::blank()
{
	{
		std::lock_guard<std::mutex> lk(m_mu);
		if (!m_free.empty()) {
			std::unique_ptr<Piece> p = std::move(m_free.front());
			m_free.pop_front();
			p->len = 0; p->member_end = false; p->status = 0; p->checked = false; p->have_crc = false; p->ready = true;
			return p;
		}
	}
	std::unique_ptr<Piece> p(new Piece());
	p->data.resize(kPiece + kSlack);
	return p;
}

bool GzStream::push(std::unique_ptr<Piece> p)
{
	std::unique_lock<std::mutex> lk(m_mu);
	m_cv.wait(lk, [&]() { return m_stop || m_ready.size() < kQueue; });
	if (m_stop) return false;
	m_ready.push_back(std::move(p));
	lk.unlock();
	m_cv.notify_all();
	return true;
}

/* Compressed bytes [lo, hi) of the mapped input have been decoded: drop their page-table entries now, from the worker that is
 * done with them (the page cache keeps the data; a late reader -- the in-order decoder after a dropped chunk -- only faults
 * them back in).  Left mapped, a 2 GB input is half a million entries that the kernel tears down single-threaded when the
 * process exits: 0.15 s of wall time after `Time:` has been printed.  NTSM_KEEP_MAPPED=1 turns it off (measurement). */
void GzStream::release_input(const uint8_t *lo, const uint8_t *hi)
{
	static const bool keep = getenv("NTSM_KEEP_MAPPED") != nullptr;
	if (keep) return;
	const uintptr_t a = ((uintptr_t) lo + 4095u) & ~(uintptr_t) 4095u, b = (uintptr_t) hi & ~(uintptr_t) 4095u;
	if (b > a) madvise((void *) a, b - a, MADV_DONTNEED);
}

/* decoder thread */
void GzStream::produce()
{
	std::vector<uint8_t> work(kWindow + kPiece + kSlack);
	Inflate inf;
	const uint8_t *p = m_map, *const end = m_map + m_size;
	std::unique_ptr<Parallel> par;
	g_par_spliced = 0;
	g_par_dropped = 0;
	auto finish = [&](int status) {
		if (par) { g_par_spliced = par->spliced; g_par_dropped = par->dropped; }
		std::unique_ptr<Piece> e = blank();
		e->status = status;
		push(std::move(e));
	};
	bool first = true;
	const unsigned n_threads = m_nThreads;
	if (n_threads > 1 && bgzf_member_size(p, (size_t) (end - p))) {
		const uint8_t *q = produce_bgzf(p, n_threads);
		if (!q) return;
		if (q != p) first = false;
		p = q;
	}
	/* one long deflate stream (everything that is not BGZF): chunk workers decode ahead, this thread splices (gz_parallel.hpp) */
	const size_t chunk = m_chunkBytes;
	if (n_threads > 1 && (size_t) (end - p) >= 2 * chunk) par.reset(new Parallel(this, n_threads, chunk));
	for (;;) {
		/* member header (the first one was recognised by its magic; later ones: anything else is trailing garbage) */
		if (p == end) { finish(1); return; }
		if (!first && !(end - p > 1 && p[0] == 0x1f && p[1] == 0x8b)) { finish(1); return; }
		first = false;
		const size_t h = skip_header(p, (size_t) (end - p));
		if (h == SIZE_MAX) { finish(1); return; }             /* truncated inside the header */
		if (h == 0) { finish(-1); return; }
		inf.reset(p + h, end);
		size_t out = 0, sent = 0;                              /* work[0, out) = this member's recent output */
		bool member_done = false;
		for (;;) {
			if (par) inf.set_stop(m_map, par->target(inf.bit_pos(m_map)));
			const Inflate::Status st = inf.run(work.data(), &out, kWindow + kPiece);
			if (out > sent) {
				std::unique_ptr<Piece> pc = blank();
				pc->len = out - sent;
				if (pc->data.size() < pc->len) pc->data.resize(pc->len);
				memcpy(pc->data.data(), work.data() + sent, pc->len);
				sent = out;
				if (!push(std::move(pc))) return;
			}
			if (st == Inflate::MORE) {
				if (out >= kWindow + kPiece) {                   /* keep the last 32 KiB as history */
					memmove(work.data(), work.data() + out - kWindow, kWindow);
					out = sent = kWindow;
				}
				continue;
			}
			if (st == Inflate::BLOCK_STOP) {
				/* A block ended at or beyond the stop bit.  If a finished chunk starts at exactly this bit and a full window of
				 * this member's output is at hand, its symbols become bytes (markers -> window bytes) and decoding continues where
				 * the chunk ended; otherwise the next target is asked for at the top of the loop. */
				const uint64_t pos = inf.bit_pos(m_map);
				if (par->target(pos) != pos) continue;
				std::unique_ptr<Parallel::Chunk> c = par->take();
				if (inf.total_out() < kWindow || out < kWindow) {           /* starts here, but this member has no full window yet: dropped */
					par->drop(std::move(c));
					continue;
				}
				const uint8_t *const win = work.data() + out - kWindow;
				const size_t n = c->n_sym;
				const uint16_t *const sym = c->sym.data() + kWindow;
				uint8_t next_win[kWindow];
				if (n >= kWindow) SpecInflate::resolve(sym + n - kWindow, kWindow, win, kWindow, next_win);
				else {
					memcpy(next_win, win + n, kWindow - n);
					SpecInflate::resolve(sym, n, win, kWindow, next_win + kWindow - n);
				}
				const uint64_t e_bit = c->e_bit, total = inf.total_out() + n;
				const bool hit_final = c->hit_final;
				if (n) {
					std::unique_ptr<Piece> pc = blank();
					pc->len = n;
					if (pc->data.size() < n) pc->data.resize(n);
					pc->have_crc = true;
					pc->ready = false;
					Piece *raw = pc.get();
					/* queued for the reader first (not ready: it waits), handed to the workers second: a reader that has gone
					 * away (push refuses, the piece dies here) must never leave a worker writing into a freed piece.  Once
					 * queued, the piece lives until close() has joined this thread and, through it, the workers. */
					if (!push(std::move(pc))) return;
					par->resolve_async(std::move(c), win, raw);    /* copies the window; fills raw->data, raw->crc, sets ready */
				}
				memcpy(work.data(), next_win, kWindow);
				out = sent = kWindow;
				if (hit_final) {                                   /* the chunk ran to the end of the member's last block */
					p = m_map + (e_bit + 7) / 8;
					member_done = true;
					break;
				}
				inf.reset_at(m_map + e_bit / 8, (unsigned) (e_bit & 7u), end, total);
				continue;
			}
			if (st == Inflate::TRUNCATED) { finish(1); return; }
			if (st == Inflate::DATA_ERROR) { finish(-1); return; }
			break;                                              /* STREAM_END */
		}
		if (!member_done) p = inf.in();
		if (end - p < 8) { finish(1); return; }               /* truncated trailer: like a truncated stream */
		std::unique_ptr<Piece> t = blank();
		t->member_end = true;
		t->crc = le32(p);
		t->isize = le32(p + 4);
		if (!push(std::move(t))) return;
		p += 8;
	}
}

/* Parallel phase over a run of complete BGZF members starting at p. */
const uint8_t *GzStream::produce_bgzf(const uint8_t *p, unsigned n_threads)
{
	constexpr size_t kGroupBytes = 2u << 20;                 /* uncompressed bytes per work item */
	struct Member { const uint8_t *data, *trailer; uint32_t isize; };
	struct Group { uint64_t id; std::vector<Member> members; size_t out_bytes; const uint8_t *begin; };
	struct Done { std::unique_ptr<Piece> piece; bool failed; const uint8_t *resume; };
	const uint8_t *const end = m_map + m_size;
	std::mutex mu;
	std::condition_variable cv;
	std::deque<Group> work;
	std::map<uint64_t, Done> done;
	bool no_more = false, abort = false;
	auto worker = [&]() {
		Inflate inf;
		for (;;) {
			Group g;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&]() { return abort || no_more || !work.empty(); });
				if (abort || work.empty()) return;
				g = std::move(work.front());
				work.pop_front();
			}
			Done d;
			d.piece = blank();
			d.failed = false;
			d.resume = nullptr;
			if (d.piece->data.size() < g.out_bytes + kSlack) d.piece->data.resize(g.out_bytes + kSlack);
			d.piece->checked = true;
			size_t at = 0;
			const uint8_t *member_begin = g.begin;
			for (const Member &m : g.members) {
				inf.reset(m.data, m.trailer);
				size_t out = 0;
				Inflate::Status st = Inflate::MORE;
				while (st == Inflate::MORE && out <= m.isize) st = inf.run(d.piece->data.data() + at, &out, (size_t) m.isize + 1);
				if (st != Inflate::STREAM_END || out != m.isize || inf.in() != m.trailer ||
				    crc32_fast(0, d.piece->data.data() + at, out) != le32(m.trailer)) {
					d.failed = true;                            /* the sequential decoder repeats this member: same bytes, same verdict as zlib */
					d.resume = member_begin;
					break;
				}
				at += out;
				member_begin = m.trailer + 8;
			}
			d.piece->len = at;
			if (!d.failed) release_input(g.begin, member_begin);
			{
				std::lock_guard<std::mutex> lk(mu);
				done.emplace(g.id, std::move(d));
			}
			cv.notify_all();
		}
	};
	std::vector<std::thread> pool;
	for (unsigned i = 0; i < n_threads; ++i) pool.emplace_back(worker);
	auto shutdown = [&](bool hard) {
		{
			std::lock_guard<std::mutex> lk(mu);
			if (hard) abort = true;
			no_more = true;
		}
		cv.notify_all();
		for (auto &t : pool) t.join();
	};
	uint64_t issued = 0, delivered = 0;
	const uint8_t *q = p, *resume = nullptr;
	bool scan_done = false;
	for (;;) {
		/* keep the workers fed: up to 3 groups per thread in flight */
		while (!scan_done && issued - delivered < 3ull * n_threads) {
			Group g;
			g.id = issued;
			g.out_bytes = 0;
			g.begin = q;
			while (g.out_bytes < kGroupBytes) {
				const size_t sz = bgzf_member_size(q, (size_t) (end - q));
				if (!sz) { scan_done = true; break; }
				const size_t xlen = (size_t) q[10] | ((size_t) q[11] << 8);
				Member m { q + 12 + xlen, q + sz - 8, le32(q + sz - 4) };
				if (m.isize > 65536) { scan_done = true; break; }   /* not BGZF after all */
				g.members.push_back(m);
				g.out_bytes += m.isize;
				q += sz;
			}
			if (g.members.empty()) break;
			{
				std::lock_guard<std::mutex> lk(mu);
				work.push_back(std::move(g));
			}
			cv.notify_all();
			++issued;
		}
		if (delivered == issued) { resume = q; break; }         /* nothing in flight and nothing more to issue */
		Done d;
		{
			std::unique_lock<std::mutex> lk(mu);
			cv.wait(lk, [&]() { return done.count(delivered) != 0; });
			d = std::move(done[delivered]);
			done.erase(delivered);
		}
		++delivered;
		if (d.piece->len || !d.failed) {
			if (!push(std::move(d.piece))) { shutdown(true); return nullptr; }
		}
		if (d.failed) { resume = d.resume; break; }
	}
	shutdown(true);
	return resume;
}

/* Next piece in stream order, with what read() used to do as the bytes went by: CRC-32 and length of the member so far (a
 * spliced chunk brings its own CRC: combined; BGZF groups were verified by their workers), the check at a member's end, the
 * final status. */
std::unique_ptr<GzStream::Piece> GzStream::pop()
{
	if (m_final) return nullptr;
	std::unique_ptr<Piece> p;
	{
		std::unique_lock<std::mutex> lk(m_mu);
		m_cv.wait(lk, [&]() { return !m_ready.empty() && m_ready.front()->ready; });
		p = std::move(m_ready.front());
		m_ready.pop_front();
	}
	m_cv.notify_all();
	if (p->len && !p->checked) {
		if (p->have_crc) m_crc = (uint32_t) crc32_combine(m_crc, p->crc, (z_off_t) p->len);
		else m_crc = crc32_fast(m_crc, p->data.data(), p->len);
		m_len += p->len;
	}
	if (p->member_end) {
		if (m_crc != p->crc || (uint32_t) m_len != p->isize) m_final = -1;   /* zlib: incorrect data / length check */
		m_crc = 0;
		m_len = 0;
	}
	if (p->status && !m_final) m_final = p->status;
	return p;
}

void GzStream::give_back(std::unique_ptr<Piece> p)
{
	if (!p) return;
	std::lock_guard<std::mutex> lk(m_mu);
	if (m_free.size() < 2 * kQueue) m_free.push_back(std::move(p));
}

std::unique_ptr<GzStream::Piece> GzStream::take()
{
	for (;;) {
		if (!m_stash.empty()) {                                /* handed back by unread(): these have been through pop() already */
			std::unique_ptr<Piece> p = std::move(m_stash.front());
			m_stash.pop_front();
			if (m_stashOff) {
				const size_t off = std::min(m_stashOff, p->len);
				memmove(p->data.data(), p->data.data() + off, p->len - off);
				p->len -= off;
				m_stashOff = 0;
			}
			if (p->len) return p;
			give_back(std::move(p));
			continue;
		}
		std::unique_ptr<Piece> p = pop();
		if (!p) return nullptr;
		if (p->len) return p;
		const bool last = m_final != 0;
		give_back(std::move(p));
		if (last) return nullptr;
	}
}

void GzStream::unread(std::deque<std::unique_ptr<Piece>> pieces, size_t offset)
{
	if (m_cur) { give_back(std::move(m_cur)); m_cur.reset(); }
	/* what an earlier unread() left stays behind the new pieces (a second parallel phase on the same stream, early_ingest.cpp) */
	if (!m_stash.empty() && m_stashOff) {
		Piece &f = *m_stash.front();
		const size_t off = std::min(m_stashOff, f.len);
		memmove(f.data.data(), f.data.data() + off, f.len - off);
		f.len -= off;
	}
	for (auto &p : m_stash) pieces.push_back(std::move(p));
	m_stash = std::move(pieces);
	m_stashOff = offset;
}

int GzStream::read(void *dst, unsigned len)
{
	uint8_t *d = (uint8_t *) dst;
	unsigned got = 0;
	while (got < len) {
		if (!m_cur) {
			if (!m_stash.empty()) {
				m_cur = std::move(m_stash.front());
				m_stash.pop_front();
				m_off = m_stashOff;
				m_stashOff = 0;
			} else {
				m_cur = pop();
				if (!m_cur) break;
				m_off = 0;
			}
		}
		const size_t n = std::min<size_t>(len - got, m_cur->len - m_off);
		if (n) {
			memcpy(d + got, m_cur->data.data() + m_off, n);
			m_off += n;
			got += (unsigned) n;
		}
		if (m_off >= m_cur->len) {
			give_back(std::move(m_cur));
			m_cur.reset();
		}
	}
	if (got) return (int) got;
	return m_final < 0 ? -1 : 0;
}

} // namespace ntsm
