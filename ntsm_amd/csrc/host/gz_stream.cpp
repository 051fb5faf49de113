#include "gz_stream.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cstring>

#include "crc32_fast.hpp"
#include "inflate.hpp"

namespace ntsm {

namespace {
constexpr size_t kPiece = 1u << 20, kWindow = 32768, kSlack = 512, kQueue = 8;
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
} // namespace

bool GzStream::is_gzip(const std::string &path)
{
	const int fd = ::open(path.c_str(), O_RDONLY);
	if (fd < 0) return false;
	struct stat st;
	unsigned char magic[2] = { 0, 0 };
	const bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 2 && pread(fd, magic, 2, 0) == 2 &&
	                magic[0] == 0x1f && magic[1] == 0x8b;
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
			p->len = 0; p->member_end = false; p->status = 0;
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

/* decoder thread */
void GzStream::produce()
{
	std::vector<uint8_t> work(kWindow + kPiece + kSlack);
	Inflate inf;
	const uint8_t *p = m_map, *const end = m_map + m_size;
	auto finish = [&](int status) {
		std::unique_ptr<Piece> e = blank();
		e->status = status;
		push(std::move(e));
	};
	bool first = true;
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
		for (;;) {
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
			if (st == Inflate::TRUNCATED) { finish(1); return; }
			if (st == Inflate::DATA_ERROR) { finish(-1); return; }
			break;                                              /* STREAM_END */
		}
		p = inf.in();
		if (end - p < 8) { finish(1); return; }               /* truncated trailer: like a truncated stream */
		std::unique_ptr<Piece> t = blank();
		t->member_end = true;
		t->crc = le32(p);
		t->isize = le32(p + 4);
		if (!push(std::move(t))) return;
		p += 8;
	}
}

int GzStream::read(void *dst, unsigned len)
{
	uint8_t *d = (uint8_t *) dst;
	unsigned got = 0;
	while (got < len) {
		if (!m_cur) {
			if (m_final) break;
			std::unique_lock<std::mutex> lk(m_mu);
			m_cv.wait(lk, [&]() { return !m_ready.empty(); });
			m_cur = std::move(m_ready.front());
			m_ready.pop_front();
			lk.unlock();
			m_cv.notify_all();
			m_off = 0;
		}
		const size_t n = std::min<size_t>(len - got, m_cur->len - m_off);
		if (n) {
			memcpy(d + got, m_cur->data.data() + m_off, n);
			m_crc = crc32_fast(m_crc, d + got, n);
			m_len += n;
			m_off += n;
			got += (unsigned) n;
		}
		if (m_off == m_cur->len) {
			if (m_cur->member_end) {
				if (m_crc != m_cur->crc || (uint32_t) m_len != m_cur->isize) m_final = -1;   /* zlib: incorrect data / length check */
				m_crc = 0;
				m_len = 0;
			}
			if (m_cur->status) m_final = m_cur->status;
			{
				std::lock_guard<std::mutex> lk(m_mu);
				m_free.push_back(std::move(m_cur));
			}
			m_cur.reset();
			if (m_final) break;
		}
	}
	if (got) return (int) got;
	return m_final < 0 ? -1 : 0;
}

} // namespace ntsm
