#include "parallel_fastq.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>

namespace ntsm {

ParallelFastq::~ParallelFastq()
{
	if (m_data) munmap(const_cast<char *>(m_data), m_size);
	if (m_fd >= 0) close(m_fd);
}

bool ParallelFastq::open(const std::string &path, uint64_t block_bytes)
{
	if (block_bytes < 4096) block_bytes = 4096;
	struct stat st;                                                       /* stat before open: never touch a FIFO / pipe here */
	if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode) || (uint64_t) st.st_size < 2 * block_bytes) return false;   /* small: not worth it */
	m_fd = ::open(path.c_str(), O_RDONLY);
	if (m_fd < 0) return false;
	m_size = (uint64_t) st.st_size;
	void *m = mmap(nullptr, m_size, PROT_READ, MAP_PRIVATE, m_fd, 0);
	if (m == MAP_FAILED) return false;
	m_data = (const char *) m;
	madvise(m, m_size, MADV_SEQUENTIAL);
	if (m_data[0] != '@') return false;                                   /* gzip (0x1f 0x8b), FASTA, junk before the first header */
	m_block = block_bytes;
	m_nBlocks = (size_t) ((m_size + block_bytes - 1) / block_bytes);
	return true;
}

const char *ParallelFastq::strict_record(const char *p, const char *e, const char **seq, uint64_t *len)
{
	if (p >= e || *p != '@') return nullptr;
	const char *l1 = (const char *) memchr(p, '\n', (size_t) (e - p));
	if (!l1 || l1[-1] == '\r' || l1 + 1 >= e) return nullptr;
	const char *s = l1 + 1;
	const char *l2 = (const char *) memchr(s, '\n', (size_t) (e - s));
	if (!l2) return nullptr;
	const int64_t slen = l2 - s;
	if (slen < 1 || *s == '@' || *s == '>' || *s == '+' || l2[-1] == '\r') return nullptr;
	if (l2 + 1 >= e || l2[1] != '+') return nullptr;
	const char *l3 = (const char *) memchr(l2 + 1, '\n', (size_t) (e - (l2 + 1)));
	if (!l3) return nullptr;
	const char *q = l3 + 1;
	if (e - q < slen + 1) return nullptr;                                 /* quality + its newline must be inside the file */
	if (q[slen] != '\n' || q[slen - 1] == '\r') return nullptr;
	if (memchr(q, '\n', (size_t) slen)) return nullptr;                   /* quality shorter than the sequence */
	*seq = s;
	*len = (uint64_t) slen;
	return q + slen + 1;
}

uint64_t ParallelFastq::find_start(uint64_t lo, uint64_t hi) const
{
	const char *const e = m_data + m_size, *const lim = m_data + hi;
	const char *p = m_data + lo;
	if (p[-1] != '\n') {
		const char *nl = (const char *) memchr(p, '\n', (size_t) (e - p));
		p = nl ? nl + 1 : e;
	}
	/* First line start in [lo, hi) that begins a strict record followed by another '@' (or EOF).  In a strict file
	 * the only other lines that can start with '@' are quality lines, and those are rejected because the line after
	 * them (the next header) would be a sequence starting with '@'.  A wrong guess in a non-strict file cannot slip
	 * through: wait_start() requires it to coincide with the end of the previous block's last record. */
	while (p < lim) {
		if (*p == '@') {
			const char *seq;
			uint64_t len;
			const char *r = strict_record(p, e, &seq, &len);
			if (r && (r == e || *r == '@')) return (uint64_t) (p - m_data);
		}
		const char *nl = (const char *) memchr(p, '\n', (size_t) (e - p));
		p = nl ? nl + 1 : e;
	}
	return kNone;
}

bool ParallelFastq::wait_start(size_t b, uint64_t first, uint64_t *prev_end)
{
	std::unique_lock<std::mutex> lk(m_mu);
	if (b == 0) {
		*prev_end = 0;
		return true;                                                      /* block 0 starts at byte 0 by construction */
	}
	m_cv.wait(lk, [&]() { return m_done[b - 1] || m_failBlock < b; });
	if (m_failBlock < b) return false;                                    /* an earlier block stopped the parallel phase */
	*prev_end = m_end[b - 1];
	if (first == kNone || first == *prev_end) return true;
	if (b < m_failBlock) {                                                /* guessed start is not where b-1 ended */
		m_failBlock = b;
		m_resume = *prev_end;
		m_cv.notify_all();
	}
	return false;
}

void ParallelFastq::release(size_t b) const
{
	const uint64_t page = 4096;
	const uint64_t lo = (uint64_t) b * m_block, hi = std::min(m_size, lo + m_block);
	const uint64_t a = (lo + page - 1) & ~(page - 1), z = hi & ~(page - 1);
	if (z > a) (void) madvise(const_cast<char *>(m_data) + a, z - a, MADV_DONTNEED);
}

void ParallelFastq::publish(size_t b, uint64_t end)
{
	std::lock_guard<std::mutex> lk(m_mu);
	m_end[b] = end;
	m_done[b] = 1;
	m_cv.notify_all();
}

void ParallelFastq::fail(size_t b, uint64_t resume)
{
	std::lock_guard<std::mutex> lk(m_mu);
	if (b < m_failBlock) {
		m_failBlock = b;
		m_resume = resume;
	}
	m_cv.notify_all();
}

} // namespace ntsm
