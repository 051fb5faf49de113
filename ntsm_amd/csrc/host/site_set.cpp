#include "site_set.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <ostream>
#include <thread>

#include "kmer.hpp"
#include "seq_reader.hpp"

namespace ntsm {

namespace {

/* Stable LSD radix sort of (key, idx) pairs by key, 11 bits per pass; passes whose digit is constant are skipped. */
void radix_sort_pairs(std::vector<uint64_t> &key, std::vector<uint32_t> &idx)
{
	const size_t n = key.size();
	if (n < 2) return;
	uint64_t all_or = 0, all_and = ~0ull;
	for (uint64_t x : key) { all_or |= x; all_and &= x; }
	const uint64_t varying = all_or ^ all_and;                  /* bits that differ somewhere */
	std::vector<uint64_t> key2(n);
	std::vector<uint32_t> idx2(n);
	constexpr int B = 11;
	std::vector<size_t> count((size_t) 1 << B);
	for (int sh = 0; sh < 64; sh += B) {
		if (((varying >> sh) & ((1u << B) - 1)) == 0) continue;
		std::fill(count.begin(), count.end(), 0);
		for (size_t i = 0; i < n; ++i) count[(key[i] >> sh) & ((1u << B) - 1)]++;
		size_t run = 0;
		for (size_t d = 0; d < count.size(); ++d) { const size_t c = count[d]; count[d] = run; run += c; }
		for (size_t i = 0; i < n; ++i) {
			const size_t d = (key[i] >> sh) & ((1u << B) - 1);
			const size_t o = count[d]++;
			key2[o] = key[i];
			idx2[o] = idx[i];
		}
		key.swap(key2);
		idx.swap(idx2);
	}
}

} // namespace

namespace {

/* Occurrences of one stretch of the file: what pass 1 of SiteSet::load collects */
struct Occurrences {
	std::vector<uint64_t> code;
	std::vector<uint32_t> pos;
	std::vector<uint64_t> rec_begin;                           /* first occurrence of every record (local) */
	std::vector<std::string> rec_name;
};

/* Plain two-line FASTA (">name ...\n" "SEQ\n", no CR, SEQ not empty and not starting with > @ +) parsed by several
 * threads: such a file has exactly the records kseq finds (vendor/kseq.h:177-219), and a record is self-contained, so
 * the file can be cut at any header line.  Returns false on anything else: the caller then uses the sequential reader. */
bool parallel_two_line_fasta(const std::string &path, unsigned k, std::vector<Occurrences> &parts)
{
	if (getenv("NTSM_SITES_SEQUENTIAL")) return false;        /* tests: force the sequential reader */
	struct stat st;                                            /* stat before open: never touch a FIFO / pipe here */
	if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < (1 << 20)) return false;
	const int fd = open(path.c_str(), O_RDONLY);
	if (fd < 0) return false;
	const size_t size = (size_t) st.st_size;
	void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (m == MAP_FAILED) return false;
	const char *d = (const char *) m, *const e = d + size;
	unsigned n_thr = std::min(4u, std::max(1u, std::thread::hardware_concurrency()));
	bool ok = d[0] == '>' && e[-1] == '\n';
	std::vector<const char *> cut(n_thr + 1, e);
	cut[0] = d;
	for (unsigned t = 1; t < n_thr && ok; ++t) {                /* first header line at or after the t-th fraction */
		const char *p = d + size / n_thr * t;
		while (p < e && !(p[-1] == '\n' && p[0] == '>')) {
			const char *nl = (const char *) memchr(p, '\n', (size_t) (e - p));
			p = nl ? nl + 1 : e;
		}
		cut[t] = p;
	}
	parts.assign(n_thr, Occurrences());
	std::vector<char> good(n_thr, 1);
	auto work = [&](unsigned t) {
		Occurrences &o = parts[t];
		const char *p = cut[t], *const stop = cut[t + 1];
		while (p < stop) {
			const char *l1 = (const char *) memchr(p, '\n', (size_t) (e - p));
			if (*p != '>' || !l1 || l1[-1] == '\r' || l1 + 1 >= e) { good[t] = 0; return; }
			const char *sq = l1 + 1;
			const char *l2 = (const char *) memchr(sq, '\n', (size_t) (e - sq));
			if (!l2 || l2 == sq || *sq == '>' || *sq == '@' || *sq == '+' || l2[-1] == '\r' || (l2 + 1 < e && l2[1] != '>')) { good[t] = 0; return; }
			const char *nm = p + 1, *ne = nm;
			while (ne < l1 && !isspace((unsigned char) *ne)) ++ne;
			o.rec_begin.push_back(o.code.size());
			o.rec_name.emplace_back(nm, ne);
			for_each_kmer(sq, (uint64_t) (l2 - sq), k, [&](uint64_t code, uint64_t pos) {
				o.code.push_back(code);
				o.pos.push_back((uint32_t) pos);
			});
			p = l2 + 1;
		}
	};
	if (ok) {
		std::vector<std::thread> pool;
		for (unsigned t = 1; t < n_thr; ++t) pool.emplace_back(work, t);
		work(0);
		for (auto &th : pool) th.join();
		for (char g : good) ok = ok && g;
	}
	munmap(m, size);
	return ok;
}

} // namespace

bool SiteSet::load(const std::string &path, unsigned kk, bool allow_dupes, std::ostream &err)
{
	k = kk;
	ids.clear(); ref.clear(); var.clear(); keys.clear();
	n_erased = 0;
	SeqReader rd;
	if (!rd.open(path)) return false;
	std::vector<Occurrences> parts;
	const bool parallel = k >= 1 && parallel_two_line_fasta(path, k, parts);
	/* Pass 1: every k-mer occurrence of the file in stream order (the reference inserts them one by one into
	 * m_counts, src/FingerPrint.hpp:507-556).  "Seen before" is decided afterwards by sorting the occurrences by
	 * code -- 3-5x faster than 1.5 M dependent probes of a 50 MB hash table, with identical results: within equal
	 * codes the stable sort keeps stream order, so the first element of a group is the first-seen occurrence. */
	std::vector<uint64_t> occ_code;
	std::vector<uint32_t> occ_pos;
	std::vector<uint64_t> rec_begin;                           /* first occurrence of every record; [n_rec] = total */
	std::vector<std::string> rec_name;
	if (parallel) {                                            /* stitch the per-thread pieces together in file order */
		size_t n_o = 0, n_r = 0;
		for (const Occurrences &o : parts) { n_o += o.code.size(); n_r += o.rec_name.size(); }
		occ_code.reserve(n_o); occ_pos.reserve(n_o); rec_begin.reserve(n_r + 1); rec_name.reserve(n_r);
		for (Occurrences &o : parts) {
			const uint64_t base = occ_code.size();
			for (uint64_t b : o.rec_begin) rec_begin.push_back(base + b);
			for (std::string &nm : o.rec_name) rec_name.push_back(std::move(nm));
			occ_code.insert(occ_code.end(), o.code.begin(), o.code.end());
			occ_pos.insert(occ_pos.end(), o.pos.begin(), o.pos.end());
			Occurrences().code.swap(o.code);
		}
	} else {
		for (int64_t l = rd.next(); l >= 0; l = rd.next()) {
			rec_begin.push_back(occ_code.size());
			rec_name.push_back(rd.name());
			for_each_kmer(rd.seq_data(), (uint64_t) l, k, [&](uint64_t code, uint64_t pos) {
				occ_code.push_back(code);
				occ_pos.push_back((uint32_t) pos);
			});
		}
	}
	const size_t n_occ = occ_code.size(), n_rec = rec_name.size();
	rec_begin.push_back(n_occ);
	if (n_occ > 0xFFFFFFF0ull) { err << "too many k-mers in " << path << std::endl; return false; }
	/* Pass 2: sort (code, occurrence) and classify */
	std::vector<uint8_t> later(n_occ, 0);                      /* occurrence of a code that was seen before */
	std::vector<uint8_t> dup_first(n_occ, 0);                  /* first occurrence of a code that occurs again */
	{
		std::vector<uint64_t> sk(occ_code);
		std::vector<uint32_t> si(n_occ);
		for (size_t i = 0; i < n_occ; ++i) si[i] = (uint32_t) i;
		radix_sort_pairs(sk, si);
		for (size_t i = 0; i < n_occ;) {
			size_t j = i + 1;
			while (j < n_occ && sk[j] == sk[i]) later[si[j++]] = 1;
			if (j - i > 1) dup_first[si[i]] = 1;
			i = j;
		}
	}
	/* Pass 3: stream order again -- warnings, allele lists, keys (first-seen order minus erased duplicates) */
	ref.reserve(n_rec / 2 + 1);
	var.reserve(n_rec / 2 + 1);
	ids.reserve(n_rec / 2 + 1);
	for (size_t r = 0; r < n_rec; ++r) {
		const bool is_ref = (r % 2 == 0);
		std::vector<std::vector<int64_t>> &side = is_ref ? ref : var;
		side.emplace_back();
		std::vector<int64_t> &list = side.back();
		for (uint64_t o = rec_begin[r]; o < rec_begin[r + 1]; ++o) {
			if (later[o]) {
				err << "Warning: " << rec_name[r] << " of " << (is_ref ? "REF" : "VAR")
				    << " file has a k-mer collision at pos: " << occ_pos[o] << std::endl;
			} else if (dup_first[o] && !allow_dupes) {
				++n_erased;                                          /* :557-563: erased again, stays in this allele's list */
				list.push_back(kErased);
			} else {
				list.push_back((int64_t) keys.size());
				keys.push_back(occ_code[o]);
			}
		}
		if (is_ref) ids.push_back(std::move(rec_name[r]));
	}
	return true;
}

} // namespace ntsm
