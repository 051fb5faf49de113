#include "site_set.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <ostream>
#include <thread>

#include "kmer.hpp"
#include "seq_reader.hpp"

namespace ntsm {

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
	unsigned n_thr = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
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
	const bool prof = getenv("NTSM_SITES_PROF") != nullptr;     /* phase times of this function on stderr */
	const auto tp0 = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (prof) fprintf(stderr, "[sites] %s: %.4f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count());
	};
	k = kk;
	ids.clear(); ref.clear(); var.clear(); keys.clear();
	n_erased = 0;
	SeqReader rd;
	if (!rd.open(path)) return false;
	std::vector<Occurrences> parts;
	const bool parallel = k >= 1 && parallel_two_line_fasta(path, k, parts);
	/* Pass 1: every k-mer occurrence of the file in stream order (the reference inserts them one by one into
	 * m_counts, src/FingerPrint.hpp:507-556).  "Seen before" is decided afterwards by sorting the occurrences by
	 * (code, occurrence) -- many times faster than 1.5 M dependent probes of a 50 MB hash table, with identical results: the
	 * first element of a group of equal codes is the first-seen occurrence. */
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
	lap("pass 1 (k-mers of every record, stitched)");
	if (n_occ > 0xFFFFFFF0ull) { err << "too many k-mers in " << path << std::endl; return false; }
	/* Pass 2: group equal codes and classify.  Occurrences are scattered into 256 buckets by a hash of the code (stable: slice
	 * by slice in stream order), then every bucket -- a few thousand pairs, cache resident -- is sorted by (code, occurrence)
	 * and classified on its own; buckets are independent, so both steps run on several threads. */
	std::vector<uint8_t> later(n_occ, 0);                      /* occurrence of a code that was seen before */
	std::vector<uint8_t> dup_first(n_occ, 0);                  /* first occurrence of a code that occurs again */
	{
		struct Pair { uint64_t code; uint32_t idx; };
		constexpr unsigned NB = 256;
		const unsigned T = (unsigned) std::max<size_t>(1, std::min<size_t>(std::min(8u, std::max(1u, std::thread::hardware_concurrency())), n_occ / 65536 + 1));
		auto bucket_of = [](uint64_t code) { return (unsigned) ((code * 0x9E3779B97F4A7C15ull) >> 56); };
		std::vector<std::vector<size_t>> cnt(T, std::vector<size_t>(NB, 0));
		std::vector<Pair> pairs(n_occ);
		std::vector<size_t> start(NB + 1, 0);
		auto slice = [&](unsigned t, size_t *lo, size_t *hi) { *lo = n_occ * t / T; *hi = n_occ * (t + 1) / T; };
		auto on_threads = [&](const std::function<void(unsigned)> &f) {
			std::vector<std::thread> pool;
			for (unsigned t = 1; t < T; ++t) pool.emplace_back(f, t);
			f(0);
			for (auto &th : pool) th.join();
		};
		on_threads([&](unsigned t) {
			size_t lo, hi;
			slice(t, &lo, &hi);
			for (size_t i = lo; i < hi; ++i) cnt[t][bucket_of(occ_code[i])]++;
		});
		for (unsigned b = 0; b < NB; ++b) {
			size_t run = start[b];
			for (unsigned t = 0; t < T; ++t) { const size_t c = cnt[t][b]; cnt[t][b] = run; run += c; }
			start[b + 1] = run;
		}
		on_threads([&](unsigned t) {
			size_t lo, hi;
			slice(t, &lo, &hi);
			for (size_t i = lo; i < hi; ++i) pairs[cnt[t][bucket_of(occ_code[i])]++] = Pair { occ_code[i], (uint32_t) i };
		});
		std::atomic<unsigned> next_bucket { 0 };
		on_threads([&](unsigned) {
			for (unsigned b = next_bucket++; b < NB; b = next_bucket++) {
				Pair *const p0 = pairs.data() + start[b], *const p1 = pairs.data() + start[b + 1];
				std::sort(p0, p1, [](const Pair &x, const Pair &y) { return x.code != y.code ? x.code < y.code : x.idx < y.idx; });
				for (Pair *p = p0; p < p1;) {
					Pair *q = p + 1;
					while (q < p1 && q->code == p->code) later[(q++)->idx] = 1;
					if (q - p > 1) dup_first[p->idx] = 1;
					p = q;
				}
			}
		});
	}
	lap("pass 2 (sort + classify)");
	/* Pass 3: stream order again -- warnings, allele lists, keys (first-seen order minus erased duplicates).  The key index
	 * of a record's first kept k-mer is a prefix sum over the records; with that every record is filled independently. */
	for (size_t o = 0, r = 0; o < n_occ; ++o) {                 /* collisions are rare: one sequential scan, file order */
		if (!later[o]) continue;
		while (rec_begin[r + 1] <= o) ++r;
		err << "Warning: " << rec_name[r] << " of " << (r % 2 == 0 ? "REF" : "VAR")
		    << " file has a k-mer collision at pos: " << occ_pos[o] << std::endl;
	}
	ref.assign((n_rec + 1) / 2, std::vector<int64_t>());
	var.assign(n_rec / 2, std::vector<int64_t>());
	ids.resize((n_rec + 1) / 2);
	std::vector<uint64_t> key_base(n_rec + 1, 0);
	{
		const unsigned T = (unsigned) std::max<size_t>(1, std::min<size_t>(std::min(8u, std::max(1u, std::thread::hardware_concurrency())), n_rec / 8192 + 1));
		std::vector<uint64_t> erased(T, 0);
		auto on_threads = [&](const std::function<void(unsigned)> &f) {
			std::vector<std::thread> pool;
			for (unsigned t = 1; t < T; ++t) pool.emplace_back(f, t);
			f(0);
			for (auto &th : pool) th.join();
		};
		on_threads([&](unsigned t) {                            /* kept k-mers per record */
			for (size_t r = n_rec * t / T; r < n_rec * (t + 1) / T; ++r) {
				uint64_t kept = 0;
				for (uint64_t o = rec_begin[r]; o < rec_begin[r + 1]; ++o) {
					if (later[o]) continue;
					if (dup_first[o] && !allow_dupes) ++erased[t];
					else ++kept;
				}
				key_base[r + 1] = kept;
			}
		});
		for (size_t r = 0; r < n_rec; ++r) key_base[r + 1] += key_base[r];
		for (uint64_t e : erased) n_erased += e;
		keys.resize(key_base[n_rec]);
		on_threads([&](unsigned t) {
			for (size_t r = n_rec * t / T; r < n_rec * (t + 1) / T; ++r) {
				std::vector<int64_t> &list = (r % 2 == 0 ? ref : var)[r / 2];
				uint64_t at = key_base[r];
				for (uint64_t o = rec_begin[r]; o < rec_begin[r + 1]; ++o) {
					if (later[o]) continue;                       /* warned about above; not in the list (:507-556) */
					if (dup_first[o] && !allow_dupes) {
						list.push_back(kErased);                     /* :557-563: erased again, stays in this allele's list */
					} else {
						list.push_back((int64_t) at);
						keys[at++] = occ_code[o];
					}
				}
				if (r % 2 == 0) ids[r / 2] = std::move(rec_name[r]);
			}
		});
	}
	lap("pass 3 (allele lists, keys)");
	return true;
}

} // namespace ntsm
