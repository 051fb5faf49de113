#include "site_set.hpp"

#include <ostream>

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

bool SiteSet::load(const std::string &path, unsigned kk, bool allow_dupes, std::ostream &err)
{
	k = kk;
	ids.clear(); ref.clear(); var.clear(); keys.clear();
	n_erased = 0;
	SeqReader rd;
	if (!rd.open(path)) return false;
	/* Pass 1: every k-mer occurrence of the file in stream order (the reference inserts them one by one into
	 * m_counts, src/FingerPrint.hpp:507-556).  "Seen before" is decided afterwards by sorting the occurrences by
	 * code -- 3-5x faster than 1.5 M dependent probes of a 50 MB hash table, with identical results: within equal
	 * codes the stable sort keeps stream order, so the first element of a group is the first-seen occurrence. */
	std::vector<uint64_t> occ_code;
	std::vector<uint32_t> occ_pos;
	std::vector<uint64_t> rec_begin;                           /* first occurrence of every record; [n_rec] = total */
	std::vector<std::string> rec_name;
	for (int64_t l = rd.next(); l >= 0; l = rd.next()) {
		rec_begin.push_back(occ_code.size());
		rec_name.push_back(rd.name());
		for_each_kmer(rd.seq_data(), (uint64_t) l, k, [&](uint64_t code, uint64_t pos) {
			occ_code.push_back(code);
			occ_pos.push_back((uint32_t) pos);
		});
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
