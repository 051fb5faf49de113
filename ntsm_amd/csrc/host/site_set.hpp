/*
 * site_set.hpp -- the SNP-site k-mer set: host replacement for FingerPrint::initCountsHash
 * (src/FingerPrint.hpp:490-564).  Records of the sites file alternate REF (=AT) / VAR (=CG); the
 * name of every even record is the locus ID (:531); each record is a run of k-mers joined by 'N'.
 * A k-mer seen for the first time is appended to that allele's list and becomes a key; later
 * occurrences only raise a warning and are remembered as duplicates.  Without -d every
 * duplicated k-mer is removed from the key set again (:557-563) while staying in the first
 * allele's list, which makes the reference abort when it prints (m_counts.at throws, :282).
 */
#ifndef NTSM_SITE_SET_HPP
#define NTSM_SITE_SET_HPP
#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

namespace ntsm {

struct SiteSet {
	unsigned k = 19;
	std::vector<std::string> ids;                  /* m_alleleIDs */
	/* per site: indices into `keys` (or kErased) for the REF and VAR allele k-mers */
	std::vector<std::vector<int64_t>> ref, var;    /* m_alleleIDToKmerRef / Var */
	std::vector<uint64_t> keys;                    /* distinct, non-erased canonical codes (m_counts keys) */
	uint64_t n_erased = 0;
	static constexpr int64_t kErased = -1;

	/* Returns false if the file cannot be opened.  Collision warnings go to `err`. */
	bool load(const std::string &path, unsigned k, bool allow_dupes, std::ostream &err);
	uint64_t n_distinct() const { return keys.size(); }       /* m_counts.size() */
};

} // namespace ntsm
#endif
