/*
 * report.hpp -- output formatting of ntsmCount: printOptionalHeader / printCountsMax /
 * printInfoSummary (src/FingerPrint.hpp:261-349, :389-413), as free functions over the site set
 * and the per-k-mer counts fetched from the GPU so they can be tested without a device.
 */
#ifndef NTSM_REPORT_HPP
#define NTSM_REPORT_HPP
#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

#include "site_set.hpp"

namespace ntsm {

/* "#@TK\t<totalKmers>\n#@KS\t<k>" (no trailing newline, :261-268) */
void print_optional_header(std::ostream &out, uint64_t total_kmers, unsigned k);
/* Column header + one row per site.  Returns false where the reference's m_counts.at() /
 * vector::at() would throw (erased duplicate k-mer or REF without VAR): rows before it are written. */
bool print_counts_max(std::ostream &out, const SiteSet &sites, const std::vector<uint64_t> &counts);
/* getSitesCoveredInSample, :389-413 */
unsigned sites_covered(const SiteSet &sites, const std::vector<uint64_t> &counts);
/* the six summary lines, :313-333 */
std::string info_summary(const SiteSet &sites, const std::vector<uint64_t> &counts, uint64_t total_bases,
		uint64_t total_kmers, uint64_t total_hits);

} // namespace ntsm
#endif
