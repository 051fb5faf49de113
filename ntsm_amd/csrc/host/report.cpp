#include "report.hpp"

#include <ostream>

namespace ntsm {

void print_optional_header(std::ostream &out, uint64_t total_kmers, unsigned k)
{
	out << "#@TK\t" << total_kmers << "\n#@KS\t" << k;
}

bool print_counts_max(std::ostream &out, const SiteSet &sites, const std::vector<uint64_t> &counts)
{
	out << "\n#locusID\tcountAT\tcountCG\tsumAT\tsumCG\tdistinctAT\tdistinctCG\n";
	std::string row;
	for (size_t i = 0; i < sites.ids.size(); ++i) {
		unsigned mx[2] = { 0, 0 }, sm[2] = { 0, 0 };       /* `unsigned`: counts wrap mod 2^32 at print, :277-294 */
		size_t n[2] = { 0, 0 };
		for (int a = 0; a < 2; ++a) {
			const auto &side = a == 0 ? sites.ref : sites.var;
			if (i >= side.size()) return false;
			for (int64_t ix : side[i]) {
				if (ix == SiteSet::kErased) return false;
				const unsigned f = (unsigned) counts[(size_t) ix];
				if (mx[a] < f) mx[a] = f;
				sm[a] += f;
			}
			n[a] = side[i].size();
		}
		row.clear();
		row += sites.ids[i];
		for (unsigned v : { mx[0], mx[1], sm[0], sm[1] }) { row += '\t'; row += std::to_string(v); }
		row += '\t'; row += std::to_string(n[0]);
		row += '\t'; row += std::to_string(n[1]);
		row += '\n';
		out << row;
	}
	return true;
}

unsigned sites_covered(const SiteSet &sites, const std::vector<uint64_t> &counts)
{
	unsigned covered = 0;
	for (size_t i = 0; i < sites.ids.size(); ++i) {
		bool hit = false;
		for (int a = 0; a < 2 && !hit; ++a) {
			const auto &side = a == 0 ? sites.ref : sites.var;
			if (i >= side.size()) continue;
			for (int64_t ix : side[i])
				if (ix != SiteSet::kErased && counts[(size_t) ix] > 0) { hit = true; break; }
		}
		covered += hit ? 1u : 0u;
	}
	return covered;
}

std::string info_summary(const SiteSet &sites, const std::vector<uint64_t> &counts, uint64_t total_bases,
		uint64_t total_kmers, uint64_t total_hits)
{
	std::string s;
	s += "Total Bases Considered: " + std::to_string(total_bases) + "\n";
	s += "Total k-mers Considered: " + std::to_string(total_kmers) + "\n";
	s += "Total k-mers Recorded: " + std::to_string(total_hits) + "\n";
	s += "Distinct k-mers in initial set: " + std::to_string(sites.n_distinct()) + "\n";
	s += "Total Sites: " + std::to_string(sites.ref.size()) + "\n";
	s += "Sites Covered by at least one k-mer: " + std::to_string(sites_covered(sites, counts)) + "\n";
	return s;
}

} // namespace ntsm
