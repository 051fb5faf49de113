#include "site_set.hpp"

#include <ostream>
#include <unordered_map>
#include <unordered_set>

#include "kmer.hpp"
#include "seq_reader.hpp"

namespace ntsm {

bool SiteSet::load(const std::string &path, unsigned kk, bool allow_dupes, std::ostream &err)
{
	k = kk;
	ids.clear(); ref.clear(); var.clear(); keys.clear();
	n_erased = 0;
	SeqReader rd;
	if (!rd.open(path)) return false;
	std::unordered_map<uint64_t, int64_t> index;           /* canonical code -> provisional key index */
	std::unordered_set<uint64_t> dup;
	std::vector<uint64_t> prov;                            /* provisional keys in first-seen order */
	uint64_t entry = 0;
	for (int64_t l = rd.next(); l >= 0; l = rd.next(), ++entry) {
		const bool is_ref = (entry % 2 == 0);
		std::vector<std::vector<int64_t>> &side = is_ref ? ref : var;
		side.emplace_back();
		std::vector<int64_t> &list = side.back();
		for_each_kmer(rd.seq_data(), (uint64_t) l, k, [&](uint64_t code, uint64_t pos) {
			auto it = index.find(code);
			if (it != index.end()) {
				err << "Warning: " << rd.name() << " of " << (is_ref ? "REF" : "VAR")
				    << " file has a k-mer collision at pos: " << pos << std::endl;
				dup.insert(code);
			} else {
				index.emplace(code, (int64_t) prov.size());
				list.push_back((int64_t) prov.size());
				prov.push_back(code);
			}
		});
		if (is_ref) ids.push_back(rd.name());
	}
	/* final key set = first-seen order minus erased duplicates; remap the allele lists */
	std::vector<int64_t> remap(prov.size(), kErased);
	for (size_t i = 0; i < prov.size(); ++i) {
		if (!allow_dupes && dup.count(prov[i])) { ++n_erased; continue; }
		remap[i] = (int64_t) keys.size();
		keys.push_back(prov[i]);
	}
	for (auto *side : { &ref, &var })
		for (auto &list : *side)
			for (auto &ix : list) ix = remap[(size_t) ix];
	return true;
}

} // namespace ntsm
