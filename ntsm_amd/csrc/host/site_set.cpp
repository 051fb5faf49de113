#include "site_set.hpp"

#include <ostream>

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
	/* flat open-addressing index: canonical code -> provisional key index (first-seen order); grows by doubling */
	std::vector<uint64_t> tab_key(1u << 16, ~0ull);
	std::vector<int64_t> tab_idx(1u << 16, 0);
	uint64_t tab_mask = tab_key.size() - 1;
	auto mix = [](uint64_t x) { x ^= x >> 31; x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; return x; };
	auto lookup = [&](uint64_t code) -> int64_t {            /* slot of code, or of the empty slot where it belongs */
		uint64_t i = mix(code) & tab_mask;
		while (tab_key[i] != ~0ull && tab_key[i] != code) i = (i + 1) & tab_mask;
		return (int64_t) i;
	};
	std::vector<uint64_t> prov;                            /* provisional keys in first-seen order */
	std::vector<uint8_t> is_dup;                           /* parallel to prov: seen more than once */
	uint64_t entry = 0;
	for (int64_t l = rd.next(); l >= 0; l = rd.next(), ++entry) {
		const bool is_ref = (entry % 2 == 0);
		std::vector<std::vector<int64_t>> &side = is_ref ? ref : var;
		side.emplace_back();
		std::vector<int64_t> &list = side.back();
		for_each_kmer(rd.seq_data(), (uint64_t) l, k, [&](uint64_t code, uint64_t pos) {
			int64_t slot = lookup(code);
			if (tab_key[(size_t) slot] == code) {
				err << "Warning: " << rd.name() << " of " << (is_ref ? "REF" : "VAR")
				    << " file has a k-mer collision at pos: " << pos << std::endl;
				is_dup[(size_t) tab_idx[(size_t) slot]] = 1;
			} else {
				if (2 * (prov.size() + 1) > tab_key.size()) {          /* keep the load below 0.5 */
					std::vector<uint64_t> nk(tab_key.size() * 2, ~0ull);
					std::vector<int64_t> ni(tab_key.size() * 2, 0);
					const uint64_t nm = nk.size() - 1;
					for (size_t j = 0; j < tab_key.size(); ++j)
						if (tab_key[j] != ~0ull) {
							uint64_t i = mix(tab_key[j]) & nm;
							while (nk[i] != ~0ull) i = (i + 1) & nm;
							nk[i] = tab_key[j];
							ni[i] = tab_idx[j];
						}
					tab_key.swap(nk);
					tab_idx.swap(ni);
					tab_mask = nm;
					slot = lookup(code);
				}
				tab_key[(size_t) slot] = code;
				tab_idx[(size_t) slot] = (int64_t) prov.size();
				list.push_back((int64_t) prov.size());
				prov.push_back(code);
				is_dup.push_back(0);
			}
		});
		if (is_ref) ids.push_back(rd.name());
	}
	/* final key set = first-seen order minus erased duplicates; remap the allele lists */
	std::vector<int64_t> remap(prov.size(), kErased);
	for (size_t i = 0; i < prov.size(); ++i) {
		if (!allow_dupes && is_dup[i]) { ++n_erased; continue; }
		remap[i] = (int64_t) keys.size();
		keys.push_back(prov[i]);
	}
	for (auto *side : { &ref, &var })
		for (auto &list : *side)
			for (auto &ix : list) ix = remap[(size_t) ix];
	return true;
}

} // namespace ntsm
