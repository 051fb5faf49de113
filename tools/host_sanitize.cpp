/*
 * tools/host_sanitize.cpp -- drives the host-side code (reader, site loader, report formatting) over a list of
 * files so that it can run under AddressSanitizer / UBSan on the CPU build (GPU sanitizers are not available on
 * this pool).  Usage: host_sanitize <sites.fa> <k> <dupes 0|1> <reads...>; prints record/byte totals.
 */
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../ntsm_amd/csrc/host/report.hpp"
#include "../ntsm_amd/csrc/host/seq_reader.hpp"
#include "../ntsm_amd/csrc/host/site_set.hpp"

int main(int argc, char **argv)
{
	if (argc < 4) return 2;
	ntsm::SiteSet sites;
	std::ostringstream warn;
	if (!sites.load(argv[1], (unsigned) atoi(argv[2]), atoi(argv[3]) != 0, warn)) return 3;
	unsigned long long records = 0, bytes = 0;
	for (int i = 4; i < argc; ++i) {
		ntsm::SeqReader rd;
		if (!rd.open(argv[i])) return 4;
		long long l;
		while ((l = rd.next()) >= 0) {
			++records;
			for (long long j = 0; j < l; ++j) bytes += (unsigned char) rd.seq_data()[j] != 0;
		}
	}
	std::vector<uint64_t> counts(sites.keys.size());
	for (size_t i = 0; i < counts.size(); ++i) counts[i] = i * 2654435761u;       /* exercise 32-bit truncation paths */
	std::ostringstream out;
	ntsm::print_optional_header(out, records, sites.k);
	const bool ok = ntsm::print_counts_max(out, sites, counts);
	const std::string summary = ntsm::info_summary(sites, counts, bytes, records, 0);
	printf("sites=%zu keys=%zu erased=%llu records=%llu bytes=%llu printed=%d out=%zu summary=%zu\n", sites.ids.size(),
			sites.keys.size(), (unsigned long long) sites.n_erased, records, bytes, (int) ok, out.str().size(), summary.size());
	return 0;
}
