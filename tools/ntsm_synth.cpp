/*
 * tools/ntsm_synth.cpp -- command-line front end of the synthetic workload generator.
 *   ntsm_synth sites --seed S --n-sites N [--k 19] --out sites.fa[.gz]
 *   ntsm_synth reads --seed S --sites-seed S0 --n-sites N [--k 19] [--len 150] [--r0 0] --n-reads R
 *                    [--p-embed 0.1] [--p-sub 0.01] [--p-n 0.0005] [--qual-model 0|1] --out reads.fq[.gz]
 *   ntsm_synth long  --seed S --sites-seed S0 --n-sites N [--spacing 20000] [--mu 9.6] [--sigma 0.6]
 *                    [--lo 200] [--hi 200000] [--p-sub 0.05] [--r0 0] --n-reads R --out reads.fq[.gz]
 */
#include "../include/ntsm_synth.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

int main(int argc, char **argv)
{
	if (argc < 2) { fprintf(stderr, "usage: ntsm_synth sites|reads|long --key value ...\n"); return 2; }
	std::string mode = argv[1];
	std::map<std::string, std::string> a;
	for (int i = 2; i + 1 < argc; i += 2) a[argv[i]] = argv[i + 1];
	auto S = [&](const char *k, const char *d) { return a.count(k) ? a[k] : std::string(d); };
	auto U = [&](const char *k, const char *d) { return strtoull(S(k, d).c_str(), nullptr, 10); };
	auto D = [&](const char *k, const char *d) { return strtod(S(k, d).c_str(), nullptr); };
	const uint64_t seed = U("--seed", "1");
	const uint32_t n_sites = (uint32_t) U("--n-sites", "96287");
	const unsigned k = (unsigned) U("--k", "19");
	const std::string out = S("--out", "");
	if (out.empty()) { fprintf(stderr, "--out required\n"); return 2; }
	if (mode == "sites") {
		uint64_t nk = 0;
		int rc = ntsm_synth_sites(seed, n_sites, k, nullptr, out.c_str(), &nk);
		fprintf(stderr, "sites=%u distinct_kmers=%llu rc=%d\n", n_sites, (unsigned long long) nk, rc);
		return rc ? 1 : 0;
	}
	std::vector<uint8_t> win((size_t) n_sites * 2 * NTSM_SYNTH_WSTRIDE);
	if (ntsm_synth_sites(U("--sites-seed", "1"), n_sites, k, win.data(), nullptr, nullptr)) return 1;
	const uint64_t r0 = U("--r0", "0"), n_reads = U("--n-reads", "1000");
	if (mode == "reads") {
		ntsm_synth_short p;
		ntsm_synth_short_params(&p, seed, (uint32_t) U("--len", "150"), n_sites, D("--p-embed", "0.1"),
				D("--p-sub", "0.01"), D("--p-n", "0.0005"));
		return ntsm_synth_short_write_fastq_q(&p, win.data(), r0, n_reads, out.c_str(), (unsigned) U("--qual-model", "0")) ? 1 : 0;
	}
	if (mode == "long") {
		ntsm_synth_long p;
		ntsm_synth_long_params(&p, seed, n_sites, (uint32_t) U("--spacing", "20000"), D("--p-sub", "0.05"),
				D("--p-n", "0.0005"));
		uint32_t q[257];
		ntsm_synth_long_qtable(D("--mu", "9.6"), D("--sigma", "0.6"), (uint32_t) U("--lo", "200"),
				(uint32_t) U("--hi", "200000"), q);
		return ntsm_synth_long_write_fastq(&p, win.data(), q, r0, n_reads, out.c_str()) ? 1 : 0;
	}
	fprintf(stderr, "unknown mode %s\n", mode.c_str());
	return 2;
}
