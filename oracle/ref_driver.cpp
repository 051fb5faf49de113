/*
 * oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin driver around the UNMODIFIED reference hot path.  It #includes the reference's
 * header-only FingerPrint class where it lies under /root/reference (nothing is copied into
 * this repo) and replays the five calls that the reference main makes
 * (src/ntSeqMatchCount.cpp:175-182):
 *
 *     FingerPrint fp; fp.computeCounts(files); fp.printOptionalHeader();
 *     fp.printCountsMax(); cerr << fp.printInfoSummary() << endl;
 *
 * Why a driver instead of the reference's own main: src/ntSeqMatchCount.cpp includes the
 * autoconf-generated "config.h" (PACKAGE_NAME / GIT_REVISION for --version); autotools is not
 * in this image and we do not write stand-ins for generated code, so main() itself is treated
 * as unbuildable.  Everything on the counting path (FingerPrint.hpp, KseqHashIterator.hpp,
 * tsl/robin_*, kseq.h, Options.h) builds as-is.
 *
 * Flags accepted here are the subset of the reference CLI that reaches the hot path:
 *   -s FILE  -k INT  -m FLOAT  -t INT  -d  -o FILE  -v      (same meaning as
 *   src/ntSeqMatchCount.cpp:75-136).  Output goes to stdout/stderr exactly like the reference.
 * With NTSM_REF_TIME_SCAN set, the wall time of computeCounts() alone is also printed ("SCAN_SECONDS x").
 * With NTSM_REF_INSERT_MULTIPLIER=M set, computeCounts() is replaced by its own loop written out here -- gzopen, kseq_read,
 * one call of the reference's public FingerPrint::insertCount(seq, len, M) per record (src/FingerPrint.hpp:89, its third
 * parameter) -- so that per-k-mer counts reach and pass 2^32 on a small input: the only way to see what the reference's
 * printCountsMax() does with such counts (`unsigned`, :282/:289) without feeding it four billion reads
 * (tests/golden/make_wrap.py).
 */
#include <cassert>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include <omp.h>

#include "src/FingerPrint.hpp"

int main(int argc, char **argv)
{
	std::vector<std::string> files;
	for (int i = 1; i < argc; ++i) {
		std::string a(argv[i]);
		if (a == "-s" && i + 1 < argc) opt::snp = argv[++i];
		else if (a == "-k" && i + 1 < argc) { std::stringstream c(argv[++i]); c >> opt::k; }
		else if (a == "-m" && i + 1 < argc) { std::stringstream c(argv[++i]); c >> opt::covThresh; }
		else if (a == "-t" && i + 1 < argc) { std::stringstream c(argv[++i]); c >> opt::threads; }
		else if (a == "-o" && i + 1 < argc) opt::summary = argv[++i];
		else if (a == "-d") opt::dupes = true;
		else if (a == "-v") opt::verbose++;
		else files.push_back(a);
	}
	if (opt::threads > 0) omp_set_num_threads(opt::threads);   /* ntSeqMatchCount.cpp:138-141 */
	if (opt::snp.empty() || files.empty()) {
		std::cerr << "usage: ref_ntsmCount -s sites.fa [-k K] [-m M] [-t T] [-d] [-o F] reads..." << std::endl;
		return 1;
	}
	double time = omp_get_wtime();
	FingerPrint fp;                                             /* :177 */
	const double scan0 = omp_get_wtime();
	if (const char *mul = getenv("NTSM_REF_INSERT_MULTIPLIER")) {
		const unsigned m = (unsigned) strtoul(mul, nullptr, 10);
		for (const std::string &f : files) {                    /* the loop of computeCounts (:49-69) around insertCount(.., m) */
			gzFile fh = gzopen(f.c_str(), "r");
			if (fh == Z_NULL) { std::cerr << "file " << f << " cannot be opened" << std::endl; return 1; }
			kseq_t *seq = kseq_init(fh);
			while (kseq_read(seq) >= 0) fp.insertCount(seq->seq.s, seq->seq.l, m);
			kseq_destroy(seq);
			gzclose(fh);
		}
	} else
		fp.computeCounts(files);                                /* :178 */
	if (getenv("NTSM_REF_TIME_SCAN"))                           /* bench.py cpu_baseline: the scan alone, site-table build excluded */
		std::cerr << "SCAN_SECONDS " << omp_get_wtime() - scan0 << std::endl;
	fp.printOptionalHeader();                                   /* :179 */
	fp.printCountsMax();                                        /* :180 */
	std::cerr << fp.printInfoSummary() << std::endl;            /* :181 */
	std::cerr << "Time: " << omp_get_wtime() - time << " s" << std::endl;
	return 0;
}
