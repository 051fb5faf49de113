/*
 * ntsm_eval_main.cpp -- host mirror of the reference's ntsmEval for its all-to-all path (src/ntSeqMatchEval.cpp:86-349,
 * src/CompareCounts.hpp): same flags, same stdout bytes; the pair loop of CompareCounts::computeScore (:591-624) is one
 * call into the HIP library (include/ntsm_eval_hip.h).  A single input file prints the QC table (computeScoreSingle,
 * :541-585) without touching the GPU; -e FILE writes the merged counts (mergeCounts, :626-674), -o skips the analysis.
 * Not built: the PCA / kd-tree search (-p, -n and its radii); asking for it is an error instead of a silent all-to-all run.  Parity with the reference is unpinned (DESIGN.md
 * section 9): the reference's scoring class cannot be compiled in this image.
 */
#include <getopt.h>

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/ntsm_eval_hip.h"

#define PROGRAM "ntsmEval"

namespace {

struct Opt {                                 /* src/Options.h:44-55 */
	double scoreThresh = 0.5, covSkew = 0.2;
	bool all = false;
	unsigned minCov = 1, threads = 1;
	uint64_t genomeSize = 6200000000ull;
	int verbose = 0, device = 0;
	std::string pca, merge;
	bool onlyMerge = false;
};

struct Counts {                              /* the members of CompareCounts the all-to-all path reads */
	std::vector<std::string> files, locus;
	std::vector<uint32_t> distinct;          /* [site][2] */
	std::vector<uint32_t> counts, sums;      /* [sample][site][2] */
	std::vector<uint64_t> rawTotal, total;
	std::vector<unsigned> kmerSize;
	size_t nSites() const { return locus.size(); }
};

/* src/CompareCounts.hpp:934-940 */
std::pair<unsigned, unsigned> loadPair(std::stringstream &ss, std::string &item)
{
	const unsigned a = (unsigned) std::stoul(item);
	std::getline(ss, item, '\t');
	const unsigned b = (unsigned) std::stoul(item);
	std::getline(ss, item, '\t');
	return std::make_pair(a, b);
}

/* src/CompareCounts.hpp:30-114: the first file fixes loci and distinct counts, then every file fills its row */
void load(Counts &c)
{
	std::unordered_map<std::string, unsigned> index;
	{
		std::ifstream fh(c.files.at(0));
		std::string line;
		while (fh.is_open() && std::getline(fh, line)) {
			if (line.empty() || line[0] == '#') continue;
			std::stringstream ss(line);
			std::string item;
			std::getline(ss, item, '\t');
			index[item] = (unsigned) c.locus.size();
			c.locus.push_back(item);
			for (int s = 0; s < 5; ++s) std::getline(ss, item, '\t');
			const auto d = loadPair(ss, item);
			c.distinct.push_back(d.first);
			c.distinct.push_back(d.second);
		}
	}
	const size_t n = c.files.size(), m = c.nSites();
	c.counts.assign(n * m * 2, 0u);
	c.sums.assign(n * m * 2, 0u);
	c.rawTotal.assign(n, 0);
	c.total.assign(n, 0);
	c.kmerSize.assign(n, 0);
	for (size_t i = 0; i < n; ++i) {
		std::ifstream fh(c.files[i]);
		std::string line;
		while (fh.is_open() && std::getline(fh, line)) {
			if (line.empty()) continue;
			std::stringstream ss(line);
			std::string item;
			std::getline(ss, item, '\t');
			if (line[0] == '#') {
				if (item == "#@TK") { std::getline(ss, item, '\t'); c.rawTotal[i] = std::stoull(item); }
				else if (item == "#@KS") { std::getline(ss, item, '\t'); c.kmerSize[i] = (unsigned) std::stoull(item); }
				continue;
			}
			const unsigned s = index.at(item);                     /* unknown locus: std::out_of_range, as in the reference */
			std::getline(ss, item, '\t');
			const auto cnt = loadPair(ss, item);
			c.counts[(i * m + s) * 2] = cnt.first;
			c.counts[(i * m + s) * 2 + 1] = cnt.second;
			c.total[i] += (uint64_t) cnt.first + cnt.second;
			const auto sm = loadPair(ss, item);
			c.sums[(i * m + s) * 2] = sm.first;
			c.sums[(i * m + s) * 2 + 1] = sm.second;
		}
	}
}

struct Genotype { unsigned hets = 0, homs = 0, miss = 0; double errorRate = 0, cov = 0; };

/* calcHomHetMiss (:742-767), computeErrorRate (:1198-1216), cov (:597-598) */
std::vector<Genotype> summaries(const Counts &c, const Opt &opt)
{
	const size_t m = c.nSites();
	std::vector<Genotype> g(c.files.size());
	for (size_t i = 0; i < g.size(); ++i) {
		const uint32_t *cnt = &c.counts[i * m * 2];
		for (size_t s = 0; s < m; ++s) {
			if (cnt[2 * s] > opt.minCov) { if (cnt[2 * s + 1] > opt.minCov) ++g[i].hets; else ++g[i].homs; }
			else if (cnt[2 * s + 1] > opt.minCov) ++g[i].homs;
			else ++g[i].miss;
		}
		if (c.rawTotal[i] > 0 && c.kmerSize[i] > 0) {
			uint64_t sum = 0, distinct = 0;
			for (size_t s = 0; s < m; ++s) {
				sum += c.sums[(i * m + s) * 2] + c.sums[(i * m + s) * 2 + 1];
				distinct += c.distinct[2 * s] + c.distinct[2 * s + 1];
			}
			const double expected = double(c.rawTotal[i]) * double(distinct) / double(opt.genomeSize);
			g[i].errorRate = 1.0 - std::pow(double(sum) / expected, 1.0 / double(c.kmerSize[i]));
		} else g[i].errorRate = -1.0;
		g[i].cov = double(c.total[i]) / double(m);
	}
	return g;
}

void printHelpDialog()
{
	const Opt d;
	std::cerr << "Usage: " PROGRAM " [FILES...]\n"
	    "Processes sets of counts files and compares their similarity.\n"
	    "If only a single file is provided general QC information returned.\n"
	    "  -t, --threads              Number of threads to run.[1]\n"
	    "  -s, --score_thresh = FLOAT Score threshold [" << std::to_string(d.scoreThresh) << "]\n"
	    "  -a, --all                  Output results of all tests tried, not just those that\n"
	    "                             pass the score threshold.\n"
	    "  -w, --skew = FLOAT         Divides the score by coverage. Formula: (cov1*cov2)^skew\n"
	    "                             Set to zero for no skew.[" << std::to_string(d.covSkew) << "]\n"
	    "  -c, --min_cov = INT        Keep only sites with this coverage and above.[" << std::to_string(d.minCov) << "]\n"
	    "  -g, --genome_size = INT    Diploid genome size for error rate estimation.\n"
	    "                             [" << std::to_string(d.genomeSize) << "]\n"
	    "  -G, --gpu = INT            HIP device [0] (this build only)\n"
	    "  -h, --help                 Display this dialog.\n"
	    "  -v, --verbose              Display verbose output.\n"
	    "  -e, --merge = STR          After analysis merge counts and output to file.\n"
	    "  -o, --only_merge           Do not perform an analysis. Only functions when\n"
	    "                             -e (--merge) option is specified.\n"
	    "Not in this build: -p/-n/-d/-r/-1/-2/-S/-l (PCA search).\n" << std::endl;
	exit(EXIT_SUCCESS);
}

template <typename T> bool parse(const char *s, T &out) { std::stringstream c(s); return bool(c >> out); }

}  // namespace

int main(int argc, char **argv)
{
	Opt opt;
	bool die = false;
	static struct option long_options[] = {
		{ "score_thresh", required_argument, nullptr, 's' }, { "all", no_argument, nullptr, 'a' },
		{ "min_cov", required_argument, nullptr, 'c' }, { "skew", required_argument, nullptr, 'w' },
		{ "genome_size", required_argument, nullptr, 'g' }, { "threads", required_argument, nullptr, 't' },
		{ "merge", required_argument, nullptr, 'e' }, { "only_merge", required_argument, nullptr, 'o' },
		{ "help", no_argument, nullptr, 'h' }, { "pca", required_argument, nullptr, 'p' }, { "norm", required_argument, nullptr, 'n' },
		{ "gpu", required_argument, nullptr, 'G' }, { "verbose", no_argument, nullptr, 'v' }, { nullptr, 0, nullptr, 0 } };
	int ch;
	while ((ch = getopt_long(argc, argv, "t:vhs:c:m:aw:g:p:n:d:r:e:o1:2:S:l:b:G:", long_options, nullptr)) != -1) {
		switch (ch) {
		case 'h': printHelpDialog(); break;
		case 'a': opt.all = true; break;
		case 's': if (!parse(optarg, opt.scoreThresh)) { std::cerr << "Error - Invalid parameter s: " << optarg << std::endl; return 0; } break;
		case 'w': if (!parse(optarg, opt.covSkew)) { std::cerr << "Error - Invalid parameter w: " << optarg << std::endl; return 0; } break;
		case 'c': if (!parse(optarg, opt.minCov)) { std::cerr << "Error - Invalid parameter c: " << optarg << std::endl; return 0; } break;
		case 'g': if (!parse(optarg, opt.genomeSize)) { std::cerr << "Error - Invalid parameter g: " << optarg << std::endl; return 0; } break;
		case 't': if (!parse(optarg, opt.threads)) { std::cerr << "Error - Invalid parameter t: " << optarg << std::endl; return 0; } break;
		case 'G': if (!parse(optarg, opt.device)) { std::cerr << "Error - Invalid parameter G: " << optarg << std::endl; return 0; } break;
		case 'e': opt.merge = optarg; break;
		case 'o': opt.onlyMerge = true; break;
		case 'p': opt.pca = optarg; break;
		case 'v': opt.verbose++; break;
		case '?': die = true; break;
		default: break;                              /* m n d r 1 2 S l b: read by the reference, without effect on this path */
		}
	}
	Counts c;
	while (optind < argc) c.files.emplace_back(argv[optind++]);
	for (const std::string &f : c.files)
		if (!std::ifstream(f).good()) {                  /* the reference asserts (src/ntSeqMatchEval.cpp:286) */
			std::cerr << PROGRAM ": input file " << f << " does not exist" << std::endl;
			abort();
		}
	if (c.files.empty()) { std::cerr << "Error: Need Input File" << std::endl; die = true; }
	if (!opt.pca.empty()) {
		std::cerr << "Error: the PCA search (-p) is not part of this build" << std::endl;
		die = true;
	}
	if (die) { std::cerr << "Try '--help' for more information.\n"; exit(EXIT_FAILURE); }
	const auto t0 = std::chrono::steady_clock::now();
	if (opt.verbose > 0) std::cerr << "Reading count files" << std::endl;
	load(c);
	const std::vector<Genotype> g = summaries(c, opt);
	if (c.files.size() == 1) {                           /* computeScoreSingle, :541-585 */
		if (opt.verbose > 1) std::cerr << "Detected only 1 file, providing only QC information." << std::endl;
		std::cout << "sample\tcov\terrorRate\tmiss\thom\thet" << std::endl;
		std::cout << c.files[0] << "\t" << std::to_string(g[0].cov) << "\t" << std::to_string(g[0].errorRate) << "\t" << std::to_string(g[0].miss)
		          << "\t" << std::to_string(g[0].homs) << "\t" << std::to_string(g[0].hets);
	} else if (opt.onlyMerge) {                          /* src/ntSeqMatchEval.cpp:314-322 */
		if (opt.verbose > 1) std::cerr << "Finished loading files. Now comparing all samples." << std::endl;
		if (opt.merge.empty()) { std::cerr << "(-l) cannot be used without --merge (-e) option." << std::endl; exit(EXIT_FAILURE); }
		std::cerr << " (-l) option detected. Not performing analysis, only merging." << std::endl;
	} else {                                             /* computeScore, :591-624 */
		if (opt.verbose > 1) std::cerr << "Finished loading files. Now comparing all samples." << std::endl;
		std::cerr << "Performing all-to-all score computation.\nSpecify -p (--pca) to enable faster comparisons." << std::endl;
		const uint32_t n = (uint32_t) c.files.size();
		std::vector<ntsm_eval_record> rec((size_t) n * (n - 1) / 2);
		double ms = 0;
		const int rc = ntsm_eval_pairs(opt.device, c.counts.data(), n, (uint32_t) c.nSites(), opt.minCov, rec.data(), &ms);
		if (rc) { std::cerr << PROGRAM ": scoring on the GPU failed (" << rc << "); there is no CPU path" << std::endl; return 3; }
		if (opt.verbose > 1) std::cerr << "pair kernel: " << ms << " ms for " << rec.size() << " pairs" << std::endl;
		std::cout << "sample1\tsample2\tscore\tsame\tdist\trelate\tibs0\tibs2\thomConcord\thet1\thet2\tsharedHet\thom1\thom2\tsharedHom\tn"
		             "\tcov1\tcov2\terrorRate1\terrorRate2\tmiss1\tmiss2\tallHom1\tallHom2\tallHet1\tallHet2";
		std::cout << "\n";
		std::string temp;
		for (uint32_t i = 0; i < n; ++i)
			for (uint32_t j = i + 1; j < n; ++j) {
				const ntsm_eval_record &r = rec[ntsm_eval_pair_index(i, j, n)];
				double score = DBL_MAX;
				if (r.n_valid > 0) {
					score = -2.0 * (r.sum_joint - (r.sum_single1 + r.sum_single2));            /* :1093-1099 */
					score = score / std::pow(g[i].cov * g[j].cov, opt.covSkew);                 /* :1081-1083 */
					score /= double(r.n_valid);
				}
				if (!(opt.all || score < opt.scoreThresh)) continue;
				const double homConcord = (double(r.shared_homs) - 2.0 * double(r.ibs0)) / double(r.homs1 < r.homs2 ? r.homs1 : r.homs2);
				const double relate = (double(r.shared_hets) - 2.0 * double(r.ibs0)) / double(r.hets1 < r.hets2 ? r.hets1 : r.hets2);
				temp.clear();                                /* resultsStr, :843-905 */
				temp += c.files[i]; temp += "\t"; temp += c.files[j]; temp += "\t"; temp += std::to_string(score);
				temp += opt.all ? (score < opt.scoreThresh ? "\t1\t" : "\t0\t") : "\t1\t";
				temp += "-1"; temp += "\t"; temp += std::to_string(relate);
				temp += "\t"; temp += std::to_string(r.ibs0); temp += "\t"; temp += std::to_string(r.ibs2);
				temp += "\t"; temp += std::to_string(homConcord);
				temp += "\t"; temp += std::to_string(r.hets1); temp += "\t"; temp += std::to_string(r.hets2); temp += "\t"; temp += std::to_string(r.shared_hets);
				temp += "\t"; temp += std::to_string(r.homs1); temp += "\t"; temp += std::to_string(r.homs2); temp += "\t"; temp += std::to_string(r.shared_homs);
				temp += "\t"; temp += std::to_string((uint64_t) r.n_valid);
				temp += "\t"; temp += std::to_string(g[i].cov); temp += "\t"; temp += std::to_string(g[j].cov);
				temp += "\t"; temp += std::to_string(g[i].errorRate); temp += "\t"; temp += std::to_string(g[j].errorRate);
				temp += "\t"; temp += std::to_string(g[i].miss); temp += "\t"; temp += std::to_string(g[j].miss);
				temp += "\t"; temp += std::to_string(g[i].homs); temp += "\t"; temp += std::to_string(g[j].homs);
				temp += "\t"; temp += std::to_string(g[i].hets); temp += "\t"; temp += std::to_string(g[j].hets);
				temp += "\n";
				std::cout << temp;
			}
	}
	std::cout.flush();
	if (c.files.size() > 1 && !opt.merge.empty()) {      /* mergeCounts, :626-674 */
		for (size_t i = 0; i < c.kmerSize.size(); ++i)
			for (size_t j = i + 1; j < c.kmerSize.size(); ++j)
				if (c.kmerSize[i] != c.kmerSize[j]) { std::cerr << PROGRAM ": counts files with different k cannot be merged" << std::endl; abort(); }   /* assert, :631-635 */
		std::ofstream out(opt.merge);
		uint64_t tk = 0;
		for (uint64_t v : c.rawTotal) tk += v;
		out << "#@TK\t" << std::to_string(tk) << "\n#@KS\t" << std::to_string(c.kmerSize[0])
		    << "\n#locusID\tcountAT\tcountCG\tsumAT\tsumCG\tdistinctAT\tdistinctCG\n";
		const size_t m = c.nSites();
		for (size_t s = 0; s < m; ++s) {
			unsigned cAT = 0, cCG = 0, sAT = 0, sCG = 0;
			for (size_t j = 0; j < c.files.size(); ++j) {
				cAT += c.counts[(j * m + s) * 2]; cCG += c.counts[(j * m + s) * 2 + 1];
				sAT += c.sums[(j * m + s) * 2]; sCG += c.sums[(j * m + s) * 2 + 1];
			}
			out << c.locus[s] << "\t" << cAT << "\t" << cCG << "\t" << sAT << "\t" << sCG << "\t" << c.distinct[2 * s] << "\t" << c.distinct[2 * s + 1] << "\n";
		}
	}
	std::cerr << "Time: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " s" << std::endl;
	return 0;
}
