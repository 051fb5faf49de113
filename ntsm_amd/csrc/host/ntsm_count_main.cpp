/*
 * ntsmCount -- same command line and outputs as the reference tool (src/ntSeqMatchCount.cpp:53-185):
 *   ntsmCount -s sites.fa [-t N] [-m COV] [-o summary] [-d] [-k K] [-v] reads.fq[.gz] ... > counts.txt
 * stdout: "#@TK", "#@KS" headers + one row per site (countAT/countCG = max per-k-mer count of the
 * allele).  stderr: collision warnings, the six summary lines, "Time: .. s Memory: .. kbytes".
 * New, optional: -g/--gpu INT[,INT...] selects the HIP device(s) (default 0): one context per listed device, with -t N the host
 * threads are spread round-robin over them and the per-k-mer counts are merged by one RCCL SUM (ntsm_allreduce).  A device listed
 * twice gets two contexts, merged on the device without RCCL -- `-g 0,0` runs the whole multi-context path on a one-GPU host.
 */
#include <getopt.h>
#include <sched.h>
#include <signal.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "fingerprint.hpp"
#include "gz_stream.hpp"
#include "pack2.hpp"

#define PROGRAM "ntsmCount"

/* src/Util.h:24-29 opens the file with an ifstream to see whether it is readable; access() answers the same question
 * without opening it, which matters for FIFOs and process substitution (an extra open/close can end the writer) */
static bool fexists(const std::string &fn) { return access(fn.c_str(), R_OK) == 0; }

static size_t rss_kb()                                  /* src/Util.h:32-49 */
{
	std::ifstream f("/proc/self/status");
	std::string line;
	while (std::getline(f, line))
		if (line.compare(0, 6, "VmRSS:") == 0) return (size_t) strtoull(line.c_str() + 6, nullptr, 10);
	return 0;
}

static void printVersion()
{
	std::cerr << PROGRAM " (ntsm-mi355x) " << ntsm_version() << "\n"
	          << "MI355X-native implementation of the ntsmCount k-mer counting path\n" << std::endl;
	exit(EXIT_SUCCESS);
}

static void printHelpDialog()
{
	static const char dialog[] =
		"Usage: " PROGRAM " -s [FASTA] [OPTION]... [FILES...]\n"
		"  -t, --threads = INT    Number of threads to run.[1]\n"
		"  -m, --maxCov = INT     k-mer coverage threshold for early\n"
		"                         termination. [inf]\n"
		"  -o, --output = STR     Output for summary file.\n"
		"  -d, --dupes            Allow shared k-mers between sites to\n"
		"                         be counted.\n"
		"  -s, --snp = STR        Interleaved fasta of SNP sites to\n"
		"                         k-merize. [required]\n"
		"  -k, --kmer = INT       k-mer size used. [19]\n"
		"  -g, --gpu = INT[,INT]  HIP device(s) to run on; with -t N the host\n"
		"                         threads are spread over them. [0]\n"
		"  -h, --help             Display this dialog.\n"
		"  -v, --verbose          Display verbose output.\n"
		"      --version          Print version information.\n";
	std::cerr << dialog << std::endl;
	exit(EXIT_SUCCESS);
}

template <class T>
static bool parse(const char *arg, T &out)
{
	std::stringstream convert(arg);
	return (bool) (convert >> out);
}

/* What is left once everything is printed is the kernel's work: unmapping the queues of the HIP runtime, releasing its
 * pinned and device memory, the address space -- 0.10-0.15 s on the MI355X box for this process, all of it inside exit(2),
 * i.e. between `Time:` and the moment the caller's wait() returns (tools/exit_cost.hip: 0.06-0.09 s for a process that
 * only initialised the runtime, +4 ms per stream, +0.1 s per GB of pinned memory).  The reference simply returns from main
 * (src/ntSeqMatchCount.cpp:182-185) and so does this tool BY DEFAULT: when the caller's wait() returns, the process, its
 * HBM, its pinned memory and its /dev/kfd handles are gone.
 *
 * NTSM_FAST_EXIT=1 (opt-in, round 4's default) moves that work out of the caller's sight: the teardown starts when the LAST
 * user of the address space goes, so the last user is made somebody else -- a child that shares the address space
 * (clone(CLONE_VM), its own copy of the descriptor table, so the /dev/kfd and render-node files are released by it as well),
 * closes its copies of stdin/stdout/stderr, waits until this process is gone and then leaves.  What that costs, and why it is
 * not the default: for 0.1-0.2 s after the CLI has "finished" a stray process still owns the context's HBM, pinned memory and
 * queues (a cgroup / Slurm step clean-up or a back-to-back run sees it), and the child is re-parented to whatever reaps
 * orphans -- under a PID 1 that never waits (a container without an init) every run leaves a zombie.  It is refused when this
 * process's parent is PID 1 for that reason.  The child makes raw system calls only (no glibc wrapper: it shares the parent
 * thread's TLS and must not write its errno) on a static stack.  NTSM_CLEAN_EXIT=1: run the destructors as well. */
static inline long raw_syscall3(long nr, long a, long b, long c)
{
	long ret;
	__asm__ volatile("syscall" : "=a"(ret) : "a"(nr), "D"(a), "S"(b), "d"(c) : "rcx", "r11", "memory");
	return ret;
}

static int teardown_child(void *arg)
{
	const long parent = (long) (intptr_t) arg;
	for (long fd = 0; fd < 3; ++fd) raw_syscall3(SYS_close, fd, 0, 0);
	struct timespec ts = { 0, 100000 };
	for (int i = 0; i < 20000 && raw_syscall3(SYS_getppid, 0, 0, 0) == parent; ++i) raw_syscall3(SYS_nanosleep, (long) &ts, 0, 0);   /* <= 2 s */
	raw_syscall3(SYS_exit_group, 0, 0, 0);
	return 0;
}

static bool hand_over_teardown()
{
	if (getppid() == 1) return false;                      /* nobody we can count on to reap the child */
	alignas(64) static char stack[64 << 10];
	return clone(teardown_child, stack + sizeof stack, CLONE_VM | CLONE_UNTRACED | SIGCHLD, (void *) (intptr_t) getpid()) > 0;
}

int main(int argc, char *argv[])
{
	ntsm::Options opt;
	bool die = false;
	int OPT_VERSION = 0;
	/* "dupes" takes an argument in its long form only, as in the reference (:66 vs :75) */
	static struct option long_options[] = {
		{ "threads", required_argument, NULL, 't' }, { "maxCov", required_argument, NULL, 'm' },
		{ "output", required_argument, NULL, 'o' },  { "dupes", required_argument, NULL, 'd' },
		{ "snp", required_argument, NULL, 's' },     { "kmer", required_argument, NULL, 'k' },
		{ "gpu", required_argument, NULL, 'g' },     { "help", no_argument, NULL, 'h' },
		{ "version", no_argument, &OPT_VERSION, 1 }, { "verbose", no_argument, NULL, 'v' },
		{ "debug-fault", required_argument, NULL, 1000 },      /* tests only, not in the help text: KIND:NTH -> ntsm_debug_fail_after */
		{ "debug-kernel", required_argument, NULL, 1001 },     /* tests only: force a kernel variant (ntsm_set_kernel) */
		{ NULL, 0, NULL, 0 } };
	int c, option_index = 0;
	while ((c = getopt_long(argc, argv, "s:t:vhk:m:do:g:", long_options, &option_index)) != -1) {
		switch (c) {
		case 'h': printHelpDialog(); break;
		case 'o': if (!parse(optarg, opt.summary)) { std::cerr << "Error - Invalid parameter o: " << optarg << std::endl; return 0; } break;
		case 'd': opt.dupes = true; break;
		case 's': if (!parse(optarg, opt.snp)) { std::cerr << "Error - Invalid parameter s: " << optarg << std::endl; return 0; } break;
		case 'm': if (!parse(optarg, opt.covThresh)) { std::cerr << "Error - Invalid parameter m: " << optarg << std::endl; return 0; } break;
		case 'k': if (!parse(optarg, opt.k)) { std::cerr << "Error - Invalid parameter k: " << optarg << std::endl; return 0; } break;
		case 't': if (!parse(optarg, opt.threads)) { std::cerr << "Error - Invalid parameter t: " << optarg << std::endl; return 0; } break;
		case 'g': {                                    /* one device or a comma-separated list */
			std::stringstream list(optarg);
			std::string item;
			opt.devices.clear();
			while (std::getline(list, item, ',')) {
				int d;
				if (!parse(item.c_str(), d) || d < 0) { std::cerr << "Error - Invalid parameter g: " << optarg << std::endl; return 0; }
				opt.devices.push_back(d);
			}
			if (opt.devices.empty()) { std::cerr << "Error - Invalid parameter g: " << optarg << std::endl; return 0; }
			opt.device = opt.devices[0];
			break;
		}
		case 'v': opt.verbose++; break;
		case 1000: {                                   /* fault injection for tests/test_gpu_parity.py: the NTH device allocation (KIND 1),
		                                                * host-to-device copy (2) or pinned allocation (3) of this process fails */
			int kind = 0;
			long long nth = 0;
			if (sscanf(optarg, "%d:%lld", &kind, &nth) != 2 || ntsm_debug_fail_after(kind, nth) < 0) {
				std::cerr << "Error - Invalid parameter debug-fault: " << optarg << std::endl;
				return 0;
			}
			break;
		}
		case 1001:                                     /* tools/soak.py: the same inputs through every kernel form, against the oracle */
			if (!parse(optarg, opt.debug_kernel) || opt.debug_kernel < 0) { std::cerr << "Error - Invalid parameter debug-kernel: " << optarg << std::endl; return 0; }
			break;
		case '?': die = true; break;
		}
	}
	if (OPT_VERSION) printVersion();
	if (opt.k > 32) {
		die = true;
		std::cerr << "Error: k cannot be greater than 32" << std::endl;
	}
	if (opt.k == 0) {
		die = true;
		std::cerr << "Error: k must be at least 1" << std::endl;
	}
	if (opt.snp.empty()) {
		die = true;
		std::cerr << "Error: Missing variants (-s) file" << std::endl;
	}
	std::vector<std::string> inputFiles;
	while (optind < argc) {
		inputFiles.emplace_back(argv[optind]);
		if (!fexists(inputFiles.back())) {             /* the reference asserts here (:160) */
			std::cerr << PROGRAM ": input file " << inputFiles.back() << " does not exist" << std::endl;
			abort();
		}
		optind++;
	}
	if (inputFiles.size() == 0) {
		std::cerr << "Error: Need input files" << std::endl;
		die = true;
	}
	if (die) {
		std::cerr << "Try '--help' for more information.\n";
		exit(EXIT_FAILURE);
	}
	if (const char *bb = getenv("NTSM_BATCH_BYTES")) opt.batch_bytes = strtoull(bb, nullptr, 10);   /* staging slot size */
	if (getenv("NTSM_NO_PACK")) opt.pack = false;                                                  /* lanes send raw bytes instead of 2-bit codes + validity */
	if (const char *pi = getenv("NTSM_PACK_IMPL")) ntsm::pack2_force_impl(atoi(pi));               /* A/B of the packer: 0 best the CPU has (AVX-512 VBMI), 1 portable, 2 at most AVX2 */
	if (const char *pb = getenv("NTSM_BLOCK_BYTES")) opt.block_bytes = strtoull(pb, nullptr, 10);   /* block-parallel ingest block size */
	if (const char *gm = getenv("NTSM_GZ_PARALLEL_MIN")) opt.gz_parallel_min_bytes = strtoull(gm, nullptr, 10);   /* smallest gzip file that takes the decoder pool + piece-parallel parse (-t N) */
	if (const char *gd = getenv("NTSM_GZ_DECODERS")) opt.gz_decoders = (unsigned) strtoul(gd, nullptr, 10);     /* decoder threads of that route */
	if (const char *gc = getenv("NTSM_GZ_CHUNK")) ntsm::GzStream::set_parallel_chunk(strtoull(gc, nullptr, 10));  /* compressed bytes per chunk of the parallel gzip decoder */
	const auto t0 = std::chrono::steady_clock::now();
	const bool phases = opt.phase_times = getenv("NTSM_PHASE_TIMES") != nullptr;
	if (phases) {                                          /* time between exec and main: loader + static initialisers of the HIP runtime */
		std::ifstream st("/proc/self/stat");
		std::string tok, all;
		std::getline(st, all);
		std::stringstream ss(all.substr(all.rfind(')') + 2));
		unsigned long long start_ticks = 0;
		for (int i = 3; i <= 22 && (ss >> tok); ++i) if (i == 22) start_ticks = strtoull(tok.c_str(), nullptr, 10);
		struct timespec bt;
		clock_gettime(CLOCK_BOOTTIME, &bt);
		const double now = (double) bt.tv_sec + 1e-9 * (double) bt.tv_nsec;
		std::cerr << "[phase] exec -> main: " << now - (double) start_ticks / (double) sysconf(_SC_CLK_TCK) << " s (clock-tick resolution)" << std::endl;
	}     /* diagnostics: where the wall time goes */
	auto lap = [&](const char *what) {
		if (phases) std::cerr << "[phase] " << what << ": " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " s" << std::endl;
	};
	opt.inputs = inputFiles;
	if (getenv("NTSM_NO_EARLY")) opt.early = false;        /* do not start parsing the first input while the sites load */
	if (const char *ek = getenv("NTSM_EARLY")) opt.early_kinds = !strcmp(ek, "plain") ? 1 : !strcmp(ek, "gz") ? 2 : !strcmp(ek, "all") ? 3 : opt.early_kinds;
	ntsm::FingerPrint fp(opt);
	lap("sites loaded + first GPU context");
	fp.computeCounts(inputFiles);
	lap("reads counted");
	fp.printOptionalHeader(std::cout);
	fp.printCountsMax(std::cout);
	lap("counts printed");
	std::cerr << fp.printInfoSummary() << std::endl;
	const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	std::cerr << "Time: " << secs << " s Memory: " << rss_kb() << " kbytes" << std::endl;
	/* Everything is printed.  Tearing down the context, the lanes, the pinned pool and the HIP runtime in order costs
	 * 0.13 s (tools/e2e_threads.py) and gives nothing back that the kernel driver does not reclaim at exit anyway, so
	 * leave without it; NTSM_CLEAN_EXIT=1 runs the destructors (leak checks).  The kernel's own teardown stays inside
	 * this process's exit unless NTSM_FAST_EXIT=1 asks for the hand-over above (NTSM_SYNC_EXIT=1 overrides it). */
	if (!getenv("NTSM_CLEAN_EXIT")) {
		const char *fast = getenv("NTSM_FAST_EXIT");
		if (fast && fast[0] == '1' && !getenv("NTSM_SYNC_EXIT")) (void) hand_over_teardown();
		std::cout.flush();
		std::cerr.flush();
		fflush(nullptr);
		_exit(0);
	}
	return 0;
}
