/*
 * host_shape.hpp -- thread counts of the ingest pipeline from the CPUs this process is GRANTED, not from the CPUs the
 * machine has (the reference's -t is omp_set_num_threads(N) over files, src/ntSeqMatchCount.cpp:138-141
 * -- it takes what it is told; here -t N is a ceiling).
 *
 * granted_cpus() = min(CPUs in the affinity mask, the cgroup's CPU quota rounded up): the GPU pod of this project reports 256
 * CPUs (hardware_concurrency) and runs under `cpu.max = 1600000 100000` = 16 CPUs' worth of time -- every thread sweep of
 * round 4 peaked at 16 and fell off behind it.  Rounds 1-4 had the numbers that were best THERE as constants in
 * fingerprint.cpp (16 feeders, 16 decoders, 14 early decoders); they are now one row of a small table keyed on the grant.
 * Rows: the 16-CPU row is measured (feeders 8 / 16 / 32 -> 0.25 / 0.12 / 0.15 s for the 12.6 GB FASTQ;
 * decoders 8 / 12 / 16 / 20 / 24 / 32 -> 0.74 / 0.51 / 0.42 / 0.48 / 0.47 / 0.52 s; early decoders 8 ... 16 -> best at 14);
 * beyond 16 the same row applies (more feeders queue up on the HIP submission path: 40 Gbases/s at 32, 25 at 64, against
 * 50 at 16); the smaller rows follow the rule the measured row obeys -- feeders = CPUs, decoders = CPUs - 1 (the stream's
 * in-order producer is a decoding thread too; feeders wait while the text is not there yet, so feeders + decoders + producer
 * <= 2 x CPUs never asks for more than two runnable threads per CPU), early decoders = CPUs - 2 (the start-up has two busy
 * threads of its own).  tests/test_host_cpu.py drives the host library under affinity masks of 2 / 4 / 8 CPUs: identical
 * packed bytes, ingest threads <= 2 x CPUs (+ one coordinator that only waits).
 */
#ifndef NTSM_HOST_SHAPE_HPP
#define NTSM_HOST_SHAPE_HPP
#include <sched.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

namespace ntsm {

/* CPU quota of this process's cgroup in CPUs (0 = none / unknown): cgroup v2 cpu.max along the path of /proc/self/cgroup
 * (the tightest ancestor counts), else cgroup v1 cpu.cfs_quota_us / cpu.cfs_period_us */
inline double cgroup_cpu_quota()
{
	double best = 0;
	auto take = [&](double q) { if (q > 0 && (best == 0 || q < best)) best = q; };
	std::string rel;
	{
		std::ifstream f("/proc/self/cgroup");
		std::string line;
		while (std::getline(f, line))
			if (line.compare(0, 3, "0::") == 0) rel = line.substr(3);
	}
	for (std::string p = rel;; p = p.substr(0, p.rfind('/'))) {            /* v2: own group and every ancestor up to the mount point */
		std::ifstream f("/sys/fs/cgroup" + p + "/cpu.max");
		std::string a, b;
		if (f >> a >> b && a != "max") take(atof(a.c_str()) / std::max(1.0, atof(b.c_str())));
		if (p.empty() || p == "/") break;
	}
	{
		std::ifstream q("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), per("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
		double qv = 0, pv = 0;
		if (q >> qv && per >> pv && qv > 0 && pv > 0) take(qv / pv);
	}
	return best;
}

inline unsigned granted_cpus()
{
	unsigned n = 0;
	cpu_set_t set;
	CPU_ZERO(&set);
	if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned) CPU_COUNT(&set);
	if (n == 0) n = 1;
	const double q = cgroup_cpu_quota();
	if (q > 0) n = std::min<unsigned>(n, (unsigned) std::max(1.0, std::ceil(q - 1e-9)));
	return std::max(1u, n);
}

struct IngestPlan {
	unsigned cpus;              /* the row's key: CPUs granted */
	unsigned feeders;           /* threads that parse text into lanes (one plain FASTQ in blocks, the pieces of an inflated .gz, or one file each) */
	unsigned decoders;          /* decoder pool of one big .gz that is read by itself */
	unsigned early_decoders;    /* the same while the sites still load and the context is being created */
};

/* asked = -t (>= 1): a ceiling for the feeders; cpus = granted_cpus() (a parameter so that tests can walk the table) */
inline IngestPlan ingest_plan(unsigned asked, unsigned cpus)
{
	static const IngestPlan rows[] = {           /* cpus, feeders, decoders, early decoders */
		{ 1, 1, 1, 1 }, { 2, 2, 1, 1 }, { 3, 3, 2, 1 }, { 4, 4, 3, 2 }, { 6, 6, 5, 4 }, { 8, 8, 7, 6 }, { 12, 12, 11, 10 },
		{ 16, 16, 16, 14 },                      /* measured; also every larger grant */
	};
	asked = std::max(1u, asked);
	cpus = std::max(1u, cpus);
	IngestPlan p = rows[0];
	for (const IngestPlan &r : rows) if (r.cpus <= cpus) p = r;
	p.cpus = cpus;
	p.feeders = std::min(p.feeders, asked);
	p.decoders = std::max(1u, std::min(p.decoders, 2 * p.feeders));
	p.early_decoders = std::max(1u, std::min(p.early_decoders, 2 * asked));
	return p;
}

} // namespace ntsm
#endif
