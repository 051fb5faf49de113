/*
 * oracle/ref_gpu_binding.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product; the product never links this).
 *
 * INTEGRATION.md section 2 written out and compiled: the reference-side binding a maintainer of JustinChu/ntsm would add so
 * that the reference's OWN FingerPrint class counts on an MI355X through the C ABI of include/ntsm_hip.h.
 *
 * What is the reference's and what is the library's in this program:
 *   reference, unmodified, #included where it lies under /root/reference (nothing is copied into this repo):
 *       FingerPrint()            site loader initCountsHash (src/FingerPrint.hpp:490-564), m_maxCounts (:41-43)
 *       kseq_read                the record reader (vendor/kseq.h:177-219 through gzread)
 *       printOptionalHeader / printCountsMax / printInfoSummary   (:261-349, :389-413)
 *   libntsm_hip.so (this repo's product):
 *       everything insertCount() does per read (:89-103) and the -m check behind it (:476-487)
 *
 * The class keeps its members private; a maintainer would add the few lines below INSIDE the class.  A test harness outside
 * the file cannot, so the header is included with `private` spelled `public` -- after every standard and vendored header it
 * pulls in has already been included normally, so the only code that sees the macro is FingerPrint.hpp, Options.h and
 * KseqHashIterator.hpp themselves.  The file on disk is not touched.
 *
 * The flags are ref_driver.cpp's (the subset of src/ntSeqMatchCount.cpp:75-136 that reaches the hot path); output goes to
 * stdout / stderr exactly like the reference's main (:175-182).  tests/test_reference_binding.py replays every recording of
 * tests/golden/ through this binary on the GPU box and expects the reference's recorded bytes.
 *
 * Built by oracle/Makefile (target refgpu) into oracle/_ref/ref_gpu_ntsmCount, only where /root/reference exists.
 */
#include <cassert>
#include <cfloat>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include <omp.h>
#include <zlib.h>

#include "vendor/tsl/robin_map.h"
#include "vendor/tsl/robin_set.h"

#define private public
#include "src/FingerPrint.hpp"
#undef private

#include "ntsm_hip.h"

namespace {

/* INTEGRATION.md section 2, member for member (m_gpu..., gpuInit, gpuFlush, processSingleRead, gpuFinish); `fp.` is where the
 * stub inside the class would say `this->`. */
struct GpuBinding {
	FingerPrint &fp;
	ntsm_ctx *m_gpu = nullptr;
	std::vector<uint64_t> m_gpuKeys;                 /* m_counts' keys in a fixed order */
	uint8_t *m_stageBases = nullptr;
	uint64_t *m_stageEnds = nullptr;
	uint64_t m_stageCapB = 0, m_stageCapR = 0, m_fill = 0;
	uint32_t m_nReads = 0;

	explicit GpuBinding(FingerPrint &f) : fp(f) {}

	static void die(const char *what, int rc)
	{
		std::cerr << what << ": " << ntsm_strerror(rc) << std::endl;
		exit(1);
	}

	void gpuInit()                                   /* at the end of FingerPrint() */
	{
		/* robin_map iteration = hash order (:466): the library's choice of kernel form does not depend on it */
		for (auto it = fp.m_counts.begin(); it != fp.m_counts.end(); ++it) m_gpuKeys.push_back(it->first);
		/* the default covThresh = DBL_MAX (Options.h:32) leaves an out-of-range double -> uint64 conversion in m_maxCounts
		 * (:41-43; 2^63 on x86-64): a threshold no run reaches = not armed */
		const uint64_t maxHits = fp.m_maxCounts < (1ull << 63) ? fp.m_maxCounts : 0;
		const int rc = ntsm_create(&m_gpu, 0, (int) opt::k, m_gpuKeys.data(), (uint32_t) m_gpuKeys.size(), NTSM_KEYS_HASH64, maxHits);
		if (rc != NTSM_OK) die("no GPU", rc);
		m_armed = maxHits != 0;
		if (const char *b = getenv("NTSM_REF_GPU_BATCH")) {     /* tests: small staging slots, so that batch boundaries fall inside the inputs */
			m_slotBytes = strtoull(b, nullptr, 10);
			const int rb = ntsm_set_batch_capacity(m_gpu, m_slotBytes, m_slotBytes / 64 + 16);
			if (rb != NTSM_OK) die("ntsm_set_batch_capacity", rb);
		}
	}

	void gpuFlush()
	{
		if (!m_stageBases) return;
		const int rc = ntsm_submit_staged(m_gpu, m_fill, m_nReads);
		if (rc != NTSM_OK) die("ntsm_submit_staged", rc);
		m_stageBases = nullptr; m_fill = 0; m_nReads = 0;
		if (m_armed) {
			ntsm_totals t;
			const int rs = ntsm_sync(m_gpu, &t);
			if (rs != NTSM_OK) die("ntsm_sync", rs);
			fp.m_earlyTerm = t.early_stop != 0;
		}
	}

	void processSingleRead(kseq_t *seq)              /* replaces :473-488 */
	{
		if (m_stageBases && (m_fill + seq->seq.l + 1 > m_stageCapB || m_nReads >= m_stageCapR)) gpuFlush();
		if (fp.m_earlyTerm) return;
		if (!m_stageBases) {
			if (seq->seq.l + 1 > m_slotBytes) {          /* a read longer than a staging slot (default 64 MiB): grow both */
				m_slotBytes = 2 * (seq->seq.l + 1);
				const int rc = ntsm_set_batch_capacity(m_gpu, m_slotBytes, 1u << 20);
				if (rc != NTSM_OK) die("ntsm_set_batch_capacity", rc);
			}
			const int rc = ntsm_staging_acquire(m_gpu, &m_stageBases, &m_stageCapB, &m_stageEnds, &m_stageCapR);
			if (rc != NTSM_OK) die("ntsm_staging_acquire", rc);
			m_slotBytes = m_stageCapB;
		}
		memcpy(m_stageBases + m_fill, seq->seq.s, seq->seq.l);
		m_fill += seq->seq.l;
		m_stageBases[m_fill] = 'N';
		m_stageEnds[m_nReads++] = m_fill++;
	}

	void gpuFinish()                                 /* at the end of computeCounts(), before printing */
	{
		gpuFlush();
		ntsm_totals t;
		const int rs = ntsm_sync(m_gpu, &t);
		if (rs != NTSM_OK) die("ntsm_sync", rs);
		fp.m_totalKmers = t.total_kmers; fp.m_totalCounts = t.total_hits; fp.m_totalBases = t.total_bases;
		fp.m_earlyTerm = t.early_stop != 0;
		std::vector<uint64_t> c(m_gpuKeys.size());
		const int rc = ntsm_counts(m_gpu, c.data());
		if (rc != NTSM_OK) die("ntsm_counts", rc);
		for (size_t i = 0; i < c.size(); ++i) fp.m_counts[m_gpuKeys[i]] = c[i];      /* printCountsMax() unchanged */
		ntsm_destroy(m_gpu);
		m_gpu = nullptr;
	}

	/* computeCounts (:46-87) with the binding's processSingleRead: the same file loop, on one thread (the stop of -m is defined
	 * on one ordered stream of reads; INTEGRATION.md section 2 names the lane calls for the `omp parallel for` form) */
	void computeCounts(const std::vector<std::string> &filenames)
	{
		for (unsigned i = 0; i < filenames.size(); ++i) {
			gzFile f = gzopen(filenames[i].c_str(), "r");
			if (f == Z_NULL) {
				std::cerr << "file " << filenames[i] << " cannot be opened" << std::endl;
				exit(1);
			} else if (opt::verbose) {
				std::cerr << "Opening " << filenames[i] << std::endl;
			}
			kseq_t *seq = kseq_init(f);
			int l = kseq_read(seq);
			while (l >= 0 && !fp.m_earlyTerm) {
				processSingleRead(seq);
				l = kseq_read(seq);
			}
			kseq_destroy(seq);
			gzclose(f);
		}
		gpuFinish();
		if (fp.m_earlyTerm) std::cerr << "Reached desired (-m) threshold" << std::endl;
	}

	/* The `#pragma omp parallel for` form of computeCounts (:47) that INTEGRATION.md section 2 describes below its stub: every
	 * OpenMP thread stages into its own producer lane of the ONE shared context (= the reference's shared m_counts with
	 * `#pragma omp atomic`, :94-99).  Only without -m: the stop is defined on one ordered stream of reads. */
	void computeCountsLanes(const std::vector<std::string> &filenames)
	{
#pragma omp parallel
		{
			ntsm_lane *lane = nullptr;
			int rc = ntsm_lane_open(m_gpu, 0, 0, &lane);             /* once per thread */
			if (rc != NTSM_OK) die("ntsm_lane_open", rc);
			uint8_t *bases = nullptr;
			uint64_t *ends = nullptr, capB = 0, capR = 0, fill = 0;
			uint32_t nReads = 0;
#pragma omp for
			for (unsigned i = 0; i < filenames.size(); ++i) {
				gzFile f = gzopen(filenames[i].c_str(), "r");
				if (f == Z_NULL) {
#pragma omp critical (stderr)
					std::cerr << "file " << filenames[i] << " cannot be opened" << std::endl;
					exit(1);
				}
				kseq_t *seq = kseq_init(f);
				while (kseq_read(seq) >= 0) {
					if (bases && (fill + seq->seq.l + 1 > capB || nReads >= capR)) {
						if ((rc = ntsm_lane_submit(lane, fill, nReads)) != NTSM_OK) die("ntsm_lane_submit", rc);
						bases = nullptr;
					}
					if (!bases) {
						if ((rc = ntsm_lane_acquire(lane, &bases, &capB, &ends, &capR)) != NTSM_OK) die("ntsm_lane_acquire", rc);
						fill = 0; nReads = 0;
					}
					memcpy(bases + fill, seq->seq.s, seq->seq.l);
					fill += seq->seq.l;
					bases[fill] = 'N';
					ends[nReads++] = fill++;
				}
				kseq_destroy(seq);
				gzclose(f);
			}
			if (bases && (rc = ntsm_lane_submit(lane, fill, nReads)) != NTSM_OK) die("ntsm_lane_submit", rc);
			if ((rc = ntsm_lane_close(lane)) != NTSM_OK) die("ntsm_lane_close", rc);   /* before gpuFinish */
		}
		gpuFinish();
	}

	bool armed() const { return m_armed; }

private:
	bool m_armed = false;
	uint64_t m_slotBytes = 64ull << 20;              /* ntsm_hip.h: default capacity of a staging slot */
};

} // namespace

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
		std::cerr << "usage: ref_gpu_ntsmCount -s sites.fa [-k K] [-m M] [-t T] [-d] [-o F] reads..." << std::endl;
		return 1;
	}
	double time = omp_get_wtime();
	FingerPrint fp;                                             /* :177 -- the reference's site loader */
	GpuBinding gpu(fp);
	gpu.gpuInit();
	if (opt::threads > 1 && !gpu.armed()) gpu.computeCountsLanes(files);   /* :178 -- counting on the device, `omp parallel for` form */
	else gpu.computeCounts(files);                              /* :178 -- counting on the device */
	fp.printOptionalHeader();                                   /* :179 -- the reference's printing, unchanged */
	fp.printCountsMax();                                        /* :180 */
	std::cerr << fp.printInfoSummary() << std::endl;            /* :181 */
	std::cerr << "Time: " << omp_get_wtime() - time << " s" << std::endl;
	return 0;
}
