/*
 * fingerprint.hpp -- host mirror of the reference's FingerPrint class for the ntsmCount path
 * (src/FingerPrint.hpp): same public calls in the same order as src/ntSeqMatchCount.cpp:177-181,
 * same stdout/stderr bytes; the per-read insertCount loop is replaced by batched submission to
 * the HIP library (include/ntsm_hip.h).
 */
#ifndef NTSM_FINGERPRINT_HPP
#define NTSM_FINGERPRINT_HPP
#include <cstdint>
#include <iosfwd>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/ntsm_hip.h"
#include "site_set.hpp"

namespace ntsm {

struct Options {                           /* the opt:: fields ntsmCount reads (src/Options.h:21-62) */
	int verbose = 0;
	unsigned threads = 1;
	unsigned k = 19;
	std::string snp, summary;
	float siteCovThreshold = 0.75f;
	double covThresh = 1.7976931348623157e308;   /* DBL_MAX: never stop */
	bool dupes = false;
	int device = 0;                        /* HIP device (new; the reference has no device concept) */
	std::vector<int> devices;              /* -g 0,1,...: host threads (-t) are spread round-robin over these devices */
	uint64_t batch_bytes = 64ull << 20;    /* staging capacity per slot */
};

/* One GPU context plus the staging batch being filled for it.  A Feeder is driven by one thread. */
class Feeder {
public:
	Feeder(const Options &opt, const SiteSet &sites, uint64_t max_hits, int device);
	~Feeder();
	Feeder(const Feeder &) = delete;
	Feeder &operator=(const Feeder &) = delete;
	/* Count every record of one file (src/FingerPrint.hpp:49-81); stops early once the -m threshold tripped. */
	void feedFile(const std::string &path);
	void flush();
	bool earlyTerm() const { return m_earlyTerm; }
	ntsm_ctx *ctx() const { return m_ctx; }

private:
	[[noreturn]] void die(int rc, const char *what) const;
	const Options &m_opt;
	ntsm_ctx *m_ctx = nullptr;
	uint64_t m_maxCounts = 0;
	uint8_t *m_bases = nullptr;
	uint64_t *m_readEnd = nullptr;
	uint64_t m_capBytes = 0, m_capReads = 0, m_fill = 0, m_cfgBytes = 0;
	uint32_t m_nReads = 0;
	bool m_earlyTerm = false;
};

class FingerPrint {
public:
	explicit FingerPrint(const Options &opt);            /* FingerPrint(), :35-44 */
	~FingerPrint();
	void computeCounts(const std::vector<std::string> &filenames);   /* :46-87 */
	void printOptionalHeader(std::ostream &out) const;   /* :261-268 */
	void printCountsMax(std::ostream &out) const;        /* :270-311 */
	std::string printInfoSummary();                      /* :313-349 */
	uint64_t maxCounts() const { return m_maxCounts; }

private:
	void fetchResults();

	Options m_opt;
	SiteSet m_sites;
	uint64_t m_maxCounts = 0;
	std::vector<std::unique_ptr<Feeder>> m_feeders;      /* [0] always exists; more with -t N and several files */
	/* results */
	bool m_fetched = false;
	ntsm_totals m_totals {};
	std::vector<uint64_t> m_counts;
};

} // namespace ntsm
#endif
