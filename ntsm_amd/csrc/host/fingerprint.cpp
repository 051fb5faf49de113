#include "fingerprint.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <mutex>
#include <sstream>
#include <thread>

#include "report.hpp"
#include "seq_reader.hpp"

namespace ntsm {

/* `(m_counts.size() * opt::covThresh) / 2` assigned to a uint64_t (src/FingerPrint.hpp:41-43).
 * Out-of-range doubles (the DBL_MAX default, negatives) are undefined behaviour in the
 * reference; every such value yields a threshold that never trips, which is what is kept. */
static uint64_t threshold_from(double n_distinct, double cov)
{
	if (cov == 0) return 0;
	const double x = (n_distinct * cov) / 2;
	if (!(x == x) || x >= 18446744073709551616.0) return 0;
	if (x < 0) return UINT64_MAX;
	return (uint64_t) x;
}

void Feeder::die(int rc, const char *what) const
{
	std::cerr << "ntsmCount: " << what << ": " << ntsm_strerror(rc);
	if (rc == NTSM_ERR_HIP) std::cerr << " (hipError " << ntsm_last_hip_error() << ")";
	std::cerr << std::endl;
	exit(1);
}

static std::mutex g_stderr;

Feeder::Feeder(const Options &opt, const SiteSet &sites, uint64_t max_hits, int device) : m_opt(opt), m_maxCounts(max_hits)
{
	if (sites.keys.size() > 0xFFFFFFFFull) die(NTSM_ERR_ARG, "too many site k-mers");
	int rc = ntsm_create(&m_ctx, device, (int) m_opt.k, sites.keys.data(), (uint32_t) sites.keys.size(),
			NTSM_KEYS_CANONICAL, m_maxCounts);
	if (rc) die(rc, "cannot create GPU context");
	m_cfgBytes = m_opt.batch_bytes < 4096 ? 4096 : m_opt.batch_bytes;
	rc = ntsm_set_batch_capacity(m_ctx, m_cfgBytes, m_cfgBytes / 64 + 16);
	if (rc) die(rc, "cannot size staging buffers");
}

Feeder::~Feeder() { ntsm_destroy(m_ctx); }

void Feeder::flush()
{
	if (!m_bases) return;
	int rc = ntsm_submit_staged(m_ctx, m_fill, m_nReads);
	if (rc) die(rc, "submit failed");
	m_bases = nullptr;
	m_fill = 0;
	m_nReads = 0;
	if (m_maxCounts != 0) {                              /* armed: submission was synchronous */
		ntsm_totals t;
		rc = ntsm_sync(m_ctx, &t);
		if (rc) die(rc, "sync failed");
		if (t.early_stop) {
			if (m_opt.verbose > 0)
				std::cerr << "max count reached at " << t.reads_consumed << " reads, " << t.total_kmers
				          << " k-mers, " << t.total_hits << " total counts, and " << t.total_bases
				          << " total bases " << std::endl;
			m_earlyTerm = true;
		}
	}
}

void Feeder::feedFile(const std::string &fn)
{
	SeqReader rd;
	if (!rd.open(fn)) {
		std::lock_guard<std::mutex> lk(g_stderr);
		std::cerr << "file " << fn << " cannot be opened" << std::endl;
		exit(1);
	} else if (m_opt.verbose) {
		std::lock_guard<std::mutex> lk(g_stderr);
		std::cerr << "Opening " << fn << std::endl;
	}
	int64_t l = rd.next();
	while (l >= 0 && !m_earlyTerm) {
		const uint64_t len = (uint64_t) l;
		if (m_bases && (m_fill + len + 1 > m_capBytes || m_nReads >= m_capReads)) flush();
		if (m_earlyTerm) break;
		if (!m_bases) {
			if (len + 1 > m_cfgBytes) {                              /* a read longer than a slot: grow both slots */
				m_cfgBytes = (len + 1) + (len + 1) / 2;
				int rc = ntsm_set_batch_capacity(m_ctx, m_cfgBytes, m_cfgBytes / 64 + 16);
				if (rc) die(rc, "cannot grow staging buffers");
			}
			int rc = ntsm_staging_acquire(m_ctx, &m_bases, &m_capBytes, &m_readEnd, &m_capReads);
			if (rc) die(rc, "cannot acquire staging");
		}
		memcpy(m_bases + m_fill, rd.seq_data(), len);
		m_fill += len;
		m_bases[m_fill] = 'N';                               /* read terminator */
		m_readEnd[m_nReads++] = m_fill;
		m_fill += 1;
		l = rd.next();
	}
}

FingerPrint::FingerPrint(const Options &opt) : m_opt(opt)
{
	if (!m_sites.load(m_opt.snp, m_opt.k, m_opt.dupes, std::cerr)) {
		std::cerr << "file " << m_opt.snp << " cannot be opened" << std::endl;   /* :493-499 */
		exit(1);
	}
	if (m_opt.verbose) std::cerr << "Opening " << m_opt.snp << std::endl;
	m_maxCounts = threshold_from((double) m_sites.n_distinct(), m_opt.covThresh);
	if (m_opt.devices.empty()) m_opt.devices.push_back(m_opt.device);
	m_feeders.emplace_back(new Feeder(m_opt, m_sites, m_maxCounts, m_opt.devices[0]));
}

FingerPrint::~FingerPrint() { }

void FingerPrint::computeCounts(const std::vector<std::string> &filenames)
{
	/* The reference runs this loop under `omp parallel for` over files (:47): -t N means N files at a time.
	 * Same here when no -m threshold is armed: N host threads, each with its own GPU context on the same
	 * device, pull files from a shared index; per-k-mer counts are summed afterwards (order cannot matter).
	 * With -m the reference's parallel schedule is a race; the only defined semantics is
	 * argv order on one thread, which is what an armed run always uses. */
	const size_t n_threads = m_maxCounts != 0 ? 1 : std::min<size_t>(std::max(1u, m_opt.threads), filenames.size());
	if (n_threads <= 1) {
		Feeder &f = *m_feeders[0];
		for (const std::string &fn : filenames) f.feedFile(fn);    /* after a stop: still opened, nothing counted (:66) */
		f.flush();
		if (f.earlyTerm()) std::cerr << "Reached desired (-m) threshold" << std::endl;   /* :84-86 */
		return;
	}
	m_feeders.resize(n_threads);
	std::atomic<size_t> next(0);
	std::vector<std::thread> pool;
	for (size_t t = 0; t < n_threads; ++t)
		pool.emplace_back([&, t]() {
			if (!m_feeders[t])                                  /* contexts are built in parallel too, spread over the -g devices */
				m_feeders[t].reset(new Feeder(m_opt, m_sites, 0, m_opt.devices[t % m_opt.devices.size()]));
			Feeder &f = *m_feeders[t];
			for (size_t i = next++; i < filenames.size(); i = next++) f.feedFile(filenames[i]);
			f.flush();
		});
	for (auto &th : pool) th.join();
}

void FingerPrint::fetchResults()
{
	if (m_fetched) return;
	m_counts.assign(m_sites.keys.size(), 0);
	m_totals = ntsm_totals();
	std::vector<uint64_t> part(m_sites.keys.size());
	for (auto &f : m_feeders) {
		ntsm_totals t;
		int rc = ntsm_sync(f->ctx(), &t);
		if (rc == 0) rc = ntsm_counts(f->ctx(), part.data());
		if (rc) {
			std::cerr << "ntsmCount: cannot fetch counts: " << ntsm_strerror(rc) << std::endl;
			exit(1);
		}
		for (size_t i = 0; i < part.size(); ++i) m_counts[i] += part[i];
		m_totals.total_kmers += t.total_kmers;
		m_totals.total_hits += t.total_hits;
		m_totals.total_bases += t.total_bases;
		m_totals.reads_consumed += t.reads_consumed;
		m_totals.early_stop |= t.early_stop;
	}
	m_fetched = true;
}

void FingerPrint::printOptionalHeader(std::ostream &out) const
{
	const_cast<FingerPrint *>(this)->fetchResults();
	print_optional_header(out, m_totals.total_kmers, m_opt.k);
}

void FingerPrint::printCountsMax(std::ostream &out) const
{
	const_cast<FingerPrint *>(this)->fetchResults();
	if (!print_counts_max(out, m_sites, m_counts)) {
		/* the reference's m_counts.at()/vector::at() throws here and the process aborts (exit 134) */
		out.flush();
		std::cerr << "terminate called after throwing an instance of 'std::out_of_range'\n"
		             "  what():  Couldn't find key.\n"
		             "ntsmCount: the sites file has duplicate k-mers (rerun with -d) or an odd number of records"
		          << std::endl;
		abort();
	}
}

std::string FingerPrint::printInfoSummary()
{
	fetchResults();
	const std::string s = info_summary(m_sites, m_counts, m_totals.total_bases, m_totals.total_kmers, m_totals.total_hits);
	if (!m_opt.summary.empty()) {
		std::ofstream fh(m_opt.summary);
		fh << s;
	}
	const double covPer = double(sites_covered(m_sites, m_counts)) / double(m_sites.ref.size());
	if (covPer < m_opt.siteCovThreshold)
		std::cerr << "Warning: site coverage is : " << covPer
		          << "(<75%). Data may be sorted or sparse along the genome. Any PCA projection may be inaccurate."
		          << std::endl;
	return s;
}

} // namespace ntsm
