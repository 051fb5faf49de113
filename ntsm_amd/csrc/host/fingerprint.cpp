#include "fingerprint.hpp"

#include <algorithm>
#include <set>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <mutex>
#include <sstream>
#include <thread>
#include <sys/stat.h>
#include <unistd.h>

#include "early_ingest.hpp"
#include "host_shape.hpp"
#include "gz_stream.hpp"
#include "pack2.hpp"
#include "parallel_fastq.hpp"
#include "parallel_gz_fastq.hpp"
#include <chrono>
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

static std::mutex g_stderr;

/* A run that cannot go on ends with ONE message and exit status 1 (the reference: `exit(1)` where it cannot open a file,
 * src/FingerPrint.hpp:51-57).  The caller may be one of several feeder threads that all see the same failure (a lost lane
 * batch marks the whole context failed): the first one reports, the others wait for it.  _exit, not exit: the other threads
 * are inside the HIP runtime and nothing has been written to stdout yet (counts are printed only after everything is
 * counted), so there is nothing to flush and no destructor worth racing them for. */
[[noreturn]] static void fatal(const std::string &message)
{
	static std::atomic<bool> dying { false };
	if (dying.exchange(true)) for (;;) std::this_thread::sleep_for(std::chrono::seconds(1));
	{
		std::lock_guard<std::mutex> lk(g_stderr);
		std::cerr << message << std::endl;
	}
	fflush(nullptr);
	_exit(1);
}

void Feeder::die(int rc, const char *what) const
{
	std::ostringstream m;
	m << "ntsmCount: " << what << ": " << ntsm_strerror(rc);
	if (rc == NTSM_ERR_HIP) m << " (hipError " << ntsm_last_hip_error() << ")";
	fatal(m.str());
}

/* Staging slot of a producer lane (-t N): 8 MiB, less when many threads would pin more than 512 MiB in total
 * (pinning costs 0.16 ms/MiB and competes with the table upload for the runtime's lock) */
static uint64_t lane_bytes(unsigned threads)
{
	uint64_t b = 8ull << 20;
	while (b > (1ull << 20) && 2ull * threads * b > (512ull << 20)) b >>= 1;
	return b;
}

Feeder::Feeder(const Options &opt, ntsm_ctx *ctx, uint64_t max_hits, bool lane) : m_opt(opt), m_ctx(ctx), m_useLane(lane), m_maxCounts(max_hits)
{
	m_cfgBytes = m_opt.batch_bytes < 4096 ? 4096 : m_opt.batch_bytes;
	if (m_useLane) {
		/* N producers share the GPU: smaller slots keep the pinned footprint (and its allocation time) flat */
		m_cfgBytes = std::max<uint64_t>(4096, std::min<uint64_t>(m_cfgBytes, lane_bytes(m_opt.threads)));
		m_packed = m_opt.pack;
		openLane();
	} else {
		int rc = ntsm_set_batch_capacity(m_ctx, m_cfgBytes, m_cfgBytes / 64 + 16);
		if (rc) die(rc, "cannot size staging buffers");
	}
}

void Feeder::openLane()
{
	int rc = m_packed ? ntsm_lane_open_packed(m_ctx, m_cfgBytes, &m_lane) : ntsm_lane_open(m_ctx, m_cfgBytes, m_cfgBytes / 64 + 16, &m_lane);
	if (rc) die(rc, "cannot open a producer lane");
}

Feeder::~Feeder() { if (m_lane) ntsm_lane_close(m_lane); }

void Feeder::finish()
{
	flush();
	if (m_lane) {
		int rc = ntsm_lane_close(m_lane);
		m_lane = nullptr;
		if (rc) die(rc, "cannot close a producer lane");
	}
}

void Feeder::flush()
{
	if (m_packed) {
		if (!m_codes || m_nReads == 0) return;
		int rc = ntsm_lane_submit_packed(m_lane, m_pos, m_nReads, m_nBases);
		if (rc) die(rc, "submit failed");
		m_codes = m_valid = nullptr;
		m_pos = m_nBases = 0;
		m_nReads = 0;
		return;
	}
	if (!m_bases || m_nReads == 0) return;
	int rc = m_useLane ? ntsm_lane_submit(m_lane, m_fill, m_nReads) : ntsm_submit_staged(m_ctx, m_fill, m_nReads);
	if (rc) die(rc, "submit failed");
	m_bases = nullptr;
	m_fill = 0;
	m_nReads = 0;
	if (m_maxCounts != 0) {                              /* armed: submission was synchronous */
		ntsm_totals t;
		rc = ntsm_sync(m_ctx, &t);
		if (rc) die(rc, "sync failed");
		if (t.early_stop) {
			/* the reference prints m_totalReads here, a counter it only advances under -vvv and only after a read has been
			 * processed (src/FingerPrint.hpp:70-72): 0 for -v / -vv, the reads before the crossing one for -vvv */
			if (m_opt.verbose > 0)
				std::cerr << "max count reached at " << (m_opt.verbose > 2 ? t.reads_consumed - 1 : 0) << " reads, " << t.total_kmers
				          << " k-mers, " << t.total_hits << " total counts, and " << t.total_bases
				          << " total bases " << std::endl;
			m_earlyTerm = true;
		}
	}
}

void Feeder::feedFile(const std::string &fn, uint64_t offset)
{
	SeqReader rd;
	if (!rd.open(fn, offset)) {
		fatal("file " + fn + " cannot be opened");
	} else if (m_opt.verbose && offset == 0) {
		std::lock_guard<std::mutex> lk(g_stderr);
		std::cerr << "Opening " << fn << std::endl;
	}
	int64_t l = rd.next();
	while (l >= 0 && !m_earlyTerm) {
		feedRead(rd.seq_data(), (uint64_t) l);
		l = rd.next();
		/* -vvv: "Current Total" after every 1,000,000th read (src/FingerPrint.hpp:70-78: m_totalReads only advances at this
		 * verbosity, after the read has been processed and the next one fetched).  The totals have to be those after exactly
		 * that many reads, so the batch is submitted and waited for here -- a debugging verbosity, run on one thread. */
		if (m_opt.verbose > 2 && !m_useLane && (++m_totalReads % 1000000) == 0) progressLine();
	}
}

void Feeder::feedStream(std::unique_ptr<GzStream> gz)
{
	SeqReader rd;
	if (!rd.open_stream(std::move(gz))) return;
	int64_t l = rd.next();
	while (l >= 0 && !m_earlyTerm) {
		feedRead(rd.seq_data(), (uint64_t) l);
		l = rd.next();
	}
}

void Feeder::submitChunk(const PackedChunk &c)
{
	if (!m_packed || c.n_reads == 0) return;
	flush();                                                   /* own staging first: the slot must be free */
	if (m_codes) {                                             /* held but empty (after discard()): hand it back */
		int rc = ntsm_lane_submit_packed(m_lane, 0, 0, 0);
		if (rc) die(rc, "cannot return an empty staging slot");
		m_codes = m_valid = nullptr;
	}
	const uint64_t need = (c.pos + 31) & ~31ull;               /* pack2 writes whole groups of 32 positions */
	if (need > (m_cfgBytes & ~31ull)) {                        /* a chunk grown for a very long read: grow both slots */
		m_cfgBytes = need + need / 2;
		int rc = ntsm_lane_close(m_lane);
		m_lane = nullptr;
		if (rc) die(rc, "cannot grow staging buffers");
		openLane();
	}
	int rc = ntsm_lane_acquire_packed(m_lane, &m_codes, &m_valid, &m_capPos);
	if (rc) die(rc, "cannot acquire staging");
	if (m_capPos < need) die(NTSM_ERR_ARG, "staging slot smaller than an early chunk");
	memcpy(m_codes, c.codes, need / 4);
	memcpy(m_valid, c.valid, need / 8);
	rc = ntsm_lane_submit_packed(m_lane, c.pos, c.n_reads, c.n_bases);
	if (rc) die(rc, "submit failed");
	m_codes = m_valid = nullptr;
	m_pos = m_nBases = 0;
	m_nReads = 0;
}

void Feeder::progressLine()
{
	flush();                                               /* may trip the -m threshold: the line is printed all the same, like the reference's */
	ntsm_totals t;
	int rc = ntsm_sync(m_ctx, &t);
	if (rc) die(rc, "sync failed");
	std::cerr << "Current Total: " << m_totalReads << " reads, " << t.total_kmers << " k-mers, " << t.total_hits
	          << " total counts, and " << t.total_bases << " total bases " << std::endl;
}

/* one read into a packed lane: the same slot logic as below, in positions */
void Feeder::feedPacked(const char *seq, uint64_t len)
{
	if (m_codes && packedExtent(len) > m_capPos) flush();
	if (m_codes && m_nReads == 0 && packedExtent(len) > m_capPos) {   /* held but empty (after discard()) and too small: hand it back */
		int rc = ntsm_lane_submit_packed(m_lane, 0, 0, 0);
		if (rc) die(rc, "cannot return an empty staging slot");
		m_codes = m_valid = nullptr;
	}
	if (!m_codes) {
		if (len + 64 > (m_cfgBytes & ~31ull)) {                          /* a read longer than a slot: grow both slots */
			m_cfgBytes = (len + 64) + (len + 64) / 2;
			int rc = ntsm_lane_close(m_lane);
			m_lane = nullptr;
			if (rc) die(rc, "cannot grow staging buffers");
			openLane();
		}
		int rc = ntsm_lane_acquire_packed(m_lane, &m_codes, &m_valid, &m_capPos);
		if (rc) die(rc, "cannot acquire staging");
		m_pos = m_nBases = 0;
	}
	m_pos = pack2_append(m_codes, m_valid, m_pos, seq, len);
	m_nBases += len;
	++m_nReads;
}

void Feeder::feedRead(const char *seq, uint64_t len)
{
	if (m_packed) { feedPacked(seq, len); return; }
	if (m_bases && (m_fill + len + 1 > m_capBytes || m_nReads >= m_capReads)) flush();
	if (m_earlyTerm) return;
	if (m_bases && m_nReads == 0 && len + 1 > m_capBytes) {
		/* an acquired but empty slot (after discard(): flush() has nothing to submit and keeps it) that is too small for
		 * this read: hand it back empty so that the grow path below runs instead of writing past its end */
		int rc = m_useLane ? ntsm_lane_submit(m_lane, 0, 0) : ntsm_submit_staged(m_ctx, 0, 0);
		if (rc) die(rc, "cannot return an empty staging slot");
		m_bases = nullptr;
		m_fill = 0;
	}
	if (!m_bases) {
		if (len + 1 > m_cfgBytes) {                              /* a read longer than a slot: grow both slots */
			m_cfgBytes = (len + 1) + (len + 1) / 2;
			if (m_useLane) {
				int rc = ntsm_lane_close(m_lane);
				m_lane = nullptr;
				if (rc) die(rc, "cannot grow staging buffers");
				openLane();
			} else {
				int rc = ntsm_set_batch_capacity(m_ctx, m_cfgBytes, m_cfgBytes / 64 + 16);
				if (rc) die(rc, "cannot grow staging buffers");
			}
		}
		int rc = m_useLane ? ntsm_lane_acquire(m_lane, &m_bases, &m_capBytes, &m_readEnd, &m_capReads)
		                   : ntsm_staging_acquire(m_ctx, &m_bases, &m_capBytes, &m_readEnd, &m_capReads);
		if (rc) die(rc, "cannot acquire staging");
	}
	memcpy(m_bases + m_fill, seq, len);
	m_fill += len;
	m_bases[m_fill] = 'N';                               /* read terminator */
	m_readEnd[m_nReads++] = m_fill;
	m_fill += 1;
}

namespace {
/* a regular file of at least min_bytes that starts with the gzip magic: gets the decoder pool (plain or BGZF) */
bool big_gzip_input(const std::string &fn, uint64_t min_bytes)
{
	struct stat st;
	return stat(fn.c_str(), &st) == 0 && S_ISREG(st.st_mode) && (uint64_t) st.st_size >= min_bytes && GzStream::is_gzip(fn);
}
/* from this many big .gz inputs on they are read side by side, one reader per file, instead of one after the other with the whole pool */
size_t side_by_side_from(unsigned threads) { return std::max<size_t>(3, threads / 4); }
} // namespace

FingerPrint::FingerPrint(const Options &opt) : m_opt(opt)
{
	/* -t N is a ceiling: the threads that parse follow the CPUs this process is granted (affinity mask, cgroup quota), and the
	 * decoder pools follow them (host_shape.hpp); counting into one table does not depend on the number */
	m_plan = ingest_plan(std::max(1u, m_opt.threads), granted_cpus());
	if (m_opt.threads > 1) m_opt.threads = std::max(2u, std::min(m_opt.threads, std::max(m_plan.feeders, 2u)));   /* 2 at least: keeps the lane path (and its tests) on a 1-CPU grant */
	if (m_opt.devices.empty()) m_opt.devices.push_back(m_opt.device);
	m_ctxDevice = m_opt.devices;                          /* one context per LISTED device: `-g 0,0` = two contexts on device 0, merged on the device (ntsm_allreduce) */
	/* Side threads, one per device, prepare everything that does not depend on the sites while this thread parses
	 * them: runtime + device context, the three streams of a context, the pinned staging pool (first device),
	 * the two streams its lanes share.  They are joined when the first batch is about to be staged (computeCounts). */
	const auto tc0 = std::chrono::steady_clock::now();
	{
		const bool maybe_armed = m_opt.covThresh != 0 && m_opt.covThresh < 1e300;
		const bool lanes = m_opt.threads > 1 && !maybe_armed && m_opt.verbose <= 2;
		const uint64_t slot = std::max<uint64_t>(4096, m_opt.batch_bytes);
		const uint64_t lane_slot = std::min<uint64_t>(slot, lane_bytes(m_opt.threads));
		const uint64_t pool_bytes = lanes ? (uint64_t) m_opt.threads * 2 * ((m_opt.pack ? lane_slot * 3 / 8 : lane_slot) + 8192)   /* packed lanes pin 3/8 byte per position */
		                                  : 2 * (slot + 8192 + (slot / 64 + 16) * 8 + 8192);
		const int lanes_per_dev = lanes ? (int) ((m_opt.threads + m_ctxDevice.size() - 1) / m_ctxDevice.size()) : 0;
		for (size_t i = 0; i < m_ctxDevice.size(); ++i) {
			const int d = m_ctxDevice[i];
			const bool first = i == 0;
			m_prep.emplace_back([d, first, pool_bytes, lanes_per_dev]() {
				if (ntsm_warmup(d, lanes_per_dev ? 6 : 4) != NTSM_OK) return;   /* a context's 4 streams (two slots, resident, copy) + its 2 lane streams; ntsm_create reports failures */
				if (first) (void) ntsm_staging_pool(pool_bytes);
			});
		}
	}
	/* the first input file starts being parsed now, into ordinary memory (early_ingest.hpp): -t N, no -m, no -vvv */
	{
		const bool maybe_armed = m_opt.covThresh != 0 && m_opt.covThresh < 1e300;
		/* several big .gz inputs are read side by side (computeCounts), the first one included: measured with 8 / 4 files of
		 * 4e7 reads in total, -t 16: 0.97 / 1.03 s against 1.07 / 1.35 s with the first file taken early and alone */
		size_t n_big_gz = 0;
		for (const std::string &fn : m_opt.inputs) n_big_gz += big_gzip_input(fn, m_opt.gz_parallel_min_bytes) ? 1 : 0;
		const bool side_by_side = !getenv("NTSM_ZLIB_ONLY") && n_big_gz >= side_by_side_from(m_opt.threads);
		if (m_opt.early && !side_by_side && m_opt.pack && m_opt.threads > 1 && !maybe_armed && m_opt.verbose <= 2 && !m_opt.inputs.empty()) {
			const unsigned n_par = std::min(m_opt.threads, std::max(1u, m_plan.feeders));
			/* fewer than later, the start-up has threads of its own -- but the stream keeps these decoders to its end, also behind
			 * the hand-over: 8 / 10 / 12 / 14 / 16 of them take the 12.6 GB file through in 1.29 / 1.14 / 1.00 / 0.91 / 0.94 s
			 * (medians of five, one box, interleaved: profiles/r04_gz3/decoders_ab.txt) */
			const unsigned n_dec = m_opt.gz_decoders ? m_opt.gz_decoders : m_plan.early_decoders;
			const uint64_t chunk_pos = std::max<uint64_t>(4096, std::min<uint64_t>(std::max<uint64_t>(4096, m_opt.batch_bytes), lane_bytes(m_opt.threads))) & ~31ull;
			/* 1.5 GiB of packed reads at most (4 Gbases): a gzip stream is handed over when the context is
			 * there, so the chunks only ever hold what was parsed during the start-up -- 0.6 GB for the 12.6 GB file at 12 GB/s
			 * of text and 0.25 s; a slower start-up makes the parsers wait, not the host swap */
			const size_t max_chunks = (size_t) std::max<uint64_t>(4 * n_par, (3ull << 29) / (chunk_pos * 3 / 8 + 1));
			m_early.reset(new EarlyIngest(m_opt.inputs[0], n_par, n_dec, std::min<uint64_t>(m_opt.block_bytes, 2 * lane_bytes(m_opt.threads)),
			                              m_opt.gz_parallel_min_bytes, chunk_pos, max_chunks, m_opt.early_kinds));
			if (!m_early->taken()) m_early.reset();
		}
	}
	const bool loaded = m_sites.load(m_opt.snp, m_opt.k, m_opt.dupes, std::cerr);
	const auto tc1 = std::chrono::steady_clock::now();
	if (!loaded) {
		std::cerr << "file " << m_opt.snp << " cannot be opened" << std::endl;   /* :493-499 */
		joinPrep();                                            /* never exit() under a thread that is inside the HIP runtime */
		exit(1);
	}
	if (m_opt.verbose) std::cerr << "Opening " << m_opt.snp << std::endl;
	m_maxCounts = threshold_from((double) m_sites.n_distinct(), m_opt.covThresh);
	if (m_sites.keys.size() > 0xFFFFFFFFull) {
		std::cerr << "ntsmCount: too many site k-mers" << std::endl;
		joinPrep();
		exit(1);
	}
	/* one context per listed device; with -m everything runs on the first one */
	if (m_maxCounts != 0) m_ctxDevice.resize(1);
	m_ctx.assign(m_ctxDevice.size(), nullptr);
	std::vector<int> rcs(m_ctxDevice.size(), 0);
	std::vector<std::thread> mk;
	for (size_t i = 0; i < m_ctxDevice.size(); ++i)
		mk.emplace_back([&, i]() {
			rcs[i] = ntsm_create(&m_ctx[i], m_ctxDevice[i], (int) m_opt.k, m_sites.keys.data(), (uint32_t) m_sites.keys.size(),
					NTSM_KEYS_CANONICAL, m_maxCounts);
		});
	for (auto &t : mk) t.join();
	/* several devices end with one RCCL SUM (fetchResults): bind the library now, so that a host without it hears
	 * about it before the work, not after (the host-side sum of fetchResults takes over in that case) */
	const size_t n_distinct_devices = std::set<int>(m_ctxDevice.begin(), m_ctxDevice.end()).size();   /* contexts on ONE device are merged without RCCL */
	if (n_distinct_devices > 1 && ntsm_rccl_probe() != NTSM_OK)
		std::cerr << "ntsmCount: warning: RCCL could not be loaded; the devices' counts will be summed on the host" << std::endl;
	if (m_opt.phase_times)
		std::cerr << "[phase] sites parsed " << std::chrono::duration<double>(tc1 - tc0).count() << " s, contexts (tables + upload) "
		          << std::chrono::duration<double>(std::chrono::steady_clock::now() - tc1).count() << " s" << std::endl;
	for (size_t i = 0; i < rcs.size(); ++i)
		if (rcs[i]) {
			std::cerr << "ntsmCount: cannot create GPU context: " << ntsm_strerror(rcs[i]);
			if (rcs[i] == NTSM_ERR_HIP) std::cerr << " (hipError " << ntsm_last_hip_error() << ")";
			std::cerr << std::endl;
			joinPrep();
			exit(1);
		}
	if (m_opt.debug_kernel >= 0)
		for (auto *ctx : m_ctx) {
			const int rc = ntsm_set_kernel(ctx, m_opt.debug_kernel);
			if (rc) {
				std::cerr << "ntsmCount: --debug-kernel " << m_opt.debug_kernel << ": " << ntsm_strerror(rc) << std::endl;
				joinPrep();
				exit(1);
			}
		}
}

FingerPrint::~FingerPrint()
{
	joinPrep();
	for (auto &t : m_retire) if (t.joinable()) t.join();
	m_main.reset();
	m_lanes.clear();
	for (ntsm_ctx *c : m_ctx) ntsm_destroy(c);
}

void FingerPrint::joinPrep()
{
	for (auto &t : m_prep) if (t.joinable()) t.join();
	m_prep.clear();
}

Feeder &FingerPrint::feederFor(size_t t)
{
	if (!m_lanes[t]) {
		m_lanes[t].reset(new Feeder(m_opt, m_ctx[t % m_ctx.size()], 0, true));   /* threads round-robin over the contexts */
	}
	return *m_lanes[t];
}

void FingerPrint::closeLanes()
{
	for (auto &f : m_lanes) if (f) f->finish();
	m_lanes.clear();
}

void FingerPrint::computeCounts(const std::vector<std::string> &filenames)
{
	/* The reference runs this loop under `omp parallel for` over files (:47) with ONE shared m_counts and atomic
	 * increments: -t N means N files at a time.  Same here when no -m threshold is armed: N host threads, each
	 * with its own producer lane (pinned staging + stream) of the SAME GPU context, pull work from a shared
	 * index; the counts meet in the context's table (one context per device with -g a,b: summed at the end).
	 * With -m the reference's parallel schedule is a race; the only defined semantics is
	 * argv order on one thread, which is what an armed run always uses. */
	/* -vvv prints running totals at exact read counts (Feeder::progressLine): one ordered stream as well */
	const size_t want = m_maxCounts != 0 || m_opt.verbose > 2 ? 1 : std::max(1u, m_opt.threads);
	/* BGZF (bgzip) input is inflated block-parallel: share the -t threads among the files that are read at once */
	GzStream::set_decoder_threads((unsigned) std::max<size_t>(1, m_opt.threads / std::max<size_t>(1, std::min(want, filenames.size()))));
	if (!m_prep.empty()) {
		const auto tj = std::chrono::steady_clock::now();
		joinPrep();
		if (m_opt.phase_times) std::cerr << "[phase] waited " << std::chrono::duration<double>(std::chrono::steady_clock::now() - tj).count() << " s for streams + pinned pool" << std::endl;
	}
	if (want <= 1) {
		if (!m_main) m_main.reset(new Feeder(m_opt, m_ctx[0], m_maxCounts, false));
		Feeder &f = *m_main;
		for (const std::string &fn : filenames) f.feedFile(fn);    /* after a stop: still opened, nothing counted (:66) */
		f.flush();
		if (f.earlyTerm()) std::cerr << "Reached desired (-m) threshold" << std::endl;   /* :84-86 */
		return;
	}
	m_lanes.resize(want);
	std::vector<std::string> todo(filenames);
	if (m_early && !todo.empty() && todo[0] == m_opt.inputs[0]) {
		drainEarly();                                            /* the first file has been in the works since the process started */
		todo.erase(todo.begin());
	}
	retireLater(std::move(m_early));
	const std::vector<std::string> &files_left = todo;
	/* Big plain FASTQ files are cut into blocks and parsed by all threads (parallel_fastq.hpp); files that are not
	 * eligible (gzip, FASTA, wrapped or CR lines, small) are taken whole, one thread per file. */
	std::vector<std::string> rest;
	for (const std::string &fn : files_left) {
		ParallelFastq pf;
		/* a block's sequences + terminators (at most half its bytes: a record is header + SEQ + '+' line + QUAL) must
		 * fit one lane slot, so that no thread waits for its predecessor in the middle of a block */
		if (!pf.open(fn, std::min<uint64_t>(m_opt.block_bytes, 2 * lane_bytes(m_opt.threads)))) { rest.push_back(fn); continue; }
		const auto tp0 = std::chrono::steady_clock::now();
		if (m_opt.verbose) std::cerr << "Opening " << fn << "\n" << "block-parallel: " << pf.n_blocks() << " blocks, " << std::min<size_t>(want, m_plan.feeders) << " threads" << std::endl;
		std::vector<Feeder *> sinks;
		/* One plain FASTQ is parsed by at most 16 threads however many -t asks for (host_shape.hpp's largest row): measured
		 * on a 256-thread host, 16 feeders parse + count at 50 Gbases/s, 32 at 40, 64 at 25 (they queue up on the runtime's
		 * submission path and on the memory of the socket that holds the page cache); the result does not depend on the number. */
		const size_t n_par = std::min<size_t>(want, std::max(1u, m_plan.feeders));
		{
			std::vector<std::thread> mk;                         /* lanes (pinned staging) are allocated in parallel */
			for (size_t t = 0; t < n_par; ++t) mk.emplace_back([this, t]() { (void) feederFor(t); });
			for (auto &th : mk) th.join();
		}
		for (size_t t = 0; t < n_par; ++t) sinks.push_back(&feederFor(t));
		const auto tp1 = std::chrono::steady_clock::now();
		const ParallelFastq::Result r = pf.run(sinks);
		const auto tp2 = std::chrono::steady_clock::now();
		if (!r.complete) {                                      /* the rest of the file is not plain 4-line FASTQ */
			if (m_opt.verbose) std::cerr << "block-parallel: sequential from byte " << r.resume << std::endl;
			feederFor(0).feedFile(fn, r.resume);
			feederFor(0).flush();
		}
		if (m_opt.phase_times)
			std::cerr << "[phase] " << fn << ": lanes " << std::chrono::duration<double>(tp1 - tp0).count() << " s, parse+count "
			          << std::chrono::duration<double>(tp2 - tp1).count() << " s (" << r.records << " records in parallel)" << std::endl;
	}
	/* A big gzip file (plain or BGZF) is inflated by a pool of decoder threads (gz_stream.hpp) and the text is parsed piece-
	 * parallel by the same feeders (parallel_gz_fastq.hpp); small ones and NTSM_ZLIB_ONLY stay one thread per file. */
	if (!getenv("NTSM_ZLIB_ONLY")) {
		std::vector<std::string> small;
		auto is_big_gz = [&](const std::string &fn) { return big_gzip_input(fn, m_opt.gz_parallel_min_bytes); };
		/* Many big files (a lane's worth of .fq.gz: four or more left with -t 16) are better off side by side, one reader per file
		 * with the -t threads' worth of decoders shared out among them (the route below), than one after the other with the
		 * whole pool each: every file pays the pool's start and its drain, and the in-order share of the decoding is the
		 * cheapest (no block search, no marker pass).  Measured, 8 x 262 MB of .gz, -t 16: 1.49 s one after the other. */
		size_t n_big = 0;
		for (const std::string &fn : rest) n_big += is_big_gz(fn) ? 1 : 0;
		const bool side_by_side = n_big >= side_by_side_from((unsigned) want);
		for (const std::string &fn : rest) {
			if (side_by_side || !is_big_gz(fn)) { small.push_back(fn); continue; }
			/* decoder threads: as many as the grant has CPUs, at most twice the feeders (host_shape.hpp) -- measured under a 16-CPU
			 * quota on a 2 x 64-core host (6.3 GB of text, 1 MiB chunks, 16 feeders): 8 / 12 / 16 / 20 / 24 / 32 decoders inflate +
			 * parse + count in 0.74 / 0.51 / 0.42 / 0.48 / 0.47 / 0.52 s */
			const size_t n_feed = std::min<size_t>(want, std::max(1u, m_plan.feeders));
			unsigned n_dec = (unsigned) std::max<size_t>(1, std::min<size_t>(m_plan.decoders, 2 * n_feed));
			if (m_opt.gz_decoders) n_dec = m_opt.gz_decoders;
			GzStream::set_decoder_threads(n_dec);
			std::unique_ptr<GzStream> gz(new GzStream());
			if (!gz->open(fn)) { small.push_back(fn); continue; }
			if (m_opt.verbose) std::cerr << "Opening " << fn << "\n" << "parallel gzip: " << n_dec << " decoder threads, " << n_feed << " parsing threads" << std::endl;
			countGzStream(std::move(gz), fn, 0, n_feed);
		}
		rest.swap(small);
		GzStream::set_decoder_threads((unsigned) std::max<size_t>(1, m_opt.threads / std::max<size_t>(1, std::min(want, std::max<size_t>(1, rest.size())))));
	}
	if (!rest.empty()) {
		const size_t n_threads = std::min<size_t>(want, rest.size());
		std::atomic<size_t> next(0);
		std::vector<std::thread> pool;
		for (size_t t = 0; t < n_threads; ++t)
			pool.emplace_back([&, t]() {
				Feeder &f = feederFor(t);
				for (size_t i = next++; i < rest.size(); i = next++) f.feedFile(rest[i]);
				f.flush();
			});
		for (auto &th : pool) th.join();
	}
	const auto tc0 = std::chrono::steady_clock::now();
	closeLanes();
	if (m_opt.phase_times) std::cerr << "[phase] lanes closed in " << std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count() << " s" << std::endl;
}

/* An open gzip stream nobody has read from yet (or that stands at a record boundary): its pieces are parsed by the feeders
 * in parallel (parallel_gz_fastq.hpp), what they cannot take by the sequential reader on the same object. */
void FingerPrint::countGzStream(std::unique_ptr<GzStream> gz, const std::string &fn, size_t first, size_t n_par)
{
	const auto tp0 = std::chrono::steady_clock::now();
	{
		std::vector<std::thread> mk;
		for (size_t t = first; t < first + n_par; ++t) mk.emplace_back([this, t]() { (void) feederFor(t); });
		for (auto &th : mk) th.join();
	}
	std::vector<Feeder *> sinks;
	for (size_t t = first; t < first + n_par; ++t) sinks.push_back(&feederFor(t));
	const auto tp1 = std::chrono::steady_clock::now();
	ParallelGzFastq pg(gz.get());
	const ParallelGzFastq::Result r = pg.run(sinks);
	const auto tp2 = std::chrono::steady_clock::now();
	if (!r.complete) {                                      /* what is left is not plain 4-line FASTQ (or the last record has no newline) */
		if (m_opt.verbose) std::cerr << "parallel gzip: sequential after " << r.records << " records" << std::endl;
		feederFor(first).feedStream(std::move(gz));
		feederFor(first).flush();
	}
	if (m_opt.phase_times) {
		uint64_t ps[2];
		GzStream::last_parallel_stats(ps);
		std::cerr << "[phase] " << fn << ": lanes " << std::chrono::duration<double>(tp1 - tp0).count() << " s, inflate+parse+count "
		          << std::chrono::duration<double>(tp2 - tp1).count() << " s (" << r.records << " records in " << r.pieces << " pieces in parallel, "
		          << ps[0] << " chunks spliced, " << ps[1] << " dropped), rest "
		          << std::chrono::duration<double>(std::chrono::steady_clock::now() - tp2).count() << " s" << std::endl;
	}
	retireLater(std::move(gz));                                 /* null if the sequential reader took it over (and closed it) */
}

void FingerPrint::drainEarly()
{
	const auto t0 = std::chrono::steady_clock::now();
	const size_t n_par = std::min<size_t>(m_lanes.size(), std::max(1u, m_plan.feeders));
	if (m_opt.verbose) std::cerr << "Opening " << m_opt.inputs[0] << "\n" << "early ingest (" << m_early->how() << "): parsed while the sites were loading" << std::endl;
	{
		std::vector<std::thread> mk;                         /* lanes (pinned staging) are allocated in parallel */
		for (size_t t = 0; t < n_par; ++t) mk.emplace_back([this, t]() { (void) feederFor(t); });
		for (auto &th : mk) th.join();
	}
	const auto t1 = std::chrono::steady_clock::now();
	/* The consumers are there.  A gzip stream stops being parsed into chunks at the next record boundary: a quarter of the
	 * feeders submit the chunks that are waiting (a copy into a lane each) while the others parse the rest of the stream
	 * straight into their lanes -- one copy and gigabytes of first-touched memory less than taking the whole file through
	 * the chunks (14.3 -> 13.x CPU-seconds for the 12.6 GB file under the pod's 16-CPU quota). */
	const size_t n_drain = n_par >= 8 ? n_par / 4 : n_par;
	m_early->hand_over();
	std::atomic<uint64_t> chunks(0);
	std::vector<std::thread> pool;
	for (size_t t = 0; t < n_drain; ++t)
		pool.emplace_back([this, t, &chunks]() {
			Feeder &f = feederFor(t);
			std::unique_ptr<PackedChunk> c;
			while (m_early->next(&c)) {
				f.submitChunk(*c);
				m_early->recycle(std::move(c));
				++chunks;
			}
		});
	std::unique_ptr<GzStream> rest = m_early->release_stream();   /* waits for the parsers to finish what they hold */
	const auto t2 = std::chrono::steady_clock::now();
	if (rest && !m_early->failed()) {
		if (n_drain < n_par) {
			countGzStream(std::move(rest), m_opt.inputs[0], n_drain, n_par - n_drain);
			for (auto &th : pool) th.join();
		} else {                                                /* few threads: one after the other on the same lanes */
			for (auto &th : pool) th.join();
			countGzStream(std::move(rest), m_opt.inputs[0], 0, n_par);
		}
		pool.clear();
	}
	for (auto &th : pool) th.join();
	if (m_early->failed()) {                                    /* reads were lost (no memory for a chunk, the file's rest unreadable): never print counts */
		fatal("ntsmCount: " + m_early->error());                /* the message names its own cause (allocation or I/O) */
	}
	if (m_opt.phase_times)
		std::cerr << "[phase] " << m_opt.inputs[0] << ": early ingest (" << m_early->how() << ") parsed " << m_early->records() << " records ("
		          << m_early->parallel_records() << " in parallel) in " << m_early->parse_seconds() << " s beside the start-up; lanes "
		          << std::chrono::duration<double>(t1 - t0).count() << " s, " << chunks.load() << " chunks submitted, the stream handed over "
		          << std::chrono::duration<double>(t2 - t1).count() << " s and everything done "
		          << std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() << " s after the context was ready" << std::endl;
}

void FingerPrint::fetchResults()
{
	if (m_fetched) return;
	m_main.reset();
	closeLanes();
	m_counts.assign(m_sites.keys.size(), 0);
	m_totals = ntsm_totals();
	/* Several devices (-g a,b,...): one RCCL SUM of the dense per-k-mer vectors + totals over xGMI, after which every
	 * context reports the job-wide result -- SUM, not MAX: the per-site maxima are taken from the summed counts, which is
	 * what one reference run over all reads computes (src/FingerPrint.hpp:281-294). */
	int rc = m_ctx.size() > 1 ? ntsm_allreduce(m_ctx.data(), (int) m_ctx.size()) : NTSM_OK;
	if (rc == NTSM_ERR_RCCL) {
		/* RCCL missing or failing must not cost a finished run its result: every context still holds its own counts
		 * (a failed ntsm_allreduce imports nothing), so they are summed here instead -- same SUM, over PCIe. */
		std::cerr << "ntsmCount: RCCL unavailable (" << ntsm_strerror(rc) << "), summing the " << m_ctx.size()
		          << " devices' counts on the host" << std::endl;
		std::vector<uint64_t> part(m_counts.size());
		rc = NTSM_OK;
		for (size_t d = 0; d < m_ctx.size() && rc == 0; ++d) {
			ntsm_totals t;
			rc = ntsm_sync(m_ctx[d], &t);
			if (rc == 0) rc = ntsm_counts(m_ctx[d], part.data());
			if (rc) break;
			for (size_t i = 0; i < part.size(); ++i) m_counts[i] += part[i];
			m_totals.total_kmers += t.total_kmers;
			m_totals.total_hits += t.total_hits;
			m_totals.total_bases += t.total_bases;
			m_totals.reads_consumed += t.reads_consumed;
			m_totals.early_stop |= t.early_stop;
		}
		if (rc == 0) { m_fetched = true; return; }
	}
	if (rc == 0) rc = ntsm_sync(m_ctx[0], &m_totals);
	if (rc == 0) rc = ntsm_counts(m_ctx[0], m_counts.data());
	if (rc) {
		fatal(std::string("ntsmCount: cannot fetch counts: ") + ntsm_strerror(rc));
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
