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
#include <thread>
#include <vector>

#include "../../../include/ntsm_hip.h"
#include "host_shape.hpp"
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
	int debug_kernel = -1;                 /* --debug-kernel V (tests only): ntsm_set_kernel(ctx, V) on every context, -1 = the library's choice */
	std::vector<int> devices;              /* -g 0,1,...: host threads (-t) are spread round-robin over these devices */
	uint64_t batch_bytes = 64ull << 20;    /* staging capacity per slot */
	bool phase_times = false;              /* NTSM_PHASE_TIMES: print where the wall time goes (stderr) */
	/* Block size of the block-parallel FASTQ ingest (-t N, plain files).  A block's sequences (< half its bytes) fit
	 * one 16 MiB lane slot, so a thread never has to wait for its predecessor in the middle of a block. */
	uint64_t block_bytes = 16ull << 20;
	/* Producer lanes send 2-bit codes + a validity bit per position (3/8 byte instead of 1 over PCIe; pack2.hpp) and the
	 * device unpacks them.  NTSM_NO_PACK=1 sends the raw bytes instead (same counts: A/B of the two ingest forms). */
	bool pack = true;
	/* gzip inputs of at least this many (compressed) bytes take the parallel route with -t N: decoder pool + piece-parallel
	 * parsing (gz_stream.hpp, parallel_gz_fastq.hpp); smaller ones are read one thread per file */
	uint64_t gz_parallel_min_bytes = 8ull << 20;
	unsigned gz_decoders = 0;              /* NTSM_GZ_DECODERS: decoder threads of that route (0 = automatic) */
	/* The input files, known before the sites are loaded: with -t N and no -m the first one is parsed into ordinary memory
	 * while the sites load and the tables build (early_ingest.hpp; NTSM_NO_EARLY=1 switches that off) */
	std::vector<std::string> inputs;
	bool early = true;
	/* which kind of first file: 1 plain FASTQ, 2 gzip, 3 both (NTSM_EARLY=plain|gz|all).  Default gzip only: measured on a
	 * 12.6 GB FASTQ, the early path parses into gigabytes of memory touched for the first time at 1/6 of the speed of the lane
	 * slots (reused, pinned, cache-warm) and loses (0.85 s against 0.41 s whole process); a .gz, whose inflate dominates, gains
	 * (2e7 / 4e7 reads: 0.68 / 1.22 s against 0.80 / 1.29 s) */
	int early_kinds = 2;
};

/* The staging batch one host thread is filling for a GPU context: the context's own slots (single-threaded and
 * -m runs) or a producer lane of it (-t N: all threads count into the same context).  Driven by one thread. */
class Feeder {
public:
	Feeder(const Options &opt, ntsm_ctx *ctx, uint64_t max_hits, bool lane);
	~Feeder();
	Feeder(const Feeder &) = delete;
	Feeder &operator=(const Feeder &) = delete;
	/* Count every record of one file (src/FingerPrint.hpp:49-81); stops early once the -m threshold tripped. */
	void feedFile(const std::string &path, uint64_t offset = 0);
	/* the same on an open gzip stream positioned at a record boundary (what a parallel phase left, parallel_gz_fastq.hpp) */
	void feedStream(std::unique_ptr<class GzStream> gz);
	/* a batch that was packed in ordinary memory before this lane existed (early_ingest.hpp): copied into a slot and submitted */
	void submitChunk(const struct PackedChunk &c);
	/* One read (insertCount(seq.s, seq.l), src/FingerPrint.hpp:89-103): append to the staging batch. */
	void feedRead(const char *seq, uint64_t len);
	void flush();
	/* Sink interface of the block-parallel ingest (parallel_fastq.hpp) */
	bool has_room(uint64_t len) const
	{
		if (m_packed) return !(m_codes && packedExtent(len) > m_capPos);
		return !(m_bases && (m_fill + len + 1 > m_capBytes || m_nReads >= m_capReads));
	}
	void feed(const char *seq, uint64_t len) { feedRead(seq, len); }
	/* Drop what is staged.  The slot stays acquired (it is handed back by the next submit), so feedRead() must still
	 * be able to grow it: it checks the capacity whenever the batch is empty, not only when no slot is held. */
	void discard() { m_fill = 0; m_nReads = 0; m_pos = 0; m_nBases = 0; }
	void begin_block(size_t) { }
	/* flush + close the lane (its totals fold into the context); the Feeder must not be fed afterwards */
	void finish();
	bool earlyTerm() const { return m_earlyTerm; }

private:
	[[noreturn]] void die(int rc, const char *what) const;
	void progressLine();                       /* -vvv: "Current Total: ..." (src/FingerPrint.hpp:70-78) */
	uint64_t m_totalReads = 0;                 /* the reference's m_totalReads: advanced under -vvv only */
	const Options &m_opt;
	void openLane();
	ntsm_ctx *m_ctx = nullptr;
	ntsm_lane *m_lane = nullptr;
	bool m_useLane = false;
	uint64_t m_maxCounts = 0;
	uint8_t *m_bases = nullptr;
	uint64_t *m_readEnd = nullptr;
	uint64_t m_capBytes = 0, m_capReads = 0, m_fill = 0, m_cfgBytes = 0;
	uint32_t m_nReads = 0;
	bool m_earlyTerm = false;
	/* packed lane (Options::pack): the two planes of the slot, its capacity in positions, the position the next read
	 * starts at and the sum of the read lengths staged so far */
	bool m_packed = false;
	uint8_t *m_codes = nullptr, *m_valid = nullptr;
	uint64_t m_capPos = 0, m_pos = 0, m_nBases = 0;
	uint64_t packedExtent(uint64_t len) const { return m_pos + (len & ~31ull) + 32; }   /* pack2_extent */
	void feedPacked(const char *seq, uint64_t len);
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
	IngestPlan m_plan { 1, 1, 1, 1 };                       /* thread counts from the CPUs granted (host_shape.hpp) */
	SiteSet m_sites;
	uint64_t m_maxCounts = 0;
	Feeder &feederFor(size_t t);                         /* thread t's lane on device devices[t % n] (created on first use) */
	void closeLanes();
	void joinPrep();
	std::vector<std::thread> m_prep;                     /* per device: GPU bring-up, streams, pinned pool while the sites are parsed */
	std::vector<ntsm_ctx *> m_ctx;                       /* one GPU context per distinct -g device, [0] = first device */
	std::vector<int> m_ctxDevice;
	std::unique_ptr<Feeder> m_main;                      /* context [0]'s own staging: single-threaded and -m runs */
	std::vector<std::unique_ptr<Feeder>> m_lanes;        /* -t N: one producer lane per host thread */
	std::unique_ptr<class EarlyIngest> m_early;          /* the first input file, parsed while the sites load (early_ingest.hpp) */
	/* A finished gzip stream holds 0.5-0.8 GB of buffers it touched (symbol buffers of the decoder pool, pieces): giving them
	 * back costs the kernel 0.07-0.1 s per stream, which used to sit between two files and in front of the last line.  Streams
	 * are therefore destroyed on a side thread while the next file is read (joined by the destructor; _exit does not wait). */
	std::vector<std::thread> m_retire;
	template <class T> void retireLater(std::unique_ptr<T> p)
	{
		if (p) m_retire.emplace_back([q = std::shared_ptr<T>(std::move(p))]() mutable { q.reset(); });
	}
	void drainEarly();
	void countGzStream(std::unique_ptr<class GzStream> gz, const std::string &fn, size_t first_feeder, size_t n_feeders);                                   /* its chunks -> the lanes */
	/* results */
	bool m_fetched = false;
	ntsm_totals m_totals {};
	std::vector<uint64_t> m_counts;
};

} // namespace ntsm
#endif
