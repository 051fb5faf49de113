/*
 * tools/feed_bench.cpp -- the HOST-FED count path on a PCIe roofline ("next" item 1).
 *
 * The resident-stream figure of bench.py starts with the reads already in HBM.  A caller of the C ABI that replaces the
 * reference's read loop (src/FingerPrint.hpp:66-69 over vendor/kseq.h:229) starts with reads in HOST memory; what it gets is
 * bounded by the host-to-device link, and this tool measures every way include/ntsm_hip.h offers to cross it, on pre-parsed
 * reads (no file, no parser), each leg checked against the resident path's counts:
 *
 *   h2d_ceiling          pinned hipMemcpyAsync of the same batches, nothing else (two streams / one stream): the roofline
 *   submit_1thread       ntsm_submit from pageable memory, staging copy on the submitting thread alone (rounds 1-5)
 *   submit               ntsm_submit from pageable memory, staging copy on several threads (default)
 *   staged               ntsm_staging_acquire / ntsm_submit_staged, the caller filling the slot on several threads
 *   submit_pinned        ntsm_submit_pinned from memory the caller pinned: no host-side copy at all
 *   lanes_raw_T          T producer lanes, raw bytes (each thread memcpy's its reads into its lane's pinned slot)
 *   lanes_packed_T       T producer lanes, 2-bit codes + validity bit (3/8 byte per base over the link), packed by the lane threads
 *
 * One JSON object on stdout.  Usage: ntsm_feed_bench [--reads 2e7] [--batch-mib 64] [--device 0] [--lanes 1,4,16]
 *                                                    [--legs all|comma list] [--reps 2] [--n-sites 96287] [--sites-seed S]
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>
#include <unistd.h>

#include "../include/ntsm_hip.h"
#include "../include/ntsm_host.h"
#include "../include/ntsm_synth.h"
#include "../ntsm_amd/csrc/host/pack2.hpp"

namespace {

double now_s()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define HIPOK(call)                                                                                       \
	do {                                                                                                  \
		hipError_t e_ = (call);                                                                           \
		if (e_ != hipSuccess) { fprintf(stderr, "feed_bench: %s: %s\n", #call, hipGetErrorString(e_)); exit(1); } \
	} while (0)
#define NTOK(call)                                                                                        \
	do {                                                                                                  \
		int r_ = (call);                                                                                  \
		if (r_ != NTSM_OK) { fprintf(stderr, "feed_bench: %s: %s\n", #call, ntsm_strerror(r_)); exit(1); }  \
	} while (0)

struct Result { uint64_t kmers = 0, hits = 0, bases = 0, reads = 0; std::vector<uint64_t> counts; };

Result fetch(ntsm_ctx *ctx, uint32_t n_kmers)
{
	Result r;
	ntsm_totals t;
	NTOK(ntsm_sync(ctx, &t));
	r.kmers = t.total_kmers; r.hits = t.total_hits; r.bases = t.total_bases; r.reads = t.reads_consumed;
	r.counts.resize(n_kmers);
	NTOK(ntsm_counts(ctx, r.counts.data()));
	return r;
}

bool same(const Result &a, const Result &b, uint64_t reps)
{
	if (a.kmers != reps * b.kmers || a.hits != reps * b.hits || a.bases != reps * b.bases || a.reads != reps * b.reads) return false;
	for (size_t i = 0; i < a.counts.size(); ++i) if (a.counts[i] != reps * b.counts[i]) return false;
	return true;
}

void parallel(unsigned n, const std::function<void(unsigned)> &f)
{
	std::vector<std::thread> th;
	for (unsigned t = 1; t < n; ++t) th.emplace_back(f, t);
	f(0);
	for (auto &x : th) x.join();
}

std::string json;
void jadd(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void jadd(const char *fmt, ...)
{
	char buf[2048];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	json += buf;
}

} // namespace

int main(int argc, char **argv)
{
	uint64_t n_reads = 20000000ull, batch_mib = 64, lane_batch_mib = 8, sites_seed = 20241218ull, read_seed = 7;
	std::string sweep_arg = "";

	uint32_t n_sites = 96287, read_len = 150;
	int device = 0, reps = 2;
	std::string lanes_arg = "1,4,16", legs_arg = "all";
	for (int i = 1; i + 1 < argc; i += 2) {
		const std::string k = argv[i], v = argv[i + 1];
		if (k == "--reads") n_reads = (uint64_t) strtod(v.c_str(), nullptr);
		else if (k == "--batch-mib") batch_mib = strtoull(v.c_str(), nullptr, 10);
		else if (k == "--lane-batch-mib") lane_batch_mib = strtoull(v.c_str(), nullptr, 10);
		else if (k == "--submit-threads") sweep_arg = v;            /* comma list: one extra `submit` leg per thread count */
		else if (k == "--device") device = atoi(v.c_str());
		else if (k == "--reps") reps = std::max(1, atoi(v.c_str()));
		else if (k == "--lanes") lanes_arg = v;
		else if (k == "--legs") legs_arg = v;
		else if (k == "--n-sites") n_sites = (uint32_t) strtoull(v.c_str(), nullptr, 10);
		else if (k == "--sites-seed") sites_seed = strtoull(v.c_str(), nullptr, 10);
		else if (k == "--read-seed") read_seed = strtoull(v.c_str(), nullptr, 10);
		else { fprintf(stderr, "feed_bench: unknown option %s\n", k.c_str()); return 2; }
	}
	auto want = [&](const char *leg) { return legs_arg == "all" || ("," + legs_arg + ",").find(std::string(",") + leg + ",") != std::string::npos; };
	std::vector<unsigned> lane_counts;
	for (size_t p = 0; p < lanes_arg.size();) {
		size_t q = lanes_arg.find(',', p);
		if (q == std::string::npos) q = lanes_arg.size();
		const unsigned v = (unsigned) atoi(lanes_arg.substr(p, q - p).c_str());
		if (v) lane_counts.push_back(v);
		p = q + 1;
	}
	const char *diag_env = getenv("FEED_DIAG");                  /* packed-lane legs only: packonly | nopack (see the lane loop) */
	const int diag = !diag_env ? 0 : !strcmp(diag_env, "packonly") ? 1 : !strcmp(diag_env, "nopack") ? 2 : 0;
	cpu_set_t aff;
	const unsigned cpus = sched_getaffinity(0, sizeof aff, &aff) == 0 ? (unsigned) CPU_COUNT(&aff) : 1u;
	const unsigned gen_threads = std::max(1u, std::min(32u, cpus));

	/* ---- site set + context */
	char sites_path[] = "/tmp/ntsm_feed_sites_XXXXXX";
	{ const int fd = mkstemp(sites_path); if (fd < 0) { perror("mkstemp"); return 1; } close(fd); }
	std::vector<uint8_t> windows((size_t) n_sites * 2 * NTSM_SYNTH_WSTRIDE);
	uint64_t nk = 0;
	if (ntsm_synth_sites(sites_seed, n_sites, 19, windows.data(), sites_path, &nk)) { fprintf(stderr, "feed_bench: site generation failed\n"); return 1; }
	ntsm_sites *sites = nullptr;
	if (ntsm_sites_load(sites_path, 19, 0, &sites)) { fprintf(stderr, "feed_bench: cannot load %s\n", sites_path); return 1; }
	unlink(sites_path);
	const uint32_t n_kmers = (uint32_t) ntsm_sites_n_keys(sites);
	HIPOK(hipSetDevice(device));
	ntsm_ctx *ctx = nullptr;
	NTOK(ntsm_create(&ctx, device, 19, ntsm_sites_keys(sites), n_kmers, NTSM_KEYS_CANONICAL, 0));

	/* ---- the reads, pre-parsed, in ordinary (pageable) host memory: flat stream layout of include/ntsm_hip.h */
	const uint64_t stride = read_len + 1, n_bytes = n_reads * stride, n_bases = n_reads * read_len;
	uint8_t *stream = (uint8_t *) aligned_alloc(4096, (n_bytes + 4095) & ~4095ull);
	uint64_t *read_end = (uint64_t *) malloc(n_reads * sizeof(uint64_t));
	if (!stream || !read_end) { fprintf(stderr, "feed_bench: no memory for %llu reads\n", (unsigned long long) n_reads); return 1; }
	ntsm_synth_short sp;
	ntsm_synth_short_params(&sp, read_seed, read_len, n_sites, 0.10, 0.01, 0.0005);
	const double tg0 = now_s();
	parallel(gen_threads, [&](unsigned t) {
		const uint64_t r0 = n_reads * t / gen_threads, r1 = n_reads * (t + 1) / gen_threads;
		ntsm_synth_short_fill_host(&sp, windows.data(), r0 * stride, (r1 - r0) * stride, stream + r0 * stride);
		for (uint64_t r = r0; r < r1; ++r) read_end[r] = r * stride + read_len;
	});
	const double gen_s = now_s() - tg0;
	const uint64_t batch_reads = std::max<uint64_t>(1, (batch_mib << 20) / stride);
	const uint64_t n_batches = (n_reads + batch_reads - 1) / batch_reads;
	std::vector<uint64_t> rel_end(batch_reads);             /* read_end of a batch, relative to its first byte */
	for (uint64_t r = 0; r < batch_reads; ++r) rel_end[r] = r * stride + read_len;
	NTOK(ntsm_set_batch_capacity(ctx, batch_reads * stride, batch_reads));

	jadd("{\"workload\": \"%llu pre-parsed synthetic 150 bp reads (%.2f GB flat stream) in host memory, hs_n10_like sites (%u site 19-mers), batches of %llu MiB\"",
		(unsigned long long) n_reads, n_bytes / 1e9, n_kmers, (unsigned long long) batch_mib);
	jadd(", \"lane_batch_MiB\": %llu", (unsigned long long) lane_batch_mib);
	jadd(", \"reads\": %llu, \"bases\": %llu, \"stream_bytes\": %llu, \"batch_bytes\": %llu, \"reps\": %d, \"cpus_in_affinity_mask\": %u, \"generate_s\": %.3f",
		(unsigned long long) n_reads, (unsigned long long) n_bases, (unsigned long long) n_bytes, (unsigned long long) (batch_reads * stride), reps, cpus, gen_s);

	/* ---- expected result: the same stream resident in device memory, one launch (the path bench.py's headline times) */
	Result expect;
	{
		uint8_t *d = nullptr;
		HIPOK(hipMalloc((void **) &d, n_bytes + 64));
		const double t0 = now_s();
		HIPOK(hipMemcpy(d, stream, n_bytes, hipMemcpyHostToDevice));
		const double pageable_s = now_s() - t0;
		NTOK(ntsm_count_resident(ctx, d, n_bytes, nullptr, n_reads, +1));
		expect = fetch(ctx, n_kmers);
		HIPOK(hipFree(d));
		jadd(", \"pageable_hipMemcpy_GBps\": %.2f", n_bytes / pageable_s / 1e9);
		if (expect.reads != n_reads || expect.bases != n_bases) { fprintf(stderr, "feed_bench: resident pass consumed %llu reads\n", (unsigned long long) expect.reads); return 1; }
	}

	/* ---- the roofline of this path: pinned hipMemcpyAsync of the same batches and nothing else */
	double ceiling = 0;
	{
		const uint64_t bb = batch_reads * stride;
		uint8_t *h[2], *d[2];
		hipStream_t st[2];
		for (int i = 0; i < 2; ++i) {
			HIPOK(hipHostMalloc((void **) &h[i], bb, hipHostMallocPortable));
			memcpy(h[i], stream + (uint64_t) i * bb, std::min(bb, n_bytes - std::min(n_bytes, (uint64_t) i * bb)));
			HIPOK(hipMalloc((void **) &d[i], bb));
			HIPOK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
		}
		double best[2] = { 0, 0 };
		for (int n_streams = 2; n_streams >= 1; --n_streams)
			for (int rep = 0; rep < 3; ++rep) {
				HIPOK(hipDeviceSynchronize());
				const double t0 = now_s();
				for (uint64_t b = 0; b < n_batches; ++b) {
					const uint64_t len = std::min(bb, n_bytes - b * bb);
					HIPOK(hipMemcpyAsync(d[b & 1], h[b & 1], len, hipMemcpyHostToDevice, st[n_streams == 2 ? (b & 1) : 0]));
				}
				HIPOK(hipDeviceSynchronize());
				best[n_streams - 1] = std::max(best[n_streams - 1], n_bytes / (now_s() - t0) / 1e9);
			}
		for (int i = 0; i < 2; ++i) { HIPOK(hipHostFree(h[i])); HIPOK(hipFree(d[i])); HIPOK(hipStreamDestroy(st[i])); }
		ceiling = std::max(best[0], best[1]);
		jadd(", \"h2d_ceiling\": {\"pinned_hipMemcpyAsync_two_streams_GBps\": %.2f, \"pinned_hipMemcpyAsync_one_stream_GBps\": %.2f, \"GBps\": %.2f, \"nominal_link_GBps\": 64.0, "
			"\"what\": \"the same batches from two pinned buffers to two device buffers, no kernel, no host-side copy: best of 3\"}", best[1], best[0], ceiling);
	}

	bool all_ok = true;
	bool first_leg = true;
	double pin_once_s = -1;
	jadd(", \"legs\": {");
	/* run `body` reps times between ntsm_reset and the final sync; link_bytes = what one repetition hands to the H2D copies */
	auto leg = [&](const std::string &name, uint64_t link_bytes, const char *bound_by, const std::function<void()> &body, const std::function<void()> &before = nullptr,
			const std::function<void()> &after = nullptr, const std::function<void()> &finish_timed = nullptr) {
		/* both staging slots exist before the clock starts (pinning 2 x 64 MiB is 20-30 ms, once per context, not per batch) */
		NTOK(ntsm_submit(ctx, stream, stride, rel_end.data(), 1));
		NTOK(ntsm_submit(ctx, stream, stride, rel_end.data(), 1));
		NTOK(ntsm_reset(ctx));
		if (before) before();
		const double t0 = now_s();
		for (int r = 0; r < reps; ++r) body();
		if (finish_timed) finish_timed();                       /* lanes: closed (drained) inside the timed region, ntsm_sync needs them closed */
		ntsm_totals t;
		NTOK(ntsm_sync(ctx, &t));
		const double s = (now_s() - t0) / reps;
		if (after) after();
		const Result got = fetch(ctx, n_kmers);
		const bool ok = same(got, expect, (uint64_t) reps);
		all_ok = all_ok && ok;
		jadd("%s\"%s\": {\"seconds\": %.4f, \"gbases_per_s\": %.2f, \"link_bytes\": %llu, \"link_GBps\": %.2f, \"frac_of_h2d_ceiling\": %.3f, \"frac_of_nominal_64GBps\": %.3f, "
			"\"bound_by\": \"%s\", \"counts_equal_resident_path\": %s}",
			first_leg ? "" : ", ", name.c_str(), s, n_bases / s / 1e9, (unsigned long long) link_bytes, link_bytes / s / 1e9, link_bytes / s / 1e9 / ceiling,
			link_bytes / s / 1e9 / 64.0, bound_by, ok ? "true" : "false");
		first_leg = false;
		fprintf(stderr, "feed_bench: %-18s %.3f s  %.1f Gbases/s  %.1f GB/s over the link (%.0f %% of the ceiling)%s\n", name.c_str(), s, n_bases / s / 1e9, link_bytes / s / 1e9,
			100.0 * link_bytes / s / 1e9 / ceiling, ok ? "" : "  COUNTS DIFFER");
	};
	auto batch_of = [&](uint64_t b, uint64_t *r0, uint64_t *nr) { *r0 = b * batch_reads; *nr = std::min(batch_reads, n_reads - *r0); };

	if (want("submit_1thread"))
		leg("submit_1thread", n_bytes, "the submitting thread's memcpy into the pinned slot", [&] {
			for (uint64_t b = 0; b < n_batches; ++b) {
				uint64_t r0, nr; batch_of(b, &r0, &nr);
				NTOK(ntsm_submit(ctx, stream + r0 * stride, nr * stride, rel_end.data(), (uint32_t) nr));
			}
		}, [&] { NTOK(ntsm_set_submit_threads(ctx, 1)); }, [&] { NTOK(ntsm_set_submit_threads(ctx, 0)); });
	if (want("submit"))
		leg("submit", n_bytes, "the link (staging copy on several threads, overlapped with the previous batch's DMA)", [&] {
			for (uint64_t b = 0; b < n_batches; ++b) {
				uint64_t r0, nr; batch_of(b, &r0, &nr);
				NTOK(ntsm_submit(ctx, stream + r0 * stride, nr * stride, rel_end.data(), (uint32_t) nr));
			}
		});
	for (size_t p = 0; p < sweep_arg.size();) {
		size_t q = sweep_arg.find(',', p);
		if (q == std::string::npos) q = sweep_arg.size();
		const int th = atoi(sweep_arg.substr(p, q - p).c_str());
		p = q + 1;
		if (th < 1) continue;
		leg("submit_" + std::to_string(th) + "threads", n_bytes, "staging copy on that many threads", [&] {
			for (uint64_t b = 0; b < n_batches; ++b) {
				uint64_t r0, nr; batch_of(b, &r0, &nr);
				NTOK(ntsm_submit(ctx, stream + r0 * stride, nr * stride, rel_end.data(), (uint32_t) nr));
			}
		}, [&] { NTOK(ntsm_set_submit_threads(ctx, th)); }, [&] { NTOK(ntsm_set_submit_threads(ctx, 0)); });
	}
	if (want("staged")) {
		const unsigned fill_threads = std::max(1u, std::min(4u, cpus));
		leg("staged", n_bytes, "the link (the caller fills the pinned slot itself, here on several threads)", [&] {
			for (uint64_t b = 0; b < n_batches; ++b) {
				uint64_t r0, nr; batch_of(b, &r0, &nr);
				uint8_t *hb; uint64_t cb, *hr, cr;
				NTOK(ntsm_staging_acquire(ctx, &hb, &cb, &hr, &cr));
				const uint64_t len = nr * stride;
				parallel(fill_threads, [&](unsigned t) {
					const uint64_t lo = (len * t / fill_threads) & ~4095ull, hi = t + 1 == fill_threads ? len : (len * (t + 1) / fill_threads) & ~4095ull;
					memcpy(hb + lo, stream + r0 * stride + lo, hi - lo);
				});
				memcpy(hr, rel_end.data(), nr * sizeof(uint64_t));
				NTOK(ntsm_submit_staged(ctx, len, (uint32_t) nr));
			}
		});
	}
	if (want("submit_pinned")) {
		double pin_s = 0;
		leg("submit_pinned", n_bytes, "the link (no host-side copy: the DMA reads the caller's pinned memory)", [&] {
			for (uint64_t b = 0; b < n_batches; ++b) {
				uint64_t r0, nr; batch_of(b, &r0, &nr);
				NTOK(ntsm_submit_pinned(ctx, stream + r0 * stride, nr * stride, rel_end.data(), (uint32_t) nr));
			}
		}, [&] { const double t0 = now_s(); NTOK(ntsm_host_pin(stream, (n_bytes + 4095) & ~4095ull)); pin_s = now_s() - t0; }, [&] { NTOK(ntsm_host_unpin(stream)); });
		pin_once_s = pin_s;
	}
	for (int packed = 0; packed < 2; ++packed)
		for (unsigned T : lane_counts) {
			const std::string name = std::string(packed ? "lanes_packed_" : "lanes_raw_") + std::to_string(T);
			if (!want(name.c_str()) && !want(packed ? "lanes_packed" : "lanes_raw")) continue;
			std::vector<ntsm_lane *> lanes(T, nullptr);
			/* positions per packed batch: every read starts at a multiple of 8 -> 152 positions per 150 bp read */
			const uint64_t cap_pos = ((lane_batch_mib << 20) + 31) & ~31ull;
			const uint64_t lane_reads = std::max<uint64_t>(1, (lane_batch_mib << 20) / stride);
			uint64_t link = 0;
			if (packed) link = (n_reads * 152 + 31) / 32 * 12;          /* 3/8 byte per position */
			else link = n_bytes;
			leg(name, link, packed ? (std::string("the lane threads' packing (pack2_append, ") + ntsm::pack2_impl() + ") up to ~8 lanes, then the device side of small batches (FEED_DIAG=packonly|nopack), not the link").c_str() : "the lane threads' memcpy into their pinned slots / the link",
				[&] {
					std::atomic<int> err(0);
					parallel(T, [&](unsigned t) {
						ntsm_lane *ln = lanes[t];
						const uint64_t r0 = n_reads * t / T, r1 = n_reads * (t + 1) / T;
						if (packed && diag == 1) {
							/* diagnostic (FEED_DIAG=packonly): the packing alone, into a private buffer -- what the host side can produce */
							std::vector<uint8_t> priv(cap_pos * 3 / 8 + 64);
							for (uint64_t r = r0; r < r1;) {
								uint64_t pos = 0;
								while (r < r1 && ntsm::pack2_extent(pos, read_len) <= cap_pos) {
									pos = ntsm::pack2_append(priv.data(), priv.data() + cap_pos / 4, pos, (const char *) stream + r * stride, read_len);
									++r;
								}
							}
						} else if (packed) {
							uint64_t filled = 0, pos_full = 0, nb_full = 0; uint32_t nr_full = 0;
							for (uint64_t r = r0; r < r1;) {
								uint8_t *codes, *valid; uint64_t cap;
								if (ntsm_lane_acquire_packed(ln, &codes, &valid, &cap)) { err = 1; return; }
								uint64_t pos = 0; uint32_t nr = 0; uint64_t nb = 0;
								if (diag == 2 && filled >= 2 && r + nr_full <= r1) {
									/* diagnostic (FEED_DIAG=nopack): both slots hold a full batch already -- hand the same bytes over again: what
									 * the device side + the HIP calls take without the packing (counts are wrong by construction) */
									pos = pos_full; nr = nr_full; nb = nb_full; r += nr_full;
								} else {
									while (r < r1 && ntsm::pack2_extent(pos, read_len) <= cap) {
										pos = ntsm::pack2_append(codes, valid, pos, (const char *) stream + r * stride, read_len);
										++r; ++nr; nb += read_len;
									}
									if (r < r1) { ++filled; pos_full = pos; nr_full = nr; nb_full = nb; }
								}
								if (ntsm_lane_submit_packed(ln, pos, nr, nb)) { err = 1; return; }
							}
						} else {
							for (uint64_t r = r0; r < r1;) {
								uint8_t *hb; uint64_t cb, *hr, cr;
								if (ntsm_lane_acquire(ln, &hb, &cb, &hr, &cr)) { err = 1; return; }
								const uint64_t nr = std::min<uint64_t>(std::min(cb / stride, cr), r1 - r);
								memcpy(hb, stream + r * stride, nr * stride);
								hr[nr - 1] = rel_end[nr - 1];                    /* lanes are never armed: only the last offset is looked at */
								if (ntsm_lane_submit(ln, nr * stride, (uint32_t) nr)) { err = 1; return; }
								r += nr;
							}
						}
					});
					if (err) { fprintf(stderr, "feed_bench: a lane call failed\n"); exit(1); }
				},
				[&] {
					for (unsigned t = 0; t < T; ++t) {
						if (packed) NTOK(ntsm_lane_open_packed(ctx, cap_pos, &lanes[t]));
						else NTOK(ntsm_lane_open(ctx, lane_reads * stride, lane_reads, &lanes[t]));
					}
				},
				nullptr,
				[&] { for (unsigned t = 0; t < T; ++t) NTOK(ntsm_lane_close(lanes[t])); });
		}
	jadd("}");
	if (pin_once_s >= 0) jadd(", \"submit_pinned_pin_once_s\": %.3f", pin_once_s);
	jadd(", \"all_counts_equal_resident_path\": %s}", all_ok ? "true" : "false");
	puts(json.c_str());
	ntsm_destroy(ctx);
	ntsm_sites_free(sites);
	free(stream);
	free(read_end);
	return all_ok ? 0 : 1;
}
