/*
 * synth_host.cpp -- host side of the synthetic workload generator (include/ntsm_synth.h).
 * Site-set generation with reject-and-redraw, FASTQ/FASTA writers, host twins of the device fills.
 */
#include "../../include/ntsm_synth.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

namespace {

/* plain or gzip output chosen by file suffix */
struct Out {
	FILE *fp = nullptr;
	gzFile gz = nullptr;
	bool open(const char *path)
	{
		size_t n = strlen(path);
		if (n > 3 && !strcmp(path + n - 3, ".gz")) gz = gzopen(path, "wb1");
		else fp = fopen(path, "wb");
		return fp || gz;
	}
	void write(const void *p, size_t n)
	{
		if (gz) gzwrite(gz, p, (unsigned) n);
		else fwrite(p, 1, n, fp);
	}
	void close()
	{
		if (gz) gzclose(gz);
		if (fp) fclose(fp);
	}
};

/* open-addressing set of canonical k-mer codes (value+1 stored, 0 = empty) */
struct CodeSet {
	std::vector<uint64_t> slot;
	uint64_t mask = 0, size = 0;
	explicit CodeSet(uint64_t expect)
	{
		uint64_t n = 64;
		while (n < expect * 2) n <<= 1;
		slot.assign(n, 0);
		mask = n - 1;
	}
	bool contains(uint64_t c) const
	{
		for (uint64_t i = ntsm_synth_mix64(c) & mask;; i = (i + 1) & mask) {
			if (slot[i] == 0) return false;
			if (slot[i] == c + 1) return true;
		}
	}
	void insert(uint64_t c)
	{
		for (uint64_t i = ntsm_synth_mix64(c) & mask;; i = (i + 1) & mask) {
			if (slot[i] == c + 1) return;
			if (slot[i] == 0) { slot[i] = c + 1; ++size; return; }
		}
	}
};

uint64_t canon_code(const uint8_t *codes, unsigned k)
{
	uint64_t fw = 0, rv = 0;
	for (unsigned i = 0; i < k; ++i) {
		fw = (fw << 2) | codes[i];
		rv = (rv >> 2) | ((uint64_t) (3 - codes[i]) << (2 * (k - 1)));
	}
	return fw < rv ? fw : rv;
}

} // namespace

extern "C" {

int ntsm_synth_sites(uint64_t seed, uint32_t n_sites, unsigned k, uint8_t *windows,
		const char *fasta_path, uint64_t *n_kmers_out)
{
	return ntsm_synth_sites_keep(seed, n_sites, k, 0, windows, fasta_path, n_kmers_out);
}

int ntsm_synth_sites_keep(uint64_t seed, uint32_t n_sites, unsigned k, unsigned min_keep_req, uint8_t *windows,
		const char *fasta_path, uint64_t *n_kmers_out)
{
	if (k == 0 || k > 31 || k > NTSM_SYNTH_W) return -1;
	const unsigned n_starts = NTSM_SYNTH_W - k + 1;          /* 13 for k = 19 */
	const unsigned min_keep = min_keep_req ? (min_keep_req < n_starts ? min_keep_req : n_starts) : (n_starts < 3 ? n_starts : 3);
	Out out;
	if (fasta_path && !out.open(fasta_path)) return -2;
	CodeSet seen((uint64_t) n_sites * 2 * n_starts);
	std::string rec;
	uint64_t redraws = 0;
	for (uint32_t i = 0; i < n_sites; ++i) {
		for (uint32_t attempt = 0;; ++attempt) {
			const uint64_t base_ctr = ((uint64_t) i << 20) | attempt;
			uint8_t win[2][NTSM_SYNTH_WSTRIDE] = {};
			for (unsigned p = 0; p < NTSM_SYNTH_W; ++p)
				win[0][p] = win[1][p] = (uint8_t) (ntsm_synth_rnd(seed, 10, base_ctr * 64 + p) & 3);
			const uint64_t hc = ntsm_synth_rnd(seed, 11, base_ctr);
			win[0][NTSM_SYNTH_W / 2] = (hc & 1) ? 3 : 0;     /* ref in {A,T} */
			win[1][NTSM_SYNTH_W / 2] = (hc & 2) ? 2 : 1;     /* var in {C,G} */
			/* keep n in [min_keep, n_starts] start positions, same subset for both alleles */
			unsigned n_keep = min_keep + ntsm_synth_range(hc >> 32, n_starts - min_keep + 1);
			unsigned order[NTSM_SYNTH_W];
			for (unsigned s = 0; s < n_starts; ++s) order[s] = s;
			for (unsigned s = 0; s < n_keep; ++s) {          /* partial Fisher-Yates */
				unsigned j = s + ntsm_synth_range(ntsm_synth_rnd(seed, 12, base_ctr * 64 + s), n_starts - s);
				unsigned t = order[s]; order[s] = order[j]; order[j] = t;
			}
			bool keep[NTSM_SYNTH_W] = {};
			for (unsigned s = 0; s < n_keep; ++s) keep[order[s]] = true;
			/* reject the site if any kept k-mer collides with another of this site or an earlier one */
			uint64_t codes[2 * NTSM_SYNTH_W];
			unsigned nc = 0;
			bool clash = false;
			for (unsigned a = 0; a < 2 && !clash; ++a)
				for (unsigned s = 0; s < n_starts && !clash; ++s) {
					if (!keep[s]) continue;
					uint64_t c = canon_code(&win[a][s], k);
					if (seen.contains(c)) clash = true;
					for (unsigned q = 0; q < nc && !clash; ++q) if (codes[q] == c) clash = true;
					codes[nc++] = c;
				}
			if (clash) { ++redraws; continue; }
			for (unsigned q = 0; q < nc; ++q) seen.insert(codes[q]);
			if (windows) memcpy(windows + (uint64_t) i * 2 * NTSM_SYNTH_WSTRIDE, win, sizeof win);
			if (fasta_path) {
				for (unsigned a = 0; a < 2; ++a) {
					rec = ">rs" + std::to_string(i) + (a == 0 ? " ref\n" : " var\n");
					bool first = true;
					for (unsigned s = 0; s < n_starts; ++s) {
						if (!keep[s]) continue;
						if (!first) rec += 'N';
						first = false;
						for (unsigned p = 0; p < k; ++p) rec += (char) ntsm_synth_letter(win[a][s + p]);
					}
					rec += '\n';
					out.write(rec.data(), rec.size());
				}
			}
			break;
		}
	}
	out.close();
	if (n_kmers_out) *n_kmers_out = seen.size;
	(void) redraws;
	return 0;
}

static uint32_t thr24(double p) { double v = p * 16777216.0; return v <= 0 ? 0u : (v >= 16777216.0 ? 16777216u : (uint32_t) (v + 0.5)); }

void ntsm_synth_short_params(ntsm_synth_short *p, uint64_t seed, uint32_t read_len, uint32_t n_sites,
		double p_embed, double p_sub, double p_n)
{
	memset(p, 0, sizeof *p);
	p->seed = seed;
	p->read_len = read_len;
	p->n_sites = n_sites;
	double e = p_embed * 4294967296.0;
	p->embed_thr = n_sites == 0 ? 0u : (e >= 4294967295.0 ? 4294967295u : (e <= 0 ? 0u : (uint32_t) e));
	p->sub_thr = thr24(p_sub);
	p->n_thr = thr24(p_n);
}

void ntsm_synth_long_params(ntsm_synth_long *p, uint64_t seed, uint32_t n_sites, uint32_t spacing,
		double p_sub, double p_n)
{
	memset(p, 0, sizeof *p);
	p->seed = seed;
	p->n_sites = n_sites;
	p->spacing = spacing < NTSM_SYNTH_W ? NTSM_SYNTH_W : spacing;
	p->genome_len = (uint64_t) n_sites * p->spacing;
	p->sub_thr = thr24(p_sub);
	p->n_thr = thr24(p_n);
}

/* inverse normal CDF (Acklam's rational approximation); only used to tabulate quantiles */
static double inv_norm(double p)
{
	static const double a[] = { -3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
		1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00 };
	static const double b[] = { -5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
		6.680131188771972e+01, -1.328068155288572e+01 };
	static const double c[] = { -7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
		-2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00 };
	static const double d[] = { 7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00,
		3.754408661907416e+00 };
	double q, r;
	if (p < 0.02425) {
		q = sqrt(-2 * log(p));
		return (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) /
			((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
	}
	if (p > 1 - 0.02425) return -inv_norm(1 - p);
	q = p - 0.5; r = q * q;
	return (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
		(((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1);
}

void ntsm_synth_long_qtable(double mu, double sigma, uint32_t lo, uint32_t hi, uint32_t *q)
{
	for (int i = 0; i <= 256; ++i) {
		double p = i == 0 ? 1e-4 : (i == 256 ? 1 - 1e-4 : i / 256.0);
		double v = exp(mu + sigma * inv_norm(p));
		if (v < lo) v = lo;
		if (v > hi) v = hi;
		q[i] = (uint32_t) v;
		if (i && q[i] < q[i - 1]) q[i] = q[i - 1];
	}
}

uint64_t ntsm_synth_long_layout(uint64_t seed, const uint32_t *q, uint64_t r0, uint64_t n_reads, uint64_t *read_end)
{
	uint64_t off = 0;
	for (uint64_t i = 0; i < n_reads; ++i) {
		off += ntsm_synth_long_len(seed, q, r0 + i);
		read_end[i] = off;
		off += 1;
	}
	return off;
}

void ntsm_synth_short_fill_host(const ntsm_synth_short *p, const uint8_t *windows, uint64_t g0,
		uint64_t n, uint8_t *out)
{
	for (uint64_t i = 0; i < n; ++i) out[i] = ntsm_synth_short_byte(p, windows, g0 + i);
}

void ntsm_synth_long_fill_host(const ntsm_synth_long *p, const uint8_t *windows, const uint32_t *q,
		uint64_t r0, uint64_t n_reads, const uint64_t *read_end, uint8_t *out)
{
	(void) q;
	uint64_t start = 0;
	for (uint64_t i = 0; i < n_reads; ++i) {
		const uint32_t len = (uint32_t) (read_end[i] - start);
		for (uint32_t j = 0; j < len; ++j) out[start + j] = ntsm_synth_long_byte(p, windows, r0 + i, len, j);
		out[read_end[i]] = 'N';
		start = read_end[i] + 1;
	}
}

static void short_record(const ntsm_synth_short *p, const uint8_t *windows, unsigned qual_model, uint64_t r, std::vector<uint8_t> &seq,
		std::string &buf)
{
	const uint32_t L = p->read_len;
	ntsm_synth_short_fill_host(p, windows, r * ((uint64_t) L + 1), L, seq.data());
	buf += "@r";
	buf += std::to_string(r);
	buf += '\n';
	buf.append((const char *) seq.data(), L);
	buf += "\n+\n";
	if (qual_model == 0) buf.append(L, 'I');
	else for (uint32_t j = 0; j < L; ++j) buf += (char) ntsm_synth_qual_char(p->seed, qual_model, r, j, L);
	buf += '\n';
}

int ntsm_synth_short_write_fastq_q(const ntsm_synth_short *p, const uint8_t *windows, uint64_t r0,
		uint64_t n_reads, const char *path, unsigned qual_model)
{
	Out out;
	if (!out.open(path)) return -2;
	std::vector<uint8_t> seq((size_t) p->read_len + 1);
	std::string rec;
	for (uint64_t r = r0; r < r0 + n_reads; ++r) {
		rec.clear();
		short_record(p, windows, qual_model, r, seq, rec);
		out.write(rec.data(), rec.size());
	}
	out.close();
	return 0;
}

int ntsm_synth_short_write_fastq(const ntsm_synth_short *p, const uint8_t *windows, uint64_t r0,
		uint64_t n_reads, const char *path)
{
	return ntsm_synth_short_write_fastq_q(p, windows, r0, n_reads, path, 0);
}

/* The same bytes as ntsm_synth_short_write_fastq_q (plain output only), written by n_threads threads: the record of read
 * r is 2 + digits(r) + 1 + L + 3 + L + 1 bytes, so every thread knows where its slice of reads starts in the file and
 * writes it there with pwrite().  For the multi-gigabyte inputs of the end-to-end CLI measurements. */
int ntsm_synth_short_write_fastq_mt_q(const ntsm_synth_short *p, const uint8_t *windows, uint64_t r0,
		uint64_t n_reads, const char *path, unsigned n_threads, unsigned qual_model)
{
	size_t plen = strlen(path);
	if (n_threads <= 1 || (plen > 3 && !strcmp(path + plen - 3, ".gz")))
		return ntsm_synth_short_write_fastq_q(p, windows, r0, n_reads, path, qual_model);
	auto digits_sum = [](uint64_t a, uint64_t b) {             /* sum of the decimal lengths of a .. b-1 */
		uint64_t s = 0, lo = 1, d = 1;
		for (; d <= 20; ++d) {
			const uint64_t hi = d == 20 ? UINT64_MAX : lo * 10;   /* numbers with d digits: [lo, hi), and 0 counts as one digit */
			const uint64_t x = std::max(a, d == 1 ? 0 : lo), y = std::min(b, hi);
			if (y > x) s += (y - x) * d;
			if (d == 20 || hi > b) break;
			lo = hi;
		}
		return s;
	};
	const uint64_t L = p->read_len, fixed = 2 + 1 + L + 3 + L + 1;
	const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
	if (fd < 0) return -2;
	const uint64_t total = n_reads * fixed + digits_sum(r0, r0 + n_reads);
	if (ftruncate(fd, (off_t) total) != 0) { close(fd); return -2; }
	std::vector<std::thread> pool;
	std::vector<int> rcs(n_threads, 0);
	for (unsigned t = 0; t < n_threads; ++t)
		pool.emplace_back([&, t]() {
			const uint64_t a = r0 + n_reads * t / n_threads, b = r0 + n_reads * (t + 1) / n_threads;
			uint64_t off = (a - r0) * fixed + digits_sum(r0, a);
			std::vector<uint8_t> seq(L + 1);
			std::string buf;
			buf.reserve((8u << 20) + 2 * fixed);
			auto flush = [&]() {
				size_t done = 0;
				while (done < buf.size()) {
					const ssize_t w = pwrite(fd, buf.data() + done, buf.size() - done, (off_t) (off + done));
					if (w <= 0) { rcs[t] = -3; return; }
					done += (size_t) w;
				}
				off += buf.size();
				buf.clear();
			};
			for (uint64_t r = a; r < b && rcs[t] == 0; ++r) {
				short_record(p, windows, qual_model, r, seq, buf);
				if (buf.size() >= (8u << 20)) flush();
			}
			if (rcs[t] == 0) flush();
		});
	for (auto &th : pool) th.join();
	close(fd);
	for (int rc : rcs) if (rc) return rc;
	return 0;
}

int ntsm_synth_short_write_fastq_mt(const ntsm_synth_short *p, const uint8_t *windows, uint64_t r0,
		uint64_t n_reads, const char *path, unsigned n_threads)
{
	return ntsm_synth_short_write_fastq_mt_q(p, windows, r0, n_reads, path, n_threads, 0);
}

int ntsm_synth_long_write_fastq(const ntsm_synth_long *p, const uint8_t *windows, const uint32_t *q,
		uint64_t r0, uint64_t n_reads, const char *path)
{
	Out out;
	if (!out.open(path)) return -2;
	std::string rec;
	std::vector<uint8_t> seq;
	for (uint64_t r = r0; r < r0 + n_reads; ++r) {
		const uint32_t len = ntsm_synth_long_len(p->seed, q, r);
		seq.resize(len);
		for (uint32_t j = 0; j < len; ++j) seq[j] = ntsm_synth_long_byte(p, windows, r, len, j);
		rec = "@lr" + std::to_string(r) + "\n";
		rec.append((const char *) seq.data(), len);
		rec += "\n+\n";
		rec.append(len, 'I');
		rec += "\n";
		out.write(rec.data(), rec.size());
	}
	out.close();
	return 0;
}

} // extern "C"
