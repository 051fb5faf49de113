/*
 * synth.h -- counter-based synthetic workload generator shared by host (gcc/g++) and device
 * (hipcc) code.  Every output byte is a pure function of (seed, global byte offset), so the
 * host tool, the on-device generator used by bench.py, and the parity tests all see the same
 * reads without a 300 GB FASTQ ever existing.  Workload definitions follow SURVEY.md 8(d):
 *
 *   sites  "hs_n10_like": n_sites SNP sites, window w = 31, k = 19; centre base ref in {A,T},
 *          var in {C,G}; 3..13 of the 13 k-mer start positions kept (same subset for both
 *          alleles); records ">rs<i> ref" / ">rs<i> var", k-mers joined by 'N'
 *          (format: ntsm-scripts/filterRepetiveSNP.pl:41-51,96-99); duplicate-free by
 *          reject-and-redraw (the reference aborts on duplicates, src/FingerPrint.hpp:557-563).
 *   short reads: fixed length L (150); with probability p_embed a read carries one 31-bp allele
 *          window (ref/var 50/50, uniform offset, reverse-complemented w.p. 0.5); 1 %
 *          substitutions, 'N' w.p. 5e-4 per base.
 *   long reads: variable length from a 256-quantile table, cut from an implicit mini-genome in
 *          which every `spacing` bases start with a site window; substitution errors.
 *
 * Flat stream layout (the layout the C ABI consumes, include/ntsm_hip.h): read i is followed by
 * exactly one terminator byte 'N'; read_end[i] is the offset of that terminator.
 */
#ifndef NTSM_SYNTH_H
#define NTSM_SYNTH_H
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NTSM_HD __host__ __device__ __forceinline__
#else
#define NTSM_HD static inline
#endif

#define NTSM_SYNTH_W 31            /* site window, Options.h:59 (opt::window) */
#define NTSM_SYNTH_WSTRIDE 32      /* bytes per stored window (31 codes + pad) */

typedef struct {
	uint64_t seed;
	uint32_t read_len;             /* L */
	uint32_t n_sites;
	uint32_t embed_thr;            /* P(embed) * 2^32 */
	uint32_t sub_thr;              /* P(substitution) * 2^24 */
	uint32_t n_thr;                /* P(N) * 2^24 */
	uint32_t pad;
} ntsm_synth_short;

typedef struct {
	uint64_t seed;
	uint64_t genome_len;           /* n_sites * spacing */
	uint32_t n_sites;
	uint32_t spacing;              /* one site window every `spacing` genome bases */
	uint32_t sub_thr;              /* P(substitution) * 2^24 */
	uint32_t n_thr;                /* P(N) * 2^24 */
} ntsm_synth_long;

NTSM_HD uint64_t ntsm_synth_mix64(uint64_t x)
{
	x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
	x ^= x >> 27; x *= 0x94D049BB133111EBULL;
	x ^= x >> 31;
	return x;
}

/* one 64-bit draw addressed by (seed, stream, counter) */
NTSM_HD uint64_t ntsm_synth_rnd(uint64_t seed, uint64_t stream, uint64_t ctr)
{
	uint64_t s = ntsm_synth_mix64(seed + (stream + 1) * 0x9E3779B97F4A7C15ULL);
	return ntsm_synth_mix64(s ^ (ctr * 0xD1342543DE82EF95ULL + 0x2545F4914F6CDD1DULL));
}

NTSM_HD uint32_t ntsm_synth_range(uint64_t r32, uint32_t n)   /* uniform in [0,n) from 32 bits */
{
	return (uint32_t) (((r32 & 0xFFFFFFFFULL) * (uint64_t) n) >> 32);
}

NTSM_HD unsigned char ntsm_synth_letter(unsigned code) { return (unsigned char) ("ACGT"[code & 3]); }

/* Byte g of the short-read flat stream.  windows = n_sites * 2 * NTSM_SYNTH_WSTRIDE 2-bit codes. */
NTSM_HD unsigned char ntsm_synth_short_byte(const ntsm_synth_short *p, const unsigned char *windows,
		uint64_t g)
{
	const uint64_t stride = (uint64_t) p->read_len + 1;
	const uint64_t r = g / stride;
	const uint32_t j = (uint32_t) (g - r * stride);
	if (j == p->read_len) return 'N';                           /* read terminator */
	const uint64_t hr = ntsm_synth_rnd(p->seed, 1, r);
	const uint64_t hb = ntsm_synth_rnd(p->seed, 3, g);
	unsigned base = (unsigned) (hb & 3);
	if ((uint32_t) hr < p->embed_thr && p->read_len >= NTSM_SYNTH_W) {
		const uint64_t hs = ntsm_synth_rnd(p->seed, 2, r);
		const uint32_t off = ntsm_synth_range(hs >> 32, p->read_len - NTSM_SYNTH_W + 1);
		if (j >= off && j < off + NTSM_SYNTH_W) {
			const uint32_t site = ntsm_synth_range(hs, p->n_sites);
			const unsigned allele = (unsigned) (hr >> 32) & 1u;
			const unsigned rc = (unsigned) (hr >> 33) & 1u;
			const unsigned char *w = windows + ((uint64_t) site * 2 + allele) * NTSM_SYNTH_WSTRIDE;
			const uint32_t t = j - off;
			base = rc ? 3u - w[NTSM_SYNTH_W - 1 - t] : w[t];
		}
	}
	if (((hb >> 8) & 0xFFFFFF) < p->sub_thr) base = (base + 1 + (unsigned) ((hb >> 2) & 3) % 3) & 3;
	if (((hb >> 32) & 0xFFFFFF) < p->n_thr) return 'N';
	return ntsm_synth_letter(base);
}

/* Quality character of base j of short read r under quality model `model` (FASTQ text only: the count path never sees it).
 *   0  constant 'I' (rounds 1-4; compresses 6:1 and is a copy of a copy ... of one line under DEFLATE)
 *   1  "Illumina-like", 8-level binned (the instrument's Q-score binning: Phred 2, 6, 15, 22, 27, 33, 37, 40 = # ' 0 7 < B F I).
 *      A read draws a level (one read in eight is a poor one) and a decay strength; the expected score falls with the square
 *      of the position (late cycles are worse); every base draws its own deviation (mostly 0, sometimes -2 / -5 / -10 / -20,
 *      rarely the floor) and a deviation is held over the next base with probability 1/2, so low scores come in short runs
 *      as they do on an instrument; the result is rounded down to its bin.  All eight characters occur, the text compresses
 *      about 3.5 : 1 under gzip -6 like real binned short-read FASTQ (constant 'I': 6 : 1), and a DEFLATE decoder meets literals
 *      and short matches on the quality lines instead of one 150-byte copy per record.
 *   2  the same without binning (Phred 2..41, ~39 distinct characters, 4.4 bits per score: 2.6 : 1 -- older instruments).
 * Counter-based like the bases: a pure function of (seed, r, j). */
NTSM_HD unsigned char ntsm_synth_qual_char(uint64_t seed, unsigned model, uint64_t r, uint32_t j, uint32_t read_len)
{
	if (model == 0) return 'I';
	const uint64_t hr = ntsm_synth_rnd(seed, 10, r);
	const int level = 41 - (((hr & 7) == 0) ? 6 + (int) ((hr >> 3) & 7) : 0);                           /* poor reads 28..35 */
	const uint32_t decay = 4 + (uint32_t) ((hr >> 8) & 15);                                              /* 4..19 Phred lost by the last cycle */
	const uint64_t jj = (uint64_t) j * j;
	const int mean = level - (int) ((decay * jj) / ((uint64_t) read_len * read_len));
	/* deviation of base j; with probability 1/2 the deviation of base j-1 is held instead (runs) */
	uint64_t hb = ntsm_synth_rnd(seed, 9, r * 0x100000001B3ULL + j);
	if (j > 0 && (hb >> 63)) hb = ntsm_synth_rnd(seed, 9, r * 0x100000001B3ULL + j - 1);
	const unsigned t = (unsigned) (hb & 0xFF);
	int q = mean - (t < 200 ? 0 : t < 224 ? 2 : t < 240 ? 5 : t < 249 ? 10 : t < 254 ? 20 : 64);
	if (q < 2) q = 2;
	if (q > 41) q = 41;
	if (model == 1) q = q >= 40 ? 40 : q >= 37 ? 37 : q >= 33 ? 33 : q >= 27 ? 27 : q >= 22 ? 22 : q >= 15 ? 15 : q >= 6 ? 6 : 2;
	return (unsigned char) (33 + q);
}

/* Length of long read r from a 257-entry quantile table (host-built, monotone). */
NTSM_HD uint32_t ntsm_synth_long_len(uint64_t seed, const uint32_t *qtable, uint64_t r)
{
	const uint64_t u = ntsm_synth_rnd(seed, 4, r);
	const uint32_t q = (uint32_t) (u >> 56);
	const uint64_t lo = qtable[q], hi = qtable[q + 1];
	return (uint32_t) (lo + (((hi - lo) * ((u >> 24) & 0xFFFFFFFFULL)) >> 32));
}

/* Base j of long read r (length len). */
NTSM_HD unsigned char ntsm_synth_long_byte(const ntsm_synth_long *p, const unsigned char *windows,
		uint64_t r, uint32_t len, uint32_t j)
{
	const uint64_t hr = ntsm_synth_rnd(p->seed, 5, r);
	const uint64_t start = (hr >> 1) % (p->genome_len - len);
	const unsigned rc = (unsigned) hr & 1u;
	const uint64_t gp = start + (rc ? (uint64_t) (len - 1 - j) : (uint64_t) j);
	const uint64_t block = gp / p->spacing;
	const uint32_t o = (uint32_t) (gp - block * p->spacing);
	unsigned base;
	if (o < NTSM_SYNTH_W) {
		const uint32_t site = (uint32_t) (block % p->n_sites);
		const unsigned allele = (unsigned) (ntsm_synth_rnd(p->seed, 7, r * 0x100000001B3ULL + block) & 1u);
		base = windows[((uint64_t) site * 2 + allele) * NTSM_SYNTH_WSTRIDE + o];
	} else {
		base = (unsigned) (ntsm_synth_rnd(p->seed, 6, gp) & 3);
	}
	if (rc) base = 3u - base;
	const uint64_t hb = ntsm_synth_rnd(p->seed, 8, (r << 24) ^ (uint64_t) j ^ (r >> 40));
	if (((hb >> 8) & 0xFFFFFF) < p->sub_thr) base = (base + 1 + (unsigned) ((hb >> 2) & 3) % 3) & 3;
	if (((hb >> 32) & 0xFFFFFF) < p->n_thr) return 'N';
	return ntsm_synth_letter(base);
}

#endif /* NTSM_SYNTH_H */
