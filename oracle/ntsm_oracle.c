/*
 * oracle/ntsm_oracle.c -- TEST INFRASTRUCTURE ONLY (see ntsm_oracle.h).
 *
 * CPU restatement of the reference ntsmCount counting path.  Parity is pinned against the
 * unmodified reference compiled in place (oracle/_ref, tests/golden/).  Single-threaded:
 * the reference's only parallelism is an OpenMP loop over input files
 * (src/FingerPrint.hpp:47), and one thread in argv order is its only deterministic schedule.
 */
#define _GNU_SOURCE
#include "ntsm_oracle.h"

#include <ctype.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ============================================================================================
 * k-mer arithmetic -- vendor/KseqHashIterator.hpp
 * ========================================================================================== */

/* vendor/KseqHashIterator.hpp:114-127.  The table maps bytes 0..3 to themselves, the letters
 * A C G T U (either case) to 0 1 2 3 3, and every other byte to 4 (= invalid, "N"). */
int ntsm_oracle_nt4(unsigned char b)
{
	if (b < 4) return b;
	switch (b) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': case 'U': case 'u': return 3;
	default: return 4;
	}
}

/* vendor/KseqHashIterator.hpp:29.  k = 32 shifts by 64 (undefined in C); the reference binary on
 * x86-64 evaluates it as a shift by 0, giving mask = 0, which is what is mirrored here. */
uint64_t ntsm_oracle_mask(unsigned k)
{
	unsigned sh = (k * 2u) & 63u;
	return (1ULL << sh) - 1;
}

/* vendor/KseqHashIterator.hpp:129-139: seven masked mixing steps, a bijection on 2k bits. */
uint64_t ntsm_oracle_hash64(uint64_t key, uint64_t mask)
{
	key = (~key + (key << 21)) & mask;
	key ^= key >> 24;
	key = (key + (key << 3) + (key << 8)) & mask;
	key ^= key >> 14;
	key = (key + (key << 2) + (key << 4)) & mask;
	key ^= key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

/* vendor/KseqHashIterator.hpp:28-33 and init() :80-85 */
void ntsm_oracle_iter_init(ntsm_oracle_iter *it, const char *seq, uint64_t len, unsigned k)
{
	memset(it, 0, sizeof(*it));
	it->seq = (const unsigned char *) seq;
	it->len = len;
	it->k = k;
	it->mask = ntsm_oracle_mask(k);
	it->shift = ((uint64_t) k - 1) * 2;            /* :30; k = 0 wraps exactly like the reference */
}

/* vendor/KseqHashIterator.hpp:87-112.  One call == one `++itr` (or the ctor's first next()):
 * consume bytes until a window of k consecutive valid bases ends at the byte just read. */
int ntsm_oracle_iter_next(ntsm_oracle_iter *it)
{
	while (it->pos < it->len) {
		int c = ntsm_oracle_nt4(it->seq[it->pos++]);
		if (c < 4) {
			it->fw = ((it->fw << 2) | (uint64_t) c) & it->mask;                          /* :99  */
			it->rv = (it->rv >> 2) | ((uint64_t) (3 - c) << (it->shift & 63));            /* :100 */
			if (++it->run >= it->k) {                                                    /* :101 */
				it->canon = it->fw < it->rv ? it->fw : it->rv;                           /* :102 */
				it->hv = ntsm_oracle_hash64(it->canon, it->mask);
				return 1;
			}
		} else {                                                                         /* :106 */
			it->run = 0;
			it->fw = it->rv = 0;
		}
	}
	return 0;
}

uint64_t ntsm_oracle_kmers(const char *seq, uint64_t len, unsigned k, uint64_t *out_canon,
		uint64_t *out_hv, uint64_t *out_pos, uint64_t cap)
{
	ntsm_oracle_iter it;
	uint64_t n = 0;
	ntsm_oracle_iter_init(&it, seq, len, k);
	while (ntsm_oracle_iter_next(&it)) {
		if (n < cap) {
			if (out_canon) out_canon[n] = it.canon;
			if (out_hv) out_hv[n] = it.hv;
			if (out_pos) out_pos[n] = it.pos;
		}
		++n;
	}
	return n;
}

/* ============================================================================================
 * Record reader -- vendor/kseq.h over gzread
 * ========================================================================================== */

#define RD_BUFSIZE 16384                     /* vendor/kseq.h:229 */

typedef struct {
	char *s;
	size_t l, m;
} rd_str;

struct ntsm_oracle_reader {
	gzFile f;
	unsigned char *buf;
	int beg, end, eof;                       /* kstream_t, vendor/kseq.h:40-45 */
	int last_char;                           /* kseq_t.last_char, :223 */
	rd_str name, comment, seq, qual;
};

static void rd_reserve(rd_str *s, size_t need)
{
	if (s->m < need) {
		size_t m = s->m ? s->m : 64;
		while (m < need) m *= 2;
		s->s = (char *) realloc(s->s, m);
		s->m = m;
	}
}

/* ks_getc, vendor/kseq.h:67-79 */
static int rd_getc(ntsm_oracle_reader *r)
{
	if (r->end < 0) return -3;
	if (r->eof && r->beg >= r->end) return -1;
	if (r->beg >= r->end) {
		r->beg = 0;
		r->end = gzread(r->f, r->buf, RD_BUFSIZE);
		if (r->end == 0) { r->eof = 1; return -1; }
		if (r->end < 0) { r->eof = 1; return -3; }
	}
	return r->buf[r->beg++];
}

enum { SEP_SPACE = 0, SEP_LINE = 2 };

/* ks_getuntil2, vendor/kseq.h:93-144, for the two separators kseq_read uses.
 * Quirks kept on purpose: the "-1 at EOF" return happens before the trailing-CR strip; the CR
 * strip looks at the whole accumulated string and needs length > 1. */
static int64_t rd_getuntil(ntsm_oracle_reader *r, int sep, rd_str *str, int *dret, int append)
{
	int gotany = 0;
	if (dret) *dret = 0;
	if (!append) str->l = 0;
	for (;;) {
		int i;
		if (r->end < 0) return -3;
		if (r->beg >= r->end) {
			if (r->eof) break;
			r->beg = 0;
			r->end = gzread(r->f, r->buf, RD_BUFSIZE);
			if (r->end == 0) { r->eof = 1; break; }
			if (r->end == -1) { r->eof = 1; return -3; }
		}
		if (sep == SEP_LINE) {
			for (i = r->beg; i < r->end; ++i) if (r->buf[i] == '\n') break;
		} else {
			for (i = r->beg; i < r->end; ++i) if (isspace(r->buf[i])) break;
		}
		rd_reserve(str, str->l + (size_t) (i - r->beg) + 1);
		gotany = 1;
		memcpy(str->s + str->l, r->buf + r->beg, (size_t) (i - r->beg));
		str->l += (size_t) (i - r->beg);
		r->beg = i + 1;
		if (i < r->end) {
			if (dret) *dret = r->buf[i];
			break;
		}
	}
	if (!gotany && r->eof && r->beg >= r->end) return -1;
	if (str->s == NULL) rd_reserve(str, 1);
	else if (sep == SEP_LINE && str->l > 1 && str->s[str->l - 1] == '\r') --str->l;
	str->s[str->l] = '\0';
	return (int64_t) str->l;
}

ntsm_oracle_reader *ntsm_oracle_reader_open(const char *path)
{
	gzFile f = gzopen(path, "r");
	if (f == Z_NULL) return NULL;
	ntsm_oracle_reader *r = (ntsm_oracle_reader *) calloc(1, sizeof(*r));
	r->f = f;
	r->buf = (unsigned char *) malloc(RD_BUFSIZE);
	return r;
}

void ntsm_oracle_reader_close(ntsm_oracle_reader *r)
{
	if (!r) return;
	free(r->name.s); free(r->comment.s); free(r->seq.s); free(r->qual.s);
	free(r->buf);
	gzclose(r->f);
	free(r);
}

const char *ntsm_oracle_reader_seq(const ntsm_oracle_reader *r) { return r->seq.s ? r->seq.s : ""; }
const char *ntsm_oracle_reader_name(const ntsm_oracle_reader *r) { return r->name.s ? r->name.s : ""; }

/* kseq_read, vendor/kseq.h:177-219 */
int64_t ntsm_oracle_reader_next(ntsm_oracle_reader *r)
{
	int c;
	int64_t ret;
	if (r->last_char == 0) {                                  /* :181-185 jump to next header */
		while ((c = rd_getc(r)) >= 0 && c != '>' && c != '@') { }
		if (c < 0) return c;
		r->last_char = c;
	}
	r->comment.l = r->seq.l = r->qual.l = 0;                  /* :186 */
	if ((ret = rd_getuntil(r, SEP_SPACE, &r->name, &c, 0)) < 0) return ret;          /* :187 */
	if (c != '\n') rd_getuntil(r, SEP_LINE, &r->comment, NULL, 0);                    /* :188 */
	rd_reserve(&r->seq, 256);                                 /* :189-192 */
	while ((c = rd_getc(r)) >= 0 && c != '>' && c != '+' && c != '@') {              /* :193 */
		if (c == '\n') continue;                              /* :194 */
		rd_reserve(&r->seq, r->seq.l + 2);
		r->seq.s[r->seq.l++] = (char) c;                      /* :195 */
		rd_getuntil(r, SEP_LINE, &r->seq, NULL, 1);           /* :196 */
	}
	if (c == '>' || c == '@') r->last_char = c;               /* :198 */
	rd_reserve(&r->seq, r->seq.l + 2);
	r->seq.s[r->seq.l] = '\0';                                /* :204 */
	if (c != '+') return (int64_t) r->seq.l;                  /* :205-206 FASTA */
	while ((c = rd_getc(r)) >= 0 && c != '\n') { }            /* :211 rest of '+' line */
	if (c == -1) return -2;                                   /* :212 */
	while (rd_getuntil(r, SEP_LINE, &r->qual, NULL, 1) >= 0 && r->qual.l < r->seq.l) { } /* :213 */
	r->last_char = 0;                                         /* :215 */
	if (r->seq.l != r->qual.l) return -2;                     /* :216 */
	return (int64_t) r->seq.l;
}

/* ============================================================================================
 * Count table -- tsl::robin_map<uint64_t,size_t> as the reference instantiates it
 * (src/FingerPrint.hpp:466): std::hash<uint64_t> (identity) & (bucket_count-1)
 * (vendor/tsl/robin_growth_policy.h:114-115), max load factor 0.5, Robin-Hood linear probing,
 * 24-byte buckets {int16 dist; bool last; pair<u64,size_t>} (vendor/tsl/robin_hash.h:1157-1175).
 * The geometry is kept so that this restatement is also a fair CPU timing baseline.
 * ========================================================================================== */

typedef struct {
	int16_t dist;                            /* -1 = empty */
	uint8_t last;
	uint64_t key;
	uint64_t val;
} rh_bucket;                                 /* 24 bytes */

typedef struct {
	rh_bucket *b;
	uint64_t nb, mask, size;
} rh_map;

static void rh_alloc(rh_map *m, uint64_t nb)
{
	m->b = (rh_bucket *) malloc((size_t) nb * sizeof(rh_bucket));
	for (uint64_t i = 0; i < nb; ++i) { m->b[i].dist = -1; m->b[i].last = 0; }
	m->nb = nb;
	m->mask = nb - 1;
	m->size = 0;
}

static inline rh_bucket *rh_find(const rh_map *m, uint64_t key)
{
	if (m->nb == 0) return NULL;
	uint64_t i = key & m->mask;
	int16_t d = 0;
	while (d <= m->b[i].dist) {
		if (m->b[i].key == key) return &m->b[i];
		i = (i + 1) & m->mask;
		++d;
	}
	return NULL;
}

static void rh_insert_raw(rh_map *m, uint64_t key, uint64_t val)
{
	uint64_t i = key & m->mask;
	int16_t d = 0;
	for (;;) {
		rh_bucket *b = &m->b[i];
		if (b->dist < 0) { b->dist = d; b->key = key; b->val = val; return; }
		if (b->dist < d) {                   /* rob the richer entry */
			uint64_t tk = b->key, tv = b->val; int16_t td = b->dist;
			b->key = key; b->val = val; b->dist = d;
			key = tk; val = tv; d = td;
		}
		i = (i + 1) & m->mask;
		++d;
	}
}

static void rh_insert(rh_map *m, uint64_t key, uint64_t val)
{
	if (m->nb == 0) rh_alloc(m, 16);
	if ((double) (m->size + 1) > (double) m->nb * 0.5) {          /* max_load_factor 0.5 */
		rh_map n;
		rh_alloc(&n, m->nb * 2);
		for (uint64_t i = 0; i < m->nb; ++i)
			if (m->b[i].dist >= 0) rh_insert_raw(&n, m->b[i].key, m->b[i].val);
		n.size = m->size;
		free(m->b);
		*m = n;
	}
	rh_insert_raw(m, key, val);
	m->size++;
}

static void rh_erase(rh_map *m, uint64_t key)                      /* backward-shift deletion */
{
	rh_bucket *b = rh_find(m, key);
	if (!b) return;
	uint64_t i = (uint64_t) (b - m->b);
	for (;;) {
		uint64_t n = (i + 1) & m->mask;
		if (m->b[n].dist <= 0) break;
		m->b[i] = m->b[n];
		m->b[i].dist--;
		i = n;
	}
	m->b[i].dist = -1;
	m->size--;
}

/* ============================================================================================
 * FingerPrint -- src/FingerPrint.hpp
 * ========================================================================================== */

typedef struct {
	uint64_t *hv, *canon;
	size_t n, m;
} kvec;

static void kvec_push(kvec *v, uint64_t hv, uint64_t canon)
{
	if (v->n == v->m) {
		v->m = v->m ? v->m * 2 : 16;
		v->hv = (uint64_t *) realloc(v->hv, v->m * sizeof(uint64_t));
		v->canon = (uint64_t *) realloc(v->canon, v->m * sizeof(uint64_t));
	}
	v->hv[v->n] = hv;
	v->canon[v->n] = canon;
	v->n++;
}

struct ntsm_oracle_fp {
	unsigned k;
	uint64_t total_hits, total_kmers, max_hits, total_bases, reads_processed;
	int early_term;
	rh_map counts;                           /* m_counts */
	char **ids;                              /* m_alleleIDs */
	kvec *ref, *var;                         /* m_alleleIDToKmerRef / Var */
	size_t n_ids, n_ref, n_var, cap;
};

static void fp_grow(ntsm_oracle_fp *fp)
{
	if (fp->n_ref + 1 >= fp->cap || fp->n_var + 1 >= fp->cap || fp->n_ids + 1 >= fp->cap) {
		size_t nc = fp->cap ? fp->cap * 2 : 1024;
		fp->ids = (char **) realloc(fp->ids, nc * sizeof(char *));
		fp->ref = (kvec *) realloc(fp->ref, nc * sizeof(kvec));
		fp->var = (kvec *) realloc(fp->var, nc * sizeof(kvec));
		memset(fp->ref + fp->cap, 0, (nc - fp->cap) * sizeof(kvec));
		memset(fp->var + fp->cap, 0, (nc - fp->cap) * sizeof(kvec));
		fp->cap = nc;
	}
}

/* double -> uint64_t the way the reference binary does it for `(size * covThresh) / 2`
 * (src/FingerPrint.hpp:41-43).  Out-of-range values are undefined in C++; every such value
 * (DBL_MAX default, negatives) yields a threshold that can never trip, which is what matters. */
static uint64_t to_u64_like_ref(double x)
{
	if (!(x == x)) return 0;
	if (x >= 18446744073709551616.0) return 0;              /* inf / huge: "never" (0 = disabled) */
	if (x < 0) return UINT64_MAX;                            /* wraps to a huge value: "never"     */
	return (uint64_t) x;
}

ntsm_oracle_fp *ntsm_oracle_fp_create(const char *sites_path, unsigned k, double cov_thresh,
		int dupes, FILE *err)
{
	ntsm_oracle_reader *r = ntsm_oracle_reader_open(sites_path);   /* :491 */
	if (!r) {
		if (err) fprintf(err, "file %s cannot be opened\n", sites_path);
		return NULL;
	}
	ntsm_oracle_fp *fp = (ntsm_oracle_fp *) calloc(1, sizeof(*fp));
	fp->k = k;
	uint64_t *dup = NULL; size_t n_dup = 0, m_dup = 0;
	size_t entry = 0;
	int64_t l = ntsm_oracle_reader_next(r);                        /* :508 */
	while (l >= 0) {                                               /* :510 */
		int is_ref = (entry % 2 == 0);
		fp_grow(fp);
		kvec *vec = is_ref ? &fp->ref[fp->n_ref++] : &fp->var[fp->n_var++];
		ntsm_oracle_iter it;
		ntsm_oracle_iter_init(&it, ntsm_oracle_reader_seq(r), (uint64_t) l, k);
		while (ntsm_oracle_iter_next(&it)) {                       /* :517 / :539 */
			if (rh_find(&fp->counts, it.hv)) {                     /* :521 / :543 */
				if (err) fprintf(err, "Warning: %s of %s file has a k-mer collision at pos: %llu\n",
						ntsm_oracle_reader_name(r), is_ref ? "REF" : "VAR",
						(unsigned long long) it.pos);
				if (n_dup == m_dup) {
					m_dup = m_dup ? m_dup * 2 : 16;
					dup = (uint64_t *) realloc(dup, m_dup * sizeof(uint64_t));
				}
				dup[n_dup++] = it.hv;                              /* dupes.insert :525 */
			} else {
				kvec_push(vec, it.hv, it.canon);                   /* :527 */
				rh_insert(&fp->counts, it.hv, 0);                  /* :528 */
			}
		}
		if (is_ref) fp->ids[fp->n_ids++] = strdup(ntsm_oracle_reader_name(r));   /* :531 */
		l = ntsm_oracle_reader_next(r);
		entry++;
	}
	ntsm_oracle_reader_close(r);
	if (!dupes)                                                    /* :557-563 */
		for (size_t i = 0; i < n_dup; ++i) rh_erase(&fp->counts, dup[i]);
	free(dup);
	if (cov_thresh != 0)                                           /* :41-43 */
		fp->max_hits = to_u64_like_ref(((double) fp->counts.size * cov_thresh) / 2);
	return fp;
}

void ntsm_oracle_fp_destroy(ntsm_oracle_fp *fp)
{
	if (!fp) return;
	for (size_t i = 0; i < fp->n_ids; ++i) free(fp->ids[i]);
	for (size_t i = 0; i < fp->n_ref; ++i) { free(fp->ref[i].hv); free(fp->ref[i].canon); }
	for (size_t i = 0; i < fp->n_var; ++i) { free(fp->var[i].hv); free(fp->var[i].canon); }
	free(fp->ids); free(fp->ref); free(fp->var);
	free(fp->counts.b);
	free(fp);
}

/* src/FingerPrint.hpp:89-103 with its third parameter (`unsigned multiplier = 1`, :89): a hit adds `multiplier` to the
 * k-mer's 64-bit count and to m_totalCounts (:94-97); m_totalKmers and m_totalBases do not see it (:98-102) */
void ntsm_oracle_fp_insert_count_mult(ntsm_oracle_fp *fp, const char *seq, uint64_t len, unsigned multiplier)
{
	ntsm_oracle_iter it;
	ntsm_oracle_iter_init(&it, seq, len, fp->k);
	while (ntsm_oracle_iter_next(&it)) {
		rh_bucket *b = rh_find(&fp->counts, it.hv);
		if (b) {
			b->val += multiplier;
			fp->total_hits += multiplier;
		}
		fp->total_kmers++;
	}
	fp->total_bases += len;
}

/* src/FingerPrint.hpp:89-103, multiplier = 1 (every call the reference itself makes) */
void ntsm_oracle_fp_insert_count(ntsm_oracle_fp *fp, const char *seq, uint64_t len)
{
	ntsm_oracle_fp_insert_count_mult(fp, seq, len, 1u);
}

/* src/FingerPrint.hpp:473-488: the -m check runs after each whole read, strict '>' */
int ntsm_oracle_fp_process_read(ntsm_oracle_fp *fp, const char *seq, uint64_t len)
{
	ntsm_oracle_fp_insert_count(fp, seq, len);
	fp->reads_processed++;
	if (fp->max_hits != 0 && fp->total_hits > fp->max_hits) fp->early_term = 1;
	return fp->early_term;
}

/* src/FingerPrint.hpp:46-87.  After the threshold trips, later files are still opened and
 * their first record parsed, but nothing is counted (the `while` at :66 fails immediately). */
int ntsm_oracle_fp_compute_counts(ntsm_oracle_fp *fp, const char *const *files, int n_files, FILE *err)
{
	for (int i = 0; i < n_files; ++i) {
		ntsm_oracle_reader *r = ntsm_oracle_reader_open(files[i]);
		if (!r) {
			if (err) fprintf(err, "file %s cannot be opened\n", files[i]);
			return 1;
		}
		int64_t l = ntsm_oracle_reader_next(r);
		while (l >= 0 && !fp->early_term) {
			ntsm_oracle_fp_process_read(fp, ntsm_oracle_reader_seq(r), (uint64_t) l);
			l = ntsm_oracle_reader_next(r);
		}
		ntsm_oracle_reader_close(r);
	}
	if (fp->early_term && err) fprintf(err, "Reached desired (-m) threshold\n");   /* :84-86 */
	return 0;
}

/* src/FingerPrint.hpp:261-311.  Per-k-mer counts pass through `unsigned` (:282, :289): each
 * value is truncated mod 2^32 before max/sum, and the sums wrap mod 2^32. */
int ntsm_oracle_fp_print_counts(ntsm_oracle_fp *fp, FILE *out)
{
	fprintf(out, "#@TK\t%llu\n#@KS\t%u", (unsigned long long) fp->total_kmers, fp->k);  /* :261-268 */
	fprintf(out, "\n#locusID\tcountAT\tcountCG\tsumAT\tsumCG\tdistinctAT\tdistinctCG\n");
	for (size_t i = 0; i < fp->n_ids; ++i) {
		if (i >= fp->n_ref || i >= fp->n_var) return -1;           /* vector::at() would throw */
		const kvec *a[2] = { &fp->ref[i], &fp->var[i] };
		unsigned mx[2] = { 0, 0 }, sm[2] = { 0, 0 };
		for (int s = 0; s < 2; ++s)
			for (size_t j = 0; j < a[s]->n; ++j) {
				rh_bucket *b = rh_find(&fp->counts, a[s]->hv[j]);
				if (!b) return -1;                                 /* robin_map::at throws */
				unsigned f = (unsigned) b->val;
				if (mx[s] < f) mx[s] = f;
				sm[s] += f;
			}
		fprintf(out, "%s\t%u\t%u\t%u\t%u\t%zu\t%zu\n", fp->ids[i], mx[0], mx[1], sm[0], sm[1],
				a[0]->n, a[1]->n);
	}
	return 0;
}

/* src/FingerPrint.hpp:389-413 */
static unsigned fp_sites_covered(ntsm_oracle_fp *fp)
{
	unsigned count = 0;
	for (size_t i = 0; i < fp->n_ids; ++i) {
		int hit = 0;
		const kvec *a[2] = { i < fp->n_ref ? &fp->ref[i] : NULL, i < fp->n_var ? &fp->var[i] : NULL };
		for (int s = 0; s < 2 && !hit; ++s)
			for (size_t j = 0; a[s] && j < a[s]->n; ++j) {
				rh_bucket *b = rh_find(&fp->counts, a[s]->hv[j]);
				if (b && b->val > 0) { hit = 1; break; }
			}
		count += (unsigned) hit;
	}
	return count;
}

/* src/FingerPrint.hpp:313-349 */
int ntsm_oracle_fp_info_summary(ntsm_oracle_fp *fp, char *buf, size_t cap, FILE *err)
{
	unsigned cov = fp_sites_covered(fp);
	int n = snprintf(buf, cap,
			"Total Bases Considered: %llu\n"
			"Total k-mers Considered: %llu\n"
			"Total k-mers Recorded: %llu\n"
			"Distinct k-mers in initial set: %llu\n"
			"Total Sites: %zu\n"
			"Sites Covered by at least one k-mer: %u\n",
			(unsigned long long) fp->total_bases, (unsigned long long) fp->total_kmers,
			(unsigned long long) fp->total_hits, (unsigned long long) fp->counts.size,
			fp->n_ref, cov);
	double cov_per = (double) cov / (double) fp->n_ref;
	if (cov_per < 0.75f && err)                                    /* Options.h:31 siteCovThreshold (float) */
		fprintf(err, "Warning: site coverage is : %g(<75%%). Data may be sorted or sparse along the "
				"genome. Any PCA projection may be inaccurate.\n", cov_per);
	return n;
}

uint64_t ntsm_oracle_fp_total_kmers(const ntsm_oracle_fp *fp) { return fp->total_kmers; }
uint64_t ntsm_oracle_fp_total_hits(const ntsm_oracle_fp *fp) { return fp->total_hits; }
uint64_t ntsm_oracle_fp_total_bases(const ntsm_oracle_fp *fp) { return fp->total_bases; }
uint64_t ntsm_oracle_fp_max_hits(const ntsm_oracle_fp *fp) { return fp->max_hits; }
int ntsm_oracle_fp_early_term(const ntsm_oracle_fp *fp) { return fp->early_term; }
uint64_t ntsm_oracle_fp_reads_processed(const ntsm_oracle_fp *fp) { return fp->reads_processed; }
uint64_t ntsm_oracle_fp_n_distinct(const ntsm_oracle_fp *fp) { return fp->counts.size; }
uint64_t ntsm_oracle_fp_n_sites(const ntsm_oracle_fp *fp) { return fp->n_ids; }

uint64_t ntsm_oracle_fp_kmers(const ntsm_oracle_fp *fp, uint64_t *canon, uint64_t *hv,
		uint64_t *count, uint64_t cap)
{
	uint64_t n = 0;
	size_t recs = fp->n_ref > fp->n_var ? fp->n_ref : fp->n_var;
	for (size_t i = 0; i < recs; ++i)
		for (int s = 0; s < 2; ++s) {
			const kvec *v = s == 0 ? (i < fp->n_ref ? &fp->ref[i] : NULL) : (i < fp->n_var ? &fp->var[i] : NULL);
			for (size_t j = 0; v && j < v->n; ++j, ++n) {
				if (n >= cap) continue;
				if (canon) canon[n] = v->canon[j];
				if (hv) hv[n] = v->hv[j];
				if (count) {
					rh_bucket *b = rh_find(&fp->counts, v->hv[j]);
					count[n] = b ? b->val : 0;
				}
			}
		}
	return n;
}
