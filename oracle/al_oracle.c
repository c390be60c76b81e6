/* TEST INFRASTRUCTURE ONLY -- see al_oracle.h.  Parity: pinned against oracle/_ref/mm2ref and tests/golden/.
 *
 * Scalar restatement of the reference's `-ax sr` path.  "ref:" comments give the reference file:line
 * (relative to /root/reference/src/minimap2-master_remapping/) each block follows.
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include <math.h>
#include <zlib.h>
#include <pthread.h>
#include "al_oracle.h"

#define SEED_TANDEM   (1ULL<<42)   /* ref: mmpriv.h:16-22 */
#define SEED_SEG_SHIFT 48
#define SEED_SEG_MASK  (0xffULL<<SEED_SEG_SHIFT)
#define PARENT_UNSET   (-1)
#define PARENT_TMP_PRI (-2)
#define KSW_NEG_INF   -0x40000000
#define EZ_RIGHT      0x02
#define EZ_APPROX_MAX 0x08
#define EZ_EXTZ_ONLY  0x40
#define EZ_REV_CIGAR  0x80

/* ---------------------------------------------------------------- tables (ref: sketch.c:9-26, bseq.c) */
static uint8_t nt4[256], comp_tab[256];
static pthread_once_t tab_once = PTHREAD_ONCE_INIT;
static void init_tabs(void)
{
	int i;
	const char *a = "ACGTUMRWSYKVHDBN", *b = "TGCAAKYWSRMBDHVN";
	for (i = 0; i < 256; ++i) nt4[i] = 4, comp_tab[i] = (uint8_t)i;
	nt4['A'] = nt4['a'] = 0; nt4['C'] = nt4['c'] = 1; nt4['G'] = nt4['g'] = 2;
	nt4['T'] = nt4['t'] = 3; nt4['U'] = nt4['u'] = 3;
	nt4[0] = 0; nt4[1] = 1; nt4[2] = 2; nt4[3] = 3;       /* ref table maps bytes 0..3 to themselves */
	for (i = 0; a[i]; ++i) {                              /* IUPAC complement (ref: bseq.c seq_comp_table) */
		comp_tab[(uint8_t)a[i]] = b[i];
		comp_tab[(uint8_t)(a[i] + 32)] = b[i] + 32;
	}
}
static inline void tabs(void) { pthread_once(&tab_once, init_tabs); }

/* ---------------------------------------------------------------- options (ref: options.c:13-49,105-122) */
void oopt_sr(oopt_t *o)
{
	memset(o, 0, sizeof(*o));
	o->k = 21; o->w = 11;
	o->seed = 11; o->max_gap_ref = -1; o->max_chain_skip = 25; o->max_chain_iter = 5000;
	o->mask_level = 0.5f; o->max_clip_ratio = 1.0f; o->sc_ambi = 1; o->min_ksw_len = 200;
	o->a = 2; o->b = 8; o->q = 12; o->e = 2; o->q2 = 24; o->e2 = 1;
	o->zdrop = o->zdrop_inv = 100; o->end_bonus = 10; o->max_frag_len = 800; o->max_gap = 100; o->bw = 100;
	o->pri_ratio = 0.5f; o->min_cnt = 2; o->min_chain_score = 25; o->min_dp_max = 40; o->best_n = 20;
	o->mid_occ = 1000; o->max_occ = 5000; o->pe_ori = 1; o->pe_bonus = 33;
}

/* ---------------------------------------------------------------- sorting (ref: ksort.h:116-161, misc.c:155-159)
 * The reference sorts with an in-place MSD radix sort that is NOT stable above 64 elements and a stable
 * insertion sort at or below 64.  Tie order matters downstream, so the same published algorithm
 * (American-flag permutation, 8-bit digits from the top byte down) is restated here. */
#define RS_MIN 64
#define DEF_RADIX(NAME, T, KEY) \
static void ins_##NAME(T *beg, T *end) { T *i; \
	for (i = beg + 1; i < end; ++i) if (KEY(*i) < KEY(*(i-1))) { T *j, tmp = *i; \
		for (j = i; j > beg && KEY(tmp) < KEY(*(j-1)); --j) { *j = *(j-1); } \
		*j = tmp; } } \
static void rs_##NAME(T *beg, T *end, int s) { \
	struct { T *b, *e; } bk[256], *k, *be = bk + 256; T *i; \
	for (k = bk; k != be; ++k) k->b = k->e = beg; \
	for (i = beg; i != end; ++i) ++bk[KEY(*i)>>s&255].e; \
	for (k = bk + 1; k != be; ++k) k->e += (k-1)->e - beg, k->b = (k-1)->e; \
	for (k = bk; k != be;) { \
		if (k->b != k->e) { \
			typeof(bk[0]) *l; \
			if ((l = bk + (KEY(*k->b)>>s&255)) != k) { \
				T tmp = *k->b, swap; \
				do { swap = tmp; tmp = *l->b; *l->b++ = swap; l = bk + (KEY(tmp)>>s&255); } while (l != k); \
				*k->b++ = tmp; \
			} else ++k->b; \
		} else ++k; \
	} \
	for (bk->b = beg, k = bk + 1; k != be; ++k) k->b = (k-1)->e; \
	if (s) { s = s > 8? s - 8 : 0; \
		for (k = bk; k != be; ++k) \
			if (k->e - k->b > RS_MIN) rs_##NAME(k->b, k->e, s); \
			else if (k->e - k->b > 1) ins_##NAME(k->b, k->e); } } \
static void radix_##NAME(T *beg, T *end) { if (end - beg <= RS_MIN) ins_##NAME(beg, end); else rs_##NAME(beg, end, 56); }

#define KEY128(a) ((a).x)
#define KEY64(a) (a)
DEF_RADIX(128x, o128_t, KEY128)
DEF_RADIX(64, uint64_t, KEY64)
typedef struct { int s, rev; uint64_t key; oreg_t *r; } opair_t;
#define KEYPAIR(a) ((a).key)
DEF_RADIX(pair, opair_t, KEYPAIR)
void o_radix_sort_128x(o128_t *beg, o128_t *end) { radix_128x(beg, end); }
void o_radix_sort_64(uint64_t *beg, uint64_t *end) { radix_64(beg, end); }

/* ---------------------------------------------------------------- sketch (ref: sketch.c:28-38,77-143) */
static inline uint64_t hash64m(uint64_t key, uint64_t mask)
{
	key = (~key + (key << 21)) & mask;
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & mask;
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & mask;
	key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

static inline void push128(o128_t **a, size_t *n, size_t *m, o128_t v)
{
	if (*n == *m) { *m = *m? *m << 1 : 64; *a = (o128_t*)realloc(*a, *m * sizeof(o128_t)); }
	(*a)[(*n)++] = v;
}

void o_sketch(const char *str, int len, int w, int k, uint32_t rid, o128_t **pa, size_t *pn, size_t *pm)
{
	uint64_t shift1 = 2 * (k - 1), mask = (1ULL<<2*k) - 1, kmer[2] = {0,0};
	int i, j, l, buf_pos, min_pos;
	o128_t buf[256], min = { UINT64_MAX, UINT64_MAX };
	tabs();
	assert(len > 0 && w > 0 && w < 256 && k > 0 && k <= 28);
	memset(buf, 0xff, w * 16);
	for (i = l = buf_pos = min_pos = 0; i < len; ++i) {
		int c = nt4[(uint8_t)str[i]];
		o128_t info = { UINT64_MAX, UINT64_MAX };
		if (c < 4) {
			int z, kmer_span = l + 1 < k? l + 1 : k;                      /* ref: sketch.c:105 (non-HPC) */
			kmer[0] = (kmer[0] << 2 | c) & mask;
			kmer[1] = (kmer[1] >> 2) | (3ULL^c) << shift1;
			if (kmer[0] == kmer[1]) continue;                             /* ref: sketch.c:108 */
			z = kmer[0] < kmer[1]? 0 : 1;
			++l;
			if (l >= k && kmer_span < 256) {
				info.x = hash64m(kmer[z], mask) << 8 | kmer_span;
				info.y = (uint64_t)rid<<32 | (uint32_t)i<<1 | z;
			}
		} else l = 0;                                                      /* ref: sketch.c:115 */
		buf[buf_pos] = info;
		if (l == w + k - 1 && min.x != UINT64_MAX) {                       /* ref: sketch.c:117-122 */
			for (j = buf_pos + 1; j < w; ++j)
				if (min.x == buf[j].x && buf[j].y != min.y) push128(pa, pn, pm, buf[j]);
			for (j = 0; j < buf_pos; ++j)
				if (min.x == buf[j].x && buf[j].y != min.y) push128(pa, pn, pm, buf[j]);
		}
		if (info.x <= min.x) {                                             /* ref: sketch.c:123-125 */
			if (l >= w + k && min.x != UINT64_MAX) push128(pa, pn, pm, min);
			min = info, min_pos = buf_pos;
		} else if (buf_pos == min_pos) {                                   /* ref: sketch.c:126-138 */
			if (l >= w + k - 1 && min.x != UINT64_MAX) push128(pa, pn, pm, min);
			for (j = buf_pos + 1, min.x = UINT64_MAX; j < w; ++j)
				if (min.x >= buf[j].x) min = buf[j], min_pos = j;
			for (j = 0; j <= buf_pos; ++j)
				if (min.x >= buf[j].x) min = buf[j], min_pos = j;
			if (l >= w + k - 1 && min.x != UINT64_MAX) {
				for (j = buf_pos + 1; j < w; ++j)
					if (min.x == buf[j].x && min.y != buf[j].y) push128(pa, pn, pm, buf[j]);
				for (j = 0; j <= buf_pos; ++j)
					if (min.x == buf[j].x && min.y != buf[j].y) push128(pa, pn, pm, buf[j]);
			}
		}
		if (++buf_pos == w) buf_pos = 0;
	}
	if (min.x != UINT64_MAX) push128(pa, pn, pm, min);
}

/* ---------------------------------------------------------------- FASTA/FASTQ reader (own; semantics of kseq.h + bseq.c:60-75) */
typedef struct { gzFile fp; char *buf; int beg, end, eof; int last; } ofile_t;
static ofile_t *of_open(const char *fn)
{
	ofile_t *f; gzFile fp = gzopen(fn, "r");
	if (!fp) return 0;
	f = (ofile_t*)calloc(1, sizeof(ofile_t)); f->fp = fp; f->buf = (char*)malloc(1<<16);
	return f;
}
static void of_close(ofile_t *f) { if (f) { gzclose(f->fp); free(f->buf); free(f); } }
static inline int of_getc(ofile_t *f)
{
	if (f->beg >= f->end) {
		if (f->eof) return -1;
		f->beg = 0; f->end = gzread(f->fp, f->buf, 1<<16);
		if (f->end <= 0) { f->eof = 1; f->end = 0; return -1; }
	}
	return (unsigned char)f->buf[f->beg++];
}
typedef struct { char *s; size_t l, m; } ostr_t;
static inline void os_push(ostr_t *s, int c) { if (s->l + 2 > s->m) { s->m = s->m? s->m<<1 : 256; s->s = (char*)realloc(s->s, s->m); } s->s[s->l++] = c; s->s[s->l] = 0; }
/* returns seq length or -1 at EOF */
static int of_read(ofile_t *f, ostr_t *name, ostr_t *seq, ostr_t *qual)
{
	int c;
	name->l = seq->l = qual->l = 0;
	if (f->last == 0) { while ((c = of_getc(f)) >= 0 && c != '>' && c != '@'); if (c < 0) return -1; f->last = c; }
	while ((c = of_getc(f)) >= 0 && c != ' ' && c != '\t' && c != '\n' && c != '\r') os_push(name, c);
	if (c != '\n') while (c >= 0 && c != '\n') c = of_getc(f);             /* skip comment */
	os_push(name, 0); name->l--;
	while ((c = of_getc(f)) >= 0 && c != '>' && c != '+' && c != '@') {
		if (c == '\n') continue;
		if (c > 32) os_push(seq, c);                                       /* kseq keeps graph chars only */
	}
	if (c == '>' || c == '@') f->last = c; else f->last = 0;
	os_push(seq, 0); seq->l--;
	if (c != '+') return (int)seq->l;
	while ((c = of_getc(f)) >= 0 && c != '\n');                            /* skip rest of '+' line */
	while (qual->l < seq->l && (c = of_getc(f)) >= 0) if (c > 32) os_push(qual, c);
	os_push(qual, 0); qual->l--;
	f->last = 0;
	return (int)seq->l;
}

/* ---------------------------------------------------------------- index (ref: index.c:81-98,191-243,271-278) */
static int cmp_xy(const void *p, const void *q)
{
	const o128_t *u = (const o128_t*)p, *v = (const o128_t*)q;
	return u->x < v->x? -1 : u->x > v->x? 1 : u->y < v->y? -1 : u->y > v->y;
}

static void oidx_finish(oidx_t *mi, o128_t *a, size_t n)
{
	size_t i, j, nk = 0; uint64_t cap;
	/* bucket-local sort in the reference (by x), then positions of each key sorted by y (index.c:230):
	 * net effect = key -> ascending list of y.  We sort pairs (x>>8, y) globally. */
	for (i = 0; i < n; ++i) a[i].x >>= 8;
	qsort(a, n, sizeof(o128_t), cmp_xy);
	for (i = 0; i < n; ++i) if (i == 0 || a[i].x != a[i-1].x) ++nk;
	mi->n_keys = nk; mi->n_pos = n;
	cap = 16; while (cap < nk * 2 + 2) cap <<= 1;
	mi->tab_mask = cap - 1;
	mi->keys = (uint64_t*)calloc(cap, 8); mi->vals = (uint64_t*)calloc(cap, 8);
	mi->pos = (uint64_t*)malloc((n? n : 1) * 8);
	for (i = 0; i < n; ++i) mi->pos[i] = a[i].y;
	for (i = 0; i < n; i = j) {
		uint64_t h;
		for (j = i + 1; j < n && a[j].x == a[i].x; ++j);
		h = (a[i].x * 0x9E3779B97F4A7C15ULL) >> 20 & mi->tab_mask;
		while (mi->keys[h]) h = (h + 1) & mi->tab_mask;
		mi->keys[h] = a[i].x + 1; mi->vals[h] = (uint64_t)i << 32 | (uint32_t)(j - i);
	}
}

const uint64_t *oidx_get(const oidx_t *mi, uint64_t minier, int *n)
{
	uint64_t h = (minier * 0x9E3779B97F4A7C15ULL) >> 20 & mi->tab_mask;
	*n = 0;
	while (mi->keys[h]) {
		if (mi->keys[h] == minier + 1) { *n = (uint32_t)mi->vals[h]; return &mi->pos[mi->vals[h]>>32]; }
		h = (h + 1) & mi->tab_mask;
	}
	return 0;
}

oidx_t *oidx_build(int k, int w, int n, const char **names, const char **seqs)
{
	oidx_t *mi = (oidx_t*)calloc(1, sizeof(oidx_t));
	o128_t *a = 0; size_t na = 0, ma = 0; int i; uint64_t sum = 0, j;
	tabs();
	mi->k = k; mi->w = w; mi->n_seq = n; mi->seq = (oseq_t*)calloc(n, sizeof(oseq_t));
	for (i = 0; i < n; ++i) sum += strlen(seqs[i]);
	mi->S = (uint8_t*)malloc(sum + 1); mi->tot_len = sum;
	for (i = 0, sum = 0; i < n; ++i) {
		uint32_t len = strlen(seqs[i]);
		mi->seq[i].name = strdup(names[i]); mi->seq[i].len = len; mi->seq[i].offset = sum;
		for (j = 0; j < len; ++j) mi->S[sum + j] = nt4[(uint8_t)seqs[i][j]];
		if (len > 0) o_sketch(seqs[i], len, w, k, i, &a, &na, &ma);
		sum += len;
	}
	oidx_finish(mi, a, na);
	free(a);
	return mi;
}

oidx_t *oidx_build_file(const char *fn, int k, int w)
{
	ofile_t *f = of_open(fn); ostr_t nm = {0,0,0}, sq = {0,0,0}, ql = {0,0,0};
	char **names = 0, **seqs = 0; int n = 0, m = 0, i; oidx_t *mi;
	if (!f) return 0;
	while (of_read(f, &nm, &sq, &ql) >= 0) {
		if (n == m) { m = m? m<<1 : 16; names = (char**)realloc(names, m * sizeof(char*)); seqs = (char**)realloc(seqs, m * sizeof(char*)); }
		names[n] = strdup(nm.s); seqs[n] = strdup(sq.s? sq.s : ""); ++n;
	}
	of_close(f);
	mi = oidx_build(k, w, n, (const char**)names, (const char**)seqs);
	for (i = 0; i < n; ++i) free(names[i]), free(seqs[i]);
	free(names); free(seqs); free(nm.s); free(sq.s); free(ql.s);
	return mi;
}

void oidx_destroy(oidx_t *mi)
{
	uint32_t i;
	if (!mi) return;
	for (i = 0; i < mi->n_seq; ++i) free(mi->seq[i].name);
	free(mi->seq); free(mi->S); free(mi->keys); free(mi->vals); free(mi->pos); free(mi);
}

static inline void idx_getseq(const oidx_t *mi, uint32_t rid, uint32_t st, uint32_t en, uint8_t *seq)
{   /* ref: index.c:154-165 */
	if (rid >= mi->n_seq || st >= mi->seq[rid].len) return;
	if (en > mi->seq[rid].len) en = mi->seq[rid].len;
	memcpy(seq, mi->S + mi->seq[rid].offset + st, en - st);
}

/* ---------------------------------------------------------------- seeding (ref: map.c:64-123,149-213) */
typedef struct { uint32_t n, q_pos, q_span, seg_id, is_tandem; const uint64_t *cr; } omatch_t;

static inline void heapdown(size_t i, size_t n, o128_t *l)
{   /* ref: ksort.h:43-53 with heap_lt(a,b) = a.x > b.x (map.c:80) */
	size_t k = i; o128_t tmp = l[i];
	while ((k = (k << 1) + 1) < n) {
		if (k != n - 1 && l[k].x > l[k+1].x) ++k;
		if (l[k].x > tmp.x) break;
		l[i] = l[k]; i = k;
	}
	l[i] = tmp;
}

o128_t *o_collect_seeds(const oidx_t *mi, int max_occ, const o128_t *mv, size_t n_mv, int qlen, int64_t *n_a, int *rep_len)
{
	int rep_st = 0, rep_en = 0, n_m = 0, heap_size = 0; size_t i; int64_t j, n_for = 0, n_rev = 0;
	omatch_t *m = (omatch_t*)malloc((n_mv? n_mv : 1) * sizeof(omatch_t));
	o128_t *a, *heap;
	for (i = 0, *rep_len = 0, *n_a = 0; i < n_mv; ++i) {                   /* ref: map.c:98-121 */
		const o128_t *p = &mv[i]; int t;
		uint32_t q_pos = (uint32_t)p->y, q_span = p->x & 0xff;
		const uint64_t *cr = oidx_get(mi, p->x>>8, &t);
		if (t >= max_occ) {
			int en = (q_pos >> 1) + 1, st = en - q_span;
			if (st > rep_en) { *rep_len += rep_en - rep_st; rep_st = st, rep_en = en; }
			else rep_en = en;
		} else {
			omatch_t *q = &m[n_m++];
			q->q_pos = q_pos, q->q_span = q_span, q->cr = cr, q->n = t, q->seg_id = p->y >> 32;
			q->is_tandem = 0;
			if (i > 0 && p->x>>8 == mv[i-1].x>>8) q->is_tandem = 1;
			if (i < n_mv - 1 && p->x>>8 == mv[i+1].x>>8) q->is_tandem = 1;
			*n_a += q->n;
		}
	}
	*rep_len += rep_en - rep_st;
	heap = (o128_t*)malloc((n_m? n_m : 1) * sizeof(o128_t));
	a = (o128_t*)malloc((*n_a? *n_a : 1) * sizeof(o128_t));
	for (i = 0; i < (size_t)n_m; ++i)                                       /* ref: map.c:163-170 */
		if (m[i].n > 0) { heap[heap_size].x = m[i].cr[0]; heap[heap_size].y = (uint64_t)i<<32; ++heap_size; }
	if (heap_size > 1) for (i = (heap_size >> 1) - 1; i != (size_t)-1; --i) heapdown(i, heap_size, heap);
	while (heap_size > 0) {                                                 /* ref: map.c:171-198 */
		omatch_t *q = &m[heap->y>>32]; o128_t *p; uint64_t r = heap->x; int32_t rpos = (uint32_t)r >> 1;
		if ((r&1) == (q->q_pos&1)) {
			p = &a[n_for++];
			p->x = (r&0xffffffff00000000ULL) | rpos;
			p->y = (uint64_t)q->q_span << 32 | q->q_pos >> 1;
		} else {
			p = &a[(*n_a) - (++n_rev)];
			p->x = 1ULL<<63 | (r&0xffffffff00000000ULL) | rpos;
			p->y = (uint64_t)q->q_span << 32 | (qlen - ((q->q_pos>>1) + 1 - q->q_span) - 1);
		}
		p->y |= (uint64_t)q->seg_id << SEED_SEG_SHIFT;
		if (q->is_tandem) p->y |= SEED_TANDEM;
		if ((uint32_t)heap->y < q->n - 1) {
			++heap[0].y;
			heap[0].x = m[heap[0].y>>32].cr[(uint32_t)heap[0].y];
		} else { heap[0] = heap[heap_size - 1]; --heap_size; }
		if (heap_size > 0) heapdown(0, heap_size, heap);
	}
	free(m); free(heap);
	for (j = 0; j < n_rev>>1; ++j) {                                        /* ref: map.c:202-207 */
		o128_t t = a[(*n_a) - 1 - j];
		a[(*n_a) - 1 - j] = a[(*n_a) - (n_rev - j)];
		a[(*n_a) - (n_rev - j)] = t;
	}
	return a;  /* n_for + n_rev == *n_a always holds here (no skip_seed in sr mode) */
}

/* ---------------------------------------------------------------- chaining (ref: chain.c:15-162) */
static inline int ilog2_32(uint32_t v) { int r = -1; while (v) { ++r; v >>= 1; } return r; } /* == LogTable256 lookup; -1 for 0 */

o128_t *o_chain_dp(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc,
                   int n_segs, int64_t n, o128_t *a, int *n_u_, uint64_t **_u)
{
	int32_t k, *f, *p, *t, *v, n_u, n_v; int64_t i, j, st = 0; uint64_t *u, *u2, sum_qspan = 0; float avg_qspan; o128_t *b, *w;
	*_u = 0, *n_u_ = 0;
	if (n == 0 || a == 0) { free(a); return 0; }
	f = (int32_t*)malloc(n * 4); p = (int32_t*)malloc(n * 4); t = (int32_t*)calloc(n, 4); v = (int32_t*)malloc(n * 4);
	for (i = 0; i < n; ++i) sum_qspan += a[i].y>>32&0xff;
	avg_qspan = (float)sum_qspan / n;
	for (i = 0; i < n; ++i) {                                               /* ref: chain.c:46-85 */
		uint64_t ri = a[i].x; int64_t max_j = -1;
		int32_t qi = (int32_t)a[i].y, q_span = a[i].y>>32&0xff;
		int32_t max_f = q_span, n_skip = 0, min_d;
		int32_t sidi = (a[i].y & SEED_SEG_MASK) >> SEED_SEG_SHIFT;
		while (st < i && ri > a[st].x + max_dist_x) ++st;
		if (i - st > max_iter) st = i - max_iter;
		for (j = i - 1; j >= st; --j) {
			int64_t dr = ri - a[j].x;
			int32_t dq = qi - (int32_t)a[j].y, dd, sc, log_dd;
			int32_t sidj = (a[j].y & SEED_SEG_MASK) >> SEED_SEG_SHIFT;
			if ((sidi == sidj && dr == 0) || dq <= 0) continue;
			if ((sidi == sidj && dq > max_dist_y) || dq > max_dist_x) continue;
			dd = dr > dq? dr - dq : dq - dr;
			if (sidi == sidj && dd > bw) continue;
			if (n_segs > 1 && sidi == sidj && dr > max_dist_y) continue;
			min_d = dq < dr? dq : dr;
			sc = min_d > q_span? q_span : dq < dr? dq : dr;
			log_dd = dd? ilog2_32(dd) : 0;
			if (sidi != sidj) {
				int c_log, c_lin;
				c_lin = (int)(dd * .01 * avg_qspan);
				c_log = log_dd;
				if (dr == 0) ++sc;
				else sc -= c_lin < c_log? c_lin : c_log;                   /* "dr > dq || sidi != sidj" always true here */
			} else sc -= (int)(dd * .01 * avg_qspan) + (log_dd>>1);
			sc += f[j];
			if (sc > max_f) { max_f = sc, max_j = j; if (n_skip > 0) --n_skip; }
			else if (t[j] == i) { if (++n_skip > max_skip) break; }
			if (p[j] >= 0) t[p[j]] = i;
		}
		f[i] = max_f, p[i] = max_j;
		v[i] = max_j >= 0 && v[max_j] > max_f? v[max_j] : max_f;
	}
	memset(t, 0, n * 4);                                                    /* ref: chain.c:87-109 */
	for (i = 0; i < n; ++i) if (p[i] >= 0) t[p[i]] = 1;
	for (i = n_u = 0; i < n; ++i) if (t[i] == 0 && v[i] >= min_sc) ++n_u;
	if (n_u == 0) { free(a); free(f); free(p); free(t); free(v); return 0; }
	u = (uint64_t*)malloc(n_u * 8);
	for (i = n_u = 0; i < n; ++i)
		if (t[i] == 0 && v[i] >= min_sc) {
			j = i;
			while (j >= 0 && f[j] < v[j]) j = p[j];
			if (j < 0) j = i;
			u[n_u++] = (uint64_t)f[j] << 32 | j;
		}
	radix_64(u, u + n_u);
	for (i = 0; i < n_u>>1; ++i) { uint64_t tt = u[i]; u[i] = u[n_u-i-1], u[n_u-i-1] = tt; }
	memset(t, 0, n * 4);                                                    /* ref: chain.c:111-128 */
	for (i = n_v = k = 0; i < n_u; ++i) {
		int32_t n_v0 = n_v, k0 = k;
		j = (int32_t)u[i];
		do { v[n_v++] = j; t[j] = 1; j = p[j]; } while (j >= 0 && t[j] == 0);
		if (j < 0) { if (n_v - n_v0 >= min_cnt) u[k++] = u[i]>>32<<32 | (n_v - n_v0); }
		else if ((int32_t)(u[i]>>32) - f[j] >= min_sc) { if (n_v - n_v0 >= min_cnt) u[k++] = ((u[i]>>32) - f[j]) << 32 | (n_v - n_v0); }
		if (k0 == k) n_v = n_v0;
	}
	*n_u_ = n_u = k, *_u = u;
	free(f); free(p); free(t);
	b = (o128_t*)malloc((n_v? n_v : 1) * sizeof(o128_t));                   /* ref: chain.c:134-160 */
	for (i = 0, k = 0; i < n_u; ++i) {
		int32_t k0 = k, ni = (int32_t)u[i];
		for (j = 0; j < ni; ++j) b[k] = a[v[k0 + (ni - j - 1)]], ++k;
	}
	free(v);
	w = (o128_t*)malloc((n_u? n_u : 1) * sizeof(o128_t));
	for (i = k = 0; i < n_u; ++i) { w[i].x = b[k].x, w[i].y = (uint64_t)k<<32|i; k += (int32_t)u[i]; }
	radix_128x(w, w + n_u);
	u2 = (uint64_t*)malloc((n_u? n_u : 1) * 8);
	for (i = k = 0; i < n_u; ++i) {
		int32_t jj = (int32_t)w[i].y, nn = (int32_t)u[jj];
		u2[i] = u[jj];
		memcpy(&a[k], &b[w[i].y>>32], nn * sizeof(o128_t));
		k += nn;
	}
	if (n_u) memcpy(u, u2, n_u * 8);
	if (k) memcpy(b, a, k * sizeof(o128_t));
	free(a); free(w); free(u2);
	return b;
}

/* ---------------------------------------------------------------- hits (ref: hit.c:8-88) */
static inline uint64_t hash64(uint64_t key)
{
	key = (~key + (key << 21)); key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)); key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)); key = key ^ key >> 28;
	key = (key + (key << 31));
	return key;
}

static void reg_set_coor(oreg_t *r, int32_t qlen, const o128_t *a)
{   /* ref: hit.c:8-41 */
	int32_t k = r->as, q_span = (int32_t)(a[k].y>>32&0xff), i;
	r->rev = a[k].x>>63;
	r->rid = a[k].x<<1>>33;
	r->rs = (int32_t)a[k].x + 1 > q_span? (int32_t)a[k].x + 1 - q_span : 0;
	r->re = (int32_t)a[k + r->cnt - 1].x + 1;
	if (!r->rev) { r->qs = (int32_t)a[k].y + 1 - q_span; r->qe = (int32_t)a[k + r->cnt - 1].y + 1; }
	else { r->qs = qlen - ((int32_t)a[k + r->cnt - 1].y + 1); r->qe = qlen - ((int32_t)a[k].y + 1 - q_span); }
	r->mlen = r->blen = 0;
	if (r->cnt <= 0) return;
	r->mlen = r->blen = a[r->as].y>>32&0xff;
	for (i = r->as + 1; i < r->as + r->cnt; ++i) {
		int span = a[i].y>>32&0xff;
		int tl = (int32_t)a[i].x - (int32_t)a[i-1].x;
		int ql = (int32_t)a[i].y - (int32_t)a[i-1].y;
		r->blen += tl > ql? tl : ql;
		r->mlen += tl > span && ql > span? span : tl < ql? tl : ql;
	}
}

static oreg_t *gen_regs(uint32_t hash, int qlen, int n_u, uint64_t *u, o128_t *a)
{   /* ref: hit.c:52-88 */
	o128_t *z, tmp; oreg_t *r; int i, k;
	if (n_u == 0) return 0;
	z = (o128_t*)malloc(n_u * 16);
	for (i = k = 0; i < n_u; ++i) {
		uint32_t h = (uint32_t)hash64((hash64(a[k].x) + hash64(a[k].y)) ^ hash);
		z[i].x = u[i] ^ h;
		z[i].y = (uint64_t)k << 32 | (int32_t)u[i];
		k += (int32_t)u[i];
	}
	radix_128x(z, z + n_u);
	for (i = 0; i < n_u>>1; ++i) tmp = z[i], z[i] = z[n_u-1-i], z[n_u-1-i] = tmp;
	r = (oreg_t*)calloc(n_u, sizeof(oreg_t));
	for (i = 0; i < n_u; ++i) {
		oreg_t *ri = &r[i];
		ri->id = i; ri->parent = PARENT_UNSET;
		ri->score = ri->score0 = z[i].x >> 32;
		ri->hash = (uint32_t)z[i].x;
		ri->cnt = (int32_t)z[i].y; ri->as = z[i].y >> 32;
		ri->div = -1.0f;
		reg_set_coor(ri, qlen, a);
	}
	free(z);
	return r;
}

static void split_reg(oreg_t *r, oreg_t *r2, int n, int qlen, o128_t *a)
{   /* ref: hit.c:90-107 */
	if (n <= 0 || n >= r->cnt) return;
	*r2 = *r;
	r2->id = -1; r2->sam_pri = 0; r2->p = 0; r2->split_inv = 0;
	r2->cnt = r->cnt - n;
	r2->score = (int32_t)(r->score * ((float)r2->cnt / r->cnt) + .499);
	r2->as = r->as + n;
	if (r->parent == r->id) r2->parent = PARENT_TMP_PRI;
	reg_set_coor(r2, qlen, a);
	r->cnt -= r2->cnt; r->score -= r2->score;
	reg_set_coor(r, qlen, a);
	r->split |= 1, r2->split |= 2;
}

static void set_parent(float mask_level, int n, oreg_t *r, int sub_diff)
{   /* ref: hit.c:109-167 (hard_mask_level == 0) */
	int i, j, k, *w; uint64_t *cov;
	if (n <= 0) return;
	for (i = 0; i < n; ++i) r[i].id = i;
	cov = (uint64_t*)malloc(n * 8); w = (int*)malloc(n * sizeof(int));
	w[0] = 0, r[0].parent = 0;
	for (i = 1, k = 1; i < n; ++i) {
		oreg_t *ri = &r[i];
		int si = ri->qs, ei = ri->qe, n_cov = 0, uncov_len = 0;
		for (j = 0; j < k; ++j) {
			oreg_t *rp = &r[w[j]]; int sj = rp->qs, ej = rp->qe;
			if (ej <= si || sj >= ei) continue;
			if (sj < si) sj = si;
			if (ej > ei) ej = ei;
			cov[n_cov++] = (uint64_t)sj<<32 | ej;
		}
		if (n_cov == 0) goto set_parent_test;
		else {
			int jj, x = si;
			radix_64(cov, cov + n_cov);
			for (jj = 0; jj < n_cov; ++jj) {
				if ((int)(cov[jj]>>32) > x) uncov_len += (cov[jj]>>32) - x;
				x = (int32_t)cov[jj] > x? (int32_t)cov[jj] : x;
			}
			if (ei > x) uncov_len += ei - x;
		}
		for (j = 0; j < k; ++j) {
			oreg_t *rp = &r[w[j]]; int sj = rp->qs, ej = rp->qe, min, max, ol;
			if (ej <= si || sj >= ei) continue;
			min = ej - sj < ei - si? ej - sj : ei - si;
			max = ej - sj > ei - si? ej - sj : ei - si;
			ol = si < sj? (ei < sj? 0 : ei < ej? ei - sj : ej - sj) : (ej < si? 0 : ej < ei? ej - si : ei - si);
			if ((float)ol / min - (float)uncov_len / max > mask_level) {
				int cnt_sub = 0;
				ri->parent = rp->parent;
				rp->subsc = rp->subsc > ri->score? rp->subsc : ri->score;
				if (ri->cnt >= rp->cnt) cnt_sub = 1;
				if (rp->p && ri->p && (rp->rid != ri->rid || rp->rs != ri->rs || rp->re != ri->re || ol != min)) {
					rp->p->dp_max2 = rp->p->dp_max2 > ri->p->dp_max? rp->p->dp_max2 : ri->p->dp_max;
					if (rp->p->dp_max - ri->p->dp_max <= sub_diff) cnt_sub = 1;
				}
				if (cnt_sub) ++rp->n_sub;
				break;
			}
		}
set_parent_test:
		if (j == k) w[k++] = i, ri->parent = i, ri->n_sub = 0;
	}
	free(cov); free(w);
}

static int set_sam_pri(int n, oreg_t *r)
{   /* ref: hit.c:203-212 */
	int i, n_pri = 0;
	for (i = 0; i < n; ++i)
		if (r[i].id == r[i].parent) { ++n_pri; r[i].sam_pri = (n_pri == 1); }
		else r[i].sam_pri = 0;
	return n_pri;
}

static void sync_regs(int n_regs, oreg_t *regs)
{   /* ref: hit.c:214-236 */
	int *tmp, i, max_id = -1, n_tmp;
	if (n_regs <= 0) return;
	for (i = 0; i < n_regs; ++i) max_id = max_id > regs[i].id? max_id : regs[i].id;
	n_tmp = max_id + 1;
	tmp = (int*)malloc((n_tmp > 0? n_tmp : 1) * sizeof(int));
	for (i = 0; i < n_tmp; ++i) tmp[i] = -1;
	for (i = 0; i < n_regs; ++i) if (regs[i].id >= 0) tmp[regs[i].id] = i;
	for (i = 0; i < n_regs; ++i) {
		oreg_t *r = &regs[i];
		r->id = i;
		if (r->parent == PARENT_TMP_PRI) r->parent = i;
		else if (r->parent >= 0 && tmp[r->parent] >= 0) r->parent = tmp[r->parent];
		else r->parent = PARENT_UNSET;
	}
	free(tmp);
	set_sam_pri(n_regs, regs);
}

static void select_sub(float pri_ratio, int min_diff, int best_n, int *n_, oreg_t *r)
{   /* ref: hit.c:238-255 */
	if (pri_ratio > 0.0f && *n_ > 0) {
		int i, k, n = *n_, n_2nd = 0;
		for (i = k = 0; i < n; ++i) {
			int p = r[i].parent;
			if (p == i || r[i].inv) r[k++] = r[i];
			else if ((r[i].score >= r[p].score * pri_ratio || r[i].score + min_diff >= r[p].score) && n_2nd < best_n) {
				if (!(r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].rid == r[p].rid && r[i].rs == r[p].rs && r[i].re == r[p].re))
					r[k++] = r[i], ++n_2nd;
				else if (r[i].p) free(r[i].p);
			} else if (r[i].p) free(r[i].p);
		}
		if (k != n) sync_regs(k, r);
		*n_ = k;
	}
}

static void select_sub_multi(float pri_ratio, float pri1, float pri2, int max_gap_ref, int min_diff, int best_n, int n_segs, const int *qlens, int *n_, oreg_t *r)
{   /* ref: pe.c:6-43 */
	if (pri_ratio > 0.0f && *n_ > 0) {
		int i, k, n = *n_, n_2nd = 0;
		int max_dist = n_segs == 2? qlens[0] + qlens[1] + max_gap_ref : 0;
		for (i = k = 0; i < n; ++i) {
			int to_keep = 0;
			if (r[i].parent == i) to_keep = 1;
			else if (r[i].score + min_diff >= r[r[i].parent].score) to_keep = 1;
			else {
				oreg_t *p = &r[r[i].parent], *q = &r[i];
				if (p->rev == q->rev && p->rid == q->rid && q->re - p->rs < max_dist && p->re - q->rs < max_dist) {
					if (q->score >= p->score * pri1) to_keep = 1;
				} else {
					int is_par_both = (n_segs == 2 && p->qs < qlens[0] && p->qe > qlens[0]);
					int is_chi_both = (n_segs == 2 && q->qs < qlens[0] && q->qe > qlens[0]);
					if (is_chi_both || is_chi_both == is_par_both) { if (q->score >= p->score * pri_ratio) to_keep = 1; }
					else { if (q->score >= p->score * pri2) to_keep = 1; }
				}
			}
			if (to_keep && r[i].parent != i) { if (n_2nd++ >= best_n) to_keep = 0; }
			if (to_keep) r[k++] = r[i];
			else if (r[i].p) free(r[i].p);
		}
		if (k != n) sync_regs(k, r);
		*n_ = k;
	}
}

static void filter_regs(const oopt_t *opt, int qlen, int *n_regs, oreg_t *regs)
{   /* ref: hit.c:257-276 */
	int i, k;
	for (i = k = 0; i < *n_regs; ++i) {
		oreg_t *r = &regs[i]; int flt = 0;
		if (!r->inv && !r->seg_split && r->cnt < opt->min_cnt) flt = 1;
		if (r->p) {
			if (r->mlen < opt->min_chain_score) flt = 1;
			else if (r->p->dp_max < opt->min_dp_max) flt = 1;
			else if (r->qs > qlen * opt->max_clip_ratio && qlen - r->qe > qlen * opt->max_clip_ratio) flt = 1;
			if (flt) free(r->p);
		}
		if (!flt) { if (k < i) regs[k++] = regs[i]; else ++k; }
	}
	*n_regs = k;
}

static void hit_sort(int *n_regs, oreg_t *r)
{   /* ref: hit.c:169-201 */
	int32_t i, n_aux, n = *n_regs; o128_t *aux; oreg_t *t;
	if (n <= 1) return;
	aux = (o128_t*)malloc(n * 16); t = (oreg_t*)malloc(n * sizeof(oreg_t));
	for (i = n_aux = 0; i < n; ++i) {
		if (r[i].inv || r[i].cnt > 0) {
			if (r[i].p) aux[n_aux].x = (uint64_t)r[i].p->dp_max << 32 | r[i].hash;
			else aux[n_aux].x = (uint64_t)r[i].score << 32 | r[i].hash;
			aux[n_aux++].y = i;
		} else if (r[i].p) { free(r[i].p); r[i].p = 0; }
	}
	radix_128x(aux, aux + n_aux);
	for (i = n_aux - 1; i >= 0; --i) t[n_aux - 1 - i] = r[aux[i].y];
	memcpy(r, t, sizeof(oreg_t) * n_aux);
	*n_regs = n_aux;
	free(aux); free(t);
}

static int squeeze_a(int n_regs, oreg_t *regs, o128_t *a)
{   /* ref: hit.c:278-296 */
	int i, as = 0; uint64_t *aux = (uint64_t*)malloc((n_regs? n_regs : 1) * 8);
	for (i = 0; i < n_regs; ++i) aux[i] = (uint64_t)regs[i].as << 32 | i;
	radix_64(aux, aux + n_regs);
	for (i = 0; i < n_regs; ++i) {
		oreg_t *r = &regs[(int32_t)aux[i]];
		if (r->as != as) { memmove(&a[as], &a[r->as], r->cnt * 16); r->as = as; }
		as += r->cnt;
	}
	free(aux);
	return as;
}

typedef struct { int n_u, n_a; uint64_t *u; o128_t *a; } oseg_t;

static oseg_t *seg_gen(uint32_t hash, int n_segs, const int *qlens, int n_regs0, const oreg_t *regs0, int *n_regs, oreg_t **regs, const o128_t *a)
{   /* ref: hit.c:356-410 */
	int s, i, j, acc_qlen[256], qlen_sum = 0; oseg_t *seg;
	for (s = 1, acc_qlen[0] = 0; s < n_segs; ++s) acc_qlen[s] = acc_qlen[s-1] + qlens[s-1];
	qlen_sum = acc_qlen[n_segs - 1] + qlens[n_segs - 1];
	seg = (oseg_t*)calloc(n_segs, sizeof(oseg_t));
	for (s = 0; s < n_segs; ++s) {
		seg[s].u = (uint64_t*)malloc((n_regs0? n_regs0 : 1) * 8);
		for (i = 0; i < n_regs0; ++i) seg[s].u[i] = (uint64_t)regs0[i].score << 32;
	}
	for (i = 0; i < n_regs0; ++i) {
		const oreg_t *r = &regs0[i];
		for (j = 0; j < r->cnt; ++j) {
			int sid = (a[r->as + j].y&SEED_SEG_MASK)>>SEED_SEG_SHIFT;
			++seg[sid].u[i]; ++seg[sid].n_a;
		}
	}
	for (s = 0; s < n_segs; ++s) {
		oseg_t *sr = &seg[s];
		for (i = 0, sr->n_u = 0; i < n_regs0; ++i) if ((int32_t)sr->u[i] != 0) sr->u[sr->n_u++] = sr->u[i];
		sr->a = (o128_t*)malloc((sr->n_a? sr->n_a : 1) * sizeof(o128_t));
		sr->n_a = 0;
	}
	for (i = 0; i < n_regs0; ++i) {
		const oreg_t *r = &regs0[i];
		for (j = 0; j < r->cnt; ++j) {
			int sid = (a[r->as + j].y&SEED_SEG_MASK)>>SEED_SEG_SHIFT;
			o128_t a1 = a[r->as + j];
			a1.y -= a1.x>>63? qlen_sum - (qlens[sid] + acc_qlen[sid]) : acc_qlen[sid];
			seg[sid].a[seg[sid].n_a++] = a1;
		}
	}
	for (s = 0; s < n_segs; ++s) {
		regs[s] = gen_regs(hash, qlens[s], seg[s].n_u, seg[s].u, seg[s].a);
		n_regs[s] = seg[s].n_u;
		for (i = 0; i < n_regs[s]; ++i) regs[s][i].seg_split = 1, regs[s][i].seg_id = s;
	}
	return seg;
}

/* ---------------------------------------------------------------- ksw2 extd2 (ref: ksw2_extd2_sse.c:26-393, ksw2.h:104-176)
 * Scalar emulation of the SSE kernel INCLUDING its 16-lane block geometry: cells outside [st0,en0] but inside
 * the 16-aligned [st,en] are computed from stale score bytes exactly as the vector code does, all DP bytes are
 * wrapping int8, and the exact-max scan reproduces the 4-lane tie-break. */
static inline uint32_t *push_cigar(int *n_cigar, int *m_cigar, uint32_t *cigar, uint32_t op, int len)
{   /* ref: ksw2.h:104-114 */
	if (*n_cigar == 0 || op != (cigar[(*n_cigar) - 1]&0xf)) {
		if (*n_cigar == *m_cigar) { *m_cigar = *m_cigar? (*m_cigar)<<1 : 4; cigar = (uint32_t*)realloc(cigar, (*m_cigar) << 2); }
		cigar[(*n_cigar)++] = len<<4 | op;
	} else cigar[(*n_cigar)-1] += len<<4;
	return cigar;
}

static void ksw_backtrack_rot(int is_rev, const uint8_t *p, const int *off, const int *off_end, int n_col, int i0, int j0, int *m_cigar_, int *n_cigar_, uint32_t **cigar_)
{   /* ref: ksw2.h:119-151 (is_rot = 1, min_intron_len = 0) */
	int n_cigar = 0, m_cigar = *m_cigar_, i = i0, j = j0, r, state = 0; uint32_t *cigar = *cigar_, tmp;
	while (i >= 0 && j >= 0) {
		int force_state = -1;
		r = i + j;
		if (i < off[r]) force_state = 2;
		if (i > off_end[r]) force_state = 1;
		tmp = force_state < 0? p[(size_t)r * n_col + i - off[r]] : 0;
		if (state == 0) state = tmp & 7;
		else if (!(tmp >> (state + 2) & 1)) state = 0;
		if (state == 0) state = tmp & 7;
		if (force_state >= 0) state = force_state;
		if (state == 0) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 0, 1), --i, --j;
		else if (state == 1 || state == 3) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 2, 1), --i;
		else cigar = push_cigar(&n_cigar, &m_cigar, cigar, 1, 1), --j;
	}
	if (i >= 0) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 2, i + 1);
	if (j >= 0) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 1, j + 1);
	if (!is_rev)
		for (i = 0; i < n_cigar>>1; ++i) tmp = cigar[i], cigar[i] = cigar[n_cigar-1-i], cigar[n_cigar-1-i] = tmp;
	*m_cigar_ = m_cigar, *n_cigar_ = n_cigar, *cigar_ = cigar;
}

static inline void ksw_reset(oksw_t *ez)
{   /* ref: ksw2.h:153-158 */
	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->max = 0, ez->score = ez->mqe = ez->mte = KSW_NEG_INF;
	ez->n_cigar = 0, ez->zdropped = 0, ez->reach_end = 0;
}

static inline int ksw_zdrop(oksw_t *ez, int32_t H, int r, int t, int zdrop, int8_t e)
{   /* ref: ksw2.h:160-176 (is_rot = 1) */
	if (H > (int32_t)ez->max) { ez->max = H, ez->max_t = t, ez->max_q = r - t; }
	else if (t >= ez->max_t && r - t >= ez->max_q) {
		int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l;
		l = tl > ql? tl - ql : ql - tl;
		if (zdrop >= 0 && (int32_t)ez->max - H > zdrop + l * e) { ez->zdropped = 1; return 1; }
	}
	return 0;
}

#define I8(x) ((int8_t)(x))
void o_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                 int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, oksw_t *ez)
{
	int r, t, qe = q + e, n_col_, *off = 0, *off_end = 0, tlen_, qlen_, last_st, last_en, wl, wr, max_sc, min_sc, long_thres, long_diff;
	int approx_max = !!(flag&EZ_APPROX_MAX);
	int32_t *H = 0, H0 = 0, last_H0_t = 0;
	uint8_t *qr, *sf, *p = 0;
	int8_t *u, *v, *x, *y, *x2, *y2, *s, *mem;
	int8_t sc_mch, sc_mis, sc_N, qe_, qe2_;

	ksw_reset(ez);
	if (m <= 1 || qlen <= 0 || tlen <= 0) return;
	if (q2 + e2 < q + e) t = q, q = q2, q2 = t, t = e, e = e2, e2 = t;
	qe = q + e;
	qe_ = I8(q + e); qe2_ = I8(q2 + e2);
	sc_mch = mat[0]; sc_mis = mat[1];
	sc_N = mat[m*m-1] == 0? I8(-e2) : mat[m*m-1];
	if (w < 0) w = tlen > qlen? tlen : qlen;
	wl = wr = w;
	tlen_ = (tlen + 15) / 16;
	n_col_ = qlen < tlen? qlen : tlen;
	n_col_ = ((n_col_ < w + 1? n_col_ : w + 1) + 15) / 16 + 1;
	qlen_ = (qlen + 15) / 16;
	for (t = 1, max_sc = mat[0], min_sc = mat[1]; t < m * m; ++t) {
		max_sc = max_sc > mat[t]? max_sc : mat[t];
		min_sc = min_sc < mat[t]? min_sc : mat[t];
	}
	if (-min_sc > 2 * (q + e)) return;
	long_thres = e != e2? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	long_diff = long_thres * (e - e2) - (q2 - q) - e2;

	mem = (int8_t*)calloc((size_t)tlen_ * 8 + qlen_ + 2, 16);
	u = mem; v = u + tlen_*16; x = v + tlen_*16; y = x + tlen_*16; x2 = y + tlen_*16; y2 = x2 + tlen_*16;
	s = y2 + tlen_*16; sf = (uint8_t*)(s + tlen_*16); qr = sf + tlen_*16;
	memset(u, -q - e, tlen_*16); memset(v, -q - e, tlen_*16); memset(x, -q - e, tlen_*16); memset(y, -q - e, tlen_*16);
	memset(x2, -q2 - e2, tlen_*16); memset(y2, -q2 - e2, tlen_*16);
	if (!approx_max) { H = (int32_t*)malloc((size_t)tlen_ * 16 * 4); for (t = 0; t < tlen_ * 16; ++t) H[t] = KSW_NEG_INF; }
	p = (uint8_t*)calloc(((size_t)(qlen + tlen - 1) * n_col_ + 1) * 16, 1);
	off = (int*)malloc((qlen + tlen - 1) * sizeof(int) * 2);
	off_end = off + qlen + tlen - 1;
	for (t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
	memcpy(sf, target, tlen);

	for (r = 0, last_st = last_en = -1; r < qlen + tlen - 1; ++r) {
		int st = 0, en = tlen - 1, st0, en0, st_, en_;
		int8_t x1, x21, v1, xprev, x2prev, vprev;
		uint8_t *qrr = qr + (qlen - 1 - r), *pr;
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < (r-wr+1)>>1) st = (r-wr+1)>>1;
		if (en > (r+wl)>>1) en = (r+wl)>>1;
		if (st > en) { ez->zdropped = 1; break; }
		st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) x1 = x[st - 1], x21 = x2[st - 1], v1 = v[st - 1];
			else x1 = I8(-q - e), x21 = I8(-q2 - e2), v1 = I8(-q - e);
		} else {
			x1 = I8(-q - e), x21 = I8(-q2 - e2);
			v1 = r == 0? I8(-q - e) : r < long_thres? I8(-e) : r == long_thres? I8(long_diff) : I8(-e2);
		}
		if (en >= r) {
			y[r] = I8(-q - e), y2[r] = I8(-q2 - e2);
			u[r] = r == 0? I8(-q - e) : r < long_thres? I8(-e) : r == long_thres? I8(long_diff) : I8(-e2);
		}
		for (t = st0; t <= en0; t += 16) {                                  /* ref: :158-176, 16 bytes per step */
			int i;
			for (i = 0; i < 16; ++i) {
				uint8_t sq = sf[t + i], sq2 = qrr[t + i];
				int8_t sc = sq == sq2? sc_mch : sc_mis;
				if (sq == (uint8_t)(m - 1) || sq2 == (uint8_t)(m - 1)) sc = sc_N;
				s[t + i] = sc;
			}
		}
		st_ = st / 16, en_ = en / 16;
		assert(en_ - st_ + 1 <= n_col_);
		pr = p + ((size_t)r * n_col_ - st_) * 16;
		off[r] = st, off_end[r] = en;
		xprev = x1, x2prev = x21, vprev = v1;
		for (t = st; t <= en; ++t) {                                        /* ref: :182-306 lane by lane */
			int8_t z = s[t], a, b, a2, b2, xt1, x2t1, vt1, ut, tmp, d;
			xt1 = xprev; xprev = x[t];
			vt1 = vprev; vprev = v[t];
			x2t1 = x2prev; x2prev = x2[t];
			ut = u[t];
			a = I8(xt1 + vt1); b = I8(y[t] + ut); a2 = I8(x2t1 + vt1); b2 = I8(y2[t] + ut);
			if (!(flag & EZ_RIGHT)) {
				d = a > z? 1 : 0;  z = z > a? z : a;
				d = b > z? 2 : d;  z = z > b? z : b;
				d = a2 > z? 3 : d; z = z > a2? z : a2;
				d = b2 > z? 4 : d; z = z > b2? z : b2;
			} else {
				d = z > a? 0 : 1;  z = z > a? z : a;
				d = z > b? d : 2;  z = z > b? z : b;
				d = z > a2? d : 3; z = z > a2? z : a2;
				d = z > b2? d : 4; z = z > b2? z : b2;
			}
			z = z < sc_mch? z : sc_mch;
			u[t] = I8(z - vt1); v[t] = I8(z - ut);
			tmp = I8(z - q);  a = I8(a - tmp);  b = I8(b - tmp);
			tmp = I8(z - q2); a2 = I8(a2 - tmp); b2 = I8(b2 - tmp);
			if (!(flag & EZ_RIGHT)) {
				x[t]  = I8((a  > 0? a  : 0) - qe_);  if (a  > 0) d |= 0x08;
				y[t]  = I8((b  > 0? b  : 0) - qe_);  if (b  > 0) d |= 0x10;
				x2[t] = I8((a2 > 0? a2 : 0) - qe2_); if (a2 > 0) d |= 0x20;
				y2[t] = I8((b2 > 0? b2 : 0) - qe2_); if (b2 > 0) d |= 0x40;
			} else {
				x[t]  = I8((a  >= 0? a  : 0) - qe_);  if (a  >= 0) d |= 0x08;
				y[t]  = I8((b  >= 0? b  : 0) - qe_);  if (b  >= 0) d |= 0x10;
				x2[t] = I8((a2 >= 0? a2 : 0) - qe2_); if (a2 >= 0) d |= 0x20;
				y2[t] = I8((b2 >= 0? b2 : 0) - qe2_); if (b2 >= 0) d |= 0x40;
			}
			pr[t] = (uint8_t)d;
		}
		if (!approx_max) {                                                  /* ref: :307-361 */
			int32_t max_H, max_t;
			if (r > 0) {
				int32_t HH[4], tt[4], en1 = st0 + (en0 - st0) / 4 * 4, i;
				max_H = H[en0] = en0 > 0? H[en0-1] + u[en0] : H[en0] + v[en0];
				max_t = en0;
				for (i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
				for (t = st0; t < en1; t += 4)
					for (i = 0; i < 4; ++i) {
						H[t+i] += v[t+i];
						if (H[t+i] > HH[i]) HH[i] = H[t+i], tt[i] = t;
					}
				for (i = 0; i < 4; ++i) if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
				for (; t < en0; ++t) { H[t] += (int32_t)v[t]; if (H[t] > max_H) max_H = H[t], max_t = t; }
			} else H[0] = v[0] - qe, max_H = H[0], max_t = 0;
			if (en0 == tlen - 1 && H[en0] > ez->mte) ez->mte = H[en0], ez->mte_q = r - en;
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe) ez->mqe = H[st0], ez->mqe_t = st0;
			if (ksw_zdrop(ez, max_H, r, max_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
		} else {                                                            /* ref: :362-379 */
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int32_t d0 = v[last_H0_t], d1 = u[last_H0_t + 1];
					if (d0 > d1) H0 += d0; else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += v[last_H0_t];
				else ++last_H0_t, H0 += u[last_H0_t];
			} else H0 = v[0] - qe, last_H0_t = 0;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
		}
		last_st = st, last_en = en;
	}
	free(mem); free(H);
	{   /* ref: :384-392 */
		int rev_cigar = !!(flag & EZ_REV_CIGAR);
		if (!ez->zdropped && !(flag&EZ_EXTZ_ONLY))
			ksw_backtrack_rot(rev_cigar, p, off, off_end, n_col_*16, tlen-1, qlen-1, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
		else if (!ez->zdropped && (flag&EZ_EXTZ_ONLY) && ez->mqe + end_bonus > (int)ez->max) {
			ez->reach_end = 1;
			ksw_backtrack_rot(rev_cigar, p, off, off_end, n_col_*16, ez->mqe_t, qlen-1, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
		} else if (ez->max_t >= 0 && ez->max_q >= 0)
			ksw_backtrack_rot(rev_cigar, p, off, off_end, n_col_*16, ez->max_t, ez->max_q, &ez->m_cigar, &ez->n_cigar, &ez->cigar);
		free(p); free(off);
	}
}

/* ---------------------------------------------------------------- base-level alignment (ref: align.c) */
static void gen_simple_mat(int m, int8_t *mat, int8_t a, int8_t b, int8_t sc_ambi)
{   /* ref: align.c:9-22 */
	int i, j;
	a = a < 0? -a : a; b = b > 0? -b : b; sc_ambi = sc_ambi > 0? -sc_ambi : sc_ambi;
	for (i = 0; i < m - 1; ++i) {
		for (j = 0; j < m - 1; ++j) mat[i * m + j] = i == j? a : b;
		mat[i * m + m - 1] = sc_ambi;
	}
	for (j = 0; j < m; ++j) mat[(m - 1) * m + j] = sc_ambi;
}

static inline void seq_rev(uint32_t len, uint8_t *seq)
{
	uint32_t i; uint8_t t;
	for (i = 0; i < len>>1; ++i) t = seq[i], seq[i] = seq[len - 1 - i], seq[len - 1 - i] = t;
}

static int test_zdrop(const oopt_t *opt, const uint8_t *qseq, const uint8_t *tseq, uint32_t n_cigar, uint32_t *cigar, const int8_t *mat)
{   /* ref: align.c:33-89; the inversion branch is unreachable under MM_F_SR (align.c:72) */
	uint32_t k; int32_t score = 0, max = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
#define UPD(sc_, i_, j_) do { if ((sc_) < max) { int li = (i_) - max_i, lj = (j_) - max_j; int diff = li > lj? li - lj : lj - li; \
		int z = max - (sc_) - diff * opt->e; if (z > max_zdrop) max_zdrop = z; } else max = (sc_), max_i = (i_), max_j = (j_); } while (0)
	for (k = 0, score = 0; k < n_cigar; ++k) {
		uint32_t l, op = cigar[k]&0xf, len = cigar[k]>>4;
		if (op == 0) {
			for (l = 0; l < len; ++l) { score += mat[tseq[i + l] * 5 + qseq[j + l]]; UPD(score, i+(int)l, j+(int)l); }
			i += len, j += len;
		} else if (op == 1 || op == 2 || op == 3) {
			score -= opt->q + opt->e * len;
			if (op == 1) j += len; else i += len;
			UPD(score, i, j);
		}
	}
#undef UPD
	return max_zdrop > opt->zdrop? 1 : 0;
}

static void fix_cigar(oreg_t *r, const uint8_t *qseq, const uint8_t *tseq, int *qshift, int *tshift)
{   /* ref: align.c:91-167 */
	oextra_t *p = r->p; int32_t toff = 0, qoff = 0, to_shrink = 0; uint32_t k;
	*qshift = *tshift = 0;
	if (p->n_cigar <= 1) return;
	for (k = 0; k < p->n_cigar; ++k) {
		uint32_t op = p->cigar[k]&0xf, len = p->cigar[k]>>4;
		if (len == 0) to_shrink = 1;
		if (op == 0) toff += len, qoff += len;
		else if (op == 1 || op == 2) {
			if (k > 0 && k < p->n_cigar - 1 && (p->cigar[k-1]&0xf) == 0 && (p->cigar[k+1]&0xf) == 0) {
				int l, prev_len = p->cigar[k-1] >> 4;
				if (op == 1) { for (l = 0; l < prev_len; ++l) if (qseq[qoff - 1 - l] != qseq[qoff + len - 1 - l]) break; }
				else { for (l = 0; l < prev_len; ++l) if (tseq[toff - 1 - l] != tseq[toff + len - 1 - l]) break; }
				if (l > 0) p->cigar[k-1] -= l<<4, p->cigar[k+1] += l<<4, qoff -= l, toff -= l;
				if (l == prev_len) to_shrink = 1;
			}
			if (op == 1) qoff += len; else toff += len;
		} else if (op == 3) toff += len;
	}
	assert(qoff == r->qe - r->qs && toff == r->re - r->rs);
	for (k = 0; k + 2 < p->n_cigar; ++k) {                                  /* k < n_cigar - 2 (unsigned in ref; n_cigar >= 2 here) */
		if ((p->cigar[k]&0xf) > 0 && (p->cigar[k]&0xf) + (p->cigar[k+1]&0xf) == 3) {
			uint32_t l, s[3] = {0,0,0};
			for (l = k; l < p->n_cigar; ++l) {
				uint32_t op = p->cigar[l]&0xf;
				if (op == 1 || op == 2 || p->cigar[l]>>4 == 0) s[op] += p->cigar[l] >> 4;
				else break;
			}
			if (s[1] > 0 && s[2] > 0 && l - k > 2) {
				p->cigar[k] = s[1]<<4|1; p->cigar[k+1] = s[2]<<4|2;
				for (k += 2; k < l; ++k) p->cigar[k] &= 0xf;
				to_shrink = 1;
			}
			k = l;
		}
	}
	if (to_shrink) {
		int32_t l = 0;
		for (k = 0; k < p->n_cigar; ++k) if (p->cigar[k]>>4 != 0) p->cigar[l++] = p->cigar[k];
		p->n_cigar = l;
		for (k = l = 0; k < p->n_cigar; ++k)
			if (k == p->n_cigar - 1 || (p->cigar[k]&0xf) != (p->cigar[k+1]&0xf)) p->cigar[l++] = p->cigar[k];
			else p->cigar[k+1] += p->cigar[k]>>4<<4;
		p->n_cigar = l;
	}
	if ((p->cigar[0]&0xf) == 1 || (p->cigar[0]&0xf) == 2) {
		int32_t l = p->cigar[0] >> 4;
		if ((p->cigar[0]&0xf) == 1) { if (r->rev) r->qe -= l; else r->qs += l; *qshift = l; }
		else r->rs += l, *tshift = l;
		--p->n_cigar;
		memmove(p->cigar, p->cigar + 1, p->n_cigar * 4);
	}
}

static void update_extra(oreg_t *r, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int8_t q, int8_t e)
{   /* ref: align.c:240-286 */
	uint32_t k, l; int32_t s = 0, max = 0, qshift, tshift, toff = 0, qoff = 0; oextra_t *p = r->p;
	if (p == 0) return;
	fix_cigar(r, qseq, tseq, &qshift, &tshift);
	qseq += qshift, tseq += tshift;
	r->blen = r->mlen = 0;
	for (k = 0; k < p->n_cigar; ++k) {
		uint32_t op = p->cigar[k]&0xf, len = p->cigar[k]>>4;
		if (op == 0) {
			int n_ambi = 0, n_diff = 0;
			for (l = 0; l < len; ++l) {
				int cq = qseq[qoff + l], ct = tseq[toff + l];
				if (ct > 3 || cq > 3) ++n_ambi; else if (ct != cq) ++n_diff;
				s += mat[ct * 5 + cq];
				if (s < 0) s = 0; else max = max > s? max : s;
			}
			r->blen += len - n_ambi, r->mlen += len - (n_ambi + n_diff), p->n_ambi += n_ambi;
			toff += len, qoff += len;
		} else if (op == 1) {
			int n_ambi = 0;
			for (l = 0; l < len; ++l) if (qseq[qoff + l] > 3) ++n_ambi;
			r->blen += len - n_ambi, p->n_ambi += n_ambi;
			s -= q + e * len; if (s < 0) s = 0;
			qoff += len;
		} else if (op == 2) {
			int n_ambi = 0;
			for (l = 0; l < len; ++l) if (tseq[toff + l] > 3) ++n_ambi;
			r->blen += len - n_ambi, p->n_ambi += n_ambi;
			s -= q + e * len; if (s < 0) s = 0;
			toff += len;
		} else if (op == 3) toff += len;
	}
	p->dp_max = max;
	assert(qoff == r->qe - r->qs && toff == r->re - r->rs);
}

static void append_cigar(oreg_t *r, uint32_t n_cigar, uint32_t *cigar)
{   /* ref: align.c:288-311 */
	oextra_t *p;
	if (n_cigar == 0) return;
	if (r->p == 0) {
		uint32_t capacity = n_cigar + sizeof(oextra_t)/4 + 8;
		r->p = (oextra_t*)calloc(capacity, 4); r->p->capacity = capacity;
	} else if (r->p->n_cigar + n_cigar + sizeof(oextra_t)/4 > r->p->capacity) {
		uint32_t oc = r->p->capacity;
		r->p->capacity = (r->p->n_cigar + n_cigar + sizeof(oextra_t)/4) * 2;
		r->p = (oextra_t*)realloc(r->p, r->p->capacity * 4);
		(void)oc;
	}
	p = r->p;
	if (p->n_cigar > 0 && (p->cigar[p->n_cigar-1]&0xf) == (cigar[0]&0xf)) {
		p->cigar[p->n_cigar-1] += cigar[0]>>4<<4;
		if (n_cigar > 1) memcpy(p->cigar + p->n_cigar, cigar + 1, (n_cigar - 1) * 4);
		p->n_cigar += n_cigar - 1;
	} else {
		memcpy(p->cigar + p->n_cigar, cigar, n_cigar * 4);
		p->n_cigar += n_cigar;
	}
}

static void align_pair(const oopt_t *opt, int qlen, const uint8_t *qseq, int tlen, const uint8_t *tseq, const int8_t *mat, int w, int end_bonus, int zdrop, int flag, oksw_t *ez, ostat_t *st)
{   /* ref: align.c:313-339 (max_sw_mat = 0; q != q2 so extd2) */
	int i;
	if (opt->dbg_aln) {
		fprintf(stderr, "===> q=(%d,%d), e=(%d,%d), bw=%d, flag=%d, zdrop=%d <===\n", opt->q, opt->q2, opt->e, opt->e2, w, flag, opt->zdrop);
		for (i = 0; i < tlen; ++i) fputc("ACGTN"[tseq[i]], stderr);
		fputc('\n', stderr);
		for (i = 0; i < qlen; ++i) fputc("ACGTN"[qseq[i]], stderr);
		fputc('\n', stderr);
	}
	o_ksw_extd2(qlen, qseq, tlen, tseq, 5, mat, opt->q, opt->e, opt->q2, opt->e2, w, zdrop, end_bonus, flag, ez);
	if (st) st->n_ksw++;
	if (opt->dbg_aln) {
		fprintf(stderr, "score=%d, cigar=", ez->score);
		for (i = 0; i < ez->n_cigar; ++i) fprintf(stderr, "%d%c", ez->cigar[i]>>4, "MIDN"[ez->cigar[i]&0xf]);
		fprintf(stderr, "\n");
	}
}

static void max_stretch(const oreg_t *r, const o128_t *a, int32_t *as, int32_t *cnt)
{   /* ref: align.c:495-521 */
	int32_t i, score, max_score, len, max_i, max_len;
	*as = r->as, *cnt = r->cnt;
	if (r->cnt < 2) return;
	max_score = -1, max_i = -1, max_len = 0;
	score = a[r->as].y >> 32 & 0xff, len = 1;
	for (i = r->as + 1; i < r->as + r->cnt; ++i) {
		int32_t lq, lr, q_span = a[i].y >> 32 & 0xff;
		lr = (int32_t)a[i].x - (int32_t)a[i-1].x;
		lq = (int32_t)a[i].y - (int32_t)a[i-1].y;
		if (lq == lr) { score += lq < q_span? lq : q_span; ++len; }
		else {
			if (score > max_score) max_score = score, max_len = len, max_i = i - len;
			score = q_span, len = 1;
		}
	}
	if (score > max_score) max_score = score, max_len = len, max_i = i - len;
	*as = max_i, *cnt = max_len;
}

static void align1(const oopt_t *opt, const oidx_t *mi, int qlen, uint8_t *qseq0[2], oreg_t *r, oreg_t *r2, int n_a, o128_t *a, oksw_t *ez, ostat_t *st)
{   /* ref: align.c:565-788, is_sr branch only */
	int32_t rid = a[r->as].x<<1>>33, rev = a[r->as].x>>63, as1, cnt1;
	uint8_t *tseq, *qseq;
	int32_t i, l, bw, dropped = 0, rs0, re0, qs0, qe0, rs, re, qs, qe, rs1, qs1, re1, qe1;
	int8_t mat[25];
	(void)n_a;
	r2->cnt = 0;
	if (r->cnt == 0) return;
	gen_simple_mat(5, mat, opt->a, opt->b, opt->sc_ambi);
	bw = (int)(opt->bw * 1.5 + 1.);
	max_stretch(r, a, &as1, &cnt1);
	rs = (int32_t)a[as1].x + 1 - (int32_t)(a[as1].y>>32&0xff);
	qs = (int32_t)a[as1].y + 1 - (int32_t)(a[as1].y>>32&0xff);
	re = (int32_t)a[as1+cnt1-1].x + 1;
	qe = (int32_t)a[as1+cnt1-1].y + 1;
	assert(cnt1 > 0);
	qs0 = 0, qe0 = qlen;                                                    /* ref: align.c:613-620 */
	l = qs;
	l += l * opt->a + opt->end_bonus > opt->q? (l * opt->a + opt->end_bonus - opt->q) / opt->e : 0;
	rs0 = rs - l > 0? rs - l : 0;
	l = qlen - qe;
	l += l * opt->a + opt->end_bonus > opt->q? (l * opt->a + opt->end_bonus - opt->q) / opt->e : 0;
	re0 = re + l < (int32_t)mi->seq[rid].len? re + l : (int32_t)mi->seq[rid].len;
	assert(re0 > rs0);
	tseq = (uint8_t*)malloc(re0 - rs0 + 16);
	if (st) st->n_regs_aln++, st->n_refbases += re0 - rs0;

	if (qs > 0 && rs > 0) {                                                 /* left extension, ref: align.c:690-705 */
		qseq = &qseq0[rev][qs0];
		idx_getseq(mi, rid, rs0, rs, tseq);
		seq_rev(qs - qs0, qseq); seq_rev(rs - rs0, tseq);
		align_pair(opt, qs - qs0, qseq, rs - rs0, tseq, mat, bw, opt->end_bonus, r->split_inv? opt->zdrop_inv : opt->zdrop, EZ_EXTZ_ONLY|EZ_RIGHT|EZ_REV_CIGAR, ez, st);
		if (ez->n_cigar > 0) { append_cigar(r, ez->n_cigar, ez->cigar); r->p->dp_score += ez->max; }
		rs1 = rs - (ez->reach_end? ez->mqe_t + 1 : ez->max_t + 1);
		qs1 = qs - (ez->reach_end? qs - qs0 : ez->max_q + 1);
		seq_rev(qs - qs0, qseq);
	} else rs1 = rs, qs1 = qs;
	re1 = rs, qe1 = qs;
	assert(qs1 >= 0 && rs1 >= 0);

	for (i = cnt1 - 1; i < cnt1; ++i) {                                     /* ref: align.c:709-758 with is_sr */
		int j, zdrop_code;
		re = (int32_t)a[as1 + i].x + 1;
		qe = (int32_t)a[as1 + i].y + 1;
		re1 = re, qe1 = qe;
		qseq = &qseq0[rev][qs];
		idx_getseq(mi, rid, rs, re, tseq);
		assert(qe - qs == re - rs);
		ksw_reset(ez);
		for (j = 0, ez->score = 0; j < qe - qs; ++j) {
			if (qseq[j] >= 4 || tseq[j] >= 4) ez->score += opt->e2;
			else ez->score += qseq[j] == tseq[j]? opt->a : -opt->b;
		}
		ez->cigar = push_cigar(&ez->n_cigar, &ez->m_cigar, ez->cigar, 0, qe - qs);
		if ((zdrop_code = test_zdrop(opt, qseq, tseq, ez->n_cigar, ez->cigar, mat)) != 0)
			align_pair(opt, qe - qs, qseq, re - rs, tseq, mat, bw, -1, opt->zdrop, 0, ez, st);
		if (ez->n_cigar > 0) append_cigar(r, ez->n_cigar, ez->cigar);
		if (ez->zdropped) {
			for (j = i - 1; j >= 0; --j) if ((int32_t)a[as1 + j].x <= rs + ez->max_t) break;
			dropped = 1;
			if (j < 0) j = 0;
			r->p->dp_score += ez->max;
			re1 = rs + (ez->max_t + 1);
			qe1 = qs + (ez->max_q + 1);
			if (cnt1 - (j + 1) >= opt->min_cnt) split_reg(r, r2, as1 + j + 1 - r->as, qlen, a);
			break;
		} else r->p->dp_score += ez->score;
		rs = re, qs = qe;
	}

	if (!dropped && qe < qe0 && re < re0) {                                 /* right extension, ref: align.c:760-771 */
		qseq = &qseq0[rev][qe];
		idx_getseq(mi, rid, re, re0, tseq);
		align_pair(opt, qe0 - qe, qseq, re0 - re, tseq, mat, bw, opt->end_bonus, opt->zdrop, EZ_EXTZ_ONLY, ez, st);
		if (ez->n_cigar > 0) { append_cigar(r, ez->n_cigar, ez->cigar); r->p->dp_score += ez->max; }
		re1 = re + (ez->reach_end? ez->mqe_t + 1 : ez->max_t + 1);
		qe1 = qe + (ez->reach_end? qe0 - qe : ez->max_q + 1);
	}
	assert(qe1 <= qlen);
	r->rs = rs1, r->re = re1;
	if (rev) r->qs = qlen - qe1, r->qe = qlen - qs1;
	else r->qs = qs1, r->qe = qe1;
	assert(re1 - rs1 <= re0 - rs0);
	if (r->p) {
		idx_getseq(mi, rid, rs1, re1, tseq);
		update_extra(r, &qseq0[r->rev][qs1], tseq, mat, opt->q, opt->e);
		if (st) st->n_cigar += r->p->n_cigar;
	}
	free(tseq);
}

static oreg_t *align_skeleton(const oopt_t *opt, const oidx_t *mi, int qlen, const char *qstr, int *n_regs_, oreg_t *regs, o128_t *a, ostat_t *st)
{   /* ref: align.c:857-913 (no splice, no inversion under sr) */
	int32_t i, n_regs = *n_regs_, n_a; uint8_t *qseq0[2]; oksw_t ez;
	qseq0[0] = (uint8_t*)malloc(qlen * 2 + 32); qseq0[1] = qseq0[0] + qlen;
	for (i = 0; i < qlen; ++i) {
		qseq0[0][i] = nt4[(uint8_t)qstr[i]];
		qseq0[1][qlen - 1 - i] = qseq0[0][i] < 4? 3 - qseq0[0][i] : 4;
	}
	n_a = squeeze_a(n_regs, regs, a);
	memset(&ez, 0, sizeof(ez));
	for (i = 0; i < n_regs; ++i) {
		oreg_t r2;
		align1(opt, mi, qlen, qseq0, &regs[i], &r2, n_a, a, &ez, st);
		if (r2.cnt > 0) {                                                   /* ref: align.c:847-855 mm_insert_reg */
			regs = (oreg_t*)realloc(regs, (n_regs + 1) * sizeof(oreg_t));
			if (i + 1 != n_regs) memmove(&regs[i + 2], &regs[i + 1], sizeof(oreg_t) * (n_regs - i - 1));
			regs[i + 1] = r2; ++n_regs;
		}
	}
	*n_regs_ = n_regs;
	free(qseq0[0]); free(ez.cigar);
	filter_regs(opt, qlen, n_regs_, regs);
	hit_sort(n_regs_, regs);
	return regs;
}

/* ---------------------------------------------------------------- MAPQ + pairing (ref: hit.c:446-491, pe.c:45-177) */
static void set_mapq(int n_regs, oreg_t *regs, int min_chain_sc, int match_sc, int rep_len)
{
	static const float q_coef = 40.0f; int64_t sum_sc = 0; float uniq_ratio; int i;
	if (n_regs == 0) return;
	for (i = 0; i < n_regs; ++i) if (regs[i].parent == regs[i].id) sum_sc += regs[i].score;
	uniq_ratio = (float)sum_sc / (sum_sc + rep_len);
	for (i = 0; i < n_regs; ++i) {
		oreg_t *r = &regs[i];
		if (r->inv) r->mapq = 0;
		else if (r->parent == r->id) {
			int mapq, subsc;
			float pen_s1 = (r->score > 100? 1.0f : 0.01f * r->score) * uniq_ratio;
			float pen_cm = r->cnt > 10? 1.0f : 0.1f * r->cnt;
			pen_cm = pen_s1 < pen_cm? pen_s1 : pen_cm;
			subsc = r->subsc > min_chain_sc? r->subsc : min_chain_sc;
			if (r->p && r->p->dp_max2 > 0 && r->p->dp_max > 0) {
				float identity = (float)r->mlen / r->blen;
				float x = (float)r->p->dp_max2 * subsc / r->p->dp_max / r->score0;
				mapq = (int)(identity * pen_cm * q_coef * (1.0f - x * x) * logf((float)r->p->dp_max / match_sc));
			} else {
				float x = (float)subsc / r->score0;
				if (r->p) {
					float identity = (float)r->mlen / r->blen;
					mapq = (int)(identity * pen_cm * q_coef * (1.0f - x) * logf((float)r->p->dp_max / match_sc));
				} else mapq = (int)(pen_cm * q_coef * (1.0f - x) * logf(r->score));
			}
			mapq -= (int)(4.343f * logf(r->n_sub + 1) + .499f);
			mapq = mapq > 0? mapq : 0;
			r->mapq = mapq < 60? mapq : 60;
			if (r->p && r->p->dp_max > r->p->dp_max2 && r->mapq == 0) r->mapq = 1;
		} else r->mapq = 0;
	}
}

static void set_pe_thru(const int *qlens, int *n_regs, oreg_t **regs)
{   /* ref: pe.c:45-64 */
	int s, i, n_pri[2] = {0,0}, pri[2] = {-1,-1};
	for (s = 0; s < 2; ++s)
		for (i = 0; i < n_regs[s]; ++i)
			if (regs[s][i].id == regs[s][i].parent) ++n_pri[s], pri[s] = i;
	if (n_pri[0] == 1 && n_pri[1] == 1) {
		oreg_t *p = &regs[0][pri[0]], *q = &regs[1][pri[1]];
		if (p->rid == q->rid && p->rev == q->rev && abs(p->rs - q->rs) < 3 && abs(p->re - q->re) < 3
			&& ((p->qs == 0 && qlens[1] - q->qe == 0) || (q->qs == 0 && qlens[0] - p->qe == 0)))
			p->pe_thru = q->pe_thru = 1;
	}
}

static void pair_regs(int max_gap_ref, int pe_bonus, int sub_diff, int match_sc, const int *qlens, int *n_regs, oreg_t **regs)
{   /* ref: pe.c:76-177 */
	int i, j, s, n, last[2], dp_thres, segs = 0, max_idx[2]; int64_t max; opair_t *a;
	uint64_t *sc = 0; size_t n_sc = 0, m_sc = 0;
	a = (opair_t*)malloc((n_regs[0] + n_regs[1] + 1) * sizeof(opair_t));
	for (s = n = 0, dp_thres = 0; s < 2; ++s) {
		int mx = 0;
		for (i = 0; i < n_regs[s]; ++i) {
			a[n].s = s; a[n].r = &regs[s][i]; a[n].rev = a[n].r->rev;
			a[n].key = (uint64_t)a[n].r->rid << 32 | a[n].r->rs<<1 | (s^a[n].rev);
			mx = mx > a[n].r->p->dp_max? mx : a[n].r->p->dp_max;
			++n; segs |= 1<<s;
		}
		dp_thres += mx;
	}
	if (segs != 3) { free(a); return; }
	dp_thres -= pe_bonus;
	if (dp_thres < 0) dp_thres = 0;
	radix_pair(a, a + n);
	max = -1; max_idx[0] = max_idx[1] = -1; last[0] = last[1] = -1;
	for (i = 0; i < n; ++i) {
		if (a[i].key & 1) {
			oreg_t *q, *r;
			if (last[a[i].rev] < 0) continue;
			r = a[i].r; q = a[last[a[i].rev]].r;
			if (r->rid != q->rid || r->rs - q->re > max_gap_ref) continue;
			for (j = last[a[i].rev]; j >= 0; --j) {
				int64_t score;
				if (a[j].rev != a[i].rev || a[j].s == a[i].s) continue;
				q = a[j].r;
				if (r->rid != q->rid || r->rs - q->re > max_gap_ref) break;
				if (r->p->dp_max + q->p->dp_max < dp_thres) continue;
				score = (int64_t)(r->p->dp_max + q->p->dp_max) << 32 | (r->hash + q->hash);
				if (score > max) max = score, max_idx[a[j].s] = j, max_idx[a[i].s] = i;
				if (n_sc == m_sc) { m_sc = m_sc? m_sc<<1 : 16; sc = (uint64_t*)realloc(sc, m_sc * 8); }
				sc[n_sc++] = score;
			}
		} else last[a[i].rev] = i;
	}
	if (n_sc > 1) radix_64(sc, sc + n_sc);
	if (n_sc > 0 && max > 0) {
		int n_sub = 0, mapq_pe; oreg_t *r[2];
		r[0] = a[max_idx[0]].r, r[1] = a[max_idx[1]].r;
		r[0]->proper_frag = r[1]->proper_frag = 1;
		for (s = 0; s < 2; ++s) {
			if (r[s]->id != r[s]->parent) {
				oreg_t *p = &regs[s][r[s]->parent];
				for (i = 0; i < n_regs[s]; ++i) if (regs[s][i].parent == p->id) regs[s][i].parent = r[s]->id;
				p->mapq = 0;
			}
			if (!r[s]->sam_pri) {
				for (i = 0; i < n_regs[s]; ++i) regs[s][i].sam_pri = 0;
				r[s]->sam_pri = 1;
			}
		}
		mapq_pe = r[0]->mapq > r[1]->mapq? r[0]->mapq : r[1]->mapq;
		for (i = 0; i < (int)n_sc; ++i) if ((sc[i]>>32) + sub_diff >= (uint64_t)max>>32) ++n_sub;
		if (n_sc > 1) {
			int mapq_pe_alt = (int)(6.02f * ((max>>32) - (sc[n_sc - 2]>>32)) / match_sc - 4.343f * logf(n_sub));
			mapq_pe = mapq_pe < mapq_pe_alt? mapq_pe : mapq_pe_alt;
		}
		if ((int)r[0]->mapq < mapq_pe) r[0]->mapq = (int)(.2f * r[0]->mapq + .8f * mapq_pe + .499f);
		if ((int)r[1]->mapq < mapq_pe) r[1]->mapq = (int)(.2f * r[1]->mapq + .8f * mapq_pe + .499f);
		if (n_sc == 1) { if (r[0]->mapq < 2) r[0]->mapq = 2; if (r[1]->mapq < 2) r[1]->mapq = 2; }
		else if ((uint64_t)max>>32 > sc[n_sc - 2]>>32) { if (r[0]->mapq < 1) r[0]->mapq = 1; if (r[1]->mapq < 1) r[1]->mapq = 1; }
	}
	free(a); free(sc);
	set_pe_thru(qlens, n_regs, regs);
}

/* ---------------------------------------------------------------- per-fragment driver (ref: map.c:272-424) */
static inline uint32_t wang_hash(uint32_t key)
{   /* ref: khash.h:400-409 */
	key += ~(key << 15); key ^= (key >> 10); key += (key << 3); key ^= (key >> 6); key += ~(key << 11); key ^= (key >> 16);
	return key;
}
uint32_t o_qname_hash(const char *qname, int qlen_sum, int seed)
{   /* ref: map.c:291-293, khash.h:383-388 */
	uint32_t h = 0;
	if (qname) { const char *s = qname; h = (uint32_t)*s; if (h) for (++s; *s; ++s) h = (h << 5) - h + (uint32_t)*s; }
	h ^= wang_hash(qlen_sum) + wang_hash(seed);
	return wang_hash(h);
}

static void dbg_print_seeds(const oidx_t *mi, const char *tag, int id, const o128_t *a, int64_t st, int64_t en)
{
	int64_t i;
	for (i = st; i < en; ++i) {
		if (id >= 0) fprintf(stderr, "%s\t%d\t", tag, id); else fprintf(stderr, "%s\t", tag);
		fprintf(stderr, "%s\t%d\t%c\t%d\t%d\t%d\n", mi->seq[a[i].x<<1>>33].name, (int32_t)a[i].x, "+-"[a[i].x>>63], (int32_t)a[i].y, (int32_t)(a[i].y>>32&0xff),
				i == st? 0 : ((int32_t)a[i].y - (int32_t)a[i-1].y) - ((int32_t)a[i].x - (int32_t)a[i-1].x));
	}
}

static oreg_t *align_regs(const oopt_t *opt, const oidx_t *mi, int qlen, const char *seq, int *n_regs, oreg_t *regs, o128_t *a, ostat_t *st)
{   /* ref: map.c:260-270 */
	regs = align_skeleton(opt, mi, qlen, seq, n_regs, regs, a, st);
	set_parent(opt->mask_level, *n_regs, regs, opt->a * 2 + opt->b);
	select_sub(opt->pri_ratio, mi->k*2, opt->best_n, n_regs, regs);
	set_sam_pri(*n_regs, regs);
	return regs;
}

void o_map_frag(const oidx_t *mi, const oopt_t *opt, int n_segs, const int *qlens, const char **seqs,
                int *n_regs, oreg_t **regs, const char *qname, int *rep_len_, ostat_t *st)
{
	int i, j, rep_len, qlen_sum, n_regs0, max_chain_gap_qry, max_chain_gap_ref, sum = 0;
	uint32_t hash; int64_t n_a; uint64_t *u; o128_t *a, *mv = 0; size_t n_mv = 0, m_mv = 0, n0 = 0; oreg_t *regs0;
	for (i = 0, qlen_sum = 0; i < n_segs; ++i) qlen_sum += qlens[i], n_regs[i] = 0, regs[i] = 0;
	*rep_len_ = 0;
	if (qlen_sum == 0 || n_segs <= 0 || n_segs > 255) return;
	hash = o_qname_hash(qname, qlen_sum, opt->seed);
	for (i = 0; i < n_segs; ++i) {                                          /* ref: map.c:64-77 */
		if (qlens[i] > 0) o_sketch(seqs[i], qlens[i], mi->w, mi->k, i, &mv, &n_mv, &m_mv);
		for (j = n0; j < (int)n_mv; ++j) mv[j].y += (uint64_t)sum << 1;
		sum += qlens[i], n0 = n_mv;
	}
	a = o_collect_seeds(mi, opt->mid_occ, mv, n_mv, qlen_sum, &n_a, &rep_len);
	if (st) st->n_mini += n_mv, st->n_anchor += n_a;
	if (opt->dbg_seeds) { fprintf(stderr, "RS\t%d\n", rep_len); dbg_print_seeds(mi, "SD", -1, a, 0, n_a); }
	max_chain_gap_qry = qlen_sum > opt->max_gap? qlen_sum : opt->max_gap;   /* ref: map.c:341-351 */
	if (opt->max_gap_ref > 0) max_chain_gap_ref = opt->max_gap_ref;
	else if (opt->max_frag_len > 0) { max_chain_gap_ref = opt->max_frag_len - qlen_sum; if (max_chain_gap_ref < opt->max_gap) max_chain_gap_ref = opt->max_gap; }
	else max_chain_gap_ref = opt->max_gap;
	a = o_chain_dp(max_chain_gap_ref, max_chain_gap_qry, opt->bw, opt->max_chain_skip, opt->max_chain_iter, opt->min_cnt, opt->min_chain_score, n_segs, n_a, a, &n_regs0, &u);
	if (opt->max_occ > opt->mid_occ && rep_len > 0) {                       /* ref: map.c:353-375 */
		int rechain = 0;
		if (n_regs0 > 0) {
			int n_chained_segs = 1, max = 0, max_i = -1, max_off = -1, off = 0;
			for (i = 0; i < n_regs0; ++i) { if (max < (int)(u[i]>>32)) max = u[i]>>32, max_i = i, max_off = off; off += (uint32_t)u[i]; }
			for (i = 1; i < (int32_t)u[max_i]; ++i)
				if ((a[max_off+i].y&SEED_SEG_MASK) != (a[max_off+i-1].y&SEED_SEG_MASK)) ++n_chained_segs;
			if (n_chained_segs < n_segs) rechain = 1;
		} else rechain = 1;
		if (rechain) {
			free(a); free(u);
			a = o_collect_seeds(mi, opt->max_occ, mv, n_mv, qlen_sum, &n_a, &rep_len);
			if (st) st->n_anchor += n_a;
			a = o_chain_dp(max_chain_gap_ref, max_chain_gap_qry, opt->bw, opt->max_chain_skip, opt->max_chain_iter, opt->min_cnt, opt->min_chain_score, n_segs, n_a, a, &n_regs0, &u);
		}
	}
	*rep_len_ = rep_len;
	regs0 = gen_regs(hash, qlen_sum, n_regs0, u, a);
	if (opt->dbg_seeds) for (j = 0; j < n_regs0; ++j) dbg_print_seeds(mi, "CN", j, a, regs0[j].as, regs0[j].as + regs0[j].cnt);
	set_parent(opt->mask_level, n_regs0, regs0, opt->a * 2 + opt->b);      /* chain_post, ref: map.c:249-258 */
	if (n_segs <= 1) select_sub(opt->pri_ratio, mi->k*2, opt->best_n, &n_regs0, regs0);
	else select_sub_multi(opt->pri_ratio, 0.2f, 0.7f, max_chain_gap_ref, mi->k*2, opt->best_n, n_segs, qlens, &n_regs0, regs0);
	if (n_segs == 1) {
		regs0 = align_regs(opt, mi, qlens[0], seqs[0], &n_regs0, regs0, a, st);
		set_mapq(n_regs0, regs0, opt->min_chain_score, opt->a, rep_len);
		n_regs[0] = n_regs0, regs[0] = regs0;
	} else {
		oseg_t *seg = seg_gen(hash, n_segs, qlens, n_regs0, regs0, n_regs, regs, a);
		free(regs0);
		for (i = 0; i < n_segs; ++i) {
			set_parent(opt->mask_level, n_regs[i], regs[i], opt->a * 2 + opt->b);
			regs[i] = align_regs(opt, mi, qlens[i], seqs[i], &n_regs[i], regs[i], seg[i].a, st);
			set_mapq(n_regs[i], regs[i], opt->min_chain_score, opt->a, rep_len);
		}
		for (i = 0; i < n_segs; ++i) free(seg[i].u), free(seg[i].a);
		free(seg);
		if (n_segs == 2 && opt->pe_ori >= 0)
			pair_regs(max_chain_gap_ref, opt->pe_bonus, opt->a * 2 + opt->b, opt->a, qlens, n_regs, regs);
	}
	free(mv); free(a); free(u);
}

static void revcomp_read(oread_t *s)
{   /* ref: bseq.h:46-58 */
	int i, t, l = s->l_seq;
	for (i = 0; i < l>>1; ++i) {
		t = s->seq[l - i - 1];
		s->seq[l - i - 1] = comp_tab[(uint8_t)s->seq[i]];
		s->seq[i] = comp_tab[t];
	}
	if (l&1) s->seq[l>>1] = comp_tab[(uint8_t)s->seq[l>>1]];
	if (s->qual) for (i = 0; i < l>>1; ++i) t = s->qual[l - i - 1], s->qual[l - i - 1] = s->qual[i], s->qual[i] = t;
}

void o_map_reads(const oidx_t *mi, const oopt_t *opt, int n_segs, oread_t *reads, int *n_regs, oreg_t **regs, int *rep_len, ostat_t *st)
{   /* ref: map.c:458-498 worker_for */
	int qlens[255], j, k, pe_ori = opt->pe_ori; const char *qseqs[255];
	tabs();
	for (j = 0; j < n_segs; ++j) {
		if (n_segs == 2 && ((j == 0 && (pe_ori>>1&1)) || (j == 1 && (pe_ori&1)))) revcomp_read(&reads[j]);
		qlens[j] = reads[j].l_seq; qseqs[j] = reads[j].seq;
	}
	o_map_frag(mi, opt, n_segs, qlens, qseqs, n_regs, regs, reads[0].name, rep_len, st);
	if (st) st->n_reads += n_segs;
	for (j = 0; j < n_segs; ++j)
		if (n_segs == 2 && ((j == 0 && (pe_ori>>1&1)) || (j == 1 && (pe_ori&1)))) {
			revcomp_read(&reads[j]);
			for (k = 0; k < n_regs[j]; ++k) {
				oreg_t *r = &regs[j][k]; int t = r->qs;
				r->qs = qlens[j] - r->qe; r->qe = qlens[j] - t; r->rev = !r->rev;
			}
		}
}

int o_alser_count(const oidx_t *mi, const oopt_t *opt, int qlen, const char *seq)
{   /* ref: map.c:299-312 as driven by main.c:384-391 (mm_map => one segment, qname = NULL) */
	o128_t *mv = 0, *a; size_t n_mv = 0, m_mv = 0; int64_t n_a, i; int rep_len, seed_num = 0, cnt = 0;
	if (qlen <= 0) return 0;
	o_sketch(seq, qlen, mi->w, mi->k, 0, &mv, &n_mv, &m_mv);
	a = o_collect_seeds(mi, opt->mid_occ, mv, n_mv, qlen, &n_a, &rep_len);
	for (i = 1; i < n_a; ++i) {
		if (((int32_t)a[i].x - (int32_t)a[i-1].x) > qlen) { if (seed_num >= opt->min_cnt - 1) ++cnt; seed_num = 0; }
		else ++seed_num;
	}
	free(mv); free(a);
	return cnt;
}

/* ---------------------------------------------------------------- SAM (ref: format.c:82-135,276-302,361-544) */
void o_write_sam_hdr(FILE *fp, const oidx_t *mi, const char *rg, char *rg_id)
{
	uint32_t i;
	for (i = 0; i < mi->n_seq; ++i) fprintf(fp, "@SQ\tSN:%s\tLN:%d\n", mi->seq[i].name, mi->seq[i].len);
	if (rg_id) rg_id[0] = 0;
	if (rg && strstr(rg, "@RG") == rg && strchr(rg, '\t') == 0) {           /* ref: format.c:82-114 */
		char *line = strdup(rg), *p, *q;
		for (p = q = line; *p; ++p) {
			if (*p == '\\') { ++p; if (*p == 't') *q++ = '\t'; else if (*p == '\\') *q++ = '\\'; }
			else *q++ = *p;
		}
		*q = 0;
		if ((p = strstr(line, "\tID:")) != 0) {
			char *r = rg_id;
			for (p += 4; *p && *p != '\t' && *p != '\n'; ++p) if (rg_id) *r++ = *p;
			if (rg_id) *r = 0;
			fprintf(fp, "%s\n", line);
		}
		free(line);
	}
	fprintf(fp, "@PG\tID:minimap2\tPN:minimap2\n");
}

static int qname_len(const char *s)
{   /* ref: bseq.h:31-36 */
	int l = strlen(s);
	return l >= 3 && s[l-1] >= '0' && s[l-1] <= '9' && s[l-2] == '/'? l - 2 : l;
}

static const oreg_t *get_sam_pri(int n_regs, const oreg_t *regs)
{
	int i;
	for (i = 0; i < n_regs; ++i) if (regs[i].sam_pri) return &regs[i];
	return 0;
}

static double event_identity(const oreg_t *r)
{   /* ref: format.c:263-274 */
	int32_t i, n_gapo = 0, n_gap = 0;
	for (i = 0; i < (int32_t)r->p->n_cigar; ++i) {
		int32_t op = r->p->cigar[i] & 0xf, len = r->p->cigar[i] >> 4;
		if (op == 1 || op == 2) ++n_gapo, n_gap += len;
	}
	return (double)r->mlen / (r->blen - n_gap + n_gapo);
}

static char *put_sq(char *o, const char *seq, int l, int rev, int comp)
{   /* ref: format.c:341-353 */
	int i;
	if (rev) for (i = 0; i < l; ++i) { int c = (uint8_t)seq[l - 1 - i]; *o++ = c < 128 && comp? comp_tab[c] : c; }
	else { memcpy(o, seq, l); o += l; }
	return o;
}

int o_write_sam(char *buf, const oidx_t *mi, const oread_t *t, int seg_idx, int reg_idx, int n_seg,
                const int *n_regss, oreg_t *const *regss, const char *rg_id, int rep_len)
{
	char *o = buf; int flag, n_regs = n_regss[seg_idx];
	int this_rid = -1, this_pos = -1;
	const oreg_t *regs = regss[seg_idx], *r_prev = 0, *r_next = 0;
	const oreg_t *r = n_regs > 0 && reg_idx < n_regs && reg_idx >= 0? &regs[reg_idx] : 0;
	tabs();
	if (n_seg > 1) { int next_sid = (seg_idx + 1) % n_seg; r_next = get_sam_pri(n_regss[next_sid], regss[next_sid]); r_prev = r_next; }
	{ int l = n_seg > 1? qname_len(t->name) : (int)strlen(t->name); memcpy(o, t->name, l); o += l; }
	flag = n_seg > 1? 0x1 : 0x0;
	if (r == 0) flag |= 0x4;
	else { if (r->rev) flag |= 0x10; if (r->parent != r->id) flag |= 0x100; else if (!r->sam_pri) flag |= 0x800; }
	if (n_seg > 1) {
		if (r && r->proper_frag) flag |= 0x2;
		if (seg_idx == 0) flag |= 0x40; else if (seg_idx == n_seg - 1) flag |= 0x80;
		if (r_next == 0) flag |= 0x8; else if (r_next->rev) flag |= 0x20;
	}
	o += sprintf(o, "\t%d", flag);
	if (r == 0) {
		if (r_prev) { this_rid = r_prev->rid, this_pos = r_prev->rs; o += sprintf(o, "\t%s\t%d\t0\t*", mi->seq[this_rid].name, this_pos+1); }
		else o += sprintf(o, "\t*\t0\t0\t*");
	} else {
		uint32_t k, clip_len[2];
		this_rid = r->rid, this_pos = r->rs;
		o += sprintf(o, "\t%s\t%d\t%d\t", mi->seq[r->rid].name, r->rs+1, r->mapq);
		if (r->p == 0) *o++ = '*';
		else {                                                              /* ref: format.c:361-385 */
			int clip_char = (flag&0x800)? 'H' : 'S';
			clip_len[0] = r->rev? t->l_seq - r->qe : r->qs;
			clip_len[1] = r->rev? r->qs : t->l_seq - r->qe;
			if (clip_len[0]) o += sprintf(o, "%d%c", clip_len[0], clip_char);
			for (k = 0; k < r->p->n_cigar; ++k) o += sprintf(o, "%d%c", r->p->cigar[k]>>4, "MIDNSHP=XB"[r->p->cigar[k]&0xf]);
			if (clip_len[1]) o += sprintf(o, "%d%c", clip_len[1], clip_char);
		}
	}
	if (n_seg > 1) {                                                        /* ref: format.c:455-480 */
		int tlen = 0;
		if (this_rid >= 0 && r_next) {
			if (this_rid == r_next->rid) {
				if (r) {
					int this_pos5 = r->rev? r->re - 1 : this_pos;
					int next_pos5 = r_next->rev? r_next->re - 1 : r_next->rs;
					tlen = next_pos5 - this_pos5;
				}
				o += sprintf(o, "\t=\t");
			} else o += sprintf(o, "\t%s\t", mi->seq[r_next->rid].name);
			o += sprintf(o, "%d\t", r_next->rs + 1);
		} else if (r_next) o += sprintf(o, "\t%s\t%d\t", mi->seq[r_next->rid].name, r_next->rs + 1);
		else if (this_rid >= 0) o += sprintf(o, "\t=\t%d\t", this_pos + 1);
		else o += sprintf(o, "\t*\t0\t");
		if (tlen > 0) ++tlen; else if (tlen < 0) --tlen;
		o += sprintf(o, "%d\t", tlen);
	} else o += sprintf(o, "\t*\t0\t0\t");
	if (r == 0) {
		o = put_sq(o, t->seq, t->l_seq, 0, 0); *o++ = '\t';
		if (t->qual) o = put_sq(o, t->qual, t->l_seq, 0, 0); else *o++ = '*';
	} else if ((flag & 0x900) == 0) {
		o = put_sq(o, t->seq, t->l_seq, r->rev, r->rev); *o++ = '\t';
		if (t->qual) o = put_sq(o, t->qual, t->l_seq, r->rev, 0); else *o++ = '*';
	} else if (flag & 0x100) { *o++ = '*'; *o++ = '\t'; *o++ = '*'; }
	else {
		o = put_sq(o, t->seq + r->qs, r->qe - r->qs, r->rev, r->rev); *o++ = '\t';
		if (t->qual) o = put_sq(o, t->qual + r->qs, r->qe - r->qs, r->rev, 0); else *o++ = '*';
	}
	if (rg_id && rg_id[0]) o += sprintf(o, "\tRG:Z:%s", rg_id);
	if (r) {                                                                /* ref: format.c:276-302 write_tags */
		int type = r->id == r->parent? (r->inv? 'I' : 'P') : (r->inv? 'i' : 'S');
		if (r->p) o += sprintf(o, "\tNM:i:%d\tms:i:%d\tAS:i:%d\tnn:i:%d", r->blen - r->mlen + r->p->n_ambi, r->p->dp_max, r->p->dp_score, r->p->n_ambi);
		o += sprintf(o, "\ttp:A:%c\tcm:i:%d\ts1:i:%d", type, r->cnt, r->score);
		if (r->parent == r->id) o += sprintf(o, "\ts2:i:%d", r->subsc);
		if (r->p) {
			double div = 1.0 - event_identity(r);
			if (div == 0.0) o += sprintf(o, "\tde:f:0"); else o += sprintf(o, "\tde:f:%.4f", div);
		}
		if (r->split) o += sprintf(o, "\tzd:i:%d", r->split);
		if (r->parent == r->id && r->p && n_regs > 1) {                     /* ref: format.c:508-535 SA tag */
			int i, n_sa = 0;
			for (i = 0; i < n_regs; ++i) if (i != r - regs && regs[i].parent == regs[i].id && regs[i].p) ++n_sa;
			if (n_sa > 0) {
				o += sprintf(o, "\tSA:Z:");
				for (i = 0; i < n_regs; ++i) {
					const oreg_t *q = &regs[i]; int l_M, l_I = 0, l_D = 0, clip5, clip3;
					if (r == q || q->parent != q->id || q->p == 0) continue;
					if (q->qe - q->qs < q->re - q->rs) l_M = q->qe - q->qs, l_D = (q->re - q->rs) - l_M;
					else l_M = q->re - q->rs, l_I = (q->qe - q->qs) - l_M;
					clip5 = q->rev? t->l_seq - q->qe : q->qs;
					clip3 = q->rev? q->qs : t->l_seq - q->qe;
					o += sprintf(o, "%s,%d,%c,", mi->seq[q->rid].name, q->rs+1, "+-"[q->rev]);
					if (clip5) o += sprintf(o, "%dS", clip5);
					if (l_M) o += sprintf(o, "%dM", l_M);
					if (l_I) o += sprintf(o, "%dI", l_I);
					if (l_D) o += sprintf(o, "%dD", l_D);
					if (clip3) o += sprintf(o, "%dS", clip3);
					o += sprintf(o, ",%d,%d;", q->mapq, q->blen - q->mlen + q->p->n_ambi);
				}
			}
		}
	}
	if (rep_len >= 0) o += sprintf(o, "\trl:i:%d", rep_len);
	*o++ = '\n'; *o = 0;
	return (int)(o - buf);
}

/* ---------------------------------------------------------------- whole-file driver (ref: map.c:557-700) */
typedef struct { oread_t rd[2]; int n_seg, n_regs[2], rep_len; oreg_t *regs[2]; } ofrag_t;
typedef struct { const oidx_t *mi; const oopt_t *opt; ofrag_t *fr; long n; volatile long next; ostat_t *st; pthread_mutex_t *mtx; } owork_t;

static void *map_worker(void *arg)
{
	owork_t *w = (owork_t*)arg; ostat_t st; memset(&st, 0, sizeof(st));
	for (;;) {
		long i = __sync_fetch_and_add(&w->next, 1);
		if (i >= w->n) break;
		o_map_reads(w->mi, w->opt, w->fr[i].n_seg, w->fr[i].rd, w->fr[i].n_regs, w->fr[i].regs, &w->fr[i].rep_len, &st);
	}
	if (w->st) {
		pthread_mutex_lock(w->mtx);
		w->st->n_reads += st.n_reads; w->st->n_mini += st.n_mini; w->st->n_anchor += st.n_anchor; w->st->n_regs_aln += st.n_regs_aln;
		w->st->n_refbases += st.n_refbases; w->st->n_cigar += st.n_cigar; w->st->n_ksw += st.n_ksw;
		pthread_mutex_unlock(w->mtx);
	}
	return 0;
}

static int qname_same(const char *a, const char *b) { int l1 = qname_len(a), l2 = qname_len(b); return l1 == l2 && strncmp(a, b, l1) == 0; }

long o_map_files(const oidx_t *mi, const oopt_t *opt, const char *fn1, const char *fn2, FILE *out, const char *rg_id, int n_threads, ostat_t *st)
{
	ofile_t *f1 = of_open(fn1), *f2 = fn2? of_open(fn2) : 0;
	ostr_t nm = {0,0,0}, sq = {0,0,0}, ql = {0,0,0}; long total = 0; int done = 0;
	const long BATCH = 100000; char *buf = 0; size_t mbuf = 0; oread_t pend; int has_pend = 0;
	pthread_mutex_t mtx = PTHREAD_MUTEX_INITIALIZER;
	if (!f1 || (fn2 && !f2)) { of_close(f1); of_close(f2); return -1; }
	tabs();
	memset(&pend, 0, sizeof(pend));
	while (!done) {
		ofrag_t *fr = (ofrag_t*)calloc(BATCH, sizeof(ofrag_t)); long n = 0, i; int j, k;
		while (n < BATCH) {
			ofrag_t *f = &fr[n];
			if (f2) {                                                       /* ref: bseq.c:129-160 lock-step */
				int l1 = of_read(f1, &nm, &sq, &ql);
				if (l1 < 0) { done = 1; break; }
				f->rd[0].l_seq = l1; f->rd[0].name = strdup(nm.s); f->rd[0].seq = strdup(sq.s? sq.s : ""); f->rd[0].qual = ql.l? strdup(ql.s) : 0;
				l1 = of_read(f2, &nm, &sq, &ql);
				if (l1 < 0) { free(f->rd[0].name); free(f->rd[0].seq); free(f->rd[0].qual); done = 1; break; }
				f->rd[1].l_seq = l1; f->rd[1].name = strdup(nm.s); f->rd[1].seq = strdup(sq.s? sq.s : ""); f->rd[1].qual = ql.l? strdup(ql.s) : 0;
				f->n_seg = 2;
			} else {                                                        /* frag_mode grouping of adjacent same-name reads, ref: map.c:580-586 */
				if (!has_pend) {
					int l1 = of_read(f1, &nm, &sq, &ql);
					if (l1 < 0) { done = 1; break; }
					pend.l_seq = l1; pend.name = strdup(nm.s); pend.seq = strdup(sq.s? sq.s : ""); pend.qual = ql.l? strdup(ql.s) : 0;
				}
				f->rd[0] = pend; f->n_seg = 1; has_pend = 0;
				{
					int l1 = of_read(f1, &nm, &sq, &ql);
					if (l1 >= 0) {
						pend.l_seq = l1; pend.name = strdup(nm.s); pend.seq = strdup(sq.s? sq.s : ""); pend.qual = ql.l? strdup(ql.s) : 0;
						if (qname_same(f->rd[0].name, pend.name)) { f->rd[1] = pend; f->n_seg = 2; has_pend = 0; }
						else has_pend = 1;
					} else { done = 1; ++n; break; }
				}
			}
			for (j = 0; j < f->n_seg; ++j) { char *s = f->rd[j].seq; for (k = 0; s[k]; ++k) if (s[k] == 'u' || s[k] == 'U') --s[k]; } /* ref: bseq.c:72-74 */
			++n;
		}
		if (n > 0) {
			owork_t w; pthread_t *tid; int nt = n_threads > 1? n_threads : 1;
			w.mi = mi; w.opt = opt; w.fr = fr; w.n = n; w.next = 0; w.st = st; w.mtx = &mtx;
			if (nt == 1 || opt->dbg_seeds || opt->dbg_aln) map_worker(&w);
			else { tid = (pthread_t*)malloc(nt * sizeof(pthread_t)); for (j = 0; j < nt; ++j) pthread_create(&tid[j], 0, map_worker, &w); for (j = 0; j < nt; ++j) pthread_join(tid[j], 0); free(tid); }
			for (i = 0; i < n; ++i) {                                       /* ref: map.c:594-650 step 2 */
				ofrag_t *f = &fr[i];
				for (j = 0; j < f->n_seg; ++j) {
					size_t need = (size_t)f->rd[j].l_seq * 4 + 4096 + strlen(f->rd[j].name);
					if (need > mbuf) { mbuf = need * 2; buf = (char*)realloc(buf, mbuf); }
					if (f->n_regs[j] > 0) {
						for (k = 0; k < f->n_regs[j]; ++k) {
							oreg_t *r = &f->regs[j][k];
							if (r->id != r->parent) continue;                /* MM_F_NO_PRINT_2ND */
							if (out) { o_write_sam(buf, mi, &f->rd[j], j, k, f->n_seg, f->n_regs, f->regs, rg_id, f->rep_len); fputs(buf, out); }
						}
					} else if (!opt->sam_hit_only && out) { o_write_sam(buf, mi, &f->rd[j], j, -1, f->n_seg, f->n_regs, f->regs, rg_id, f->rep_len); fputs(buf, out); }
				}
				for (j = 0; j < f->n_seg; ++j) {
					for (k = 0; k < f->n_regs[j]; ++k) free(f->regs[j][k].p);
					free(f->regs[j]); free(f->rd[j].name); free(f->rd[j].seq); free(f->rd[j].qual);
					++total;
				}
			}
		}
		free(fr);
	}
	free(buf); free(nm.s); free(sq.s); free(ql.s);
	of_close(f1); of_close(f2);
	return total;
}
