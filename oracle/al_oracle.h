/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the AirLift re-alignment hot path.
 *
 * Plain-C restatement of the algorithm of the reference's bundled minimap2 fork
 * (/root/reference/src/minimap2-master_remapping, `-ax sr`, ALSER edits reverted).
 * Every function cites the reference file:line it follows.  Parity status: PINNED -- checked here
 * against (i) the reference itself built by oracle/Makefile (`oracle/_ref/mm2ref`, SAM + --seeds taps)
 * and (ii) the committed golden vectors under tests/golden/ that were produced by that build.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code; the product
 * (airlift_amd/) never links, imports or executes it.
 */
#ifndef AL_ORACLE_H
#define AL_ORACLE_H
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t x, y; } o128_t;

typedef struct { char *name; uint64_t offset; uint32_t len; } oseq_t;

/* Minimizer index (semantics of mm_idx_t, index.c:27-98; own layout: sorted key/offset arrays + open hash) */
typedef struct {
	int k, w;
	uint32_t n_seq;
	oseq_t *seq;
	uint8_t *S;          /* 1 byte / base, codes 0..4 (reference packs 4 bit/base, mmpriv.h:28-29) */
	uint64_t tot_len;
	uint64_t n_keys;     /* distinct minimizer hashes */
	uint64_t *keys;      /* hash table slots: key+1 (0 = empty) */
	uint64_t *vals;      /* off<<32 | n  into pos[]  */
	uint64_t tab_mask;
	uint64_t *pos;       /* rid<<32 | pos<<1 | strand, ascending within each key (index.c:230) */
	uint64_t n_pos;
} oidx_t;

/* mm_mapopt_t subset after mm_set_opt("sr") + -a (options.c:105-122, main.c:162) */
typedef struct {
	int k, w;
	int seed, bw, max_gap, max_gap_ref, max_frag_len, max_chain_skip, max_chain_iter, min_cnt, min_chain_score;
	float mask_level, pri_ratio, max_clip_ratio;
	int best_n, a, b, q, e, q2, e2, sc_ambi, zdrop, zdrop_inv, end_bonus, min_dp_max, min_ksw_len;
	int pe_ori, pe_bonus, mid_occ, max_occ;
	int sam_hit_only;
	int dbg_seeds;       /* print RS/SD/CN taps to stderr (map.c:333-338,381-385) */
	int dbg_aln;         /* print ksw call taps (align.c:315-338) */
} oopt_t;

typedef struct {
	uint32_t capacity;
	int32_t dp_score, dp_max, dp_max2;
	uint32_t n_ambi, trans_strand;
	uint32_t n_cigar;
	uint32_t cigar[];
} oextra_t;

typedef struct {             /* mm_reg1_t, minimap.h:83-98 */
	int32_t id, cnt, rid, score, qs, qe, rs, re, parent, subsc, as, mlen, blen, n_sub, score0;
	uint32_t mapq, split, rev, inv, sam_pri, proper_frag, pe_thru, seg_split, seg_id, split_inv;
	uint32_t hash;
	float div;
	oextra_t *p;
} oreg_t;

typedef struct { int l_seq; char *name, *seq, *qual; } oread_t;

typedef struct {             /* per-fragment work counters for the algorithmic-bytes formula */
	uint64_t n_reads, n_mini, n_anchor, n_regs_aln, n_refbases, n_cigar, n_ksw;
} ostat_t;

void    oopt_sr(oopt_t *o);
oidx_t *oidx_build_file(const char *fn, int k, int w);
oidx_t *oidx_build(int k, int w, int n, const char **names, const char **seqs);
void    oidx_destroy(oidx_t *mi);
const uint64_t *oidx_get(const oidx_t *mi, uint64_t minier, int *n);

void    o_sketch(const char *str, int len, int w, int k, uint32_t rid, o128_t **a, size_t *n, size_t *m);
uint32_t o_qname_hash(const char *qname, int qlen_sum, int seed);

/* map one fragment (1 or 2 segments, already in mapping orientation -- see o_map_pair for the FR flip) */
void    o_map_frag(const oidx_t *mi, const oopt_t *opt, int n_segs, const int *qlens, const char **seqs,
                   int *n_regs, oreg_t **regs, const char *qname, int *rep_len, ostat_t *st);
/* worker_for (map.c:458-498): flips mate 2, maps, flips back */
void    o_map_reads(const oidx_t *mi, const oopt_t *opt, int n_segs, oread_t *reads, int *n_regs, oreg_t **regs, int *rep_len, ostat_t *st);
/* a8: ALSER candidate counter (map.c:299-312) for a single read */
int     o_alser_count(const oidx_t *mi, const oopt_t *opt, int qlen, const char *seq);

void    o_write_sam_hdr(FILE *fp, const oidx_t *mi, const char *rg, char *rg_id);
/* writes one SAM record (format.c:387-544) into buf (must hold >= 4*l_seq+1024); returns length */
int     o_write_sam(char *buf, const oidx_t *mi, const oread_t *t, int seg_idx, int reg_idx, int n_seg,
                    const int *n_regss, oreg_t *const *regss, const char *rg_id, int rep_len);

/* whole-file driver: maps fn1[/fn2] and prints SAM to out; returns #reads processed */
long    o_map_files(const oidx_t *mi, const oopt_t *opt, const char *fn1, const char *fn2, FILE *out,
                    const char *rg, int n_threads, ostat_t *st);

/* exposed for stage-level tests */
typedef struct {
	uint32_t max, zdropped; int max_q, max_t, mqe, mqe_t, mte, mte_q, score, m_cigar, n_cigar, reach_end; uint32_t *cigar;
} oksw_t;
void    o_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                    int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, oksw_t *ez);
o128_t *o_chain_dp(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc,
                   int n_segs, int64_t n, o128_t *a, int *n_u_, uint64_t **_u);
o128_t *o_collect_seeds(const oidx_t *mi, int max_occ, const o128_t *mv, size_t n_mv, int qlen, int64_t *n_a, int *rep_len);
void    o_radix_sort_128x(o128_t *beg, o128_t *end);
void    o_radix_sort_64(uint64_t *beg, uint64_t *end);

#ifdef __cplusplus
}
#endif
#endif
