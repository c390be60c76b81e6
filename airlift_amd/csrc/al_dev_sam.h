// al_dev_sam.h -- SAM record text (mm_write_sam3, format.c:387-544; write_sam_cigar :361-385; write_tags :276-302) as
// one routine over a byte sink, compiled for the device (k_sam_len / k_sam_write, al_stream.hip) and, with AL_SAM_HOST, for the
// CPU (al_dbg_sam_selftest pins it against al_write_sam, which the golden SAM files pin against the reference).
//
// It works on the device's own records (AlReg as k_compact leaves them, still in mapping orientation: mate 2 of an FR pair was
// mapped reverse-complemented, map.c:468) and un-flips on the fly (map.c:486-497), so no host pass touches a record.
#pragma once
#include <stdint.h>
#include "al_internal.h"
#ifdef AL_SAM_HOST
#define AL_SD static inline
#define AL_SM inline
#else
#include <hip/hip_runtime.h>
#define AL_SD __device__ __forceinline__
#define AL_SM __device__ __forceinline__
#endif

// ---- "%.4f" of a double, exactly as glibc prints it (format.c:292: `de:f:%.4f`): the value's exact binary expansion times 10^4,
// rounded to nearest, ties to even.  v = M * 2^-s with a 53-bit M; M * 10^4 fits 67 bits, kept as (hi, lo).
AL_SD void al_fmt_f4(double v, bool *neg, uint64_t *q_out)
{
	uint64_t u; __builtin_memcpy(&u, &v, 8);
	*neg = (u >> 63) != 0;
	const int ex = (int)((u >> 52) & 0x7ff);
	uint64_t M = u & 0xfffffffffffffULL;
	int e;                                              // |v| = M * 2^e
	if (ex == 0) e = -1074; else { M |= 1ULL << 52; e = ex - 1075; }
	const uint64_t lo = M * 10000ULL, hi = ((M >> 32) * 10000ULL + (((M & 0xffffffffULL) * 10000ULL) >> 32)) >> 32;
	uint64_t q;
	if (e >= 0) q = e < 10 ? lo << e : ~0ULL;            // (>= 2^52: not a ratio this path prints; saturate)
	else {
		const int s = -e;
		if (s >= 128) q = 0;                            // M * 10^4 < 2^67 <= 2^(s-1): below one half
		else {
			uint64_t rem_hi, rem_lo, half_hi, half_lo;
			if (s >= 64) { q = s == 64 ? hi : hi >> (s - 64); rem_hi = s == 64 ? 0 : hi & ((1ULL << (s - 64)) - 1); rem_lo = lo; }
			else { q = (lo >> s) | (hi << (64 - s)); rem_hi = 0; rem_lo = lo & ((1ULL << s) - 1); }     // hi < 8: q fits for s >= 4; smaller s means v >= 2^48
			if (s - 1 >= 64) { half_hi = 1ULL << (s - 1 - 64); half_lo = 0; } else { half_hi = 0; half_lo = 1ULL << (s - 1); }
			const bool gt = rem_hi > half_hi || (rem_hi == half_hi && rem_lo > half_lo), eq = rem_hi == half_hi && rem_lo == half_lo;
			if (gt || (eq && (q & 1))) ++q;
		}
	}
	*q_out = q;
}

struct AlSamCfg {                     // what the formatter needs besides the records
	const char *names;                // contig names, concatenated (not NUL-terminated)
	const uint32_t *name_off;         // n_seq + 1 offsets into names
	const char *rg_id; int rg_len;    // RG:Z: value ("" = none)
	int no_print_2nd, hit_only;       // AL_F_NO_PRINT_2ND, AL_F_SAM_HIT_ONLY
	int pe_ori;
};

struct AlSamRead {                    // one read as the formatter sees it
	const AlReg *regs; int n_regs;    // its records (mapping orientation)
	const uint32_t *arena;            // CIGAR words of records with more than four operations
	int qlen, flip;                   // length; mapped reverse-complemented (to be un-flipped)
	uint32_t name, name_len, seq, qual;   // offsets into the read's text (qual == ~0u: none)
};

// un-flipped view of the scalar fields the text depends on
struct AlSamView { int qs, qe, rev; };
AL_SD AlSamView al_sam_view(const AlReg &r, int qlen, int flip)
{
	AlSamView v; v.qs = r.qs; v.qe = r.qe; v.rev = (r.flags & ALR_REV) ? 1 : 0;
	if (flip) { v.qs = qlen - r.qe; v.qe = qlen - r.qs; v.rev = !v.rev; }
	return v;
}
AL_SD uint32_t al_sam_ncig(const AlReg &r) { return (r.flags & ALR_HAS_P) ? r.n_cigar : 0u; }
AL_SD const uint32_t *al_sam_cig(const AlReg &r, const uint32_t *arena) { return r.cigar_off == AL_CIG_INLINE ? r.cig_inl : arena + r.cigar_off; }
AL_SD int al_sam_pri_idx(const AlReg *r, int n) { for (int i = 0; i < n; ++i) if (r[i].flags & ALR_SAM_PRI) return i; return -1; }

// One record.  reg_idx < 0: the unmapped record of a read without hits.  `me` is the read, `mate` the other read of the pair
// (nullptr for single-end), seg_idx its position in the fragment.  Sink: ch(c), lit("..."), num(v), txt(off, len) (bytes of the
// read's text), cname(rid), seqfld(off, len, rev, comp, is_seq) (SEQ / QUAL: the bulk of a record).
template <class S>
AL_SD void al_sam_record(S &o, const AlSamCfg &C, const AlSamRead &me, const AlSamRead *mate, int seg_idx, int n_seg, int reg_idx, int rep_len)
{
	const AlReg *regs = me.regs; const int n_regs = me.n_regs;
	const AlReg *r = n_regs > 0 && reg_idx >= 0 && reg_idx < n_regs ? &regs[reg_idx] : nullptr;
	const AlReg *r_next = nullptr; AlSamView vn{0, 0, 0};
	if (n_seg > 1 && mate) { const int p = al_sam_pri_idx(mate->regs, mate->n_regs); if (p >= 0) { r_next = &mate->regs[p]; vn = al_sam_view(*r_next, mate->qlen, mate->flip); } }
	const AlReg *r_prev = r_next;
	AlSamView v{0, 0, 0}; if (r) v = al_sam_view(*r, me.qlen, me.flip);
	const int l_seq = me.qlen;
	int this_rid = -1, this_pos = -1, flag;
	{   // qname without a trailing /1 /2 in paired mode (format.c:413, bseq.h:31-36)
		uint32_t l = me.name_len;
		if (n_seg > 1 && l >= 3) { const char c1 = o.peek(me.name + l - 1), c2 = o.peek(me.name + l - 2); if (c1 >= '0' && c1 <= '9' && c2 == '/') l -= 2; }
		o.txt(me.name, l);
	}
	flag = n_seg > 1 ? 0x1 : 0x0;
	if (!r) flag |= 0x4;
	else { if (v.rev) flag |= 0x10; if (r->parent != r->id) flag |= 0x100; else if (!(r->flags & ALR_SAM_PRI)) flag |= 0x800; }
	if (n_seg > 1) {
		if (r && (r->flags & ALR_PROPER)) flag |= 0x2;
		if (seg_idx == 0) flag |= 0x40; else if (seg_idx == n_seg - 1) flag |= 0x80;
		if (!r_next) flag |= 0x8; else if (vn.rev) flag |= 0x20;
	}
	o.ch('\t'); o.num(flag);
	const uint32_t n_cig = r ? al_sam_ncig(*r) : 0u;
	const uint32_t *cig = r && n_cig ? al_sam_cig(*r, me.arena) : nullptr;
	if (!r) {
		if (r_prev) { this_rid = r_prev->rid; this_pos = r_prev->rs; o.ch('\t'); o.cname(this_rid); o.ch('\t'); o.num(this_pos + 1); o.lit("\t0\t*"); }
		else o.lit("\t*\t0\t0\t*");
	} else {
		this_rid = r->rid; this_pos = r->rs;
		o.ch('\t'); o.cname(r->rid); o.ch('\t'); o.num(r->rs + 1); o.ch('\t'); o.num((int)(r->mapq & 0xff)); o.ch('\t');
		if (n_cig == 0) o.ch('*');
		else {
			const char clip_char = (flag & 0x800) ? 'H' : 'S';
			const int c0 = v.rev ? l_seq - v.qe : v.qs, c1 = v.rev ? v.qs : l_seq - v.qe;
			if (c0) { o.num(c0); o.ch(clip_char); }
			for (uint32_t k = 0; k < n_cig; ++k) { o.num((int)(cig[k] >> 4)); o.ch("MIDNSHP=XB"[cig[k] & 0xf]); }
			if (c1) { o.num(c1); o.ch(clip_char); }
		}
	}
	if (n_seg > 1) {
		int tlen = 0;
		if (this_rid >= 0 && r_next) {
			if (this_rid == r_next->rid) {
				if (r) { const int a5 = v.rev ? r->re - 1 : this_pos, b5 = vn.rev ? r_next->re - 1 : r_next->rs; tlen = b5 - a5; }
				o.lit("\t=\t");
			} else { o.ch('\t'); o.cname(r_next->rid); o.ch('\t'); }
			o.num(r_next->rs + 1); o.ch('\t');
		} else if (r_next) { o.ch('\t'); o.cname(r_next->rid); o.ch('\t'); o.num(r_next->rs + 1); o.ch('\t'); }
		else if (this_rid >= 0) { o.lit("\t=\t"); o.num(this_pos + 1); o.ch('\t'); }
		else o.lit("\t*\t0\t");
		if (tlen > 0) ++tlen; else if (tlen < 0) --tlen;
		o.num(tlen); o.ch('\t');
	} else o.lit("\t*\t0\t0\t");
	const bool hq = me.qual != ~0u;
	if (!r) { o.seqfld(me.seq, l_seq, 0, 0, 1); o.ch('\t'); if (hq) o.seqfld(me.qual, l_seq, 0, 0, 0); else o.ch('*'); }
	else if ((flag & 0x900) == 0) { o.seqfld(me.seq, l_seq, v.rev, v.rev, 1); o.ch('\t'); if (hq) o.seqfld(me.qual, l_seq, v.rev, 0, 0); else o.ch('*'); }
	else if (flag & 0x100) o.lit("*\t*");
	else { o.seqfld(me.seq + v.qs, v.qe - v.qs, v.rev, v.rev, 1); o.ch('\t'); if (hq) o.seqfld(me.qual + v.qs, v.qe - v.qs, v.rev, 0, 0); else o.ch('*'); }
	if (C.rg_len > 0) { o.lit("\tRG:Z:"); o.mem(C.rg_id, C.rg_len); }
	if (r) {
		const char type = r->id == r->parent ? 'P' : 'S';                       // inversions do not occur on this path (inv = 0)
		if (n_cig) { o.lit("\tNM:i:"); o.num(r->blen - r->mlen + (int)r->n_ambi); o.lit("\tms:i:"); o.num(r->dp_max); o.lit("\tAS:i:"); o.num(r->dp_score); o.lit("\tnn:i:"); o.num((int)r->n_ambi); }
		o.lit("\ttp:A:"); o.ch(type); o.lit("\tcm:i:"); o.num(r->cnt); o.lit("\ts1:i:"); o.num(r->score);
		if (r->parent == r->id) { o.lit("\ts2:i:"); o.num(r->subsc); }
		if (n_cig) {
			int n_gapo = 0, n_gap = 0;
			for (uint32_t i = 0; i < n_cig; ++i) { const int op = cig[i] & 0xf, len = (int)(cig[i] >> 4); if (op == 1 || op == 2) ++n_gapo, n_gap += len; }
			const double div = 1.0 - (double)r->mlen / (double)(r->blen - n_gap + n_gapo);
			if (div == 0.0) o.lit("\tde:f:0");
			else {
				bool neg; uint64_t q; al_fmt_f4(div, &neg, &q);
				o.lit("\tde:f:"); if (neg) o.ch('-');
				o.num((long long)(q / 10000)); o.ch('.');
				const int fr = (int)(q % 10000); o.ch((char)('0' + fr / 1000)); o.ch((char)('0' + fr / 100 % 10)); o.ch((char)('0' + fr / 10 % 10)); o.ch((char)('0' + fr % 10));
			}
		}
		if (r->flags & 3u) { o.lit("\tzd:i:"); o.num((int)(r->flags & 3u)); }
		if (r->parent == r->id && n_cig && n_regs > 1) {
			int n_sa = 0;
			for (int i = 0; i < n_regs; ++i) if (i != reg_idx && regs[i].parent == regs[i].id && al_sam_ncig(regs[i])) ++n_sa;
			if (n_sa > 0) {
				o.lit("\tSA:Z:");
				for (int i = 0; i < n_regs; ++i) {
					const AlReg *q = &regs[i]; int l_M, l_I = 0, l_D = 0;
					if (i == reg_idx || q->parent != q->id || al_sam_ncig(*q) == 0) continue;
					const AlSamView vq = al_sam_view(*q, me.qlen, me.flip);
					if (vq.qe - vq.qs < q->re - q->rs) l_M = vq.qe - vq.qs, l_D = (q->re - q->rs) - l_M;
					else l_M = q->re - q->rs, l_I = (vq.qe - vq.qs) - l_M;
					const int clip5 = vq.rev ? l_seq - vq.qe : vq.qs, clip3 = vq.rev ? vq.qs : l_seq - vq.qe;
					o.cname(q->rid); o.ch(','); o.num(q->rs + 1); o.ch(','); o.ch("+-"[vq.rev]); o.ch(',');
					if (clip5) { o.num(clip5); o.ch('S'); }
					if (l_M) { o.num(l_M); o.ch('M'); }
					if (l_I) { o.num(l_I); o.ch('I'); }
					if (l_D) { o.num(l_D); o.ch('D'); }
					if (clip3) { o.num(clip3); o.ch('S'); }
					o.ch(','); o.num((int)(q->mapq & 0xff)); o.ch(','); o.num(q->blen - q->mlen + (int)q->n_ambi); o.ch(';');
				}
			}
		}
	}
	if (rep_len >= 0) { o.lit("\trl:i:"); o.num(rep_len); }
	o.ch('\n');
}

// All records of one read, in the order the reference prints them (map.c:601-644): every hit (secondaries unless NO_PRINT_2ND),
// or one unmapped record (unless SAM_HIT_ONLY).  Returns the number of records.
template <class S>
AL_SD int al_sam_read_records(S &o, const AlSamCfg &C, const AlSamRead &me, const AlSamRead *mate, int seg_idx, int n_seg, int rep_len)
{
	int n = 0;
	if (me.n_regs > 0) {
		for (int k = 0; k < me.n_regs; ++k) {
			const AlReg *r = &me.regs[k];
			if (C.no_print_2nd && r->id != r->parent) continue;
			o.begin_record(); al_sam_record(o, C, me, mate, seg_idx, n_seg, k, rep_len); ++n;
		}
	} else if (!C.hit_only) { o.begin_record(); al_sam_record(o, C, me, mate, seg_idx, n_seg, -1, rep_len); ++n; }
	return n;
}

// ---- sinks ---------------------------------------------------------------------------------------------------------------
AL_SD int al_num_len(long long v)
{
	unsigned long long x = v < 0 ? 0ULL - (unsigned long long)v : (unsigned long long)v;
	int l = v < 0 ? 2 : 1;
	while (x >= 10) { x /= 10; ++l; }
	return l;
}
struct AlSamCountSink {               // pass 1: bytes only
	const AlSamCfg *C; const char *text; uint64_t n = 0;
	AL_SM char peek(uint32_t off) const { return text[off]; }
	AL_SM void begin_record() {}
	AL_SM void ch(char) { ++n; }
	AL_SM void lit(const char *s) { while (*s++) ++n; }
	AL_SM void num(long long v) { n += (uint64_t)al_num_len(v); }
	AL_SM void txt(uint32_t, uint32_t len) { n += len; }
	AL_SM void mem(const char *, int len) { n += (uint64_t)len; }
	AL_SM void cname(int rid) { n += C->name_off[rid + 1] - C->name_off[rid]; }
	AL_SM void seqfld(uint32_t, int len, int, int, int) { if (len > 0) n += (uint64_t)len; }
};
