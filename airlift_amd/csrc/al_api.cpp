// al_api.cpp -- C-ABI glue above the device pipeline: batch / single-fragment mapping entry points,
// SAM text formatting and the file-level driver that `airlift-align` uses (host side, C++).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "al_internal.h"
#include "al_runtime.h"
#include "al_io.h"

extern "C" int al_batch_fetch(al_ctx_t *c, int *n_regs, al_reg1_t **regs, int *rep_len)
{
	if (!c || !c->ran) return -1;
	return al_fetch_align(c, n_regs, regs, rep_len);
}

// The device -> host half of a fetch without building per-read arrays: the flat result block (record offsets, records, CIGAR
// arena, repeat lengths) copied into page-locked buffers the context keeps; reports how much moved.  What the file drivers do
// per batch; bench.py times upload + run + this as the PCIe-inclusive rate.
extern "C" int al_batch_fetch_flat(al_ctx_t *c, uint64_t *n_records, uint64_t *n_bytes)
{
	if (!c || !c->ran) return -1;
	static thread_local AlRawResult R;
	const int rc = al_fetch_raw(c, R);
	if (rc) return rc;
	if (n_records) *n_records = R.out.size();
	if (n_bytes) *n_bytes = R.off.size() * 8 + R.out.size() * sizeof(AlReg) + R.arena.size() * 4 + R.rep.size() * 4;
	return 0;
}

extern "C" int al_map_batch(al_ctx_t *c, int n_frag, const int *n_segs, const int *qlens, const char *const *seqs,
                            const char *const *qnames, int *n_regs, al_reg1_t **regs, int *rep_len)
{
	int rc;
	if ((rc = al_batch_upload(c, n_frag, n_segs, qlens, seqs, qnames)) != 0) return rc;
	if ((rc = al_batch_run(c)) != 0) return rc;
	return al_batch_fetch(c, n_regs, regs, rep_len);
}

extern "C" void al_map_frag(const al_idx_t *mi, int n_segs, const int *qlens, const char **seqs, int *n_regs,
                            al_reg1_t **regs, al_ctx_t *ctx, const al_mapopt_t *opt, const char *qname)
{   // mm_map_frag (map.c:272): a batch of one fragment
	(void)mi; (void)opt;
	int i, rep = 0, qsum = 0;
	for (i = 0; i < n_segs; ++i) n_regs[i] = 0, regs[i] = 0, qsum += qlens[i];
	if (qsum == 0 || n_segs <= 0 || n_segs > 2) return;
	const char *names[2] = { qname, qname };
	if (al_map_batch(ctx, 1, &n_segs, qlens, seqs, names, n_regs, regs, &rep) != 0)
		fprintf(stderr, "[airlift] al_map_frag: device pipeline failed\n");
}

// ---------------------------------------------------------------------------------------------
// SAM (format.c:82-135, 276-302, 361-544)
// The tail of the @PG line: mm_write_sam_hdr's `ver` / `argc, argv` arguments (format.c:116-135, called from main.c:369 with
// MM_VERSION and the process's own argv).  Set once by the program that owns main(); empty for library callers that pass none.
static std::string g_pg_tail;
extern "C" void al_set_program_line(const char *ver, int argc, char *const *argv)
{
	g_pg_tail.clear();
	if (ver) { g_pg_tail += "\tVN:"; g_pg_tail += ver; }
	if (argc > 1 && argv) {
		g_pg_tail += "\tCL:minimap2";
		for (int i = 1; i < argc; ++i) { g_pg_tail += ' '; g_pg_tail += argv[i]; }
	}
}

extern "C" int al_write_sam_hdr(FILE *out, const al_idx_t *mi, const char *rg, char *rg_id)
{
	for (size_t i = 0; i < mi->seq.size(); ++i) fprintf(out, "@SQ\tSN:%s\tLN:%d\n", mi->seq[i].name.c_str(), (int)mi->seq[i].len);
	if (rg_id) rg_id[0] = 0;
	if (rg) {
		if (strstr(rg, "@RG") != rg) fprintf(stderr, "[ERROR] the read group line is not started with @RG\n");
		else if (strchr(rg, '\t')) fprintf(stderr, "[ERROR] the read group line contained literal <tab> characters -- replace with escaped tabs: \\t\n");
		else {
			std::string line;
			for (const char *p = rg; *p; ++p) {
				if (*p == '\\') { ++p; if (*p == 't') line.push_back('\t'); else if (*p == '\\') line.push_back('\\'); if (!*p) break; }
				else line.push_back(*p);
			}
			size_t at = line.find("\tID:");
			if (at == std::string::npos) fprintf(stderr, "[ERROR] no ID within the read group line\n");
			else {
				size_t b = at + 4, e = b;
				while (e < line.size() && line[e] != '\t' && line[e] != '\n') ++e;
				if (e - b + 1 > 256) fprintf(stderr, "[ERROR] @RG:ID is longer than 255 characters\n");
				else { if (rg_id) { memcpy(rg_id, line.data() + b, e - b); rg_id[e - b] = 0; } fprintf(out, "%s\n", line.c_str()); }
			}
		}
	}
	fprintf(out, "@PG\tID:minimap2\tPN:minimap2%s\n", g_pg_tail.c_str());
	return 0;
}

static inline int qname_len(const char *s)
{   // bseq.h:31-36
	int l = (int)strlen(s);
	return l >= 3 && s[l-1] >= '0' && s[l-1] <= '9' && s[l-2] == '/' ? l - 2 : l;
}

struct Out {
	char *p, *end; bool ovf;
	void ch(char c) { if (p < end) *p++ = c; else ovf = true; }
	void str(const char *s) { while (*s) ch(*s++); }
	void mem(const char *s, int l) { if (p + l <= end) { memcpy(p, s, l); p += l; } else ovf = true; }
	void num(long long v) { char b[24]; int l = 0; unsigned long long x = v < 0 ? -(unsigned long long)v : (unsigned long long)v; do { b[l++] = '0' + x % 10; x /= 10; } while (x); if (v < 0) b[l++] = '-'; while (l) ch(b[--l]); }
};

static const al_reg1_t *sam_pri(int n, const al_reg1_t *r) { for (int i = 0; i < n; ++i) if (r[i].sam_pri) return &r[i]; return nullptr; }

static void put_sq(Out &o, const char *seq, int l, int rev, int comp)
{
	const unsigned char *ct = al_comp();
	if (rev) for (int i = 0; i < l; ++i) { int c = (unsigned char)seq[l - 1 - i]; o.ch((char)(c < 128 && comp ? ct[c] : c)); }
	else o.mem(seq, l);
}

extern "C" int al_write_sam(char *buf, size_t cap, const al_idx_t *mi, const char *qname, int l_seq, const char *seq, const char *qual,
                            int seg_idx, int reg_idx, int n_seg, const int *n_regss, const al_reg1_t *const *regss, const char *rg_id, int rep_len)
{
	Out o{buf, buf + cap - 1, false};
	const int n_regs = n_regss[seg_idx];
	const al_reg1_t *regs = regss[seg_idx], *r_prev = nullptr, *r_next = nullptr;
	const al_reg1_t *r = n_regs > 0 && reg_idx < n_regs && reg_idx >= 0 ? &regs[reg_idx] : nullptr;
	int this_rid = -1, this_pos = -1, flag;
	if (n_seg > 1) { int ns = (seg_idx + 1) % n_seg; r_next = sam_pri(n_regss[ns], regss[ns]); r_prev = r_next; }
	o.mem(qname, n_seg > 1 ? qname_len(qname) : (int)strlen(qname));
	flag = n_seg > 1 ? 0x1 : 0x0;
	if (!r) flag |= 0x4;
	else { if (r->rev) flag |= 0x10; if (r->parent != r->id) flag |= 0x100; else if (!r->sam_pri) flag |= 0x800; }
	if (n_seg > 1) {
		if (r && r->proper_frag) flag |= 0x2;
		if (seg_idx == 0) flag |= 0x40; else if (seg_idx == n_seg - 1) flag |= 0x80;
		if (!r_next) flag |= 0x8; else if (r_next->rev) flag |= 0x20;
	}
	o.ch('\t'); o.num(flag);
	if (!r) {
		if (r_prev) { this_rid = r_prev->rid; this_pos = r_prev->rs; o.ch('\t'); o.str(mi->seq[this_rid].name.c_str()); o.ch('\t'); o.num(this_pos + 1); o.str("\t0\t*"); }
		else o.str("\t*\t0\t0\t*");
	} else {
		this_rid = r->rid; this_pos = r->rs;
		o.ch('\t'); o.str(mi->seq[r->rid].name.c_str()); o.ch('\t'); o.num(r->rs + 1); o.ch('\t'); o.num(r->mapq); o.ch('\t');
		if (r->n_cigar == 0) o.ch('*');
		else {
			const char clip_char = (flag & 0x800) ? 'H' : 'S';
			const int c0 = r->rev ? l_seq - r->qe : r->qs, c1 = r->rev ? r->qs : l_seq - r->qe;
			if (c0) { o.num(c0); o.ch(clip_char); }
			for (uint32_t k = 0; k < r->n_cigar; ++k) { o.num(r->cigar[k] >> 4); o.ch("MIDNSHP=XB"[r->cigar[k] & 0xf]); }
			if (c1) { o.num(c1); o.ch(clip_char); }
		}
	}
	if (n_seg > 1) {
		int tlen = 0;
		if (this_rid >= 0 && r_next) {
			if (this_rid == r_next->rid) {
				if (r) { int a5 = r->rev ? r->re - 1 : this_pos, b5 = r_next->rev ? r_next->re - 1 : r_next->rs; tlen = b5 - a5; }
				o.str("\t=\t");
			} else { o.ch('\t'); o.str(mi->seq[r_next->rid].name.c_str()); o.ch('\t'); }
			o.num(r_next->rs + 1); o.ch('\t');
		} else if (r_next) { o.ch('\t'); o.str(mi->seq[r_next->rid].name.c_str()); o.ch('\t'); o.num(r_next->rs + 1); o.ch('\t'); }
		else if (this_rid >= 0) { o.str("\t=\t"); o.num(this_pos + 1); o.ch('\t'); }
		else o.str("\t*\t0\t");
		if (tlen > 0) ++tlen; else if (tlen < 0) --tlen;
		o.num(tlen); o.ch('\t');
	} else o.str("\t*\t0\t0\t");
	if (!r) { put_sq(o, seq, l_seq, 0, 0); o.ch('\t'); if (qual) put_sq(o, qual, l_seq, 0, 0); else o.ch('*'); }
	else if ((flag & 0x900) == 0) { put_sq(o, seq, l_seq, r->rev, r->rev); o.ch('\t'); if (qual) put_sq(o, qual, l_seq, r->rev, 0); else o.ch('*'); }
	else if (flag & 0x100) o.str("*\t*");
	else { put_sq(o, seq + r->qs, r->qe - r->qs, r->rev, r->rev); o.ch('\t'); if (qual) put_sq(o, qual + r->qs, r->qe - r->qs, r->rev, 0); else o.ch('*'); }
	if (rg_id && rg_id[0]) { o.str("\tRG:Z:"); o.str(rg_id); }
	if (r) {
		const char type = r->id == r->parent ? (r->inv ? 'I' : 'P') : (r->inv ? 'i' : 'S');
		if (r->n_cigar) { o.str("\tNM:i:"); o.num(r->blen - r->mlen + (int)r->n_ambi); o.str("\tms:i:"); o.num(r->dp_max); o.str("\tAS:i:"); o.num(r->dp_score); o.str("\tnn:i:"); o.num(r->n_ambi); }
		o.str("\ttp:A:"); o.ch(type); o.str("\tcm:i:"); o.num(r->cnt); o.str("\ts1:i:"); o.num(r->score);
		if (r->parent == r->id) { o.str("\ts2:i:"); o.num(r->subsc); }
		if (r->n_cigar) {
			int n_gapo = 0, n_gap = 0;
			for (uint32_t i = 0; i < r->n_cigar; ++i) { int op = r->cigar[i] & 0xf, len = r->cigar[i] >> 4; if (op == 1 || op == 2) ++n_gapo, n_gap += len; }
			const double div = 1.0 - (double)r->mlen / (r->blen - n_gap + n_gapo);
			if (div == 0.0) o.str("\tde:f:0"); else { char b[32]; snprintf(b, 32, "%.4f", div); o.str("\tde:f:"); o.str(b); }
		}
		if (r->split) { o.str("\tzd:i:"); o.num(r->split); }
		if (r->parent == r->id && r->n_cigar && n_regs > 1) {
			int n_sa = 0;
			for (int i = 0; i < n_regs; ++i) if (i != r - regs && regs[i].parent == regs[i].id && regs[i].n_cigar) ++n_sa;
			if (n_sa > 0) {
				o.str("\tSA:Z:");
				for (int i = 0; i < n_regs; ++i) {
					const al_reg1_t *q = &regs[i]; int l_M, l_I = 0, l_D = 0;
					if (r == q || q->parent != q->id || q->n_cigar == 0) continue;
					if (q->qe - q->qs < q->re - q->rs) l_M = q->qe - q->qs, l_D = (q->re - q->rs) - l_M;
					else l_M = q->re - q->rs, l_I = (q->qe - q->qs) - l_M;
					const int clip5 = q->rev ? l_seq - q->qe : q->qs, clip3 = q->rev ? q->qs : l_seq - q->qe;
					o.str(mi->seq[q->rid].name.c_str()); o.ch(','); o.num(q->rs + 1); o.ch(','); o.ch("+-"[q->rev]); o.ch(',');
					if (clip5) { o.num(clip5); o.ch('S'); }
					if (l_M) { o.num(l_M); o.ch('M'); }
					if (l_I) { o.num(l_I); o.ch('I'); }
					if (l_D) { o.num(l_D); o.ch('D'); }
					if (clip3) { o.num(clip3); o.ch('S'); }
					o.ch(','); o.num(q->mapq); o.ch(','); o.num(q->blen - q->mlen + (int)q->n_ambi); o.ch(';');
				}
			}
		}
	}
	if (rep_len >= 0) { o.str("\trl:i:"); o.num(rep_len); }
	o.ch('\n'); *o.p = 0;
	return o.ovf ? -1 : (int)(o.p - buf);
}

// The file-level driver (al_map_file_frag) lives in al_pipeline.cpp.

// ---------------------------------------------------------------------------------------------
// The device SAM formatter (al_dev_sam.h; kernels in al_stream.hip) compiled for the CPU and checked against al_write_sam above on
// random records: every flag / clipping / mate / tag branch, both read orientations, short and long CIGARs, `de:f:%.4f` against
// printf.  Returns the number of differences (0 = identical); needs no GPU.  tests/test_capi_cpu.py.
#define AL_SAM_HOST
#include "al_dev_sam.h"
namespace {
struct HostSamSink {
	const AlSamCfg *C; const char *text; std::string out;
	char peek(uint32_t off) const { return text[off]; }
	void begin_record() {}
	void ch(char c) { out.push_back(c); }
	void lit(const char *s) { out += s; }
	void num(long long v) { out += std::to_string(v); }
	void txt(uint32_t off, uint32_t len) { out.append(text + off, len); }
	void mem(const char *s, int len) { out.append(s, (size_t)len); }
	void cname(int rid) { out.append(C->names + C->name_off[rid], C->name_off[rid + 1] - C->name_off[rid]); }
	void seqfld(uint32_t off, int len, int rev, int comp, int is_seq)
	{
		const unsigned char *ct = al_comp();
		for (int j = 0; j < len; ++j) { unsigned char c = (unsigned char)text[off + (rev ? len - 1 - j : j)]; if (is_seq && (c == 'u' || c == 'U')) --c; if (comp && c < 128) c = ct[c]; out.push_back((char)c); }
	}
};
struct Rng { uint64_t s; uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; } uint32_t below(uint32_t n) { return (uint32_t)(next() % n); } };
}

extern "C" int al_dbg_sam_selftest(uint64_t seed, int n_frag)
{
	int bad = 0;
	Rng R{seed * 2654435761ULL + 88172645463325252ULL};
	// (1) "%.4f": every ratio 1 - m / d the tag can take for short reads, and random doubles
	for (int d = 1; d <= 700 && bad < 10; ++d) for (int m = 0; m <= d; ++m) {
		const double v = 1.0 - (double)m / d; char a[64], b[64]; bool neg; uint64_t q;
		snprintf(a, sizeof(a), "%.4f", v); al_fmt_f4(v, &neg, &q); snprintf(b, sizeof(b), "%s%llu.%04d", neg ? "-" : "", (unsigned long long)(q / 10000), (int)(q % 10000));
		if (strcmp(a, b)) { if (bad < 5) fprintf(stderr, "[airlift] sam selftest: %%.4f of 1 - %d/%d: printf '%s', formatter '%s'\n", m, d, a, b); ++bad; }
	}
	for (int i = 0; i < 200000 && bad < 10; ++i) {
		double v; const uint64_t u = R.next(); const int kind = i % 4;
		if (kind == 0) v = (double)(u >> 11) / 9007199254740992.0;                       // uniform in [0, 1)
		else if (kind == 1) v = ((double)(u % 200001) - 100000.0) / 20000.0;             // multiples of 0.00005: ties of the decimal text, not of the binary value
		else if (kind == 2) v = (double)(u % 4096) / 4096.0 + ((u >> 20) % 3 == 0 ? 0.0 : 1.0 / 8192.0);   // exact binary ties at the fifth digit are impossible below 2^-4; these have few bits
		else v = ldexp((double)(u >> 11), -(int)(53 + (u & 63)));                        // small values
		char a[64], b[64]; bool neg; uint64_t q;
		snprintf(a, sizeof(a), "%.4f", v); al_fmt_f4(v, &neg, &q); snprintf(b, sizeof(b), "%s%llu.%04d", neg ? "-" : "", (unsigned long long)(q / 10000), (int)(q % 10000));
		if (strcmp(a, b)) { if (bad < 5) fprintf(stderr, "[airlift] sam selftest: %%.4f of %.17g: printf '%s', formatter '%s'\n", v, a, b); ++bad; }
	}
	// (2) records
	al_idx_t mi; const char *cn[3] = {"chr1", "contig_two", "c"};
	for (int i = 0; i < 3; ++i) { AlSeq s; s.name = cn[i]; s.offset = 0; s.len = 1000000; mi.seq.push_back(s); }
	std::string names; std::vector<uint32_t> noff; for (int i = 0; i < 3; ++i) { noff.push_back((uint32_t)names.size()); names += cn[i]; } noff.push_back((uint32_t)names.size());
	const char *bases = "ACGTNUacgtun";
	for (int f = 0; f < n_frag && bad < 20; ++f) {
		const int n_seg = 1 + (int)R.below(2), rep_len = (int)R.below(3) == 0 ? -1 : (int)R.below(200);
		const bool with_qual = R.below(5) != 0; const bool with_rg = R.below(2) != 0;
		std::string text; AlSamRead rd[2]; std::vector<AlReg> regs[2]; std::vector<uint32_t> arena; std::string nm[2];
		for (int j = 0; j < n_seg; ++j) {
			const int L = 30 + (int)R.below(220);
			nm[j] = "read" + std::to_string(f) + (R.below(3) == 0 ? "/" + std::to_string(j + 1) : std::string());
			rd[j].name = (uint32_t)text.size(); rd[j].name_len = (uint32_t)nm[j].size(); text += nm[j]; text.push_back('\n');
			rd[j].seq = (uint32_t)text.size(); for (int i = 0; i < L; ++i) text.push_back(bases[R.below(12)]); text.push_back('\n');
			rd[j].qual = ~0u; if (with_qual) { rd[j].qual = (uint32_t)text.size(); for (int i = 0; i < L; ++i) text.push_back((char)(33 + R.below(60))); text.push_back('\n'); }
			rd[j].qlen = L; rd[j].flip = n_seg == 2 && j == 1 ? 1 : (int)R.below(4) == 0;
			const int n = (int)R.below(5) == 0 ? 0 : 1 + (int)R.below(4);
			int pri = n ? (int)R.below((uint32_t)n + 1) : 0;          // == n: no sam_pri record at all
			for (int k = 0; k < n; ++k) {
				AlReg r; memset(&r, 0, sizeof(r));
				r.id = k; r.parent = R.below(3) == 0 && k > 0 ? (int)R.below((uint32_t)k) : k; r.rid = (int)R.below(3); r.cnt = 1 + (int)R.below(30); r.score = (int)R.below(300);
				r.qs = (int)R.below((uint32_t)L / 2); r.qe = r.qs + 1 + (int)R.below((uint32_t)(L - r.qs)); r.rs = (int)R.below(900000); r.re = r.rs + (r.qe - r.qs) + (int)R.below(7) - 3; if (r.re <= r.rs) r.re = r.rs + 1;
				r.subsc = (int)R.below(200); r.mlen = (int)R.below(200); r.blen = r.mlen + (int)R.below(30); r.n_sub = (int)R.below(4); r.mapq = R.below(61); r.hash = (uint32_t)R.next();
				r.dp_score = (int)R.below(300); r.dp_max = (int)R.below(300); r.dp_max2 = (int)R.below(300); r.n_ambi = R.below(3);
				r.flags = R.below(4) == 0 ? R.below(4) : 0;               // split
				if (R.below(2)) r.flags |= ALR_REV; if (k == pri && r.parent == r.id) r.flags |= ALR_SAM_PRI; if (R.below(2)) r.flags |= ALR_PROPER;
				if (R.below(8) != 0) {
					r.flags |= ALR_HAS_P; r.n_cigar = 1 + R.below(R.below(4) == 0 ? 9u : 3u);
					uint32_t *cg;
					if (r.n_cigar <= 4) { r.cigar_off = AL_CIG_INLINE; cg = r.cig_inl; } else { r.cigar_off = (uint32_t)arena.size(); arena.resize(arena.size() + r.n_cigar); cg = arena.data() + r.cigar_off; }
					int bl = 0, ml = 0;
					for (uint32_t i = 0; i < r.n_cigar; ++i) { cg[i] = (1 + R.below(120)) << 4 | (i % 2 == 0 ? 0u : 1u + R.below(2)); bl += (int)(cg[i] >> 4); if ((cg[i] & 0xf) == 0) ml += (int)(cg[i] >> 4); }
					r.blen = bl; r.mlen = (int)R.below((uint32_t)ml + 1);               // what mm_update_extra leaves: blen = all columns, mlen <= matched columns
				}
				regs[j].push_back(r);
			}
		}
		for (int j = 0; j < n_seg; ++j) { rd[j].regs = regs[j].data(); rd[j].n_regs = (int)regs[j].size(); rd[j].arena = arena.data(); }
		// cigar pointers into `arena` were taken before it stopped growing: rebuild the host-side view afterwards (below)
		al_mapopt_t mo; memset(&mo, 0, sizeof(mo)); if (R.below(2)) mo.flag |= AL_F_NO_PRINT_2ND; if (R.below(3) == 0) mo.flag |= AL_F_SAM_HIT_ONLY;
		AlSamCfg C; C.names = names.data(); C.name_off = noff.data(); C.rg_id = "grp1"; C.rg_len = with_rg ? 4 : 0; C.no_print_2nd = (mo.flag & AL_F_NO_PRINT_2ND) ? 1 : 0; C.hit_only = (mo.flag & AL_F_SAM_HIT_ONLY) ? 1 : 0; C.pe_ori = 1;
		// the host records al_write_sam takes: AlReg -> al_reg1_t with the un-flip of al_reg_from_raw
		std::vector<al_reg1_t> hr[2]; int n_regss[2] = {0, 0}; const al_reg1_t *regss[2] = {nullptr, nullptr};
		for (int j = 0; j < n_seg; ++j) {
			for (const AlReg &r : regs[j]) {
				al_reg1_t q; memset(&q, 0, sizeof(q));
				q.id = r.id; q.cnt = r.cnt; q.rid = r.rid; q.score = r.score; q.qs = r.qs; q.qe = r.qe; q.rs = r.rs; q.re = r.re; q.parent = r.parent; q.subsc = r.subsc; q.mlen = r.mlen; q.blen = r.blen; q.n_sub = r.n_sub;
				q.mapq = r.mapq & 0xff; q.split = r.flags & 3; q.rev = (r.flags & ALR_REV) ? 1 : 0; q.sam_pri = (r.flags & ALR_SAM_PRI) ? 1 : 0; q.proper_frag = (r.flags & ALR_PROPER) ? 1 : 0; q.hash = r.hash;
				q.dp_score = r.dp_score; q.dp_max = r.dp_max; q.dp_max2 = r.dp_max2; q.n_ambi = r.n_ambi; q.n_cigar = (r.flags & ALR_HAS_P) ? r.n_cigar : 0;
				q.cigar = q.n_cigar ? const_cast<uint32_t *>(r.cigar_off == AL_CIG_INLINE ? r.cig_inl : arena.data() + r.cigar_off) : nullptr;
				if (rd[j].flip) { const int t = q.qs; q.qs = rd[j].qlen - q.qe; q.qe = rd[j].qlen - t; q.rev = !q.rev; }
				hr[j].push_back(q);
			}
			n_regss[j] = (int)hr[j].size(); regss[j] = hr[j].data();
		}
		for (int j = 0; j < n_seg; ++j) {
			std::string exp; std::vector<char> buf(1 << 16);
			std::string seq(text.data() + rd[j].seq, (size_t)rd[j].qlen); for (char &c : seq) if (c == 'u' || c == 'U') --c;       // what the host parsers hand to al_write_sam (bseq.c:72-74)
			std::string qual; if (with_qual) qual.assign(text.data() + rd[j].qual, (size_t)rd[j].qlen);
			auto emit = [&](int k) { const int l = al_write_sam(buf.data(), buf.size(), &mi, nm[j].c_str(), rd[j].qlen, seq.c_str(), with_qual ? qual.c_str() : nullptr, j, k, n_seg, n_regss, regss, with_rg ? "grp1" : "", rep_len); if (l > 0) exp.append(buf.data(), (size_t)l); };
			if (n_regss[j] > 0) { for (int k = 0; k < n_regss[j]; ++k) { if ((mo.flag & AL_F_NO_PRINT_2ND) && hr[j][k].id != hr[j][k].parent) continue; emit(k); } }
			else if (!(mo.flag & AL_F_SAM_HIT_ONLY)) emit(-1);
			HostSamSink o; o.C = &C; o.text = text.data();
			AlSamCountSink cnt; cnt.C = &C; cnt.text = text.data();
			al_sam_read_records(o, C, rd[j], n_seg == 2 ? &rd[1 - j] : nullptr, j, n_seg, rep_len);
			al_sam_read_records(cnt, C, rd[j], n_seg == 2 ? &rd[1 - j] : nullptr, j, n_seg, rep_len);
			if (o.out != exp || cnt.n != exp.size()) {
				if (bad < 2) fprintf(stderr, "[airlift] sam selftest: fragment %d read %d differs (counted %llu bytes)\n  al_write_sam: %s  formatter:    %s", f, j, (unsigned long long)cnt.n, exp.c_str(), o.out.c_str());
				++bad;
			}
		}
	}
	return bad;
}
