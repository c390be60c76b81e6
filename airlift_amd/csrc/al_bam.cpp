// al_bam.cpp -- BAM / BGZF output of the drop-in (product code, host C++).
//
// AirLift pipes the aligner into `samtools view -h -F4 | samtools sort -l5` (src/0-align_reads.sh:13): this file lets
// `airlift-align` emit the BAM itself -- record encoding (same fields the SAM writer prints, format.c:387-544), BGZF
// blocks deflated on worker threads, and for --sorted-bam the coordinate order (keys rid<<32|pos radix-sorted on the GPU,
// al_sort_keys in al_runtime.hip) with unmapped records dropped like `-F4`.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <string>
#include <thread>
#include <vector>
#include "al_internal.h"
#include "al_runtime.h"
#include "al_io.h"
#include "al_bam.h"

namespace {

inline int qname_len(const char *s)
{   // bseq.h:31-36
	const int l = (int)strlen(s);
	return l >= 3 && s[l - 1] >= '0' && s[l - 1] <= '9' && s[l - 2] == '/' ? l - 2 : l;
}
const al_reg1_t *sam_pri(int n, const al_reg1_t *r) { for (int i = 0; i < n; ++i) if (r[i].sam_pri) return &r[i]; return nullptr; }

struct Bw {            // little-endian byte writer into a std::vector
	std::vector<char> &v;
	void u8(uint8_t x) { v.push_back((char)x); }
	void u16(uint16_t x) { v.push_back((char)x); v.push_back((char)(x >> 8)); }
	void u32(uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((char)(x >> (8 * i))); }
	void i32(int32_t x) { u32((uint32_t)x); }
	void mem(const void *p, size_t n) { v.insert(v.end(), (const char *)p, (const char *)p + n); }
	void tag_i(const char *t, int32_t x) { v.push_back(t[0]); v.push_back(t[1]); v.push_back('i'); i32(x); }
	void tag_A(const char *t, char c) { v.push_back(t[0]); v.push_back(t[1]); v.push_back('A'); v.push_back(c); }
	void tag_Z(const char *t, const char *s) { v.push_back(t[0]); v.push_back(t[1]); v.push_back('Z'); mem(s, strlen(s) + 1); }
	void tag_f(const char *t, float f) { v.push_back(t[0]); v.push_back(t[1]); v.push_back('f'); uint32_t u; memcpy(&u, &f, 4); u32(u); }
};

inline int reg2bin(int64_t beg, int64_t end)
{   // SAM specification 5.3
	--end;
	if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
	if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
	if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
	if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
	if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
	return 0;
}

const unsigned char *seq16()
{
	static unsigned char t[256]; static bool init = false;
	if (!init) { memset(t, 15, 256); const char *c = "=ACMGRSVTWYHKDBN"; for (int i = 0; i < 16; ++i) { t[(unsigned char)c[i]] = (unsigned char)i; t[(unsigned char)(c[i] | 0x20)] = (unsigned char)i; } init = true; }
	return t;
}

} // namespace

// One alignment record (block_size included).  Same decisions as al_write_sam (al_api.cpp) for flag, mate fields, TLEN,
// clipping, SEQ/QUAL orientation and tags.  *key gets (refID << 32 | pos) for the coordinate sort, *unmapped the 0x4 bit.
int al_write_bam_rec(std::vector<char> &out, const al_idx_t *mi, const char *qname, int l_seq, const char *seq, const char *qual,
                     int seg_idx, int reg_idx, int n_seg, const int *n_regss, const al_reg1_t *const *regss, const char *rg_id, int rep_len,
                     uint64_t *key, int *unmapped)
{
	const int n_regs = n_regss[seg_idx];
	const al_reg1_t *regs = regss[seg_idx], *r_prev = nullptr, *r_next = nullptr;
	const al_reg1_t *r = n_regs > 0 && reg_idx < n_regs && reg_idx >= 0 ? &regs[reg_idx] : nullptr;
	int this_rid = -1, this_pos = -1, flag;
	if (n_seg > 1) { const int ns = (seg_idx + 1) % n_seg; r_next = sam_pri(n_regss[ns], regss[ns]); r_prev = r_next; }
	flag = n_seg > 1 ? 0x1 : 0x0;
	if (!r) flag |= 0x4;
	else { if (r->rev) flag |= 0x10; if (r->parent != r->id) flag |= 0x100; else if (!r->sam_pri) flag |= 0x800; }
	if (n_seg > 1) {
		if (r && r->proper_frag) flag |= 0x2;
		if (seg_idx == 0) flag |= 0x40; else if (seg_idx == n_seg - 1) flag |= 0x80;
		if (!r_next) flag |= 0x8; else if (r_next->rev) flag |= 0x20;
	}
	int mapq = 0;
	std::vector<uint32_t> cig;
	if (!r) { if (r_prev) { this_rid = r_prev->rid; this_pos = r_prev->rs; } }
	else {
		this_rid = r->rid; this_pos = r->rs; mapq = r->mapq;
		if (r->n_cigar) {
			const uint32_t clip_op = (flag & 0x800) ? 5 : 4;
			const int c0 = r->rev ? l_seq - r->qe : r->qs, c1 = r->rev ? r->qs : l_seq - r->qe;
			if (c0) cig.push_back((uint32_t)c0 << 4 | clip_op);
			for (uint32_t k = 0; k < r->n_cigar; ++k) cig.push_back(r->cigar[k]);
			if (c1) cig.push_back((uint32_t)c1 << 4 | clip_op);
		}
	}
	int next_rid = -1, next_pos = -1, tlen = 0;
	if (n_seg > 1) {
		if (this_rid >= 0 && r_next) {
			if (this_rid == r_next->rid && r) { const int a5 = r->rev ? r->re - 1 : this_pos, b5 = r_next->rev ? r_next->re - 1 : r_next->rs; tlen = b5 - a5; }
			next_rid = r_next->rid; next_pos = r_next->rs;
		} else if (r_next) { next_rid = r_next->rid; next_pos = r_next->rs; }
		else if (this_rid >= 0) { next_rid = this_rid; next_pos = this_pos; }
		if (tlen > 0) ++tlen; else if (tlen < 0) --tlen;
	}
	// SEQ / QUAL as printed (format.c:480-503)
	const char *sq = seq, *ql = qual; int sl = l_seq; bool rev = false, none = false;
	if (r) {
		if ((flag & 0x900) == 0) rev = r->rev;
		else if (flag & 0x100) none = true;
		else { sq = seq + r->qs; ql = qual ? qual + r->qs : nullptr; sl = r->qe - r->qs; rev = r->rev; }
	}
	if (none) sl = 0;
	int64_t ref_end = this_pos + 1;
	if (r && r->n_cigar) { int64_t e = this_pos; for (uint32_t c : cig) { const uint32_t op = c & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) e += c >> 4; } ref_end = e > this_pos ? e : this_pos + 1; }
	const int nl = n_seg > 1 ? qname_len(qname) : (int)strlen(qname);
	if (nl > 254) { fprintf(stderr, "[ERROR] airlift: read name longer than 254 characters cannot be stored in BAM: %.40s...\n", qname); return -1; }
	const size_t start = out.size();
	Bw w{out};
	w.u32(0);                                                             // block_size, patched below
	w.i32(this_rid); w.i32(this_pos);                                      // 0-based pos; -1 when absent
	w.u8((uint8_t)(nl + 1)); w.u8((uint8_t)mapq); w.u16((uint16_t)reg2bin(this_pos < 0 ? -1 : this_pos, this_pos < 0 ? 0 : ref_end));
	w.u16((uint16_t)cig.size()); w.u16((uint16_t)flag); w.u32((uint32_t)sl);
	w.i32(next_rid); w.i32(next_pos); w.i32(tlen);
	w.mem(qname, nl); w.u8(0);
	for (uint32_t c : cig) w.u32(c);
	{
		const unsigned char *t16 = seq16(), *cmp = al_comp();
		for (int i = 0; i < sl; i += 2) {
			int a, b = 0;
			if (!rev) { a = t16[(unsigned char)sq[i]]; if (i + 1 < sl) b = t16[(unsigned char)sq[i + 1]]; }
			else { unsigned char c0 = (unsigned char)sq[sl - 1 - i]; a = t16[c0 < 128 ? cmp[c0] : c0]; if (i + 1 < sl) { unsigned char c1 = (unsigned char)sq[sl - 2 - i]; b = t16[c1 < 128 ? cmp[c1] : c1]; } }
			w.u8((uint8_t)(a << 4 | b));
		}
		if (ql) { if (!rev) for (int i = 0; i < sl; ++i) w.u8((uint8_t)(ql[i] - 33)); else for (int i = 0; i < sl; ++i) w.u8((uint8_t)(ql[sl - 1 - i] - 33)); }
		else for (int i = 0; i < sl; ++i) w.u8(0xff);
	}
	if (rg_id && rg_id[0]) w.tag_Z("RG", rg_id);
	if (r) {
		const char type = r->id == r->parent ? (r->inv ? 'I' : 'P') : (r->inv ? 'i' : 'S');
		if (r->n_cigar) { w.tag_i("NM", r->blen - r->mlen + (int)r->n_ambi); w.tag_i("ms", r->dp_max); w.tag_i("AS", r->dp_score); w.tag_i("nn", (int)r->n_ambi); }
		w.tag_A("tp", type); w.tag_i("cm", r->cnt); w.tag_i("s1", r->score);
		if (r->parent == r->id) w.tag_i("s2", r->subsc);
		if (r->n_cigar) {
			int n_gapo = 0, n_gap = 0;
			for (uint32_t i = 0; i < r->n_cigar; ++i) { const int op = r->cigar[i] & 0xf, len = r->cigar[i] >> 4; if (op == 1 || op == 2) ++n_gapo, n_gap += len; }
			const double div = 1.0 - (double)r->mlen / (r->blen - n_gap + n_gapo);
			char b[32]; if (div == 0.0) strcpy(b, "0"); else snprintf(b, 32, "%.4f", div);    // the value the SAM text carries
			w.tag_f("de", (float)atof(b));
		}
		if (r->split) w.tag_i("zd", r->split);
		if (r->parent == r->id && r->n_cigar && n_regs > 1) {
			int n_sa = 0;
			for (int i = 0; i < n_regs; ++i) if (i != r - regs && regs[i].parent == regs[i].id && regs[i].n_cigar) ++n_sa;
			if (n_sa > 0) {
				std::string sa;
				for (int i = 0; i < n_regs; ++i) {
					const al_reg1_t *q = &regs[i]; int l_M, l_I = 0, l_D = 0;
					if (r == q || q->parent != q->id || q->n_cigar == 0) continue;
					if (q->qe - q->qs < q->re - q->rs) l_M = q->qe - q->qs, l_D = (q->re - q->rs) - l_M;
					else l_M = q->re - q->rs, l_I = (q->qe - q->qs) - l_M;
					const int clip5 = q->rev ? l_seq - q->qe : q->qs, clip3 = q->rev ? q->qs : l_seq - q->qe;
					sa += mi->seq[q->rid].name; sa += ','; sa += std::to_string(q->rs + 1); sa += ','; sa += "+-"[q->rev]; sa += ',';
					if (clip5) { sa += std::to_string(clip5); sa += 'S'; }
					if (l_M) { sa += std::to_string(l_M); sa += 'M'; }
					if (l_I) { sa += std::to_string(l_I); sa += 'I'; }
					if (l_D) { sa += std::to_string(l_D); sa += 'D'; }
					if (clip3) { sa += std::to_string(clip3); sa += 'S'; }
					sa += ','; sa += std::to_string(q->mapq); sa += ','; sa += std::to_string(q->blen - q->mlen + (int)q->n_ambi); sa += ';';
				}
				w.tag_Z("SA", sa.c_str());
			}
		}
	}
	if (rep_len >= 0) w.tag_i("rl", rep_len);
	const uint32_t bs = (uint32_t)(out.size() - start - 4);
	memcpy(&out[start], &bs, 4);
	if (key) *key = this_rid < 0 ? ~0ULL : ((uint64_t)(uint32_t)this_rid << 32 | (uint32_t)(this_pos < 0 ? 0 : this_pos));
	if (unmapped) *unmapped = (flag & 0x4) ? 1 : 0;
	return (int)(bs + 4);
}

// ---- BGZF ------------------------------------------------------------------------------------------------
static const size_t BGZF_IN = 0xff00;          // uncompressed bytes per block
static const unsigned char BGZF_EOF[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};

static int bgzf_block(const char *src, size_t n, int level, std::vector<unsigned char> &dst)
{
	dst.resize(n + n / 8 + 128);
	z_stream zs; memset(&zs, 0, sizeof(zs));
	if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return -1;
	zs.next_in = (Bytef *)src; zs.avail_in = (uInt)n; zs.next_out = dst.data() + 18; zs.avail_out = (uInt)(dst.size() - 18 - 8);
	if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); return -1; }
	const size_t clen = zs.total_out; deflateEnd(&zs);
	const size_t total = 18 + clen + 8;
	if (total > 65536) return -1;
	static const unsigned char hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
	memcpy(dst.data(), hdr, 16);
	dst[16] = (unsigned char)((total - 1) & 0xff); dst[17] = (unsigned char)((total - 1) >> 8);
	const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)src, (uInt)n), isz = (uint32_t)n;
	memcpy(dst.data() + 18 + clen, &crc, 4); memcpy(dst.data() + 18 + clen + 4, &isz, 4);
	dst.resize(total);
	return 0;
}

int AlBgzf::write(const char *p, size_t n)
{
	while (n) {
		const size_t take = std::min(n, cap - buf.size());
		buf.insert(buf.end(), p, p + take); p += take; n -= take;
		if (buf.size() == cap && flush_full()) return -1;
	}
	return 0;
}
int AlBgzf::flush_full()
{   // compress whole blocks of the staging buffer on the worker threads, write them in order
	const size_t nb = buf.size() / BGZF_IN;
	if (nb == 0) return 0;
	std::vector<std::vector<unsigned char>> blk(nb); std::vector<int> bad(n_threads > 1 ? n_threads : 1, 0);
	al_parallel_for(n_threads, nb, [&](size_t lo, size_t hi, int t) { for (size_t b = lo; b < hi; ++b) if (bgzf_block(buf.data() + b * BGZF_IN, BGZF_IN, level, blk[b])) bad[t] = 1; });
	for (int b : bad) if (b) return -1;
	for (size_t b = 0; b < nb; ++b) if (fwrite(blk[b].data(), 1, blk[b].size(), out) != blk[b].size()) return -1;
	buf.erase(buf.begin(), buf.begin() + nb * BGZF_IN);
	return 0;
}
int AlBgzf::flush_all()
{
	if (flush_full()) return -1;
	if (!buf.empty()) { std::vector<unsigned char> b; if (bgzf_block(buf.data(), buf.size(), level, b) || fwrite(b.data(), 1, b.size(), out) != b.size()) return -1; buf.clear(); }
	return 0;
}
const unsigned char AL_BGZF_EOF[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
int al_bgzf_blocks(const char *src, size_t n, int level, int n_threads, std::vector<char> &dst)
{
	const size_t nb = (n + BGZF_IN - 1) / BGZF_IN;
	if (nb == 0) return 0;
	std::vector<std::vector<unsigned char>> blk(nb); std::vector<int> bad(n_threads > 1 ? n_threads : 1, 0);
	al_parallel_for(n_threads, nb, [&](size_t lo, size_t hi, int t) { for (size_t b = lo; b < hi; ++b) if (bgzf_block(src + b * BGZF_IN, std::min(BGZF_IN, n - b * BGZF_IN), level, blk[b])) bad[t] = 1; });
	for (int b : bad) if (b) return -1;
	size_t tot = 0; for (auto &b : blk) tot += b.size();
	dst.reserve(dst.size() + tot);
	for (auto &b : blk) dst.insert(dst.end(), b.begin(), b.end());
	return 0;
}
int AlBgzf::finish()
{
	if (flush_full()) return -1;
	if (!buf.empty()) { std::vector<unsigned char> b; if (bgzf_block(buf.data(), buf.size(), level, b) || fwrite(b.data(), 1, b.size(), out) != b.size()) return -1; buf.clear(); }
	return fwrite(BGZF_EOF, 1, 28, out) == 28 ? 0 : -1;
}

// BAM header: magic, SAM header text, reference dictionary
int al_bam_header(AlBgzf &z, const al_idx_t *mi, const char *rg, char *rg_id, bool sorted)
{
	std::string text;
	if (sorted) text += "@HD\tVN:1.6\tSO:coordinate\n";
	{   // same text al_write_sam_hdr prints
		char *mem = nullptr; size_t len = 0;
		FILE *f = open_memstream(&mem, &len);
		if (!f) return -1;
		al_write_sam_hdr(f, mi, rg, rg_id);
		fclose(f);
		text.append(mem, len); free(mem);
	}
	std::vector<char> h; Bw w{h};
	w.mem("BAM\1", 4); w.u32((uint32_t)text.size()); w.mem(text.data(), text.size());
	w.u32((uint32_t)mi->seq.size());
	for (const AlSeq &s : mi->seq) { w.u32((uint32_t)s.name.size() + 1); w.mem(s.name.c_str(), s.name.size() + 1); w.u32(s.len); }
	return z.write(h.data(), h.size());
}
