// al_index.cpp -- host side: options, FASTA/FASTQ reading, minimizer index construction.
// Index build is outside the timed hot path ("index build ... reported separately").
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <algorithm>
#include <thread>
#include <atomic>
#include "al_internal.h"
#include "al_io.h"

// ---- base tables (meaning of seq_nt4_table, sketch.c:9-26, and seq_comp_table, bseq.c) -----------------
static unsigned char *mk_nt4()
{
	static unsigned char t[256];
	for (int i = 0; i < 256; ++i) t[i] = 4;
	t[(int)'A'] = t[(int)'a'] = 0; t[(int)'C'] = t[(int)'c'] = 1; t[(int)'G'] = t[(int)'g'] = 2;
	t[(int)'T'] = t[(int)'t'] = 3; t[(int)'U'] = t[(int)'u'] = 3;
	t[0] = 0; t[1] = 1; t[2] = 2; t[3] = 3;
	return t;
}
static unsigned char *mk_comp()
{
	static unsigned char t[256];
	const char *a = "ACGTUMRWSYKVHDBN", *b = "TGCAAKYWSRMBDHVN";
	for (int i = 0; i < 256; ++i) t[i] = (unsigned char)i;
	for (int i = 0; a[i]; ++i) { t[(int)a[i]] = b[i]; t[(int)a[i] + 32] = b[i] + 32; }
	return t;
}
static const unsigned char *g_nt4 = mk_nt4(), *g_comp = mk_comp();
const unsigned char *al_nt4() { return g_nt4; }
const unsigned char *al_comp() { return g_comp; }

// ---- options (mm_set_opt / mm_check_opt, options.c:69-192; only the preset AirLift's path uses) --------
extern "C" int al_set_opt(const char *preset, al_idxopt_t *io, al_mapopt_t *mo)
{
	if (preset == 0) {                                   // options.c:4-49 defaults
		memset(io, 0, sizeof(*io)); memset(mo, 0, sizeof(*mo));
		io->k = 15; io->w = 10; io->bucket_bits = 14; io->mini_batch_size = 50000000; io->batch_size = 4000000000ULL;
		mo->seed = 11; mo->min_cnt = 3; mo->min_chain_score = 40; mo->bw = 500; mo->max_gap = 5000; mo->max_gap_ref = -1;
		mo->max_chain_skip = 25; mo->max_chain_iter = 5000; mo->mask_level = 0.5f; mo->pri_ratio = 0.8f; mo->best_n = 5;
		mo->a = 2; mo->b = 4; mo->q = 4; mo->e = 2; mo->q2 = 24; mo->e2 = 1; mo->sc_ambi = 1; mo->zdrop = 400; mo->zdrop_inv = 200;
		mo->end_bonus = -1; mo->min_dp_max = mo->min_chain_score * mo->a; mo->max_clip_ratio = 1.0f; mo->mini_batch_size = 500000000;
		mo->pe_ori = 0; mo->pe_bonus = 33;
		return 0;
	}
	if (strcmp(preset, "sr") == 0 || strcmp(preset, "short") == 0) {   // options.c:105-122
		io->flag = 0; io->k = 21; io->w = 11;
		mo->flag |= AL_F_SR | AL_F_FRAG_MODE | AL_F_NO_PRINT_2ND | AL_F_HEAP_SORT;
		mo->pe_ori = 0<<1|1;
		mo->a = 2; mo->b = 8; mo->q = 12; mo->e = 2; mo->q2 = 24; mo->e2 = 1;
		mo->zdrop = mo->zdrop_inv = 100; mo->end_bonus = 10; mo->max_frag_len = 800; mo->max_gap = 100; mo->bw = 100;
		mo->pri_ratio = 0.5f; mo->min_cnt = 2; mo->min_chain_score = 25; mo->min_dp_max = 40; mo->best_n = 20;
		mo->mid_occ = 1000; mo->max_occ = 5000; mo->mini_batch_size = 50000000;
		return 0;
	}
	return -1;
}

extern "C" int al_check_opt(const al_idxopt_t *io, const al_mapopt_t *mo)
{
	if (io->k <= 0 || io->w <= 0) { fprintf(stderr, "[ERROR] -k and -w must be positive\n"); return -5; }
	if (io->k > 28 || io->w >= 256) { fprintf(stderr, "[ERROR] k must be <= 28 and w < 256\n"); return -5; }
	if (mo->best_n < 0) { fprintf(stderr, "[ERROR] -N must be no less than 0\n"); return -4; }
	if (mo->pri_ratio < 0.0f || mo->pri_ratio > 1.0f) { fprintf(stderr, "[ERROR] -p must be within 0 and 1 (including 0 and 1)\n"); return -4; }
	if (mo->e <= 0 || mo->q <= 0) { fprintf(stderr, "[ERROR] -O and -E must be positive\n"); return -1; }
	if ((mo->q != mo->q2 || mo->e != mo->e2) && !(mo->e > mo->e2 && mo->q + mo->e < mo->q2 + mo->e2)) {
		fprintf(stderr, "[ERROR] dual gap penalties violating E1>E2 and O1+E1<O2+E2\n"); return -2; }
	if ((mo->q + mo->e) + (mo->q2 + mo->e2) > 127) { fprintf(stderr, "[ERROR] scoring system violating ({-O}+{-E})+({-O2}+{-E2}) <= 127\n"); return -1; }
	if (mo->zdrop < mo->zdrop_inv) { fprintf(stderr, "[ERROR] Z-drop should not be less than inversion-Z-drop\n"); return -5; }
	if (!(mo->flag & AL_F_SR)) { fprintf(stderr, "[ERROR] airlift: only the short-read (-x sr) path is implemented on the GPU\n"); return -7; }
	if (mo->q == mo->q2 && mo->e == mo->e2) { fprintf(stderr, "[ERROR] airlift: single-piece gap cost not supported (needs O1,E1 != O2,E2)\n"); return -7; }
	return 0;
}

// ---- sequence file reader (kseq.h semantics: FASTA or FASTQ, multi-line, optional gzip) ---------------
struct AlSeqFile { gzFile fp; unsigned char *buf; int beg, end, eof, last; };
AlSeqFile *al_sf_open(const char *fn)
{
	gzFile fp = strcmp(fn, "-") == 0? gzdopen(0, "r") : gzopen(fn, "r");
	if (!fp) return nullptr;
	AlSeqFile *f = new AlSeqFile(); f->fp = fp; f->buf = (unsigned char*)malloc(1<<18); f->beg = f->end = f->eof = f->last = 0;
	gzbuffer(fp, 1<<18);
	return f;
}
void al_sf_close(AlSeqFile *f) { if (f) { gzclose(f->fp); free(f->buf); delete f; } }
static inline int sf_getc(AlSeqFile *f)
{
	if (f->beg >= f->end) {
		if (f->eof) return -1;
		f->beg = 0; f->end = gzread(f->fp, f->buf, 1<<18);
		if (f->end <= 0) { f->eof = 1; f->end = 0; return -1; }
	}
	return f->buf[f->beg++];
}
int al_sf_read(AlSeqFile *f, std::string &name, std::string &seq, std::string &qual)
{
	int c;
	name.clear(); seq.clear(); qual.clear();
	if (f->last == 0) { while ((c = sf_getc(f)) >= 0 && c != '>' && c != '@') {} if (c < 0) return -1; f->last = c; }
	while ((c = sf_getc(f)) >= 0 && c != ' ' && c != '\t' && c != '\n' && c != '\r') name.push_back((char)c);
	while (c >= 0 && c != '\n') c = sf_getc(f);
	while ((c = sf_getc(f)) >= 0 && c != '>' && c != '+' && c != '@') if (c > 32) seq.push_back((char)c);
	f->last = (c == '>' || c == '@')? c : 0;
	if (c != '+') return (int)seq.size();
	while ((c = sf_getc(f)) >= 0 && c != '\n') {}
	while (qual.size() < seq.size() && (c = sf_getc(f)) >= 0) if (c > 32) qual.push_back((char)c);
	f->last = 0;
	return (int)seq.size();
}

// ---- host sketch (algorithm of mm_sketch, sketch.c:77-143, non-HPC) -----------------------------------
static inline uint64_t hash64m(uint64_t key, uint64_t mask)
{
	key = (~key + (key << 21)) & mask; key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & mask; key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & mask; key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

void al_sketch_host(const uint8_t *codes, uint32_t len, int w, int k, uint32_t rid, std::vector<uint64_t> &hx, std::vector<uint64_t> &hy)
{
	struct E { uint64_t x, y; };
	const uint64_t shift1 = 2 * (k - 1), mask = (1ULL<<2*k) - 1; uint64_t kmer[2] = {0, 0};
	std::vector<E> buf(w, E{UINT64_MAX, UINT64_MAX}); E mn{UINT64_MAX, UINT64_MAX};
	int l = 0, buf_pos = 0, min_pos = 0;
	auto emit = [&](const E &e) { hx.push_back(e.x >> 8); hy.push_back(e.y); };
	for (uint32_t i = 0; i < len; ++i) {
		int c = codes[i]; E info{UINT64_MAX, UINT64_MAX};
		if (c < 4) {
			int span = l + 1 < k? l + 1 : k;
			kmer[0] = (kmer[0] << 2 | c) & mask;
			kmer[1] = (kmer[1] >> 2) | (3ULL^c) << shift1;
			if (kmer[0] == kmer[1]) continue;
			int z = kmer[0] < kmer[1]? 0 : 1;
			++l;
			if (l >= k) { info.x = hash64m(kmer[z], mask) << 8 | span; info.y = (uint64_t)rid<<32 | (uint32_t)i<<1 | z; }
		} else l = 0;
		buf[buf_pos] = info;
		if (l == w + k - 1 && mn.x != UINT64_MAX) {
			for (int j = buf_pos + 1; j < w; ++j) if (mn.x == buf[j].x && buf[j].y != mn.y) emit(buf[j]);
			for (int j = 0; j < buf_pos; ++j) if (mn.x == buf[j].x && buf[j].y != mn.y) emit(buf[j]);
		}
		if (info.x <= mn.x) {
			if (l >= w + k && mn.x != UINT64_MAX) emit(mn);
			mn = info; min_pos = buf_pos;
		} else if (buf_pos == min_pos) {
			if (l >= w + k - 1 && mn.x != UINT64_MAX) emit(mn);
			mn.x = UINT64_MAX;
			for (int j = buf_pos + 1; j < w; ++j) if (mn.x >= buf[j].x) { mn = buf[j]; min_pos = j; }
			for (int j = 0; j <= buf_pos; ++j) if (mn.x >= buf[j].x) { mn = buf[j]; min_pos = j; }
			if (l >= w + k - 1 && mn.x != UINT64_MAX) {
				for (int j = buf_pos + 1; j < w; ++j) if (mn.x == buf[j].x && mn.y != buf[j].y) emit(buf[j]);
				for (int j = 0; j <= buf_pos; ++j) if (mn.x == buf[j].x && mn.y != buf[j].y) emit(buf[j]);
			}
		}
		if (++buf_pos == w) buf_pos = 0;
	}
	if (mn.x != UINT64_MAX) emit(mn);
}

// ---- index construction (meaning of mm_idx_gen / worker_post, index.c:191-243,353-372) ----------------
static al_idx_t *idx_from_codes(int w, int k, std::vector<AlSeq> &&seqs, std::vector<std::vector<uint8_t>> &codes, int n_threads)
{
	al_idx_t *mi = new al_idx_t();
	mi->k = k; mi->w = w; mi->seq = std::move(seqs);
	uint64_t sum = 0;
	for (size_t i = 0; i < mi->seq.size(); ++i) { mi->seq[i].offset = sum; sum += mi->seq[i].len; }
	mi->tot_len = sum;
	mi->S4.assign((sum + 7) / 8 + 8, 0);
	size_t n = mi->seq.size();
	std::vector<std::vector<uint64_t>> hx(n), hy(n);
	std::atomic<size_t> next(0);
	auto work = [&]() {
		for (;;) {
			size_t i = next.fetch_add(1); if (i >= n) break;
			if (mi->seq[i].len > 0) al_sketch_host(codes[i].data(), mi->seq[i].len, w, k, (uint32_t)i, hx[i], hy[i]);
		}
	};
	int nt = n_threads > 1? n_threads : 1; if ((size_t)nt > n) nt = n? (int)n : 1;
	std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(work); for (auto &t : th) t.join();
	for (size_t i = 0; i < n; ++i) {          // 4-bit packing (index.c:320-326)
		uint64_t o = mi->seq[i].offset; const uint8_t *c = codes[i].data();
		for (uint32_t j = 0; j < mi->seq[i].len; ++j, ++o) mi->S4[o>>3] |= (uint32_t)c[j] << ((o&7)<<2);
		std::vector<uint8_t>().swap(codes[i]);
	}
	size_t tot = 0; for (size_t i = 0; i < n; ++i) tot += hx[i].size();
	struct P { uint64_t h, y; };
	std::vector<P> all; all.reserve(tot);
	for (size_t i = 0; i < n; ++i) { for (size_t j = 0; j < hx[i].size(); ++j) all.push_back(P{hx[i][j], hy[i][j]}); std::vector<uint64_t>().swap(hx[i]); std::vector<uint64_t>().swap(hy[i]); }
	std::sort(all.begin(), all.end(), [](const P &a, const P &b) { return a.h < b.h || (a.h == b.h && a.y < b.y); });
	uint64_t nk = 0; for (size_t i = 0; i < tot; ++i) if (i == 0 || all[i].h != all[i-1].h) ++nk;
	mi->n_keys = nk; mi->n_pos = tot;
	int bits = 4; while ((1ULL<<bits) < nk * 2 + 2) ++bits;
	mi->tab_bits = bits;
	mi->tab.assign((size_t)2 << bits, 0);
	mi->pos.resize(tot? tot : 1);
	uint64_t tmask = (1ULL<<bits) - 1;
	for (size_t i = 0, j; i < tot; i = j) {
		for (j = i; j < tot && all[j].h == all[i].h; ++j) mi->pos[j] = all[j].y;
		uint64_t s = al_tab_slot(all[i].h, bits);
		while (mi->tab[2*s]) s = (s + 1) & tmask;
		if (j - i == 1 && mi->seq.size() <= AL_TAB_SINGLE_MAX_SEQ) { mi->tab[2*s] = (all[i].h + 1) | AL_TAB_SINGLE; mi->tab[2*s+1] = all[i].y; }
		else { mi->tab[2*s] = all[i].h + 1; mi->tab[2*s+1] = (uint64_t)i << 32 | (uint32_t)(j - i); }
	}
	return mi;
}

extern "C" al_idx_t *al_idx_str(int w, int k, int n, const char **seq, const char **name)
{
	if (n <= 0) return nullptr;
	std::vector<AlSeq> seqs(n); std::vector<std::vector<uint8_t>> codes(n);
	for (int i = 0; i < n; ++i) {
		size_t l = strlen(seq[i]);
		seqs[i].name = name && name[i]? name[i] : std::to_string(i); seqs[i].len = (uint32_t)l;
		codes[i].resize(l);
		for (size_t j = 0; j < l; ++j) codes[i][j] = g_nt4[(unsigned char)seq[i][j]];
	}
	return idx_from_codes(w, k, std::move(seqs), codes, 1);
}

extern "C" al_idx_t *al_idx_build(const char *fn, const al_idxopt_t *io, int n_threads)
{
	AlSeqFile *f = al_sf_open(fn);
	if (!f) { fprintf(stderr, "[ERROR] airlift: failed to open '%s'\n", fn); return nullptr; }
	std::vector<AlSeq> seqs; std::vector<std::vector<uint8_t>> codes; std::string nm, sq, ql;
	while (al_sf_read(f, nm, sq, ql) >= 0) {
		AlSeq s; s.name = nm; s.len = (uint32_t)sq.size(); s.offset = 0; seqs.push_back(s);
		codes.emplace_back(sq.size());
		std::vector<uint8_t> &c = codes.back();
		for (size_t j = 0; j < sq.size(); ++j) c[j] = g_nt4[(unsigned char)sq[j]];
	}
	al_sf_close(f);
	if (seqs.empty()) { fprintf(stderr, "[ERROR] airlift: no sequences in '%s'\n", fn); return nullptr; }
	return idx_from_codes(io->w, io->k, std::move(seqs), codes, n_threads);
}

void al_idx_free_device(al_idx_t *mi);   // al_runtime.hip
extern "C" void al_idx_destroy(al_idx_t *mi) { if (mi) { al_idx_free_device(mi); delete mi; } }
extern "C" uint32_t al_idx_n_seq(const al_idx_t *mi) { return (uint32_t)mi->seq.size(); }
extern "C" int al_idx_k(const al_idx_t *mi) { return mi ? mi->k : 0; }
extern "C" int al_idx_w(const al_idx_t *mi) { return mi ? mi->w : 0; }
extern "C" const char *al_idx_seq_name(const al_idx_t *mi, uint32_t rid) { return rid < mi->seq.size()? mi->seq[rid].name.c_str() : nullptr; }
extern "C" uint32_t al_idx_seq_len(const al_idx_t *mi, uint32_t rid) { return rid < mi->seq.size()? mi->seq[rid].len : 0; }
extern "C" void al_idx_stat(const al_idx_t *mi, uint64_t *n_keys, uint64_t *n_pos, uint64_t *n_bases)
{
	if (n_keys) *n_keys = mi->n_keys; if (n_pos) *n_pos = mi->n_pos; if (n_bases) *n_bases = mi->tot_len;
}
extern "C" const char *al_version(void) { return AL_VERSION; }
