// al_mmi.cpp -- the reference's index file (.mmi): mm_idx_dump / mm_idx_load / mm_idx_is_idx (index.c:438-531), so that an index written by the fork is read here
// and one written here is read by the fork.  The file holds the reference's layout -- 2^b buckets, per bucket the position array p of the minimizers that occur more
// than once and the (key, value) pairs of its hash table, key = minimizer >> b << 1 | (occurs once), value = the position word or offset << 32 | count into p -- which
// is regrouped into this library's one open-addressing table + one position array on load, and back on dump.  The order of a bucket's pairs is the table's slot order
// here and khash's slot order there: the loader of either side inserts them one by one, so the files differ in that order only.
#include "al_internal.h"
#include <sys/stat.h>
#include <stdexcept>
#include <cstdio>
#include <cstring>
#include <algorithm>

static const char AL_MMI_MAGIC[4] = {'M', 'M', 'I', '\2'};                   // minimap.h:30-31 (MM_IDX_MAGIC)
#define AL_MMI_NO_SEQ 0x2                                                     // MM_I_NO_SEQ (minimap.h:27)
#define AL_MMI_HPC    0x1                                                     // MM_I_HPC

int al_idx_to_host(al_idx_t *mi);                                            // al_index_dev.hip: a device-built index gets its host arrays

extern "C" int64_t al_idx_is_idx(const char *fn)
{   // mm_idx_is_idx, index.c:531-552: the file size if it starts with the magic, 0 if not, -1 if it cannot be read
	if (!fn || !strcmp(fn, "-")) return 0;
	FILE *fp = fopen(fn, "rb");
	if (!fp) return -1;
	char m[4]; int64_t ret = 0;
	if (fread(m, 1, 4, fp) == 4 && memcmp(m, AL_MMI_MAGIC, 4) == 0) { fseek(fp, 0, SEEK_END); ret = (int64_t)ftell(fp); }
	fclose(fp);
	return ret;
}

extern "C" int al_idx_dump(const char *fn, al_idx_t *mi)
{
	if (!mi || !fn) return -1;
	if (mi->built_on >= 0 && mi->tab.empty() && al_idx_to_host(mi) != 0) { fprintf(stderr, "[ERROR] airlift: failed to copy the index off the device\n"); return -1; }
	FILE *fp = strcmp(fn, "-") ? fopen(fn, "wb") : stdout;
	if (!fp) { fprintf(stderr, "[ERROR] airlift: failed to write the index to '%s'\n", fn); return -1; }
	const int b = 14;                                                        // mm_idxopt_init: bucket_bits
	const uint32_t nb = 1u << b, bm = nb - 1;
	uint32_t x[5] = {(uint32_t)mi->w, (uint32_t)mi->k, (uint32_t)b, (uint32_t)mi->seq.size(), 0u};
	bool ok = fwrite(AL_MMI_MAGIC, 1, 4, fp) == 4 && fwrite(x, 4, 5, fp) == 5;
	for (const AlSeq &s : mi->seq) {
		const uint8_t l = (uint8_t)std::min<size_t>(s.name.size(), 255);
		ok = ok && fwrite(&l, 1, 1, fp) == 1 && (l == 0 || fwrite(s.name.data(), 1, l, fp) == l) && fwrite(&s.len, 4, 1, fp) == 1;
	}
	// pass 1: pairs and list entries per bucket
	const uint64_t n_slots = 1ULL << mi->tab_bits;
	std::vector<uint32_t> n_pair(nb + 1, 0); std::vector<uint64_t> n_p(nb + 1, 0);
	for (uint64_t s = 0; s < n_slots; ++s) {
		const uint64_t key = mi->tab[2 * s]; if (!key) continue;
		const uint64_t h = (key & ~AL_TAB_SINGLE) - 1; const uint32_t bk = (uint32_t)h & bm;
		++n_pair[bk];
		if (!(key & AL_TAB_SINGLE)) { const uint32_t n = (uint32_t)mi->tab[2 * s + 1]; if (n > 1) n_p[bk] += n; }
	}
	std::vector<uint64_t> pair_off(nb + 1, 0), p_off(nb + 1, 0);
	for (uint32_t i = 0; i < nb; ++i) { pair_off[i + 1] = pair_off[i] + n_pair[i]; p_off[i + 1] = p_off[i] + n_p[i]; }
	std::vector<uint64_t> pairs(2 * pair_off[nb] + 2), pl(p_off[nb] + 1);
	std::vector<uint64_t> pc(pair_off.begin(), pair_off.end() - 1), lc(nb, 0);   // fill cursors (list cursor relative to the bucket)
	for (uint64_t s = 0; s < n_slots; ++s) {
		const uint64_t key = mi->tab[2 * s]; if (!key) continue;
		const uint64_t h = (key & ~AL_TAB_SINGLE) - 1, v = mi->tab[2 * s + 1]; const uint32_t bk = (uint32_t)h & bm;
		uint64_t k2 = h >> b << 1, v2;
		if (key & AL_TAB_SINGLE) { k2 |= 1; v2 = v; }
		else {
			const uint64_t off = v >> 32; const uint32_t n = (uint32_t)v;
			if (n == 1) { k2 |= 1; v2 = mi->pos[off]; }                       // (an index of more than 65536 contigs keeps singletons as lists of one)
			else { v2 = lc[bk] << 32 | n; memcpy(&pl[p_off[bk] + lc[bk]], &mi->pos[off], (size_t)n * 8); lc[bk] += n; }
		}
		pairs[2 * pc[bk]] = k2; pairs[2 * pc[bk] + 1] = v2; ++pc[bk];
	}
	for (uint32_t i = 0; i < nb && ok; ++i) {
		const int32_t n = (int32_t)n_p[i]; const uint32_t size = n_pair[i];
		ok = fwrite(&n, 4, 1, fp) == 1 && (n == 0 || fwrite(&pl[p_off[i]], 8, (size_t)n, fp) == (size_t)n) && fwrite(&size, 4, 1, fp) == 1 &&
		     (size == 0 || fwrite(&pairs[2 * pair_off[i]], 8, 2 * (size_t)size, fp) == 2 * (size_t)size);
	}
	const size_t nw = (size_t)((mi->tot_len + 7) / 8);
	ok = ok && (nw == 0 || fwrite(mi->S4.data(), 4, nw, fp) == nw);
	ok = ok && fflush(fp) == 0;
	if (fp != stdout) ok = (fclose(fp) == 0) && ok;
	if (!ok) fprintf(stderr, "[ERROR] airlift: failed to write the index to '%s'\n", fn);
	return ok ? 0 : -1;
}

static al_idx_t *al_idx_load_impl(const char *fn);
extern "C" al_idx_t *al_idx_load(const char *fn)
{   // (a corrupt or hostile file must come back as NULL, not as an exception through the C boundary)
	try { return al_idx_load_impl(fn); }
	catch (const std::exception &e) { fprintf(stderr, "[ERROR] airlift: loading the index '%s' failed (%s)\n", fn ? fn : "(null)", e.what()); return nullptr; }
}
static al_idx_t *al_idx_load_impl(const char *fn)
{
	FILE *fp = fn && strcmp(fn, "-") ? fopen(fn, "rb") : nullptr;
	if (!fp) { fprintf(stderr, "[ERROR] airlift: failed to open '%s'\n", fn ? fn : "(null)"); return nullptr; }
	char magic[4]; uint32_t x[5];
	if (fread(magic, 1, 4, fp) != 4 || memcmp(magic, AL_MMI_MAGIC, 4) != 0 || fread(x, 4, 5, fp) != 5) { fclose(fp); return nullptr; }
	if (x[4] & AL_MMI_HPC) { fprintf(stderr, "[ERROR] airlift: '%s' is a homopolymer-compressed index (not on the short-read path)\n", fn); fclose(fp); return nullptr; }
	if (x[4] & AL_MMI_NO_SEQ) { fprintf(stderr, "[ERROR] airlift: '%s' was written without the sequences; alignment needs them\n", fn); fclose(fp); return nullptr; }
	if (x[0] > 32 && x[0] < 256) { fprintf(stderr, "[ERROR] airlift: '%s' was built with a minimizer window of w=%u; the device sketch holds windows of up to 32 k-mers\n", fn, x[0]); fclose(fp); return nullptr; }
	if (x[1] < 1 || x[1] > AL_MAX_K || x[0] < 1 || x[0] > 32 || x[2] > 30) { fprintf(stderr, "[ERROR] airlift: index parameters out of range in '%s' (w=%u k=%u b=%u)\n", fn, x[0], x[1], x[2]); fclose(fp); return nullptr; }
	// sizes in the file are checked against what the file can hold before anything is resized to them: a contig costs >= 5 bytes, an occurrence 8, a key 16
	uint64_t fsz = 0;
	{ struct stat sb; if (fstat(fileno(fp), &sb) == 0 && S_ISREG(sb.st_mode)) fsz = (uint64_t)sb.st_size; }
	if (fsz && (uint64_t)x[3] * 5 > fsz) { fprintf(stderr, "[ERROR] airlift: '%s' names %u contigs, more than its %llu bytes can hold\n", fn, x[3], (unsigned long long)fsz); fclose(fp); return nullptr; }
	al_idx_t *mi = new al_idx_t();
	mi->w = (int)x[0]; mi->k = (int)x[1];
	const int b = (int)x[2]; const uint32_t nb = 1u << b;
	bool ok = true; uint64_t sum = 0;
	mi->seq.resize(x[3]);
	for (uint32_t i = 0; i < x[3] && ok; ++i) {
		uint8_t l = 0; char nm[256];
		ok = fread(&l, 1, 1, fp) == 1 && (l == 0 || fread(nm, 1, l, fp) == l) && fread(&mi->seq[i].len, 4, 1, fp) == 1;
		mi->seq[i].name.assign(nm, l); mi->seq[i].offset = sum; sum += mi->seq[i].len;
	}
	mi->tot_len = sum;
	struct Ent { uint64_t h, v; bool single; };
	std::vector<Ent> ents; std::vector<uint64_t> buf;
	const bool single_ok = mi->seq.size() <= AL_TAB_SINGLE_MAX_SEQ;
	for (uint32_t i = 0; i < nb && ok; ++i) {
		int32_t n = 0; uint32_t size = 0;
		ok = fread(&n, 4, 1, fp) == 1 && n >= 0 && (!fsz || (uint64_t)n * 8 <= fsz);
		const uint64_t base = mi->pos.size();
		if (ok && n) { mi->pos.resize(base + (size_t)n); ok = fread(&mi->pos[base], 8, (size_t)n, fp) == (size_t)n; }
		ok = ok && fread(&size, 4, 1, fp) == 1 && (!fsz || (uint64_t)size * 16 <= fsz);
		if (ok && size) {
			buf.resize(2 * (size_t)size); ok = fread(buf.data(), 8, buf.size(), fp) == buf.size();
			for (uint32_t j = 0; j < size && ok; ++j) {
				const uint64_t k2 = buf[2 * j], v2 = buf[2 * j + 1];
				Ent e; e.h = (k2 >> 1) << b | i; e.single = (k2 & 1) != 0;
				if (e.single) e.v = v2;
				else { if ((v2 >> 32) + (uint32_t)v2 > (uint64_t)n) { ok = false; break; } e.v = (base + (v2 >> 32)) << 32 | (uint32_t)v2; }
				ents.push_back(e);
			}
		}
	}
	const size_t nw = (size_t)((sum + 7) / 8);
	ok = ok && (!fsz || (uint64_t)nw * 4 <= fsz);
	if (ok) mi->S4.assign(nw + 8, 0);
	ok = ok && (nw == 0 || fread(mi->S4.data(), 4, nw, fp) == nw);
	fclose(fp);
	if (!ok) { fprintf(stderr, "[ERROR] airlift: '%s' is truncated or not an index of this format\n", fn); delete mi; return nullptr; }
	// every occurrence has its entry in the position array, as in an index built here (n_pos counts occurrences); singletons that cannot live in the table entry
	// (more than 65536 contigs) become lists of one
	for (Ent &e : ents) if (e.single) { const uint64_t o = mi->pos.size(); mi->pos.push_back(e.v); if (!single_ok) { e.v = o << 32 | 1u; e.single = false; } }
	if (mi->pos.size() >= (1ULL << 32)) { fprintf(stderr, "[ERROR] airlift: '%s' holds more than 2^32 positions\n", fn); delete mi; return nullptr; }
	mi->n_keys = ents.size(); mi->n_pos = mi->pos.size();
	if (mi->pos.empty()) mi->pos.resize(1);
	int bits = 4; while ((1ULL << bits) < mi->n_keys * 2 + 2) ++bits;
	mi->tab_bits = bits; mi->tab.assign((size_t)2 << bits, 0);
	const uint64_t tmask = (1ULL << bits) - 1;
	for (const Ent &e : ents) {
		uint64_t s = al_tab_slot(e.h, bits);
		while (mi->tab[2 * s]) s = (s + 1) & tmask;
		mi->tab[2 * s] = (e.h + 1) | (e.single ? AL_TAB_SINGLE : 0ULL); mi->tab[2 * s + 1] = e.v;
	}
	return mi;
}
