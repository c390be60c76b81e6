// al_extract.cpp -- SURVEY.md N1: the read-extraction stage that feeds the re-alignment path, as two host routines that
// replace AirLift's per-region process spawning (product code, C++; no GPU work: this stage is I/O and set logic).
//
//   al_extract_reads      src/4-extract_reads/extract_reads.sh:8 / extract_reads_noprune.sh:7.  The scripts start
//                         `samtools view BAM chrom:B-E | convert2bed | awk` once per BED line (run_pipeline.sh:58 spreads them
//                         over xargs -P); here the BAM is read ONCE and every mapped record is tested against an index of the
//                         BED lines.  Output rows and their order are the scripts' (`sort -uk4,4`, C locale).
//   al_extract_sequence   src/4-extract_reads/extract_sequence.sh:17-19: seqtk subseq of both FASTQ files by the selected names,
//                         BBMap repair.sh (pairs / singletons) and rename.sh (realigned_<n>, realigned_singleton_<n>).
// samtools / bedops / BBMap are not in the reference tree: their behaviour is restated from the scripts and the tools'
// documentation (oracle/n1_oracle.py states the same rules in Python and is what the tests compare against).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <zlib.h>
#include <unistd.h>
#include <sys/mman.h>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "al_internal.h"
#include "al_seqio.h"

namespace {

struct GzIn {                      // BGZF is a sequence of gzip members: zlib's gzread walks it as one stream
	gzFile fp = nullptr; std::vector<unsigned char> buf; size_t beg = 0, end = 0;
	bool open(const char *fn) { fp = strcmp(fn, "-") == 0 ? gzdopen(0, "r") : gzopen(fn, "r"); if (!fp) return false; gzbuffer(fp, 1 << 20); buf.resize(8 << 20); return true; }
	~GzIn() { if (fp) gzclose(fp); }
	bool read(void *dst, size_t n)
	{
		unsigned char *d = (unsigned char *)dst;
		while (n) {
			if (beg == end) { const int k = gzread(fp, buf.data(), (unsigned)buf.size()); if (k <= 0) return false; beg = 0; end = (size_t)k; }
			const size_t t = std::min(n, end - beg);
			memcpy(d, buf.data() + beg, t); d += t; beg += t; n -= t;
		}
		return true;
	}
};

struct BedLine { int64_t b, e; uint32_t idx; };
struct ChromIdx { std::vector<BedLine> v; std::vector<int64_t> pmax; };      // sorted by b; pmax[i] = max e over v[0..i]

} // namespace

// extract_reads.sh:8 (prune != 0) / extract_reads_noprune.sh:7 (prune == 0) for every line of bed_fn against bam_fn.
// Rows: chrom, start, end, name[.1|.2], MAPQ, CIGAR.  Returns the number of rows, negative on error.
extern "C" int64_t al_extract_reads(const char *bam_fn, const char *bed_fn, int read_size, int prune, FILE *out)
{
	// BED lines: (chrom B E); a record belongs to the line if its BED row lies in [B-1, E-1]
	std::unordered_map<std::string, ChromIdx> idx;
	{
		FILE *fb = strcmp(bed_fn, "-") == 0 ? stdin : fopen(bed_fn, "r");
		if (!fb) { fprintf(stderr, "ERROR: failed to open file '%s'\n", bed_fn); return -1; }
		char line[1 << 16]; uint32_t n = 0;
		while (fgets(line, sizeof(line), fb)) {
			char chrom[1 << 12]; long long b, e;
			if (sscanf(line, "%4095s %lld %lld", chrom, &b, &e) != 3) { ++n; continue; }
			idx[chrom].v.push_back(BedLine{b, e, n}); ++n;
		}
		if (fb != stdin) fclose(fb);
		for (auto &kv : idx) {
			auto &v = kv.second.v;
			std::stable_sort(v.begin(), v.end(), [](const BedLine &x, const BedLine &y) { return x.b < y.b; });
			kv.second.pmax.resize(v.size()); int64_t m = INT64_MIN;
			for (size_t i = 0; i < v.size(); ++i) { m = std::max(m, v[i].e); kv.second.pmax[i] = m; }
		}
	}
	GzIn in;
	if (!in.open(bam_fn)) { fprintf(stderr, "ERROR: failed to open file '%s'\n", bam_fn); return -1; }
	char magic[4]; int32_t l_text, n_ref;
	if (!in.read(magic, 4) || memcmp(magic, "BAM\1", 4) != 0 || !in.read(&l_text, 4)) { fprintf(stderr, "ERROR: '%s' is not a BAM file\n", bam_fn); return -2; }
	// every length field of the file is checked before it sizes anything: a truncated or corrupt BAM is error -2, not a crash
	if (l_text < 0) return -2;
	{ std::vector<char> t((size_t)l_text); if (l_text && !in.read(t.data(), t.size())) return -2; }
	if (!in.read(&n_ref, 4) || n_ref < 0 || n_ref > (1 << 24)) return -2;
	std::vector<const ChromIdx *> ref_idx((size_t)n_ref, nullptr); std::vector<std::string> ref_name((size_t)n_ref);
	for (int i = 0; i < n_ref; ++i) {
		int32_t l; if (!in.read(&l, 4) || l < 1 || l > (1 << 20)) return -2;       // l_name counts the terminating NUL
		std::vector<char> nm((size_t)l); int32_t len;
		if (!in.read(nm.data(), nm.size()) || !in.read(&len, 4) || nm[(size_t)l - 1] != 0) return -2;
		ref_name[i] = nm.data();
		auto it = idx.find(ref_name[i]); if (it != idx.end()) ref_idx[i] = &it->second;
	}
	struct Row { uint32_t bed; uint64_t ord; std::string text; };
	std::unordered_map<std::string, Row> rows;          // name (with suffix) -> first row in (BED line, BAM) order
	const std::string rs = std::to_string(read_size) + "M";
	std::vector<unsigned char> rec; std::vector<uint32_t> cg; uint64_t ord = 0;
	for (;;) {
		int32_t bs;
		if (!in.read(&bs, 4)) break;
		if (bs < 32 || bs > (1 << 28)) return -2;
		rec.resize((size_t)bs);
		if (!in.read(rec.data(), rec.size())) return -2;
		++ord;
		int32_t rid, pos; memcpy(&rid, rec.data(), 4); memcpy(&pos, rec.data() + 4, 4);
		const uint32_t l_rn = rec[8], mapq = rec[9]; uint16_t n_cig, flag; memcpy(&n_cig, rec.data() + 12, 2); memcpy(&flag, rec.data() + 14, 2);
		if (l_rn < 1 || (size_t)32 + l_rn + 4 * (size_t)n_cig > (size_t)bs || rec[32 + l_rn - 1] != 0) return -2;   // name (NUL-terminated) and CIGAR inside the record
		if (rid < 0 || rid >= n_ref || !ref_idx[rid] || (flag & 4)) continue;          // convert2bed drops unmapped records
		cg.resize(n_cig); if (n_cig) memcpy(cg.data(), rec.data() + 32 + l_rn, 4 * (size_t)n_cig);   // (the words are not aligned in the record)
		if (n_cig == 2 && (cg[0] & 15) == 4 && (cg[1] & 15) == 3) { fprintf(stderr, "ERROR: '%s' holds a CIGAR of more than 65535 operations (CG tag): not supported\n", bam_fn); return -2; }
		int64_t rl = 0; for (int i = 0; i < n_cig; ++i) { const uint32_t op = cg[i] & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += cg[i] >> 4; }
		const int64_t s = pos, e = pos + rl;
		// smallest BED line number with B-1 <= s and e <= E-1
		const ChromIdx &ci = *ref_idx[rid];
		size_t hi = std::upper_bound(ci.v.begin(), ci.v.end(), s + 1, [](int64_t val, const BedLine &x) { return val < x.b; }) - ci.v.begin();   // lines with b <= s + 1
		uint32_t best = UINT32_MAX;
		for (size_t i = hi; i-- > 0; ) { if (ci.pmax[i] < e + 1) break; if (ci.v[i].e >= e + 1 && ci.v[i].idx < best) best = ci.v[i].idx; }
		if (best == UINT32_MAX) continue;
		std::string cs;
		for (int i = 0; i < n_cig; ++i) { cs += std::to_string(cg[i] >> 4); cs += "MIDNSHP=X"[cg[i] & 15]; }
		if (cs.empty()) cs = "*";
		if (prune && !(mapq <= 10 || cs != rs)) continue;
		std::string name((const char *)rec.data() + 32);
		if (flag & 64) name += ".1"; else if (flag & 128) name += ".2";
		auto it = rows.find(name);
		if (it != rows.end() && (it->second.bed < best || (it->second.bed == best && it->second.ord < ord))) continue;
		Row r; r.bed = best; r.ord = ord;
		r.text = ref_name[rid] + "\t" + std::to_string(s) + "\t" + std::to_string(e) + "\t" + name + "\t" + std::to_string(mapq) + "\t" + cs + "\n";
		rows[name] = std::move(r);
	}
	std::vector<const std::pair<const std::string, Row> *> order; order.reserve(rows.size());
	for (const auto &kv : rows) order.push_back(&kv);
	std::sort(order.begin(), order.end(), [](const std::pair<const std::string, Row> *a, const std::pair<const std::string, Row> *b) { return a->first < b->first; });   // sort -k4,4, C locale
	for (const auto *p : order) if (fwrite(p->second.text.data(), 1, p->second.text.size(), out) != p->second.text.size()) return -3;
	return (int64_t)order.size();
}

namespace {
struct FqRec { std::string name, seq, qual; };
// seqtk subseq FQ LIST (seqtk.c:546-620): records whose name (first word of the header) is listed, in file order
int subseq(const char *fn, const std::unordered_set<std::string> &want, std::vector<FqRec> &out)
{
	AlSeqReader rd;
	if (!rd.open(fn)) { fprintf(stderr, "ERROR: failed to open file '%s'\n", fn); return -1; }
	AlChunk c;
	for (;;) {
		c.text.clear(); c.recs.clear();
		if (!rd.read(c)) break;
		const AlRec &r = c.recs[0];
		const char *nm = c.text.data() + r.name;
		if (!want.count(nm)) continue;
		FqRec q; q.name = nm; q.seq.assign(c.text.data() + r.seq, r.len); if (r.qual != ~0u) q.qual.assign(c.text.data() + r.qual, r.len);
		out.push_back(std::move(q));
	}
	return 0;
}
std::string pair_key(const std::string &n) { const size_t l = n.size(); return l > 2 && n[l - 2] == '/' && (n[l - 1] == '1' || n[l - 1] == '2') ? n.substr(0, l - 2) : n; }
int write_fq_to(FILE *f, const std::vector<const FqRec *> &v, const char *prefix)
{
	size_t i = 0;
	for (const FqRec *r : v) {
		fprintf(f, "@%s_%zu\n", prefix, i++);
		fwrite(r->seq.data(), 1, r->seq.size(), f);
		if (!r->qual.empty()) { fputs("\n+\n", f); fwrite(r->qual.data(), 1, r->qual.size(), f); fputc('\n', f); }
		else { fputs("\n+\n", f); for (size_t j = 0; j < r->seq.size(); ++j) fputc('I', f); fputc('\n', f); }
	}
	return ferror(f) ? -1 : 0;
}
int write_fq(const std::string &path, const std::vector<const FqRec *> &v, const char *prefix)
{
	FILE *f = fopen(path.c_str(), "wb");
	if (!f) { fprintf(stderr, "ERROR: failed to write '%s'\n", path.c_str()); return -1; }
	const int w = write_fq_to(f, v, prefix);
	return fclose(f) == 0 && w == 0 ? 0 : -1;
}
// the names extract_sequence.sh's two awk lines pick from the rows (column 4 ending in 1 / 2, suffix dropped)
void names_of_rows(FILE *f, std::unordered_set<std::string> &l1, std::unordered_set<std::string> &l2)
{
	char line[1 << 16];
	while (fgets(line, sizeof(line), f)) {
		// awk '{if(substr($4, length($4), 1) == 1) print substr($4, 1, length($4)-2);}' : fields split on blanks
		char *save = nullptr, *tok = strtok_r(line, " \t\n", &save); int k = 1;
		while (tok && k < 4) { tok = strtok_r(nullptr, " \t\n", &save); ++k; }
		if (!tok) continue;
		const size_t l = strlen(tok);
		if (l < 2) continue;
		if (tok[l - 1] == '1') l1.insert(std::string(tok, l - 2)); else if (tok[l - 1] == '2') l2.insert(std::string(tok, l - 2));
	}
}
// seqtk subseq of both files, repair.sh pairing, rename.sh names: sink(which, records, prefix) with which = 0 reads_1, 1 reads_2, 2 singletons
template <class Sink> int subset_pair_rename(const char *fq1, const char *fq2, const std::unordered_set<std::string> &l1, const std::unordered_set<std::string> &l2, Sink sink, int64_t *n_pairs, int64_t *n_single)
{
	std::vector<FqRec> s1, s2;
	if (subseq(fq1, l1, s1) || subseq(fq2, l2, s2)) return -1;
	// repair.sh: pair by name (a trailing /1 or /2 is not part of it); the rest are singletons
	std::unordered_map<std::string, std::vector<size_t>> k2;
	for (size_t i = 0; i < s2.size(); ++i) k2[pair_key(s2[i].name)].push_back(i);
	std::vector<char> used2(s2.size(), 0); std::unordered_map<std::string, size_t> cursor;
	std::vector<const FqRec *> p1, p2, sg;
	for (const FqRec &r : s1) {
		auto it = k2.find(pair_key(r.name)); size_t j = SIZE_MAX;
		if (it != k2.end()) { size_t &cur = cursor[it->first]; while (cur < it->second.size()) { const size_t c = it->second[cur++]; if (!used2[c]) { j = c; break; } } }
		if (j == SIZE_MAX) sg.push_back(&r); else { used2[j] = 1; p1.push_back(&r); p2.push_back(&s2[j]); }
	}
	for (size_t i = 0; i < s2.size(); ++i) if (!used2[i]) sg.push_back(&s2[i]);
	if (sink(0, p1, "realigned") || sink(1, p2, "realigned") || sink(2, sg, "realigned_singleton")) return -3;
	if (n_pairs) *n_pairs = (int64_t)p1.size();
	if (n_single) *n_single = (int64_t)sg.size();
	return 0;
}
} // namespace

// extract_sequence.sh:17-19: rows_fn = the concatenated rows of al_extract_reads (only column 4 is used); writes
// out_dir/reads_1.fastq, reads_2.fastq (pairs, renamed realigned_<n>) and singletons.fastq (realigned_singleton_<n>).
// n_pairs / n_single receive the counts.  Returns 0, negative on error.
extern "C" int al_extract_sequence(const char *fq1, const char *fq2, const char *rows_fn, const char *out_dir, int64_t *n_pairs, int64_t *n_single)
{
	std::unordered_set<std::string> l1, l2;
	{
		FILE *f = strcmp(rows_fn, "-") == 0 ? stdin : fopen(rows_fn, "r");
		if (!f) { fprintf(stderr, "ERROR: failed to open file '%s'\n", rows_fn); return -1; }
		names_of_rows(f, l1, l2);
		if (f != stdin) fclose(f);
	}
	const std::string d = out_dir;
	static const char *const fn[3] = {"/reads_1.fastq", "/reads_2.fastq", "/singletons.fastq"};
	return subset_pair_rename(fq1, fq2, l1, l2, [&](int which, const std::vector<const FqRec *> &v, const char *prefix) { return write_fq(d + fn[which], v, prefix); }, n_pairs, n_single);
}

// N1 fused: extract_reads.sh + extract_sequence.sh in one call, nothing written to disk -- one pass over the BAM, the
// selected names, one scan of each FASTQ file, pairing and renaming; the three FASTQ texts (reads_1, reads_2, singletons) are left in
// anonymous memory files (memfd_create) whose descriptors go to fds[0..2].  A caller maps them by path (/proc/self/fd/<n>): for the
// stream driver that is file bytes in RAM, which it hands to the GPU as they are -- the device finds the records and packs the bases.
// The caller closes the descriptors.  Returns 0, negative on error.
extern "C" int al_extract_to_memory(const char *bam_fn, const char *bed_fn, int read_size, int prune, const char *fq1, const char *fq2, int fds[3], int64_t *n_pairs, int64_t *n_single)
{
	fds[0] = fds[1] = fds[2] = -1;
	char *rows = nullptr; size_t rows_len = 0;
	FILE *rf = open_memstream(&rows, &rows_len);
	if (!rf) return -1;
	const int64_t n = al_extract_reads(bam_fn, bed_fn, read_size, prune, rf);
	if (fflush(rf) == EOF || n < 0) { fclose(rf); free(rows); return n < 0 ? (int)n : -3; }
	fclose(rf);
	std::unordered_set<std::string> l1, l2;
	{ FILE *f = fmemopen(rows, rows_len ? rows_len : 1, "r"); if (!f) { free(rows); return -1; } if (rows_len) names_of_rows(f, l1, l2); fclose(f); }
	free(rows);
	static const char *const nm[3] = {"airlift_reads_1", "airlift_reads_2", "airlift_singletons"};
	const int rc = subset_pair_rename(fq1, fq2, l1, l2, [&](int which, const std::vector<const FqRec *> &v, const char *prefix) -> int {
		const int fd = memfd_create(nm[which], 0);
		if (fd < 0) { perror("[airlift] memfd_create"); return -1; }
		FILE *f = fdopen(dup(fd), "wb");
		if (!f) { close(fd); return -1; }
		setvbuf(f, nullptr, _IOFBF, 8 << 20);
		const int w = write_fq_to(f, v, prefix);
		if (fclose(f) == EOF || w) { close(fd); return -1; }
		fds[which] = fd;
		return 0;
	}, n_pairs, n_single);
	if (rc) for (int i = 0; i < 3; ++i) if (fds[i] >= 0) { close(fds[i]); fds[i] = -1; }
	return rc;
}
