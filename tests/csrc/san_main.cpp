// Host-side self-tests run under the sanitizers (airlift_amd/csrc/Makefile, target `san`; tests/test_san_cpu.py): the threaded host
// code of the drop-in that needs no GPU -- whole-file parallel FASTA loader, block-parallel FASTQ parser (regular and irregular
// files), ordered multi-lane output (offsets + pwrite, and turns), the SAM formatter, read extraction from a BAM.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/airlift.h"

static std::string g_dir;
static std::string path(const char *n) { return g_dir + "/" + n; }
static void write_file(const std::string &p, const std::string &s) { FILE *f = fopen(p.c_str(), "wb"); fwrite(s.data(), 1, s.size(), f); fclose(f); }

int main(int argc, char **argv)
{
	g_dir = argc > 1 ? argv[1] : "/tmp";
	int bad = 0;
	unsigned long long seed = 12345;
	auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
	// FASTA: several contigs, ragged lines, lower case, N runs
	{
		std::string fa;
		for (int c = 0; c < 7; ++c) {
			fa += ">contig" + std::to_string(c) + " description\n";
			const int len = 1000 + (int)(rnd() % 200000), width = 50 + (int)(rnd() % 40);
			for (int i = 0; i < len; ++i) { fa.push_back("ACGTacgtNn"[rnd() % 10]); if ((i + 1) % width == 0) fa.push_back('\n'); }
			if (fa.back() != '\n') fa.push_back('\n');
		}
		write_file(path("t.fa"), fa);
		for (int t : {1, 3, 8}) { const int r = al_dbg_fasta_selftest(path("t.fa").c_str(), t); if (r != 0) { fprintf(stderr, "FASTA loader self-test with %d threads: %d\n", t, r); ++bad; } }
	}
	// FASTQ: regular (large enough for several 16 MB blocks), then irregular half way
	for (int kind = 0; kind < 3; ++kind) {
		std::string fq; const int n = kind == 0 ? 120000 : 30000;
		for (int i = 0; i < n; ++i) {
			const int L = 100 + (int)(rnd() % 60);
			std::string s, q; for (int j = 0; j < L; ++j) { s.push_back("ACGTN"[rnd() % 5]); q.push_back((char)(33 + rnd() % 40)); }
			const bool odd = kind > 0 && i > n / 2 && i % 1000 == 7;
			if (odd && kind == 1) fq += "@r" + std::to_string(i) + "\n" + s.substr(0, L / 2) + "\n" + s.substr(L / 2) + "\n+\n" + q.substr(0, L / 2) + "\n" + q.substr(L / 2) + "\n";
			else if (odd) fq += "\n\n@r" + std::to_string(i) + " c\r\n" + s + "\r\n+\r\n" + q + "\r\n";
			else fq += "@r" + std::to_string(i) + "\n" + s + "\n+\n" + q + "\n";
		}
		if (kind == 2) fq.pop_back();                 // last record without its newline
		write_file(path("t.fq"), fq);
		for (int t : {2, 8}) { const int r = al_dbg_fastq_selftest(path("t.fq").c_str(), t); if (r != 0) { fprintf(stderr, "FASTQ parser self-test kind %d with %d threads: %d\n", kind, t, r); ++bad; } }
	}
	// multi-process range finding (threads as ranks): variable record lengths, more ranks than records, one and two files
	{
		std::string a, b;
		for (int i = 0; i < 5000; ++i) {
			const int L1 = 30 + (int)(rnd() % 200), L2 = 30 + (int)(rnd() % 200);
			a += "@frag" + std::to_string(i) + "/1\n" + std::string((size_t)L1, 'A') + "\n+\n" + std::string((size_t)L1, '@') + "\n";      // quality lines that start with '@'
			b += "@frag" + std::to_string(i * 7) + "/2 comment\n" + std::string((size_t)L2, 'C') + "\n+\n" + std::string((size_t)L2, 'I') + "\n";
		}
		write_file(path("p_1.fq"), a); write_file(path("p_2.fq"), b); write_file(path("tiny.fq"), "@x\nACGT\n+\nIIII\n@y\nAC\n+\nII\n");
		for (int w : {1, 2, 3, 8, 17}) { const int r = al_dbg_ranked_selftest(path("p_1.fq").c_str(), path("p_2.fq").c_str(), w, g_dir.c_str()); if (r != 0) { fprintf(stderr, "rank range self-test, %d ranks, two files: %d\n", w, r); ++bad; } }
		for (int w : {2, 5}) { const int r = al_dbg_ranked_selftest(path("p_2.fq").c_str(), "", w, g_dir.c_str()); if (r != 0) { fprintf(stderr, "rank range self-test, %d ranks, one file: %d\n", w, r); ++bad; } }
		{ const int r = al_dbg_ranked_selftest(path("tiny.fq").c_str(), path("tiny.fq").c_str(), 6, g_dir.c_str()); if (r != 0) { fprintf(stderr, "rank range self-test, 6 ranks on 2 records: %d\n", r); ++bad; } }
	}
	// the index file (al_mmi.cpp): host-built index of the FASTA above -> dump -> load -> dump again: same size, same key and position counts; a truncated file is an error, not a crash
	{
		al_idxopt_t io; al_mapopt_t mo; al_set_opt(nullptr, &io, &mo); al_set_opt("sr", &io, &mo);
		al_idx_t *mi = al_idx_build(path("t.fa").c_str(), &io, 3);
		if (!mi || al_idx_dump(path("t.mmi").c_str(), mi) != 0) { fprintf(stderr, "index dump failed\n"); ++bad; }
		al_idx_t *m2 = al_idx_is_idx(path("t.mmi").c_str()) > 0 ? al_idx_load(path("t.mmi").c_str()) : nullptr;
		uint64_t a[3] = {0, 0, 0}, b[3] = {1, 1, 1};
		if (mi) al_idx_stat(mi, &a[0], &a[1], &a[2]);
		if (m2) al_idx_stat(m2, &b[0], &b[1], &b[2]);
		if (!m2 || a[0] != b[0] || a[1] != b[1] || a[2] != b[2] || al_idx_dump(path("t2.mmi").c_str(), m2) != 0 || al_idx_is_idx(path("t2.mmi").c_str()) != al_idx_is_idx(path("t.mmi").c_str())) { fprintf(stderr, "index file round trip differs\n"); ++bad; }
		{   // truncated copy
			FILE *f = fopen(path("t.mmi").c_str(), "rb"); std::string buf;
			if (f) { char tmp[65536]; size_t n; while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.append(tmp, n); fclose(f); }
			write_file(path("t3.mmi"), buf.substr(0, buf.size() / 2));
			al_idx_t *m3 = al_idx_load(path("t3.mmi").c_str());
			if (m3) { fprintf(stderr, "al_idx_load accepted a truncated index\n"); ++bad; al_idx_destroy(m3); }
		}
		if (m2) al_idx_destroy(m2);
		if (mi) al_idx_destroy(mi);
	}
	// ordered output of several lanes
	for (int lanes : {1, 2, 5}) for (int off : {0, 1, 2}) { const int r = al_dbg_ordered_out_selftest(path("o.txt").c_str(), lanes, 40, off); if (r != 0) { fprintf(stderr, "ordered output self-test (%d lanes, offsets %d): %d\n", lanes, off, r); ++bad; } }
	// SAM formatter (device routine compiled for the host) against al_write_sam
	{ const int r = al_dbg_sam_selftest(3, 3000); if (r != 0) { fprintf(stderr, "SAM formatter self-test: %d differences\n", r); ++bad; } }
	// read extraction: a corrupt / truncated BAM must come back as an error, not a crash
	{
		write_file(path("bad.bam"), std::string("\x1f\x8b\x08\x04garbage that is not a BGZF block at all", 44));
		write_file(path("r.bed"), "chr1\t10\t2000\n");
		FILE *o = fopen(path("rows.txt").c_str(), "wb");
		const long long r = al_extract_reads(path("bad.bam").c_str(), path("r.bed").c_str(), 150, 1, o);
		fclose(o);
		if (r >= 0) { fprintf(stderr, "al_extract_reads accepted a corrupt BAM (%lld)\n", r); ++bad; }
	}
	fprintf(stderr, "san self-tests: %d failure(s)\n", bad);
	return bad ? 1 : 0;
}
