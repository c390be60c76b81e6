// al_seqio.h -- block-buffered FASTA/FASTQ(.gz) record reader shared by the read pipeline and the index builder (product code)
#pragma once
#include <stdint.h>
#include <string.h>
#include <zlib.h>
#include <vector>

struct AlRec { uint32_t name, seq, len, qual; };            // offsets into AlChunk::text; name NUL-terminated; qual = ~0u when absent
struct AlChunk { std::vector<char> text; std::vector<AlRec> recs; };

// kseq.h record grammar (FASTA/FASTQ, multi-line, optional gzip) on a block buffer; newline search by memchr.
class AlSeqReader {
	gzFile fp = nullptr; std::vector<unsigned char> buf; size_t beg = 0, end = 0; bool eof = false; int last = 0;
	bool fill()
	{
		if (eof) return false;
		beg = 0; const int n = gzread(fp, buf.data(), (unsigned)buf.size());
		if (n <= 0) { eof = true; end = 0; return false; }
		end = (size_t)n; return true;
	}
	int getc() { if (beg >= end && !fill()) return -1; return buf[beg++]; }
	// rest of the current line (without the terminator; one trailing '\r' dropped) appended to t.  -1: EOF before any byte.
	int line_append(std::vector<char> &t)
	{
		bool got = false;
		for (;;) {
			if (beg >= end && !fill()) break;
			got = true;
			const unsigned char *nl = (const unsigned char *)memchr(buf.data() + beg, '\n', end - beg);
			const size_t stop = nl ? (size_t)(nl - buf.data()) : end;
			t.insert(t.end(), (const char *)buf.data() + beg, (const char *)buf.data() + stop);
			beg = nl ? stop + 1 : stop;
			if (nl) { if (!t.empty() && t.back() == '\r') t.pop_back(); return 0; }
		}
		return got ? 0 : -1;
	}
public:
	bool open(const char *fn)
	{
		fp = strcmp(fn, "-") == 0 ? gzdopen(0, "r") : gzopen(fn, "r");
		if (!fp) return false;
		gzbuffer(fp, 1 << 20); buf.resize(4 << 20);
		return true;
	}
	~AlSeqReader() { if (fp) gzclose(fp); }
	// continue at a byte offset of an uncompressed file (a record must start there)
	bool seek(long long off) { if (gzseek(fp, (z_off_t)off, SEEK_SET) < 0) return false; beg = end = 0; eof = false; last = 0; return true; }
	// one record appended to c; false at end of input (a FASTQ record whose quality length differs ends the input, kseq.h -2)
	bool read(AlChunk &c)
	{
		int ch;
		if (last == 0) { while ((ch = getc()) >= 0 && ch != '>' && ch != '@') {} if (ch < 0) return false; last = ch; }
		std::vector<char> &t = c.text; const size_t mark = t.size();
		AlRec r; r.name = (uint32_t)t.size();
		while ((ch = getc()) >= 0 && ch != ' ' && ch != '\t' && ch != '\n' && ch != '\r' && ch != '\v' && ch != '\f') t.push_back((char)ch);
		if (ch < 0 && t.size() == mark) return false;
		t.push_back(0);
		if (ch >= 0 && ch != '\n') { const size_t m2 = t.size(); line_append(t); t.resize(m2); }    // comment: dropped (no -y)
		r.seq = (uint32_t)t.size();
		while ((ch = getc()) >= 0 && ch != '>' && ch != '+' && ch != '@') {
			if (ch == '\n') continue;
			t.push_back((char)ch); line_append(t);
		}
		r.len = (uint32_t)(t.size() - r.seq); r.qual = ~0u;
		for (size_t i = r.seq; i < t.size(); ++i) if (t[i] == 'u' || t[i] == 'U') --t[i];      // bseq.c:72-74
		last = (ch == '>' || ch == '@') ? ch : 0;
		if (ch == '+') {
			while ((ch = getc()) >= 0 && ch != '\n') {}
			if (ch < 0) { t.resize(mark); return false; }
			r.qual = (uint32_t)t.size();
			while (line_append(t) >= 0 && t.size() - r.qual < r.len);
			last = 0;
			if (t.size() - r.qual != r.len) { t.resize(mark); return false; }
		}
		c.recs.push_back(r);
		return true;
	}
};
