// al_fasta.cpp -- whole-file FASTA loader for the index builders: an uncompressed reference (3.1 GB for a human genome) is
// mapped, cut at its header lines and at line ends into pieces of a few megabytes, and the pieces are measured and then copied
// -- newlines dropped -- by a pool of threads.  Same record grammar as the block reader in al_seqio.h (kseq.h: a record starts
// at a line whose first byte is '>', the name ends at the first white space, sequence lines lose their terminator and one
// trailing '\r', 'u'/'U' become 't'/'T' as bseq.c:72-74 does).  Anything else (gzip, stdin, FASTQ, a line starting with '@' or
// '+') is left to the serial reader: the function returns false and touches nothing.
#include <fcntl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include "al_internal.h"

namespace {
struct Piece { size_t beg, end; uint32_t rec; uint64_t n; uint64_t out; };

template <class F> void par_for(size_t n, int n_threads, F f)
{
	std::atomic<size_t> next(0);
	std::vector<std::thread> th;
	const int nt = (int)std::min<size_t>((size_t)std::max(1, n_threads), n);
	for (int t = 0; t < nt; ++t) th.emplace_back([&]() { for (;;) { const size_t i = next.fetch_add(1); if (i >= n) return; f(i); } });
	for (auto &x : th) x.join();
}
}

// names / lengths / offsets into seq (AlSeq) and the concatenated sequence bytes (ASCII as in the file, terminators removed).
bool al_fasta_load_parallel(const char *fn, int n_threads, std::vector<AlSeq> &seqs, AlText &ascii)
{
	if (!fn || !strcmp(fn, "-") || getenv("AL_SERIAL_PARSE")) return false;
	const int fd = open(fn, O_RDONLY);
	if (fd < 0) return false;
	struct stat sb;
	if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 2) { close(fd); return false; }
	const size_t n = (size_t)sb.st_size;
	const char *d = (const char *)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (d == (const char *)MAP_FAILED) return false;
	(void)madvise((void *)d, n, MADV_WILLNEED);
	bool ok = d[0] == '>';                                                   // (gzip starts with 0x1f 0x8b, FASTQ with '@')
	std::vector<size_t> hdr;                                                 // offsets of the header lines
	if (ok) {
		const size_t CH = 32u << 20, nch = (n + CH - 1) / CH;
		std::vector<std::vector<size_t>> found(nch);
		par_for(nch, n_threads, [&](size_t c) {
			const size_t b = c * CH, e = std::min(n, b + CH);
			for (const char *p = d + b; p < d + e; ) {
				p = (const char *)memchr(p, '>', (size_t)(d + e - p));
				if (!p) break;
				const size_t at = (size_t)(p - d);
				if (at == 0 || d[at - 1] == '\n') found[c].push_back(at);
				++p;
			}
		});
		for (auto &v : found) hdr.insert(hdr.end(), v.begin(), v.end());
		ok = !hdr.empty() && hdr[0] == 0;
	}
	std::vector<Piece> pieces; std::vector<AlSeq> out;
	if (ok) {
		const size_t PC = 8u << 20;
		out.resize(hdr.size());
		for (size_t r = 0; r < hdr.size() && ok; ++r) {
			const size_t h = hdr[r], lim = r + 1 < hdr.size() ? hdr[r + 1] : n;
			const char *nl = (const char *)memchr(d + h, '\n', lim - h);
			const size_t he = nl ? (size_t)(nl - d) : lim;                     // end of the header line
			size_t q = h + 1;
			while (q < he && d[q] != ' ' && d[q] != '\t' && d[q] != '\r' && d[q] != '\v' && d[q] != '\f') ++q;
			out[r].name.assign(d + h + 1, q - (h + 1));
			if (out[r].name.empty()) { ok = false; break; }                     // (an empty name: left to the block reader)
			size_t b = nl ? he + 1 : lim;
			while (b < lim) {                                                   // pieces end after a newline
				size_t e = std::min(lim, b + PC);
				if (e < lim) { const char *x = (const char *)memchr(d + e, '\n', lim - e); e = x ? (size_t)(x - d) + 1 : lim; }
				pieces.push_back(Piece{b, e, (uint32_t)r, 0, 0});
				b = e;
			}
		}
	}
	if (ok) {
		std::atomic<bool> bad(false);
		par_for(pieces.size(), n_threads, [&](size_t i) {
			Piece &p = pieces[i]; uint64_t cnt = 0;
			for (size_t b = p.beg; b < p.end; ) {
				if (d[b] == '@' || d[b] == '+') { bad = true; return; }           // FASTQ grammar: the serial reader decides
				const char *x = (const char *)memchr(d + b, '\n', p.end - b);
				size_t e = x ? (size_t)(x - d) : p.end;
				const size_t nx = x ? e + 1 : p.end;
				if (e > b && d[e - 1] == '\r') --e;
				cnt += e - b; b = nx;
			}
			p.n = cnt;
		});
		ok = !bad;
	}
	if (ok) {
		uint64_t sum = 0; size_t i = 0;
		for (size_t r = 0; r < out.size() && ok; ++r) {
			uint64_t len = 0; out[r].offset = sum;
			for (; i < pieces.size() && pieces[i].rec == r; ++i) { pieces[i].out = sum + len; len += pieces[i].n; }
			if (len >= (1ULL << 32)) ok = false;
			out[r].len = (uint32_t)len; sum += len;
		}
		if (ok) {
			ascii.resize(sum);
			par_for(pieces.size(), n_threads, [&](size_t k) {
				const Piece &p = pieces[k]; char *o = ascii.data() + p.out;
				for (size_t b = p.beg; b < p.end; ) {
					const char *x = (const char *)memchr(d + b, '\n', p.end - b);
					size_t e = x ? (size_t)(x - d) : p.end;
					const size_t nx = x ? e + 1 : p.end;
					if (e > b && d[e - 1] == '\r') --e;
					memcpy(o, d + b, e - b);
					for (char *c = o; c < o + (e - b); ) {                         // bseq.c:72-74 (rare: look for it with memchr)
						char *u = (char *)memchr(c, 'u', (size_t)(o + (e - b) - c)), *U = (char *)memchr(c, 'U', (size_t)(o + (e - b) - c));
						if (!u && !U) break;
						char *f = u && U ? std::min(u, U) : (u ? u : U);
						--*f; c = f + 1;
					}
					o += e - b; b = nx;
				}
			});
			seqs = std::move(out);
		}
	}
	munmap((void *)d, n);
	return ok;
}

// test hook (tests/test_capi_cpu.py): 0 = the parallel loader and the block reader give the same names, lengths and bytes,
// 1 = the parallel loader left the file to the block reader, -1 = they differ
#include "al_seqio.h"
extern "C" int al_dbg_fasta_selftest(const char *fn, int n_threads)
{
	std::vector<AlSeq> ps; AlText pa;
	if (!al_fasta_load_parallel(fn, n_threads, ps, pa)) return 1;
	AlSeqReader rd;
	if (!rd.open(fn)) return -1;
	AlChunk c; size_t i = 0; uint64_t sum = 0;
	for (;; ++i) {
		c.text.clear(); c.recs.clear();
		if (!rd.read(c)) break;
		const AlRec &r = c.recs[0];
		if (i >= ps.size() || ps[i].name != std::string(c.text.data() + r.name) || ps[i].len != r.len || ps[i].offset != sum) return -1;
		if (r.len && memcmp(pa.data() + sum, c.text.data() + r.seq, r.len) != 0) return -1;
		sum += r.len;
	}
	return i == ps.size() && sum == pa.size() ? 0 : -1;
}
