// al_ranked.cpp -- one process per GPU (BASELINE.json north_star: "reads trivially sharded across the 8 GPUs of one
// node with a RCCL all-gather over xGMI only for the final merged offsets").
//
// Rank r of R maps a contiguous range of the input's fragments and nothing is parsed centrally:
//   1. every rank counts the lines of ITS byte range [r S / R, (r + 1) S / R) of each input file (a few threads, memchr speed);
//      one all-gather of the counts gives every rank the number of lines in front of each range;
//   2. rank r's first record is the first record that starts in its range of file 1; its number K_r follows from the counts, and
//      the byte where record K_r starts in file 2 is found by walking the one range of file 2 that holds line 4 K_r
//      (the reference's reader pairs the two files record by record, bseq.c:129-167); a second all-gather of the starts gives the ends;
//   3. the rank maps its range with the stream driver (al_stream_pipe.cpp) into a part file next to the output;
//   4. one all-gather of {records, bytes} per rank -- the exchange the north star names -- gives every part its offset in the
//      merged output (rank-major = input order, as the reference's serial writer keeps it, map.c:601-644); each rank copies its
//      part into place (copy_file_range), all ranks at once.
// The all-gathers carry 16-32 bytes per rank: RCCL (ncclAllGather over xGMI; communicator from ncclGetUniqueId exchanged through a
// file) when every rank has its own GPU, files in the rendezvous directory otherwise (ranks sharing a GPU, no librccl, AL_NO_RCCL).
// A rank that does not arrive within the timeout makes the others fail with a message and a non-zero status.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>
#include <dlfcn.h>
#include <sys/stat.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include <hip/hip_runtime.h>
#include "al_internal.h"
#include "al_runtime.h"
#include "al_stream_pipe.h"

namespace {

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct ProcExchange {
	virtual ~ProcExchange() {}
	virtual int allgather(const uint64_t *mine, int n_words, uint64_t *all) = 0;   // all: n_words * world; < 0 on timeout / failure
	virtual const char *name() const = 0;
};

// ---- files in a directory every rank sees: round k of rank r is "<dir>/x<k>.<r>" (written under a temporary name, then renamed) ----
// The names carry a run id every rank of one launch shares (AL_RUN_ID, else TORCHELASTIC_RUN_ID / MASTER_PORT, else the parent process id:
// the ranks of a launch are children of one launcher), so that files a crashed earlier run left behind are not taken for this run's.
// A rank removes its files of round k once round k + 1 is complete (everybody has read them by then); the last round's few bytes stay.
static std::string g_nonce;                                // set once the launch's token has been agreed on (al_map_file_frag_ranked): part of every file name below
static std::string run_id_base()
{
	for (const char *k : {"AL_RUN_ID", "TORCHELASTIC_RUN_ID", "MASTER_PORT"}) { const char *v = getenv(k); if (v && *v) { std::string s; for (const char *p = v; *p; ++p) s.push_back((*p >= '0' && *p <= '9') || (*p >= 'a' && *p <= 'z') || (*p >= 'A' && *p <= 'Z') ? *p : '_'); return s; } }
	return "p" + std::to_string((long long)getppid());
}
static std::string run_id() { return g_nonce.empty() ? run_id_base() : run_id_base() + "_" + g_nonce; }
// The launch's token, agreed on without comparing file times with a local clock (ranks reach this point seconds apart -- each after its own
// FASTA load and index build -- and a shared directory's server clock need not be ours).  Every rank r > 0 writes a fresh random word to its
// hello file; rank 0 writes the token file = a fresh token followed by the hello word it has read for every rank, and writes it again whenever a
// hello file changes (a crashed earlier run under the same MASTER_PORT / 'none' run id may have left stale ones); rank r takes the token only
// when the word behind it is its OWN hello word -- a stale token file cannot carry it -- and acknowledges under a name that contains the token;
// rank 0 is done when every rank has acknowledged.  Everything polls with the caller's timeout.
static unsigned long long random_word()
{
	unsigned long long w = (unsigned long long)getpid() * 1000003ULL ^ (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
	FILE *r = fopen("/dev/urandom", "rb"); unsigned long long x = 0; if (r) { if (fread(&x, 8, 1, r) == 1) w ^= x; fclose(r); }
	return w;
}
static int write_small_file(const std::string &fn, const std::string &body)
{
	const std::string tmp = fn + ".tmp"; FILE *f = fopen(tmp.c_str(), "wb");
	if (!f) return -1;
	const bool ok = fwrite(body.data(), 1, body.size(), f) == body.size();
	if (fclose(f) != 0 || !ok || rename(tmp.c_str(), fn.c_str()) != 0) return -1;
	return 0;
}
static std::string read_small_file(const std::string &fn)
{
	std::string s; FILE *f = fopen(fn.c_str(), "rb"); if (!f) return s;
	char buf[4096]; size_t n; while ((n = fread(buf, 1, sizeof(buf), f)) > 0) s.append(buf, n);
	fclose(f); return s;
}
static int agree_on_token(const std::string &dir, int rank, int world, double timeout)
{
	const std::string base = run_id_base(), ft = dir + "/.al_token_" + base;
	auto hello = [&](int r) { return dir + "/.al_hello_" + base + "." + std::to_string(r); };
	auto ack = [&](const std::string &tok, int r) { return dir + "/.al_ack_" + base + "_" + tok + "." + std::to_string(r); };
	char buf[32];
	const double t0 = now_s();
	if (rank == 0) {
		unlink(ft.c_str());
		snprintf(buf, sizeof(buf), "%016llx", random_word());
		const std::string tok = buf;
		std::vector<std::string> seen((size_t)world), cur((size_t)world);
		for (;;) {
			bool all = true, changed = false;
			for (int r = 1; r < world; ++r) {
				const std::string h = read_small_file(hello(r));
				if (h.size() == 16) { cur[r] = h; if (h != seen[r]) changed = true; } else all = false;
			}
			if (all && changed) {
				std::string body = tok; for (int r = 1; r < world; ++r) body += cur[r];
				if (write_small_file(ft, body) != 0) { fprintf(stderr, "[airlift] rank 0: cannot write '%s': %s\n", ft.c_str(), strerror(errno)); return -1; }
				seen = cur;
			}
			if (all) {
				bool acked = true; struct stat sb;
				for (int r = 1; r < world && acked; ++r) if (stat(ack(tok, r).c_str(), &sb) != 0) acked = false;
				if (acked) { for (int r = 1; r < world; ++r) { unlink(ack(tok, r).c_str()); unlink(hello(r).c_str()); } g_nonce = tok; return 0; }
			}
			if (now_s() - t0 > timeout) { fprintf(stderr, "[airlift] rank 0: not every rank of this launch arrived in '%s' within %.0f s\n", dir.c_str(), timeout); return -2; }
			usleep(2000);
		}
	}
	snprintf(buf, sizeof(buf), "%016llx", random_word());
	const std::string mine = buf;
	if (write_small_file(hello(rank), mine) != 0) { fprintf(stderr, "[airlift] rank %d: cannot write '%s': %s\n", rank, hello(rank).c_str(), strerror(errno)); return -1; }
	for (;;) {
		const std::string t = read_small_file(ft);
		if (t.size() == 16 * (size_t)world && t.compare(16 * (size_t)rank, 16, mine) == 0) {
			const std::string tok = t.substr(0, 16);
			if (write_small_file(ack(tok, rank), "1") != 0) { fprintf(stderr, "[airlift] rank %d: cannot acknowledge the token in '%s': %s\n", rank, dir.c_str(), strerror(errno)); return -1; }
			g_nonce = tok; return 0;                                           // (rank 0 removes the hello and acknowledgement files)
		}
		if (now_s() - t0 > timeout) { fprintf(stderr, "[airlift] rank %d: no token of this launch from rank 0 in '%s' within %.0f s\n", rank, dir.c_str(), timeout); return -2; }
		usleep(2000);
	}
}
struct FileExchange : ProcExchange {
	std::string dir, id; int rank, world; double timeout; int round = 0;
	FileExchange(const std::string &d, int r, int w, double t, const std::string &tag = std::string()) : dir(d), id(tag.empty() ? run_id() : tag), rank(r), world(w), timeout(t)
	{ for (int k = 0; k < 64; ++k) unlink(path(k, rank).c_str()); }             // my own leftovers of an earlier run under the same id
	std::string path(int k, int r) const { return dir + "/.al_x" + id + "_" + std::to_string(k) + "." + std::to_string(r); }
	int allgather(const uint64_t *mine, int n_words, uint64_t *all) override
	{
		const int k = round++;
		const std::string fin = path(k, rank), tmp = fin + ".tmp";
		FILE *f = fopen(tmp.c_str(), "wb");
		if (!f || fwrite(mine, 8, (size_t)n_words, f) != (size_t)n_words || fclose(f) != 0 || rename(tmp.c_str(), fin.c_str()) != 0) { fprintf(stderr, "[airlift] rank %d: cannot write '%s': %s\n", rank, fin.c_str(), strerror(errno)); return -1; }
		const double t0 = now_s();
		for (int r = 0; r < world; ++r) {
			const std::string fr = path(k, r);
			for (;;) {
				FILE *g = fopen(fr.c_str(), "rb");
				if (g) { const size_t n = fread(all + (size_t)r * n_words, 8, (size_t)n_words, g); fclose(g); if (n == (size_t)n_words) break; }
				if (now_s() - t0 > timeout) { fprintf(stderr, "[airlift] rank %d: rank %d did not arrive at exchange %d within %.0f s\n", rank, r, k, timeout); return -2; }
				usleep(2000);
			}
		}
		if (k > 0) unlink(path(k - 1, rank).c_str());       // round k complete: every rank has read round k - 1
		return 0;
	}
	void purge() { for (int k = 0; k < round; ++k) unlink(path(k, rank).c_str()); }   // (only behind a barrier on another channel)
	const char *name() const override { return "files in the rendezvous directory"; }
};

// ---- RCCL: one communicator over the ranks' GPUs; the unique id travels through a file ------------------------------------------------
struct RcclProcExchange : ProcExchange {
	enum { kMaxWords = 512 };                          // words per rank and all-gather (the grid's boundary table travels in pieces of this size)
	typedef struct { char internal[128]; } UniqueId;
	typedef int (*get_id_t)(UniqueId *); typedef int (*init_rank_t)(void **, int, UniqueId, int); typedef int (*allgather_t)(const void *, void *, size_t, int, void *, hipStream_t); typedef int (*destroy_t)(void *);
	void *lib = nullptr, *comm = nullptr; allgather_t f_ag = nullptr; destroy_t f_destroy = nullptr;
	hipStream_t st = nullptr; uint64_t *d_send = nullptr, *d_recv = nullptr; int rank, world, device; double timeout; bool ok = false, timed_out = false;
	RcclProcExchange(const std::string &dir, int r, int w, int dev, double t) : rank(r), world(w), device(dev), timeout(t)
	{
		for (const char *nm : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) if ((lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
		if (!lib) return;
		get_id_t f_id = (get_id_t)dlsym(lib, "ncclGetUniqueId"); init_rank_t f_init = (init_rank_t)dlsym(lib, "ncclCommInitRank");
		f_ag = (allgather_t)dlsym(lib, "ncclAllGather"); f_destroy = (destroy_t)dlsym(lib, "ncclCommDestroy");
		if (!f_id || !f_init || !f_ag || !f_destroy || hipSetDevice(device) != hipSuccess) return;
		UniqueId id; memset(&id, 0, sizeof(id));
		const std::string fid = dir + "/.al_rccl_id_" + run_id();
		if (rank == 0) {
			if (f_id(&id) != 0) return;
			const std::string tmp = fid + ".tmp"; FILE *f = fopen(tmp.c_str(), "wb");
			if (!f || fwrite(&id, sizeof(id), 1, f) != 1 || fclose(f) != 0 || rename(tmp.c_str(), fid.c_str()) != 0) return;
		} else {
			const double t0 = now_s();
			for (;;) { FILE *f = fopen(fid.c_str(), "rb"); if (f) { const size_t n = fread(&id, sizeof(id), 1, f); fclose(f); if (n == 1) break; } if (now_s() - t0 > timeout) { fprintf(stderr, "[airlift] rank %d: no RCCL id from rank 0 within %.0f s\n", rank, timeout); return; } usleep(2000); }
		}
		{   // ncclCommInitRank blocks until every rank has joined: a missing peer must not hang this rank for ever
			struct InitState { std::atomic<int> done{0}; void *comm = nullptr; int rc = -1; };
			std::shared_ptr<InitState> is(new InitState());
			const int dev = device, wld = world, rk = rank;
			std::thread th([is, f_init, id, dev, wld, rk]() { if (hipSetDevice(dev) == hipSuccess) is->rc = f_init(&is->comm, wld, id, rk); is->done = 1; });
			const double t0 = now_s();
			while (!is->done.load() && now_s() - t0 <= timeout) usleep(1000);
			if (!is->done.load()) { th.detach(); timed_out = true; fprintf(stderr, "[airlift] rank %d: ncclCommInitRank did not return within %.0f s (a rank is missing)\n", rank, timeout); return; }
			th.join();
			if (is->rc != 0) { comm = nullptr; return; }
			comm = is->comm;
		}
		if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipMalloc((void **)&d_send, 8 * kMaxWords) != hipSuccess || hipMalloc((void **)&d_recv, 8 * kMaxWords * (size_t)world) != hipSuccess) return;
		ok = true;
	}
	~RcclProcExchange() override
	{
		if (timed_out) return;                          // a collective is stuck on the stream: destroying the communicator (or the stream) could hang as well -- the process exits non-zero
		if (d_send) (void)hipFree(d_send); if (d_recv) (void)hipFree(d_recv); if (st) (void)hipStreamDestroy(st);
		if (comm && f_destroy) f_destroy(comm);
	}
	int allgather(const uint64_t *mine, int n_words, uint64_t *all) override
	{
		if (n_words > kMaxWords || hipSetDevice(device) != hipSuccess) return -1;
		if (hipMemcpyAsync(d_send, mine, 8 * (size_t)n_words, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
		if (f_ag(d_send, d_recv, (size_t)n_words, 5 /* ncclUint64 */, comm, st) != 0) return -1;
		if (hipMemcpyAsync(all, d_recv, 8 * (size_t)n_words * (size_t)world, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
		const double t0 = now_s();                     // a peer that never joins the collective must not hang this rank for ever
		for (;;) { const hipError_t e = hipStreamQuery(st); if (e == hipSuccess) return 0; if (e != hipErrorNotReady) return -1; if (now_s() - t0 > timeout) { timed_out = true; fprintf(stderr, "[airlift] rank %d: the RCCL all-gather did not complete within %.0f s (a rank is missing)\n", rank, timeout); return -2; } usleep(200); }
	}
	const char *name() const override { return "RCCL ncclAllGather"; }
};

// ---- lines of a byte range, record starts ----------------------------------------------------------------------------------------
struct FileMap { int fd = -1; long long size = 0; bool open(const char *fn) { fd = ::open(fn, O_RDONLY); struct stat sb; if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return false; size = (long long)sb.st_size; return true; } ~FileMap() { if (fd >= 0) close(fd); } };

long long count_newlines(int fd, long long lo, long long hi, int n_threads)
{
	if (hi <= lo) return 0;
	std::atomic<long long> total{0}; std::atomic<bool> bad{false};
	al_parallel_for(std::max(1, n_threads), (size_t)(hi - lo), [&](size_t a, size_t b, int) {
		std::vector<char> buf(4 << 20); long long c = 0, off = lo + (long long)a; const long long end = lo + (long long)b;
		while (off < end) {
			const ssize_t k = pread(fd, buf.data(), (size_t)std::min<long long>((long long)buf.size(), end - off), (off_t)off);
			if (k <= 0) { bad = true; break; }
			const char *p = buf.data(), *e = p + k;
			while ((p = (const char *)memchr(p, '\n', (size_t)(e - p))) != nullptr) { ++c; ++p; }
			off += k;
		}
		total += c;
	});
	return bad ? -1 : total.load();
}
// offset just behind the n-th newline (n >= 1) at or after `from`; -1 if the file ends first
long long after_nth_newline(int fd, long long from, long long size, long long n)
{
	std::vector<char> buf(4 << 20); long long off = from;
	while (off < size) {
		const ssize_t k = pread(fd, buf.data(), (size_t)std::min<long long>((long long)buf.size(), size - off), (off_t)off);
		if (k <= 0) return -1;
		const char *p = buf.data(), *e = p + k;
		while ((p = (const char *)memchr(p, '\n', (size_t)(e - p))) != nullptr) { ++p; if (--n == 0) return off + (p - buf.data()); }
		off += k;
	}
	return -1;
}

struct RankRange { long long start[2] = {0, 0}, end[2] = {-1, -1}; long long first_record = 0; };

// the byte ranges rank r maps.  Line counts, not a record grammar, decide the cuts: rank r starts at the first line of file 1 that begins in
// its byte share and whose number is a multiple of four (strict four-line FASTQ: the stream driver checks the grammar of every record
// it takes and refuses the input otherwise).
int find_ranges(const char *const *fn, int n_fn, int rank, int world, int n_threads, ProcExchange &ex, RankRange *out)
{
	// A rank that fails locally (a file it cannot open or read) still takes part in the exchanges, with a poison word: its peers fail at once
	// instead of waiting for the timeout.
	const uint64_t POISON = ~0ULL;
	FileMap f[2]; bool bad = false;
	for (int i = 0; i < n_fn; ++i) if (!f[i].open(fn[i])) { fprintf(stderr, "ERROR: failed to open file '%s'\n", fn[i]); bad = true; }
	auto share = [&](int i, int r) -> long long { return f[i].size * (long long)r / (long long)world; };
	// (1) lines of my share of every file
	uint64_t mine[2] = {0, 0}; std::vector<uint64_t> all(2 * (size_t)world);
	for (int i = 0; i < n_fn && !bad; ++i) { const long long c = count_newlines(f[i].fd, share(i, rank), share(i, rank + 1), n_threads); if (c < 0) bad = true; else mine[i] = (uint64_t)c; }
	if (bad) mine[0] = mine[1] = POISON;
	if (ex.allgather(mine, 2, all.data())) return -2;
	for (int r = 0; r < world; ++r) if (all[2 * (size_t)r] == POISON) { if (!bad) fprintf(stderr, "[airlift] rank %d: rank %d could not read its share of the input\n", rank, r); return -1; }
	std::vector<long long> before[2];                  // newlines in front of each share
	for (int i = 0; i < n_fn; ++i) { before[i].assign((size_t)world + 1, 0); for (int r = 0; r < world; ++r) before[i][(size_t)r + 1] = before[i][(size_t)r] + (long long)all[2 * (size_t)r + (size_t)i]; }
	// (2) my first record: the first line start at or behind my share's first byte whose line number is a multiple of 4
	long long start0, line0;                           // line0 = number of the line that starts at start0
	if (rank == 0) { start0 = 0; line0 = 0; }
	else {
		const long long s = share(0, rank);
		// the line that contains byte s - 1 ends at the first newline at or after s - 1; lines in front of byte s - 1 ... count from `before`, corrected for the byte s - 1 itself
		// (s == 0: a file shorter than the number of ranks -- byte 0 is rank 0's, this rank looks for a later record start like any other)
		char prev = 0; bool rd_bad = s > 0 && pread(f[0].fd, &prev, 1, (off_t)(s - 1)) != 1;
		if (rd_bad) bad = true;
		long long pos, ln;
		if (prev == '\n') { pos = s; ln = before[0][(size_t)rank]; }                                  // a line starts exactly at s
		else { pos = after_nth_newline(f[0].fd, s, f[0].size, 1); ln = before[0][(size_t)rank] + 1; }  // behind the line that straddles s
		if (pos >= 0 && (ln & 3) != 0) { pos = after_nth_newline(f[0].fd, pos, f[0].size, 4 - (ln & 3)); ln += 4 - (ln & 3); }
		if (pos < 0) { pos = f[0].size; ln = before[0][(size_t)world]; ln = (ln + 3) / 4 * 4; }        // nothing starts in my share
		start0 = pos; line0 = ln;
	}
	const long long K = line0 / 4;
	out->first_record = K; out->start[0] = start0; out->start[1] = 0;
	if (n_fn == 2) {   // the byte where record K starts in file 2 = behind newline number 4 K
		const long long want = 4 * K;
		if (want == 0) out->start[1] = 0;
		else if (want > before[1][(size_t)world]) out->start[1] = f[1].size;
		else {
			int q = 0; while (q + 1 < world && before[1][(size_t)q + 1] < want) ++q;                     // share q holds newline number `want`
			const long long pos = after_nth_newline(f[1].fd, share(1, q), f[1].size, want - before[1][(size_t)q]);
			out->start[1] = pos < 0 ? f[1].size : pos;
		}
	}
	// (3) everybody's starts -> my ends
	uint64_t st[2] = {(uint64_t)out->start[0], (uint64_t)out->start[1]};
	if (bad) st[0] = st[1] = POISON;
	if (ex.allgather(st, 2, all.data())) return -2;
	for (int r = 0; r < world; ++r) if (all[2 * (size_t)r] == POISON) { if (!bad) fprintf(stderr, "[airlift] rank %d: rank %d could not find its first record\n", rank, r); return -1; }
	for (int i = 0; i < n_fn; ++i) out->end[i] = rank + 1 < world ? (long long)all[2 * (size_t)(rank + 1) + (size_t)i] : f[i].size;
	for (int i = 0; i < n_fn; ++i) if (out->end[i] < out->start[i]) out->end[i] = out->start[i];
	return 0;
}

// ---- (round 6) the common grid of batches of a multi-process SAM run --------------------------------------------------------------------------
// The input's records are cut into batches of G records; batch k is mapped by rank k mod R, so that every ROUND of R batches is a contiguous stretch
// of the input and its text can be written in input order after ONE all-gather of the R sizes (the north star's exchange; the reference's writer keeps
// input order by being serial, map.c:601-644).  B[i][k] = the byte where record k G starts in file i (B[i][n_grid] = the file's size): every rank finds
// the boundaries whose line starts lie in ITS byte share in one pass over that share, the table is put together by all-gathers.
struct GridPlan { uint64_t n_grid = 0, G = 0, records = 0; std::vector<long long> B[2]; };
int find_grid(const char *const *fn, int n_fn, int rank, int world, int n_threads, ProcExchange &ex, GridPlan *gp)
{
	const uint64_t POISON = ~0ULL;
	FileMap f[2]; bool bad = false;
	for (int i = 0; i < n_fn; ++i) if (!f[i].open(fn[i])) { fprintf(stderr, "ERROR: failed to open file '%s'\n", fn[i]); bad = true; }
	auto share = [&](int i, int r) -> long long { return f[i].size * (long long)r / (long long)world; };
	uint64_t mine[2] = {0, 0}; std::vector<uint64_t> all(2 * (size_t)world);
	for (int i = 0; i < n_fn && !bad; ++i) { const long long c = count_newlines(f[i].fd, share(i, rank), share(i, rank + 1), n_threads); if (c < 0) bad = true; else mine[i] = (uint64_t)c; }
	if (!bad) for (int i = 0; i < n_fn; ++i) if (rank == world - 1 && f[i].size > 0) { char c = 0; if (pread(f[i].fd, &c, 1, (off_t)(f[i].size - 1)) == 1 && c != '\n') ++mine[i]; }   // (an unterminated last line still is one)
	if (bad) mine[0] = mine[1] = POISON;
	if (ex.allgather(mine, 2, all.data())) return -2;
	for (int r = 0; r < world; ++r) if (all[2 * (size_t)r] == POISON) { if (!bad) fprintf(stderr, "[airlift] rank %d: rank %d could not read its share of the input\n", rank, r); return -1; }
	std::vector<long long> before[2];
	for (int i = 0; i < n_fn; ++i) { before[i].assign((size_t)world + 1, 0); for (int r = 0; r < world; ++r) before[i][(size_t)r + 1] = before[i][(size_t)r] + (long long)all[2 * (size_t)r + (size_t)i]; }
	const uint64_t R = (uint64_t)(before[0][(size_t)world] / 4);
	bool shape_ok = before[0][(size_t)world] % 4 == 0 && (n_fn == 1 || before[1][(size_t)world] == before[0][(size_t)world]);
	if (!shape_ok) { if (rank == 0) fprintf(stderr, "[airlift] the input is not whole four-line FASTQ records (%lld lines%s): not supported in a multi-process run\n", before[0][(size_t)world], n_fn == 2 ? ", or the two files differ in their line counts" : ""); return -1; }
	// batch size: what a long input's batches hold in a single process (262144 pairs), less for a short input so that every rank has a few batches; AL_RANK_BATCH (records) overrides
	uint64_t G = n_fn == 2 ? 262144u : 524288u;
	{ const uint64_t per = (R + 4ULL * (uint64_t)world - 1) / (4ULL * (uint64_t)world); if (per < G) G = per ? per : 1; }
	if (getenv("AL_RANK_BATCH") && atoll(getenv("AL_RANK_BATCH")) > 0) G = (uint64_t)atoll(getenv("AL_RANK_BATCH"));
	if (n_fn == 1 && (G & 1)) ++G;                                   // (one interleaved file: a pair's two records stay in one batch)
	gp->G = G; gp->records = R; gp->n_grid = (R + G - 1) / G;
	const uint64_t ng = gp->n_grid;
	for (int i = 0; i < n_fn; ++i) gp->B[i].assign((size_t)ng + 1, 0);
	if (ng == 0) return 0;
	// the boundaries inside my share: record k G starts behind newline number 4 k G; one forward pass
	std::vector<uint64_t> tab((size_t)n_fn * (size_t)(ng + 1), 0);
	for (int i = 0; i < n_fn && !bad; ++i) {
		// one forward pass over my share with one buffer: the newlines are counted on from the share's first byte, and wherever the count reaches 4 k G the next byte is a boundary
		const long long lo = before[i][(size_t)rank], hi = before[i][(size_t)rank + 1];
		uint64_t k = (uint64_t)(lo / (long long)(4 * G)) + 1;                 // first boundary whose newline number is above lo
		if (k < 1) k = 1;
		if (k >= ng || (long long)(4 * k * G) > hi) continue;
		std::vector<char> buf(4 << 20); long long off = share(i, rank), cnt = lo; const long long end = f[i].size;
		while (off < end && k < ng && (long long)(4 * k * G) <= hi) {
			const ssize_t got = pread(f[i].fd, buf.data(), (size_t)std::min<long long>((long long)buf.size(), end - off), (off_t)off);
			if (got <= 0) { bad = true; break; }
			const char *p = buf.data(), *e = p + got;
			while (k < ng && (long long)(4 * k * G) <= hi && (p = (const char *)memchr(p, '\n', (size_t)(e - p))) != nullptr) {
				++p; ++cnt;
				if (cnt == (long long)(4 * k * G)) { tab[(size_t)i * (size_t)(ng + 1) + (size_t)k] = (uint64_t)(off + (p - buf.data())); ++k; }
			}
			off += got;
		}
		if (!bad && k < ng && (long long)(4 * k * G) <= hi) bad = true;             // (the share ended before its last boundary: the counts and the file disagree)
	}
	// all-gather of the table, in pieces (every entry is written by exactly one rank: the others hold 0)
	const size_t nw = tab.size(), PW = 256;
	std::vector<uint64_t> piece(PW + 1), got((PW + 1) * (size_t)world);
	for (size_t a = 0; a < nw; a += PW) {
		const size_t n = std::min(PW, nw - a);
		for (size_t j = 0; j < n; ++j) piece[j] = tab[a + j];
		piece[n] = bad ? POISON : 0;
		if (ex.allgather(piece.data(), (int)(n + 1), got.data())) return -2;
		for (int r = 0; r < world; ++r) { if (got[(size_t)r * (n + 1) + n] == POISON) { if (!bad) fprintf(stderr, "[airlift] rank %d: rank %d could not find its batch boundaries\n", rank, r); return -1; }
		                                  for (size_t j = 0; j < n; ++j) if (got[(size_t)r * (n + 1) + j]) tab[a + j] = got[(size_t)r * (n + 1) + j]; }
	}
	for (int i = 0; i < n_fn; ++i) { for (uint64_t k = 1; k < ng; ++k) gp->B[i][(size_t)k] = (long long)tab[(size_t)i * (size_t)(ng + 1) + (size_t)k]; gp->B[i][0] = 0; gp->B[i][(size_t)ng] = f[i].size; }
	for (int i = 0; i < n_fn; ++i) for (uint64_t k = 0; k < ng; ++k) if (gp->B[i][(size_t)k + 1] <= gp->B[i][(size_t)k]) { if (rank == 0) fprintf(stderr, "[airlift] batch boundaries of '%s' are not ascending (irregular input?)\n", fn[i]); return -1; }
	return 0;
}
// the sink of the stream driver (AlStreamBatchSink): one all-gather of {ok, bytes, bytes already at the start of the file} per round
struct RankSink { ProcExchange *ex; int rank, world; long long base = 0, first_off = -1; uint64_t bytes_total = 0, rounds = 0; bool failed = false; };
long long rank_sink_offset(void *ctx, uint64_t round, uint64_t bytes, uint64_t pre_bytes, int ok)
{
	RankSink *S = (RankSink *)ctx; (void)round;
	uint64_t mine[3] = {ok ? 1ULL : 0ULL, bytes, pre_bytes}; std::vector<uint64_t> all(3 * (size_t)S->world);
	++S->rounds;
	if (S->ex->allgather(mine, 3, all.data())) { S->failed = true; return -2; }
	long long off = S->base, tot = 0; bool all_ok = true;
	for (int r = 0; r < S->world; ++r) { if (!all[3 * (size_t)r]) all_ok = false; off += (long long)all[3 * (size_t)r + 2]; tot += (long long)all[3 * (size_t)r + 2]; }
	for (int r = 0; r < S->world; ++r) { if (r < S->rank) off += (long long)all[3 * (size_t)r + 1]; tot += (long long)all[3 * (size_t)r + 1]; }
	S->base += tot;
	if (!all_ok) { if (ok) fprintf(stderr, "[airlift] rank %d: another rank failed; the output is incomplete\n", S->rank); S->failed = true; return -1; }
	if (bytes && S->first_off < 0) S->first_off = off;
	S->bytes_total += bytes;
	return off;
}

int copy_into(int out_fd, long long off, const char *part_path, long long n)
{
	const int in = open(part_path, O_RDONLY);
	if (in < 0) return -1;
	long long done = 0; off_t in_off = 0, o = (off_t)off; int rc = 0;
	while (done < n) {
		ssize_t k = copy_file_range(in, &in_off, out_fd, &o, (size_t)std::min<long long>(n - done, 1LL << 30), 0);
		if (k < 0 && (errno == EXDEV || errno == EINVAL || errno == ENOSYS || errno == EOPNOTSUPP)) {   // not on this pair of file systems: read + pwrite
			std::vector<char> buf(8 << 20);
			while (done < n) { const ssize_t r = pread(in, buf.data(), (size_t)std::min<long long>((long long)buf.size(), n - done), (off_t)done); if (r <= 0) { rc = -1; break; } ssize_t w = 0; while (w < r) { const ssize_t x = pwrite(out_fd, buf.data() + w, (size_t)(r - w), (off_t)(off + done + w)); if (x <= 0) { rc = -1; break; } w += x; } if (rc) break; done += r; }
			break;
		}
		if (k <= 0) { rc = -1; break; }
		done += k;
	}
	close(in);
	return rc;
}

} // namespace

// One rank of a multi-process run.  out_path: the merged SAM file (rank 0 creates it); rendezvous: a directory every rank sees
// (default: the output's directory); device < 0: LOCAL_RANK or rank.  Returns 0, negative on error (message on stderr).
static int map_ranked(const al_idx_t *mi, int n_fn, const char **fn, const al_mapopt_t *opt, int n_threads, const char *out_path, const char *rg,
                      int device, int rank, int world, const char *rendezvous, double timeout_s, int bam, int bam_level);
extern "C" int al_map_file_frag_ranked(const al_idx_t *mi, int n_fn, const char **fn, const al_mapopt_t *opt, int n_threads, const char *out_path, const char *rg,
                                       int device, int rank, int world, const char *rendezvous, double timeout_s)
{
	return map_ranked(mi, n_fn, fn, opt, n_threads, out_path, rg, device, rank, world, rendezvous, timeout_s, 0, 0);
}
// The same with unsorted BAM output (the consumer of AirLift's realign step keeps BAM: 0-align_reads.sh:13): every rank deflates its own records into
// whole BGZF blocks -- no compressed stream crosses processes --, rank 0's part starts with the header, the last rank's ends with the EOF block, and
// the parts go behind each other at the exchanged offsets like the SAM parts.
extern "C" int al_map_file_frag_ranked_bam(const al_idx_t *mi, int n_fn, const char **fn, const al_mapopt_t *opt, int n_threads, const char *out_path, const char *rg,
                                           int device, int rank, int world, const char *rendezvous, double timeout_s, int bam_level)
{
	return map_ranked(mi, n_fn, fn, opt, n_threads, out_path, rg, device, rank, world, rendezvous, timeout_s, 1, bam_level);
}
static int map_ranked(const al_idx_t *mi, int n_fn, const char **fn, const al_mapopt_t *opt, int n_threads, const char *out_path, const char *rg,
                      int device, int rank, int world, const char *rendezvous, double timeout_s, int bam, int bam_level)
{
	if (!mi || !fn || !out_path || n_fn < 1 || n_fn > 2 || world < 1 || rank < 0 || rank >= world) return -1;
	if (timeout_s <= 0) timeout_s = getenv("AL_RANK_TIMEOUT") ? atof(getenv("AL_RANK_TIMEOUT")) : 600.0;
	std::string dir = rendezvous && *rendezvous ? rendezvous : std::string(out_path);
	if (!(rendezvous && *rendezvous)) { const size_t sl = dir.rfind('/'); dir = sl == std::string::npos ? "." : dir.substr(0, sl ? sl : 1); }
	const bool timing = getenv("AL_TIMING") != nullptr;
	int n_dev = 0; (void)hipGetDeviceCount(&n_dev);
	if (device < 0) { const char *lr = getenv("LOCAL_RANK"); device = lr ? atoi(lr) : rank; if (n_dev > 0) device %= n_dev; }
	// the launch's token (see agree_on_token), then the exchange: RCCL when the ranks sit on distinct GPUs, files otherwise
	if (world > 1 && !getenv("AL_RUN_ID")) { const int e = agree_on_token(dir, rank, world, timeout_s); if (e) return e; }   // (AL_RUN_ID: the caller vouches for a fresh id)
	std::unique_ptr<FileExchange> fex(new FileExchange(dir, rank, world, timeout_s)); std::unique_ptr<RcclProcExchange> rx;
	ProcExchange *ex = fex.get();
	if (world > 1 && !getenv("AL_NO_RCCL") && n_dev >= world) {
		// every rank tells its device first (through the files): a communicator cannot hold one GPU twice
		uint64_t d = (uint64_t)device; std::vector<uint64_t> all((size_t)world);
		if (ex->allgather(&d, 1, all.data())) return -2;
		bool distinct = true; for (int i = 0; i < world; ++i) for (int j = 0; j < i; ++j) if (all[(size_t)i] == all[(size_t)j]) distinct = false;
		if (distinct) {
			rx.reset(new RcclProcExchange(dir, rank, world, device, timeout_s));
			uint64_t okw = rx->ok ? 1 : 0; if (ex->allgather(&okw, 1, all.data())) return -2;
			bool all_ok = true; for (int i = 0; i < world; ++i) if (!all[(size_t)i]) all_ok = false;
			if (all_ok) ex = rx.get();
		}
	}
	if (timing && rank == 0) fprintf(stderr, "[airlift] %d ranks, exchanges by: %s\n", world, ex->name());
	const double t0 = now_s();
	if (!bam) {
		// (round 6) SAM: no part files and no copy at the end -- the ranks share one grid of batches, a round of `world` batches is a contiguous stretch of the
		// input, and after one all-gather of the round's sizes every rank writes its batch's text into the ONE output file at its offset (pwrite)
		GridPlan gp;
		{ const int e = find_grid(fn, n_fn, rank, world, std::max(1, n_threads / 2), *ex, &gp); if (e) return e; }
		std::vector<long long> lo[2], hi[2];
		for (uint64_t k = (uint64_t)rank; k < gp.n_grid; k += (uint64_t)world) for (int i = 0; i < n_fn; ++i) { lo[i].push_back(gp.B[i][(size_t)k]); hi[i].push_back(gp.B[i][(size_t)k + 1]); }
		if (timing) fprintf(stderr, "[airlift] rank %d of %d: %llu records in %llu batches of %llu, %zu of them this rank's; boundaries found in %.3f s\n", rank, world, (unsigned long long)gp.records, (unsigned long long)gp.n_grid, (unsigned long long)gp.G, lo[0].size(), now_s() - t0);
		// rank 0 creates the file; the others open it once it exists (an all-gather in between)
		FILE *pf = nullptr; uint64_t okw = 1; std::vector<uint64_t> allw((size_t)world);
		if (rank == 0) { pf = fopen(out_path, "wb"); if (!pf) { fprintf(stderr, "[airlift] rank 0: cannot create '%s': %s\n", out_path, strerror(errno)); okw = 0; } }
		if (ex->allgather(&okw, 1, allw.data())) { if (pf) fclose(pf); return -2; }
		if (!allw[0]) { if (pf) fclose(pf); return -3; }
		if (rank != 0) { pf = fopen(out_path, "r+b"); if (!pf) { fprintf(stderr, "[airlift] rank %d: cannot open '%s': %s\n", rank, out_path, strerror(errno)); } }
		RankSink rsk; rsk.ex = ex; rsk.rank = rank; rsk.world = world;
		AlStreamBatchSink sink; sink.offset_of = rank_sink_offset; sink.ctx = &rsk; sink.n_rounds = (gp.n_grid + (uint64_t)world - 1) / (uint64_t)world;
		if (gp.n_grid == 0) sink.n_rounds = 1;                                   // (an empty input: one round places the header)
		AlStreamRange range; range.header = rank == 0; range.list = true; range.n_ranges = (int)lo[0].size(); range.sink = &sink;
		for (int i = 0; i < n_fn; ++i) { range.rstart[i] = lo[i].data(); range.rend[i] = hi[i].data(); }
		AlStreamResume rs;
		int rc = pf ? al_stream_map_files(mi, n_fn, fn, opt, n_threads, pf, rg, &device, 1, &rs, &range) : -3;
		if (rc == AL_STREAM_NA) { fprintf(stderr, "[airlift] rank %d: a multi-process run takes plain (uncompressed, four-line) FASTQ files only\n", rank); rc = -1; }
		if (rc != 0 && !rsk.failed && rsk.rounds < sink.n_rounds) (void)rank_sink_offset(&rsk, rsk.rounds, 0, 0, 0);   // (the driver did not get as far as its rounds: the peers must still learn)
		if (pf && fclose(pf) != 0 && rc == 0) rc = -3;
		if (rc == 0 && rsk.failed) rc = -4;
		uint64_t fin = rc == 0 ? 1 : 0; std::vector<uint64_t> allf((size_t)world);
		if (!rsk.failed && ex->allgather(&fin, 1, allf.data()) == 0) { for (int r = 0; r < world; ++r) if (!allf[(size_t)r] && rc == 0) rc = -4; }
		if (ex != fex.get()) fex->purge();
		if (rank == 0) { unlink((dir + "/.al_rccl_id_" + run_id()).c_str()); unlink((dir + "/.al_token_" + run_id_base()).c_str()); if (rc != 0) unlink(out_path); }
		if (timing) fprintf(stderr, "[airlift] rank %d: %llu bytes in %llu rounds, its first batch's bytes at offset %lld of the merged output; total %.3f s\n", rank, (unsigned long long)rsk.bytes_total, (unsigned long long)rsk.rounds, rsk.first_off, now_s() - t0);
		return rc;
	}
	RankRange rr;
	{ const int e = find_ranges(fn, n_fn, rank, world, std::max(1, n_threads / 2), *ex, &rr); if (e) return e; }   // (every rank leaves here together: local failures travel as a poison word)
	if (timing) fprintf(stderr, "[airlift] rank %d of %d: records from %lld; bytes [%lld, %lld) of '%s'%s found in %.3f s\n", rank, world, rr.first_record, rr.start[0], rr.end[0], fn[0], n_fn == 2 ? " (and the matching range of the second file)" : "", now_s() - t0);
	// map my range into a part file
	// (rank 0's offset in the merged file is known: it writes there directly, no part file and no copy; the others learn theirs from the final exchange)
	const std::string part = rank == 0 ? std::string(out_path) : std::string(out_path) + ".part" + std::to_string(rank);
	FILE *pf = fopen(part.c_str(), "wb");
	AlStreamRange range; for (int i = 0; i < 2; ++i) { range.start[i] = rr.start[i]; range.end[i] = rr.end[i]; } range.header = rank == 0;
	AlStreamResume rs;
	int rc;
	if (!pf) { fprintf(stderr, "[airlift] rank %d: cannot create '%s': %s\n", rank, part.c_str(), strerror(errno)); rc = -3; }   // (still joins the exchange below, with ok = 0)
	else if (bam) rc = al_map_file_frag_bam_part(mi, n_fn, fn, opt, n_threads, pf, rg, device, bam_level, rr.start, rr.end, rank == 0, rank == world - 1);
	else rc = al_stream_map_files(mi, n_fn, fn, opt, n_threads, pf, rg, &device, 1, &rs, &range);
	if (rc == AL_STREAM_NA) { fprintf(stderr, "[airlift] rank %d: a multi-process run takes plain (uncompressed, four-line) FASTQ files only\n", rank); rc = -1; }
	if (rc == 0 && rs.resume) { fprintf(stderr, "[airlift] rank %d: the input is not strict four-line FASTQ at byte %lld: not supported in a multi-process run\n", rank, rs.off[0]); rc = -1; }
	if (pf && fflush(pf) == EOF) rc = rc ? rc : -3;
	const long long my_bytes = rc == 0 && pf ? (long long)ftello(pf) : 0;
	if (pf) fclose(pf);
	// the exchange of the north star: {ok, bytes} of every part -> offsets
	uint64_t mine[2] = {rc == 0 ? 1ULL : 0ULL, (uint64_t)my_bytes}; std::vector<uint64_t> all(2 * (size_t)world);
	int out_fd = -1;
	if (ex->allgather(mine, 2, all.data())) { if (rank != 0) unlink(part.c_str()); return -2; }
	long long off = 0; bool all_ok = true;
	for (int r = 0; r < world; ++r) { if (!all[2 * (size_t)r]) all_ok = false; if (r < rank) off += (long long)all[2 * (size_t)r + 1]; }
	if (all_ok) {
		if (rank != 0) {
			out_fd = open(out_path, O_WRONLY);
			if (out_fd < 0 || copy_into(out_fd, off, part.c_str(), my_bytes)) { fprintf(stderr, "[airlift] rank %d: writing its %lld bytes at offset %lld of '%s' failed\n", rank, my_bytes, off, out_path); rc = -3; }
		}
	} else if (rc == 0) { fprintf(stderr, "[airlift] rank %d: another rank failed; no output\n", rank); rc = -4; }
	if (out_fd >= 0) close(out_fd);
	if (rank != 0) unlink(part.c_str()); else if (!all_ok) unlink(out_path);
	// everybody is done with the exchange files
	uint64_t fin = rc == 0 ? 1 : 0;
	if (ex->allgather(&fin, 1, all.data()) == 0) { for (int r = 0; r < world; ++r) if (!all[(size_t)r] && rc == 0) rc = -4; }
	if (ex != fex.get()) fex->purge();                 // (the RCCL all-gather above was a barrier: nobody reads the files any more)
	if (rank == 0) { unlink((dir + "/.al_rccl_id_" + run_id()).c_str()); unlink((dir + "/.al_token_" + run_id_base()).c_str()); }
	if (timing) fprintf(stderr, "[airlift] rank %d: %lld bytes at offset %lld of the merged output; total %.3f s\n", rank, my_bytes, off, now_s() - t0);
	return rc;
}

// Self-test of the range finding without a GPU: `world` threads act as the ranks (file exchange in `dir`); the ranges must tile both
// files, start at records, and pair record for record.  Returns 0, or a negative code.
extern "C" int al_dbg_ranked_selftest(const char *fn1, const char *fn2, int world, const char *dir)
{
	const int n_fn = fn2 && *fn2 ? 2 : 1; const char *fn[2] = {fn1, fn2};
	std::vector<RankRange> rr((size_t)world); std::vector<int> rcs((size_t)world, 0); std::vector<std::thread> th;
	for (int r = 0; r < world; ++r) th.emplace_back([&, r]() { FileExchange ex(dir, r, world, 60.0, "selftest"); rcs[(size_t)r] = find_ranges(fn, n_fn, r, world, 2, ex, &rr[(size_t)r]); });
	for (auto &t : th) t.join();
	for (int r = 0; r < world; ++r) if (rcs[(size_t)r]) return -1;
	for (int i = 0; i < n_fn; ++i) {
		FileMap f; if (!f.open(fn[i])) return -2;
		if (rr[0].start[i] != 0 || rr[(size_t)world - 1].end[i] != f.size) return -3;
		for (int r = 0; r + 1 < world; ++r) if (rr[(size_t)r].end[i] != rr[(size_t)r + 1].start[i]) return -4;
		for (int r = 0; r < world; ++r) {
			const long long s = rr[(size_t)r].start[i];
			if (s < f.size) { char c[2] = {0, 0}; if (pread(f.fd, c, 1, (off_t)s) != 1 || c[0] != '@') return -5; if (s > 0 && (pread(f.fd, c + 1, 1, (off_t)(s - 1)) != 1 || c[1] != '\n')) return -5; }
			if (count_newlines(f.fd, 0, s, 2) != 4 * rr[(size_t)r].first_record && s < f.size) return -6;     // both files: record first_record starts here
		}
	}
	// (round 6) the grid of batches of the SAM path: the same table on every rank, every boundary at the record it names
	static std::atomic<int> call_no{0}; const std::string call_tag = "c" + std::to_string(call_no.fetch_add(1));   // (a directory reused by the caller: no call reads another's files)
	for (const char *gb : {"1", "3", "1000"}) {
		setenv("AL_RANK_BATCH", gb, 1);
		std::vector<GridPlan> gp((size_t)world); std::vector<int> rg((size_t)world, 0); std::vector<std::thread> tg;
		for (int r = 0; r < world; ++r) tg.emplace_back([&, r]() { FileExchange ex(dir, r, world, 60.0, std::string("selfgrid") + gb + call_tag); rg[(size_t)r] = find_grid(fn, n_fn, r, world, 2, ex, &gp[(size_t)r]); });   // (no purge: a rank's last files may still be unread by a slower one)
		for (auto &t : tg) t.join();
		unsetenv("AL_RANK_BATCH");
		for (int r = 0; r < world; ++r) if (rg[(size_t)r]) return -7;
		for (int r = 1; r < world; ++r) if (gp[(size_t)r].n_grid != gp[0].n_grid || gp[(size_t)r].G != gp[0].G || gp[(size_t)r].B[0] != gp[0].B[0] || gp[(size_t)r].B[1] != gp[0].B[1]) return -8;
		for (int i = 0; i < n_fn; ++i) {
			FileMap f; if (!f.open(fn[i])) return -2;
			const GridPlan &g = gp[0];
			if (g.n_grid == 0) continue;
			if (g.B[i][0] != 0 || g.B[i][(size_t)g.n_grid] != f.size) return -9;
			const uint64_t step = std::max<uint64_t>(1, g.n_grid / 24);          // (every boundary is checked for order and for starting a line; the line NUMBER -- a pass over the file -- of a sample)
			for (uint64_t k = 1; k < g.n_grid; ++k) {
				const long long s = g.B[i][(size_t)k]; char c[2] = {0, 0};
				if (s <= g.B[i][(size_t)k - 1] || s >= f.size || pread(f.fd, c, 1, (off_t)s) != 1 || c[0] != '@' || pread(f.fd, c + 1, 1, (off_t)(s - 1)) != 1 || c[1] != '\n') return -10;
				if ((k % step == 0 || k + 1 == g.n_grid) && count_newlines(f.fd, 0, s, 1) != (long long)(4 * k * g.G)) return -11;
			}
		}
	}
	return 0;
}
