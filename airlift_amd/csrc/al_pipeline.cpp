// al_pipeline.cpp -- file-level driver of the drop-in (mm_map_file_frag, map.c:672-700; worker_pipeline, map.c:532-653)
// as a host pipeline around the device batch API (product code, C++):
//
//   reader thread per input file   block gzread + kseq-style record parser into flat text chunks (bseq.c:56-130)
//        |  bounded queues
//   mapper (calling thread)        chunks -> fragment-major batch (<= mini_batch_size bases, like the reference's step 0)
//                                  -> parallel 4-bit packing -> al_batch_upload / al_batch_run -> one flat result fetch
//        |  bounded queue
//   writer thread                  records -> SAM text on n_threads workers (contiguous fragment ranges) -> ordered fwrite
//
// Output order == input order (the reference's step 2 is serial for the same reason, map.c:601-644).  No mapping work is
// done on the host: without a usable HIP device al_ctx_init() fails and this function returns an error.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <chrono>
#include <thread>
#include <vector>
#include "al_internal.h"
#include "al_runtime.h"
#include "al_io.h"
#include "al_seqio.h"
#include "al_bam.h"

extern "C" void al_ctx_set_threads(al_ctx_t *c, int n_threads);

namespace {

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class T> class Queue {
	std::mutex m; std::condition_variable cv; std::deque<T> q; size_t cap; bool closed = false;
public:
	explicit Queue(size_t cap_) : cap(cap_) {}
	void push(T &&v) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return q.size() < cap || closed; }); if (!closed) q.push_back(std::move(v)); l.unlock(); cv.notify_all(); }
	bool pop(T &v) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !q.empty() || closed; }); if (q.empty()) return false; v = std::move(q.front()); q.pop_front(); l.unlock(); cv.notify_all(); return true; }
	void close() { { std::lock_guard<std::mutex> l(m); closed = true; } cv.notify_all(); }
	void abort() { { std::lock_guard<std::mutex> l(m); closed = true; q.clear(); } cv.notify_all(); }
};

typedef std::unique_ptr<AlChunk> ChunkP;
const size_t CHUNK_READS = 1 << 15;

struct Batch {
	std::vector<ChunkP> chunks;                       // own the text the pointers below refer to
	std::vector<int> n_segs, qlens;
	std::vector<const char *> seqs, names, quals;     // per read, fragment-major
	std::unique_ptr<AlRawResult> R;                   // from the driver's pool of page-locked result buffers
	int64_t bases = 0;
	void add_read(const AlChunk &c, const AlRec &r) { seqs.push_back(c.text.data() + r.seq); names.push_back(c.text.data() + r.name); quals.push_back(r.qual == ~0u ? nullptr : c.text.data() + r.qual); qlens.push_back((int)r.len); bases += r.len; }
};
typedef std::unique_ptr<Batch> BatchP;

inline int qname_len(const char *s)
{   // bseq.h:31-36
	const int l = (int)strlen(s);
	return l >= 3 && s[l - 1] >= '0' && s[l - 1] <= '9' && s[l - 2] == '/' ? l - 2 : l;
}
inline bool qname_same(const char *a, const char *b) { const int l1 = qname_len(a), l2 = qname_len(b); return l1 == l2 && strncmp(a, b, l1) == 0; }

void reader_main(AlSeqReader *rd, Queue<ChunkP> *q)
{
	for (;;) {
		ChunkP c(new AlChunk()); c->text.reserve(CHUNK_READS * 340); c->recs.reserve(CHUNK_READS);
		while (c->recs.size() < CHUNK_READS && rd->read(*c));
		const bool last = c->recs.size() < CHUNK_READS;
		if (!c->recs.empty()) q->push(std::move(c));
		if (last) break;
	}
	q->close();
}

struct SortedStore {            // --sorted-bam: every mapped record is kept until the end of the input
	std::vector<std::vector<char>> bufs;                  // record bytes, one buffer per (batch, worker)
	std::vector<uint64_t> keys; std::vector<uint32_t> buf_id, off, len;
};
struct WriterState { const al_idx_t *mi; const al_mapopt_t *opt; FILE *out; const char *rg_id; int n_threads; int rc = 0; double t_conv = 0, t_fmt = 0, t_write = 0;
                     int mode = 0;                 // 0 SAM text, 1 BAM in input order, 2 BAM coordinate-sorted (mapped records only)
                     AlBgzf *bgzf = nullptr; SortedStore *store = nullptr; };

void write_batch(WriterState &W, Batch &b)
{
	const int nf = (int)b.n_segs.size(), nr = (int)b.seqs.size();
	const AlRawResult &R = *b.R;
	const double t0 = now_s();
	std::vector<al_reg1_t> pool(R.out.size()); std::vector<const al_reg1_t *> regs(nr); std::vector<int> n_regs(nr); std::vector<int> first(nf + 1);
	{ int r = 0; for (int f = 0; f < nf; ++f) { first[f] = r; r += b.n_segs[f]; } first[nf] = r; }
	al_parallel_for(W.n_threads, (size_t)nr, [&](size_t lo, size_t hi, int) {
		for (size_t i = lo; i < hi; ++i) {
			const int n = (int)(R.off[i + 1] - R.off[i]);
			n_regs[i] = n; regs[i] = pool.data() + R.off[i];
			for (int k = 0; k < n; ++k) al_reg_from_raw(R, (int)i, k, pool[R.off[i] + k]);
		}
	});
	const double t1 = now_s();
	const int nt = W.n_threads > 1 ? W.n_threads : 1;
	std::vector<std::vector<char>> text(nt); std::vector<int> bad(nt, 0);
	struct Ent { uint64_t key; uint32_t off, len; };
	std::vector<std::vector<Ent>> ents(nt);
	al_parallel_for(nt, (size_t)nf, [&](size_t lo, size_t hi, int t) {
		std::vector<char> &o = text[t]; size_t used = 0;
		if (W.mode == 0) o.resize((hi - lo) * 800 + 65536); else o.reserve((hi - lo) * 600 + 65536);
		auto emit = [&](size_t f, int i0, int ns, int j, int k) {
			const int i = i0 + j;
			if (W.mode == 0) {
				const int l = al_write_sam(o.data() + used, o.size() - used, W.mi, b.names[i], b.qlens[i], b.seqs[i], b.quals[i], j, k, ns, &n_regs[i0], &regs[i0], W.rg_id, R.rep[f]);
				if (l > 0) used += l; else bad[t] = 1;
			} else {
				uint64_t key = 0; int unm = 0; const size_t at = o.size();
				const int l = al_write_bam_rec(o, W.mi, b.names[i], b.qlens[i], b.seqs[i], b.quals[i], j, k, ns, &n_regs[i0], &regs[i0], W.rg_id, R.rep[f], &key, &unm);
				if (W.mode == 2) { if (unm) o.resize(at); else ents[t].push_back(Ent{key, (uint32_t)at, (uint32_t)l}); }     // -F4
			}
		};
		for (size_t f = lo; f < hi; ++f) {                                   // map.c:601-644
			const int i0 = first[f], ns = b.n_segs[f];
			for (int j = 0; j < ns; ++j) {
				const int i = i0 + j;
				if (W.mode == 0) {
					size_t need = (size_t)b.qlens[i] * 2 + strlen(b.names[i]) + 4096;
					for (int k = 0; k < n_regs[i]; ++k) need += (size_t)regs[i][k].n_cigar * 12 + 128;
					const int n_rec = n_regs[i] > 0 ? n_regs[i] : 1;
					if (o.size() - used < need * n_rec) o.resize((o.size() + need * n_rec) * 3 / 2);
				}
				if (n_regs[i] > 0) {
					for (int k = 0; k < n_regs[i]; ++k) {
						const al_reg1_t *r = &regs[i][k];
						if ((W.opt->flag & AL_F_NO_PRINT_2ND) && r->id != r->parent) continue;
						emit(f, i0, ns, j, k);
					}
				} else if (!(W.opt->flag & AL_F_SAM_HIT_ONLY)) emit(f, i0, ns, j, -1);
			}
		}
		if (W.mode == 0) o.resize(used);
	});
	const double t2 = now_s();
	for (int t = 0; t < nt; ++t) {
		if (bad[t]) W.rc = -3;
		if (text[t].empty()) continue;
		if (W.mode == 0) { if (fwrite(text[t].data(), 1, text[t].size(), W.out) != text[t].size()) W.rc = -3; }
		else if (W.mode == 1) { if (W.bgzf->write(text[t].data(), text[t].size())) W.rc = -3; }
		else {
			SortedStore &S = *W.store; const uint32_t id = (uint32_t)S.bufs.size();
			for (const Ent &e : ents[t]) { S.keys.push_back(e.key); S.buf_id.push_back(id); S.off.push_back(e.off); S.len.push_back(e.len); }
			S.bufs.push_back(std::move(text[t]));
		}
	}
	W.t_conv += t1 - t0; W.t_fmt += t2 - t1; W.t_write += now_s() - t2;
}

} // namespace

static int map_files(const al_idx_t *mi, int n_fn, const char **fn, const al_mapopt_t *opt, int n_threads, FILE *out, const char *rg, int device, int mode, int level);

extern "C" int al_map_file_frag(const al_idx_t *mi, int n_fn, const char **fn, const al_mapopt_t *opt, int n_threads,
                                FILE *out, const char *rg, int device)
{
	return map_files(mi, n_fn, fn, opt, n_threads, out, rg, device, 0, 0);
}

// BAM instead of SAM text: sorted == 0 keeps the input order (every record the SAM output would have); sorted != 0 writes the
// mapped records in coordinate order -- what AirLift gets from `| samtools view -h -F4 | samtools sort -l LEVEL`
// (src/0-align_reads.sh:13), with the keys sorted on the GPU.
extern "C" int al_map_file_frag_bam(const al_idx_t *mi, int n_fn, const char **fn, const al_mapopt_t *opt, int n_threads,
                                    FILE *out, const char *rg, int device, int sorted, int level)
{
	return map_files(mi, n_fn, fn, opt, n_threads, out, rg, device, sorted ? 2 : 1, level < 0 || level > 9 ? 5 : level);
}

static int map_files(const al_idx_t *mi, int n_fn, const char **fn, const al_mapopt_t *opt, int n_threads, FILE *out, const char *rg, int device, int mode, int level)
{
	if (n_fn < 1 || n_fn > 2) return -1;
	if (n_threads < 1) n_threads = 1;
	AlSeqReader rd[2];
	for (int i = 0; i < n_fn; ++i)
		if (!rd[i].open(fn[i])) { fprintf(stderr, "ERROR: failed to open file '%s'\n", fn[i]); return -1; }
	Queue<ChunkP> cq[2] = { Queue<ChunkP>(8), Queue<ChunkP>(8) };
	std::vector<std::thread> readers;
	for (int i = 0; i < n_fn; ++i) readers.emplace_back(reader_main, &rd[i], &cq[i]);   // parsing starts while the device comes up
	auto stop_readers = [&]() { for (int i = 0; i < n_fn; ++i) cq[i].abort(); for (auto &t : readers) t.join(); };

	const bool timing = getenv("AL_TIMING") != nullptr;
	const double T0 = now_s();
	al_ctx_t *ctx = al_ctx_init(mi, opt, device);
	if (!ctx) { stop_readers(); return -2; }
	const double T1 = now_s(); double t_asm = 0, t_up = 0, t_run = 0, t_fetch = 0, t_push = 0; int n_batch = 0;
	al_ctx_set_threads(ctx, n_threads);
	char rg_id[256]; rg_id[0] = 0;
	AlBgzf bgzf(out, level, n_threads); SortedStore store;
	if (mode == 0) { if (rg != (const char *)-1) al_write_sam_hdr(out, mi, rg, rg_id); }
	else if (al_bam_header(bgzf, mi, rg == (const char *)-1 ? nullptr : rg, rg_id, mode == 2)) { stop_readers(); al_ctx_destroy(ctx); return -3; }

	Queue<BatchP> wq(2);
	WriterState W{mi, opt, out, rg_id, n_threads};
	W.mode = mode; W.bgzf = &bgzf; W.store = &store;
	std::mutex pool_m; std::vector<std::unique_ptr<AlRawResult>> pool;
	std::thread writer([&]() {
		BatchP b;
		while (wq.pop(b)) {
			if (W.rc == 0) write_batch(W, *b);
			{ std::lock_guard<std::mutex> l(pool_m); pool.push_back(std::move(b->R)); }
			b.reset();
		}
	});

	const int64_t batch_bases = opt->mini_batch_size > 0 ? (int64_t)opt->mini_batch_size : 50000000;
	int rc = 0; bool done = false;
	ChunkP carry;                                     // single-file mode: last read of a batch, which may pair with the next one
	while (!done && rc == 0 && W.rc == 0) {
		const double ta = now_s();
		BatchP b(new Batch());
		if (n_fn == 2) {
			while (b->bases < batch_bases) {
				ChunkP a, c2;
				const bool ha = cq[0].pop(a), hb = cq[1].pop(c2);
				if (!ha || !hb) { if (ha != hb) fprintf(stderr, "[W::%s] query files have different number of records; extra records skipped.\n", __func__); done = true; break; }
				const size_t n = a->recs.size() < c2->recs.size() ? a->recs.size() : c2->recs.size();
				for (size_t i = 0; i < n; ++i) { b->add_read(*a, a->recs[i]); b->add_read(*c2, c2->recs[i]); b->n_segs.push_back(2); }
				if (a->recs.size() != c2->recs.size()) { fprintf(stderr, "[W::%s] query files have different number of records; extra records skipped.\n", __func__); done = true; }
				else if (n < CHUNK_READS) done = true;
				b->chunks.push_back(std::move(a)); b->chunks.push_back(std::move(c2));
				if (done) break;
			}
		} else {   // one file: adjacent reads with the same name form a fragment (frag_mode, map.c:580-586)
			std::vector<std::pair<const AlChunk *, const AlRec *>> rs;
			if (carry) { rs.emplace_back(carry.get(), &carry->recs[0]); b->bases += carry->recs[0].len; b->chunks.push_back(std::move(carry)); }
			while (b->bases < batch_bases) {
				ChunkP a;
				if (!cq[0].pop(a)) { done = true; break; }
				for (const AlRec &r : a->recs) { rs.emplace_back(a.get(), &r); b->bases += r.len; }
				b->chunks.push_back(std::move(a));
			}
			b->bases = 0;
			const size_t n = rs.size();
			std::vector<std::pair<size_t, int>> frags;        // (first read, segments), greedy from the front like the reference
			for (size_t i = 0; i < n; ) {
				int ns = 1;
				if (i + 1 < n && qname_same(rs[i].first->text.data() + rs[i].second->name, rs[i + 1].first->text.data() + rs[i + 1].second->name)) ns = 2;
				frags.emplace_back(i, ns); i += ns;
			}
			if (!done && !frags.empty() && frags.back().second == 1) {   // a trailing single read may pair with the first read of the next batch: carry it over
				const size_t k = frags.back().first; frags.pop_back();
				carry.reset(new AlChunk());
				const AlChunk &c0 = *rs[k].first; const AlRec &r0 = *rs[k].second; AlRec r; r.name = 0;
				carry->text.insert(carry->text.end(), c0.text.data() + r0.name, c0.text.data() + r0.name + strlen(c0.text.data() + r0.name) + 1);
				r.seq = (uint32_t)carry->text.size(); r.len = r0.len; carry->text.insert(carry->text.end(), c0.text.data() + r0.seq, c0.text.data() + r0.seq + r0.len);
				r.qual = ~0u; if (r0.qual != ~0u) { r.qual = (uint32_t)carry->text.size(); carry->text.insert(carry->text.end(), c0.text.data() + r0.qual, c0.text.data() + r0.qual + r0.len); }
				carry->recs.push_back(r);
			}
			for (const auto &fr : frags) {
				for (int j = 0; j < fr.second; ++j) b->add_read(*rs[fr.first + j].first, *rs[fr.first + j].second);
				b->n_segs.push_back(fr.second);
			}
		}
		const int nf = (int)b->n_segs.size();
		if (nf == 0) continue;
		const double tb = now_s();
		if ((rc = al_batch_upload(ctx, nf, b->n_segs.data(), b->qlens.data(), b->seqs.data(), b->names.data())) != 0) break;
		const double tc = now_s();
		if ((rc = al_batch_run(ctx)) != 0) break;
		const double td = now_s();
		{ std::lock_guard<std::mutex> l(pool_m); if (!pool.empty()) { b->R = std::move(pool.back()); pool.pop_back(); } }
		if (!b->R) b->R.reset(new AlRawResult());
		if ((rc = al_fetch_raw(ctx, *b->R)) != 0) break;
		const double te = now_s();
		wq.push(std::move(b));
		t_asm += tb - ta; t_up += tc - tb; t_run += td - tc; t_fetch += te - td; t_push += now_s() - te; ++n_batch;
	}
	wq.close(); writer.join();
	stop_readers();
	if (rc == 0 && W.rc == 0 && mode == 2) {       // coordinate order: stable radix sort of the keys on the GPU, then the records stream out
		const size_t n = store.keys.size();
		std::vector<uint32_t> perm(n);
		if (al_sort_keys(ctx, store.keys.data(), perm.data(), n)) W.rc = -3;
		else for (size_t i = 0; i < n && W.rc == 0; ++i) { const uint32_t k = perm[i]; if (bgzf.write(store.bufs[store.buf_id[k]].data() + store.off[k], store.len[k])) W.rc = -3; }
	}
	if (rc == 0 && W.rc == 0 && mode != 0 && bgzf.finish()) W.rc = -3;
	if (timing) fprintf(stderr, "[airlift] pipeline: ctx init %.3f s; %d batches; mapper: wait+assemble %.3f upload %.3f run %.3f fetch %.3f wait-writer %.3f; writer: convert %.3f format %.3f write %.3f; total %.3f s\n", T1 - T0, n_batch, t_asm, t_up, t_run, t_fetch, t_push, W.t_conv, W.t_fmt, W.t_write, now_s() - T0);
	al_ctx_destroy(ctx);
	fflush(out);
	return rc ? rc : W.rc;
}

extern "C" int al_batch_count_candidates(al_ctx_t *c, int64_t *total);

// The unmodified fork's main loop (main.c:384-391) maps every read of ONE file on its own (mm_map) and sums the counter of
// map.c:299-312; here: reader thread -> single-segment batches -> seed kernels + k_alser_count.
extern "C" int al_count_candidates_file(const al_idx_t *mi, const char *fn, const al_mapopt_t *opt, int n_threads, int device, int64_t *total)
{
	if (!total) return -1;
	*total = 0;
	AlSeqReader rd;
	if (!rd.open(fn)) { fprintf(stderr, "ERROR: failed to open file '%s'\n", fn); return -1; }
	Queue<ChunkP> cq(8);
	std::thread reader(reader_main, &rd, &cq);
	al_ctx_t *ctx = al_ctx_init(mi, opt, device);
	if (!ctx) { cq.abort(); reader.join(); return -2; }
	al_ctx_set_threads(ctx, n_threads);
	const int64_t batch_bases = opt->mini_batch_size > 0 ? (int64_t)opt->mini_batch_size : 50000000;
	int rc = 0; bool done = false;
	while (!done && rc == 0) {
		Batch b;
		while (b.bases < batch_bases) {
			ChunkP a;
			if (!cq.pop(a)) { done = true; break; }
			for (const AlRec &r : a->recs) { b.add_read(*a, r); b.n_segs.push_back(1); }
			b.chunks.push_back(std::move(a));
		}
		const int nf = (int)b.n_segs.size();
		if (nf == 0) continue;
		int64_t t = 0;
		if ((rc = al_batch_upload(ctx, nf, b.n_segs.data(), b.qlens.data(), b.seqs.data(), nullptr)) != 0) break;   // mm_map passes no query name
		if ((rc = al_batch_count_candidates(ctx, &t)) != 0) break;
		*total += t;
	}
	cq.abort(); reader.join();
	al_ctx_destroy(ctx);
	return rc;
}
