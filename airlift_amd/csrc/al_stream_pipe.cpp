// al_stream_pipe.cpp -- file driver of the drop-in with the byte work on the GPU (mm_map_file_frag, map.c:672-700; the
// reference's three-step pipeline worker_pipeline, map.c:532-653, kthread.c:130-159).
//
//   file readers      pread() of fixed-size pieces into page-locked buffers, a few threads per file, delivered in file order
//        |
//   ingest thread     per batch: the text left over by the previous batch + new pieces -> HBM of the next free slot; kernels index the
//                     lines, check the records and tell how many the batch takes (al_stream.hip); the rest is the next batch's carry
//        |
//   mapper per slot   read arrays on the device (pack, name hash) -> al_batch_run -> SAM text by kernels -> page-locked host buffer
//        |
//   writer thread     write() in batch order = input order (the reference's step 2 is serial for the same reason, map.c:601-644)
//
// A lane (one GPU) has several slots, so H2D + parsing of batch n+1 and SAM text + D2H of batch n-1 overlap the mapping kernels of
// batch n; batches are dealt to the lanes round-robin.  The host touches no record: it moves file bytes.  Batch size: workspaces
// grow with the seed hits of a batch, so the first (small) batch is measured and later ones are sized to fill the free HBM
// (mini_batch_size, -K, stays as the upper bound; results do not depend on the batching: map.c:229-400 is per read).
// Input that is not plain four-line FASTQ (gzip, FASTA, multi-line records, a pipe) goes to the host driver (al_pipeline.cpp),
// from the byte where the device parser stops.
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <hip/hip_runtime.h>
#include "al_internal.h"
#include "al_runtime.h"
#include "al_stream.h"
#include "al_stream_pipe.h"

namespace {

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- sequential file -> page-locked pieces ----------------------------------------------------------------------------------
struct Piece { char *p = nullptr; size_t n = 0; long long idx = -1; bool last = false; int buf = -1; };
class PieceReader {
	int fd = -1; long long size = 0, start = 0; size_t piece; int n_buf;
	std::vector<long long> r_lo, r_hi, r_first;  // (a list of ranges, round 6) bounds of range j and the number of its first piece; a piece never spans two ranges, `last` marks a range's last piece
	std::vector<char *> bufs; std::vector<int> free_bufs;
	std::mutex m; std::condition_variable cv;
	long long next_assign = 0, next_deliver = 0, n_pieces = 0; bool stop = false, failed = false;
	std::deque<Piece> done;                      // finished pieces (any order)
	std::vector<std::thread> workers;
	void work()
	{
		for (;;) {
			Piece pc;
			{
				std::unique_lock<std::mutex> l(m);
				cv.wait(l, [&] { return stop || (next_assign < n_pieces && !free_bufs.empty()); });
				if (stop) return;
				pc.idx = next_assign++; pc.buf = free_bufs.back(); free_bufs.pop_back();
			}
			pc.p = bufs[pc.buf];
			long long off = start + pc.idx * (long long)piece, lim = size; bool last = pc.idx == n_pieces - 1;
			if (!r_lo.empty()) {
				size_t j = (size_t)(std::upper_bound(r_first.begin(), r_first.end(), pc.idx) - r_first.begin()) - 1;
				off = r_lo[j] + (pc.idx - r_first[j]) * (long long)piece; lim = r_hi[j]; last = pc.idx + 1 == (j + 1 < r_first.size() ? r_first[j + 1] : n_pieces);
			}
			size_t want = (size_t)std::min<long long>((long long)piece, lim - off), got = 0;
			while (got < want) { const ssize_t k = pread(fd, pc.p + got, want - got, (off_t)(off + (long long)got)); if (k <= 0) break; got += (size_t)k; }
			pc.n = got; pc.last = last;
			if (pc.last && got > 0 && pc.p[got - 1] != '\n') pc.p[pc.n++] = '\n';     // an unterminated last line still counts (kseq.h)
			{ std::lock_guard<std::mutex> l(m); if (got < want) failed = true; done.push_back(pc); }
			cv.notify_all();
		}
	}
public:
	bool open(const char *fn, long long start_off, long long end_off, size_t piece_bytes, int n_buffers, int n_threads)
	{
		fd = ::open(fn, O_RDONLY);
		struct stat sb;
		if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return false;
		size = (long long)sb.st_size; if (end_off >= 0 && end_off < size) size = end_off; start = start_off; piece = piece_bytes; n_buf = n_buffers;
		n_pieces = size > start ? (size - start + (long long)piece - 1) / (long long)piece : 0;
		for (int i = 0; i < n_buf; ++i) { char *p = nullptr; if (hipHostMalloc((void **)&p, piece + 16, hipHostMallocDefault) != hipSuccess) return false; bufs.push_back(p); free_bufs.push_back(i); }
		for (int t = 0; t < n_threads; ++t) workers.emplace_back(&PieceReader::work, this);
		return true;
	}
	// the same over a list of non-empty byte ranges, read one after the other (pieces never span two of them)
	bool open_ranges(const char *fn, const long long *lo, const long long *hi, int n, size_t piece_bytes, int n_buffers, int n_threads)
	{
		long long np = 0;
		for (int j = 0; j < n; ++j) { if (hi[j] <= lo[j]) return false; r_lo.push_back(lo[j]); r_hi.push_back(hi[j]); r_first.push_back(np); np += (hi[j] - lo[j] + (long long)piece_bytes - 1) / (long long)piece_bytes; }
		if (!open(fn, 0, -1, piece_bytes, n_buffers, 0)) return false;
		{ std::lock_guard<std::mutex> l(m); n_pieces = np; }
		for (int t = 0; t < n_threads; ++t) workers.emplace_back(&PieceReader::work, this);
		return true;
	}
	long long file_size() const { return size > start ? size - start : 0; }
	// next piece in file order; false at end of file (or after a read error: failed() tells)
	bool next(Piece &pc)
	{
		std::unique_lock<std::mutex> l(m);
		if (next_deliver >= n_pieces) return false;
		for (;;) {
			for (auto it = done.begin(); it != done.end(); ++it) if (it->idx == next_deliver) { pc = *it; done.erase(it); ++next_deliver; return !failed; }
			if (failed) return false;
			cv.wait(l);
		}
	}
	void release(const Piece &pc) { { std::lock_guard<std::mutex> l(m); free_bufs.push_back(pc.buf); } cv.notify_all(); }
	bool at_end() { std::lock_guard<std::mutex> l(m); return next_deliver >= n_pieces; }
	bool has_failed() { std::lock_guard<std::mutex> l(m); return failed; }
	~PieceReader()
	{
		{ std::lock_guard<std::mutex> l(m); stop = true; } cv.notify_all();
		for (auto &t : workers) t.join();
		for (char *p : bufs) (void)hipHostFree(p);
		if (fd >= 0) close(fd);
	}
};

enum { SL_FREE = 0, SL_READY, SL_MAPPED };
struct Slot {                                           // text + SAM buffers of one batch in flight
	AlStreamSlot S; int lane = 0, device = 0;
	int state = SL_FREE; uint64_t seq = 0;
	AlIngestResult res;
	std::atomic<size_t> held{0};                        // device bytes of its buffers
	std::vector<std::vector<char>> chunks;              // SAM text of a batch that had to be cut (else it is in S.h_sam)
};
struct Mapper {                                         // a mapping context and the thread that runs batches on it
	al_ctx_t *ctx = nullptr; int lane = 0, device = 0, idx = 0;
	std::atomic<size_t> held{0};                        // device bytes of its workspaces
	std::thread th;
	double t_setup = 0, t_run = 0, t_sam = 0, t_wait = 0; int n_batch = 0;
};

// plain, uncompressed, starts like FASTQ?
bool eligible_file(const char *fn)
{
	if (!strcmp(fn, "-")) return false;
	const int fd = open(fn, O_RDONLY);
	if (fd < 0) return false;
	struct stat sb; unsigned char h[2] = {0, 0};
	const bool ok = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && (sb.st_size == 0 || (pread(fd, h, 2, 0) >= 1 && h[0] == '@' && !(h[0] == 0x1f && h[1] == 0x8b)));
	close(fd);
	return ok;
}

} // namespace

int al_stream_map_files(const al_idx_t *mi, int n_fn, const char **fn, const al_mapopt_t *opt, int n_threads, FILE *out, const char *rg,
                        const int *devices, int n_dev, AlStreamResume *rs, const AlStreamRange *range)
{
	static const AlStreamRange whole;
	if (!range) range = &whole;
	if (getenv("AL_HOST_IO") || n_fn < 1 || n_fn > 2 || n_dev < 1) return AL_STREAM_NA;
	for (int i = 0; i < n_fn; ++i) if (!eligible_file(fn[i])) return AL_STREAM_NA;
	const bool timing = getenv("AL_TIMING") != nullptr, trace = getenv("AL_TRACE") != nullptr;
	const double T0 = now_s();
	// A long input (by its file sizes, ~360 bytes per read) is mapped in batches of up to 2^20 reads on TWO contexts per GPU: since the chaining
	// scratch went (44 instead of 96 workspace bytes per seed hit: ~70 KB per read on a repeat-rich genome) two such contexts fit next to the
	// index, a batch of that size runs at the resident rate (the tails inside the launches are amortised), and a third context of that size is what
	// made the device run short of fast memory (3 x 2^20 reads on C4: 1.7 M reads/s; 2 x 2^20: 7.1 M).  Short inputs keep three contexts and the probes.
	const bool ranged = range->list;                      // (round 6) the caller's list of byte ranges, one batch each, placed by its sink (a multi-process run)
	if (ranged && (!range->sink || !range->sink->offset_of || (range->n_ranges > 0 && (!range->rstart[0] || !range->rend[0] || (n_fn == 2 && (!range->rstart[1] || !range->rend[1])))))) return -1;
	double est_reads0 = 0;
	if (ranged) { for (int i = 0; i < n_fn; ++i) for (int j = 0; j < range->n_ranges; ++j) est_reads0 += (double)(range->rend[i][j] - range->rstart[i][j]) / 360.0; est_reads0 /= (double)n_dev; }
	else { struct stat sb; for (int i = 0; i < n_fn; ++i) if (stat(fn[i], &sb) == 0) { const double hi = range->end[i] >= 0 ? (double)range->end[i] : (double)sb.st_size; est_reads0 += std::max(0.0, hi - (double)range->start[i]) / 360.0; } est_reads0 /= (double)n_dev; }
	const bool long_input = est_reads0 >= 3.0e6;
	const int n_slots_lane = std::max(2, std::min(8, getenv("AL_SLOTS") ? atoi(getenv("AL_SLOTS")) : long_input ? 4 : 5));      // text / SAM buffer sets per GPU
	const int n_ctx_lane = std::max(1, std::min(n_slots_lane, getenv("AL_CTXS") ? atoi(getenv("AL_CTXS")) : long_input ? 2 : 3));  // mapping contexts per GPU
	const int NL = n_dev, NS = NL * n_slots_lane, NM = NL * n_ctx_lane;
	const size_t PIECE = (size_t)(getenv("AL_PIECE_MB") ? std::max(1, atoi(getenv("AL_PIECE_MB"))) : 8) << 20;

	std::mutex m; std::condition_variable cv; int rc = 0;
	auto fail = [&](int code) { { std::lock_guard<std::mutex> l(m); if (rc == 0) rc = code ? code : -1; } cv.notify_all(); };

	// batch k: lane k % NL, the lane's batch number j = k / NL, slot j % n_slots_lane and context j % n_ctx_lane of that lane
	std::vector<std::unique_ptr<Slot>> slots; std::vector<std::unique_ptr<Mapper>> mappers;
	auto slot_of = [&](uint64_t k) -> Slot * { return slots[(size_t)((k % NL) * n_slots_lane + (k / NL) % n_slots_lane)].get(); };
	auto destroy_all = [&]() { for (auto &s : slots) { al_acct() = nullptr; al_stream_slot_destroy(s->S); } for (auto &mp : mappers) if (mp->ctx) al_ctx_destroy(mp->ctx); };
	for (int l = 0; l < NL; ++l) {
		for (int i = 0; i < n_ctx_lane; ++i) {
			std::unique_ptr<Mapper> mp(new Mapper()); mp->lane = l; mp->idx = i;
			al_acct() = &mp->held;
			mp->ctx = al_ctx_init(mi, opt, devices[l]);
			al_acct() = nullptr;
			if (!mp->ctx) { destroy_all(); return -2; }
			al_ctx_no_taps(mp->ctx);
			mp->device = mp->ctx->device;                // (a negative device argument means LOCAL_RANK or 0)
			mappers.push_back(std::move(mp));
		}
		for (int i = 0; i < n_slots_lane; ++i) {
			std::unique_ptr<Slot> sl(new Slot()); sl->lane = l; sl->device = mappers[(size_t)l * n_ctx_lane]->device;
			al_acct() = &sl->held;
			const int e = al_stream_slot_init(sl->S, mi, sl->device, n_fn);
			al_acct() = nullptr;
			slots.push_back(std::move(sl));
			if (e) { destroy_all(); return -2; }
		}
	}
	const double T1 = now_s();
	char rg_id[256]; rg_id[0] = 0;
	if (rg != (const char *)-1) {
		if (range->header) al_write_sam_hdr(out, mi, rg, rg_id);
		else { FILE *nul = fopen("/dev/null", "w"); if (nul) { al_write_sam_hdr(nul, mi, rg, rg_id); fclose(nul); } }     // (the read group's ID still goes into every record)
	}
	fflush(out);
	const int ofd = fileno(out);
	memcpy(rs->rg_id, rg_id, 256);

	// Batch size in reads.  Upper bound: mini_batch_size bases (-K) unless the caller leaves it to the driver.  The first batches are
	// small probes; what a probe held per read and how fast it ran decide the size: device memory costs per byte a process touches
	// for the first time, so the workspaces of a batch are sized to cost a fraction of the job's estimated mapping time, inside the
	// free memory.
	const int64_t k_bases = getenv("AL_AUTO_BATCH") ? (int64_t)1 << 40 : opt->mini_batch_size > 0 ? (int64_t)opt->mini_batch_size : 50000000;
	std::atomic<int> max_reads{0}; std::atomic<bool> sized{false};
	const int probe_reads = getenv("AL_PROBE_READS") ? std::max(2, atoi(getenv("AL_PROBE_READS"))) : 32768;
	std::atomic<int> reads_cap_k{1 << 30};                 // mini_batch_size in reads, once a read length is known
	std::atomic<long long> est_total_reads{0};             // from the file sizes and the bytes per read of the first batch
	max_reads = (int)std::max<int64_t>(2, std::min<int64_t>(probe_reads, k_bases / 64));   // batch 0 (-K is an upper bound: reads of >= 64 bases assumed until a batch has been seen); the following ones 4 x the probe until the size is decided
	double probe_held0 = 0, probe_n0 = 0; int probe_mult = 0;
	std::atomic<bool> single_probe{false};               // a long input (by its file sizes): no small first batch, the first batch already has the size the run keeps
	if (getenv("AL_BATCH_READS")) { max_reads = std::max(2, atoi(getenv("AL_BATCH_READS"))); sized = true; }    // tests / tuning: fixed batches
	if (ranged) { max_reads = 1 << 30; sized = true; }     // a range is a batch, whatever it holds (one that does not fit the device is halved by its mapper like any other)

	// SAM text leaves the device through a small ring of page-locked buffers (page-locking memory costs ~0.2 s per GB: no buffer of a
	// batch's size): the copy of piece n + 1 runs while piece n is written
	const size_t CH = (size_t)(getenv("AL_OUT_PIECE_MB") ? std::max(1, atoi(getenv("AL_OUT_PIECE_MB"))) : 32) << 20;
	// (page-locked buffers only: the 'piece copied' events belong to the slot that is drained -- AlStreamSlot::ev_out, created on the slot's device;
	//  an event of another device cannot be recorded on the slot's stream, which is what a ring-owned pair amounted to with lanes on several GPUs)
	struct OutRing { char *buf[2] = {nullptr, nullptr}; };
	auto ring_make = [&](OutRing &r) -> bool { for (int i = 0; i < 2; ++i) if (hipHostMalloc((void **)&r.buf[i], CH, hipHostMallocDefault) != hipSuccess) return false; return true; };
	auto ring_free = [&](OutRing &r) { for (int i = 0; i < 2; ++i) if (r.buf[i]) (void)hipHostFree(r.buf[i]); };
	// text [0, n) of slot S through ring r to sink(ptr, len)
	auto drain_sam = [&](AlStreamSlot &S, OutRing &r, const std::function<bool(const char *, size_t)> &sink) -> int {
		const uint64_t n = S.sam_bytes; const uint64_t np = (n + CH - 1) / CH;
		if (np && al_stream_sam_fetch(S, 0, std::min<uint64_t>(CH, n), r.buf[0], S.ev_out[0])) return -1;
		for (uint64_t c = 0; c < np; ++c) {
			if (c + 1 < np && al_stream_sam_fetch(S, (c + 1) * CH, std::min<uint64_t>(CH, n - (c + 1) * CH), r.buf[(c + 1) & 1], S.ev_out[(c + 1) & 1])) return -1;
			if (hipEventSynchronize(S.ev_out[c & 1]) != hipSuccess) return -1;
			if (!sink(r.buf[c & 1], (size_t)std::min<uint64_t>(CH, n - c * CH))) return -3;
		}
		return 0;
	};
	OutRing wring; std::vector<OutRing> mrings(mappers.size());
	if (!ring_make(wring)) { ring_free(wring); destroy_all(); return -2; }

	// ---- mappers -------------------------------------------------------------------------------------------------------------
	uint64_t n_batches = ~0ULL;                            // set by the ingest thread when the input is exhausted
	for (auto &mpp : mappers) {
		Mapper *mp = mpp.get();
		mp->th = std::thread([&, mp]() {
			al_acct() = &mp->held;
			al_ctx_t *ctx = mp->ctx;
			for (uint64_t j = (uint64_t)mp->idx;; j += (uint64_t)n_ctx_lane) {      // this context's batches of its lane, in order
				const uint64_t k = j * (uint64_t)NL + (uint64_t)mp->lane;
				Slot *sl = slot_of(k);
				{
					const double tw = now_s();
					std::unique_lock<std::mutex> l(m);
					cv.wait(l, [&] { return rc != 0 || (sl->state == SL_READY && sl->seq == k) || (n_batches != ~0ULL && k >= n_batches); });
					mp->t_wait += now_s() - tw;
					if (rc != 0 || (n_batches != ~0ULL && k >= n_batches)) return;
				}
				AlStreamSlot &S = sl->S;
				const AlIngestResult &res = sl->res;
				sl->chunks.clear();
				std::vector<uint32_t> fstart;                  // single-file input, a batch that has to be cut: fragment -> first record
				bool cut = false;
				// fragments [flo, fhi): arrays -> mapping kernels -> SAM text.  A range whose workspaces do not fit (AL_ERR_NOMEM) is cut in two
				// and each half run on its own, recursively, as the host driver does.
				std::function<int(uint32_t, uint32_t)> process = [&](uint32_t flo, uint32_t fhi) -> int {
					if (fhi <= flo) return 0;
					uint32_t rlo = flo, rhi = fhi;
					if (n_fn == 1) { if (flo == 0 && fhi == (uint32_t)res.n_frag) { rlo = 0; rhi = (uint32_t)res.n_reads; } else { rlo = fstart[flo]; rhi = fstart[fhi]; } }
					const double t0 = now_s();
					al_nomem_flag() = false;
					int r = al_stream_setup(S, ctx, rlo, rhi, flo, fhi);
					if (r == 0) { const double t1 = now_s(); mp->t_setup += t1 - t0; r = al_batch_run(ctx); mp->t_run += now_s() - t1; }
					else if (al_nomem_flag()) r = AL_ERR_NOMEM;
					if (r == AL_ERR_NOMEM && fhi - flo > 1) {
						const uint32_t mid = flo + (fhi - flo) / 2;
						fprintf(stderr, "[airlift] batch of %u fragments does not fit the device workspaces: running it as %u + %u\n", fhi - flo, mid - flo, fhi - mid);
						if (n_fn == 1 && fstart.empty() && al_stream_frag_starts(S, res, fstart)) return -1;
						cut = true;
						{ const int mr = max_reads.load(), half = std::max(2, (int)((uint64_t)(n_fn == 2 ? 2 : 1) * (fhi - flo) / 2)); if (half < mr) max_reads = half; }
						if ((r = process(flo, mid)) != 0) return r;
						return process(mid, fhi);
					}
					if (r != 0) return r;
					const double t2 = now_s();
					if ((r = al_stream_sam(S, ctx, rg_id)) != 0) return r;
					if (cut) {   // (rare: the pieces of a cut batch are kept on the host until its turn to be written)
						OutRing &mr = mrings[(size_t)(mp->lane * n_ctx_lane + mp->idx)];
						if (!mr.buf[0] && !ring_make(mr)) return -1;
						sl->chunks.emplace_back(); std::vector<char> &dst = sl->chunks.back(); dst.reserve(S.sam_bytes);
						if ((r = drain_sam(S, mr, [&](const char *p, size_t n) { dst.insert(dst.end(), p, p + n); return true; })) != 0) return r;
					}
					mp->t_sam += now_s() - t2;
					return 0;
				};
				const double tb = now_s();
				const int r = process(0, (uint32_t)res.n_frag);
				if (r != 0) { fail(r); return; }
				++mp->n_batch;
				if (trace) fprintf(stderr, "[airlift] trace: batch %llu (lane %d context %d): %d reads in %.1f ms at +%.3f s (device pipeline %.1f ms, side stream %.1f ms)\n", (unsigned long long)k, mp->lane, mp->idx, res.n_reads, (now_s() - tb) * 1e3, now_s() - T1, ctx->ms_total, ctx->ms_side);
				// Sizing (first context of the first lane): its first batch is a small probe, its second a larger one; what the two held and
				// how long the second ran give workspace bytes and time per read.
				if (!sized.load() && mp->lane == 0 && mp->idx == 0) {
					if (mp->n_batch == 1 && !single_probe.load()) { probe_held0 = (double)mp->held.load(); probe_n0 = (double)res.n_reads; }
					else {
						if (single_probe.load() && mp->n_batch == 1) { probe_n0 = 0; probe_held0 = std::min(1.0e9, 0.25 * (double)mp->held.load()); }   // one point: the part that does not grow with the batch is taken as 1 GB (measured 0.8 - 1.5)
						size_t free_b = 0, total_b = 0;
						if (hipSetDevice(mp->device) == hipSuccess && al_dev_mem_info(&free_b, &total_b) == hipSuccess) {
							size_t held_ctx = 0; for (auto &o : mappers) if (o->device == mp->device) held_ctx += o->held.load();
							const double h1 = (double)mp->held.load(), n1 = (double)res.n_reads, t_batch = now_s() - tb;
							double v = n1 > probe_n0 ? (h1 - probe_held0) / (n1 - probe_n0) : h1 / n1; if (v < 1024.0) v = 1024.0;   // workspace bytes per read ...
							const double F = std::max(0.0, h1 - v * n1);                                                               // ... on top of what does not grow with the batch
							// memory: the lane's contexts share what is free now plus what they hold; the slots' text / SAM buffers (~2 KB per read each) come out of the same
							double mr = (((double)free_b + (double)held_ctx) * 0.85 / (double)n_ctx_lane - F) / (v * 1.30 + (double)n_slots_lane * 2048.0 / (double)n_ctx_lane);
							// time: device memory costs a process ~1 s per 30 GB it obtains for the first time (AL_ALLOC_GBS), a batch costs ~25 ms of
							// launch gaps and serial tails on top of what scales with its reads (AL_BATCH_MS): with N reads over n contexts in batches
							// of B, allocation costs n (F + 1.3 v B) / A and the gaps N t0 / (n B) -- least at B = sqrt(N t0 A / (1.3 v n^2)).
							// (A short input keeps the probe's size; a genome's worth of reads is bounded by memory alone.)
							const double total_reads = (double)std::max<long long>(est_total_reads.load(), res.n_reads) / (double)NL;
							// With a reserve (al_device_reserve: the run's memory was obtained in the background during start-up) allocation costs nothing any more
							// and the bound is what the reserve still has room for; without one, the fitted model of the driver's pace stays.
							const long long room = al_dev_reserve_room();
							static const double A = (getenv("AL_ALLOC_GBS") ? atof(getenv("AL_ALLOC_GBS")) : 30.0) * 1e9, t0b = (getenv("AL_BATCH_MS") ? atof(getenv("AL_BATCH_MS")) : 25.0) * 1e-3;
							const double b_opt = sqrt(total_reads * t0b * A / (1.30 * v * (double)n_ctx_lane * (double)n_ctx_lane));
							if (room >= 0) mr = std::min(mr, (((double)room + (double)held_ctx) * 0.95 / (double)n_ctx_lane - F) / (v * 1.30 + (double)n_slots_lane * 2048.0 / (double)n_ctx_lane));
							else if (A > 0) mr = std::min(mr, b_opt);
							mr = std::min(mr, (double)reads_cap_k.load()); mr = std::min(mr, 4.0e6); mr = std::max(mr, n1);
							if (mr < 2.0 * n1) mr = n1;          // growing past the probe's size frees and re-obtains every workspace (seconds): only for at least twice the batch
							bool exp = false;
							if (sized.compare_exchange_strong(exp, true)) {
								max_reads = (int)mr;
								if (timing || trace) fprintf(stderr, "[airlift] stream driver: probe batches of %.0f / %.0f reads held %.1f / %.1f MB (%.0f bytes per read + %.1f MB; the second took %.1f ms); %.1f GB free on device %d, ~%.1f M reads to map -> batches of %d reads (%d context(s), %d slots per GPU)\n",
								                             probe_n0, n1, probe_held0 / 1e6, h1 / 1e6, v, F / 1e6, t_batch * 1e3, free_b / 1e9, mp->device, total_reads / 1e6, (int)mr, n_ctx_lane, n_slots_lane);
							}
						}
					}
				}
				{ std::lock_guard<std::mutex> l(m); sl->state = SL_MAPPED; }
				cv.notify_all();
			}
		});
	}

	// ---- writer --------------------------------------------------------------------------------------------------------------
	double t_write = 0; uint64_t bytes_out = 0, recs_out = 0;
	// a regular output file is written by a few threads at once (pwrite of the parts of a piece: one thread copies ~3 GB/s into the page cache)
	long long woff = -1; int n_wr = 1;
	{ struct stat sb; const int fl = fcntl(ofd, F_GETFL); const off_t at = lseek(ofd, 0, SEEK_CUR);
	  if (fstat(ofd, &sb) == 0 && S_ISREG(sb.st_mode) && at >= 0 && fl >= 0 && !(fl & O_APPEND) && !getenv("AL_NO_PWRITE")) { woff = (long long)at; n_wr = std::max(1, std::min(16, n_threads / 2)); } }
	uint64_t sink_rounds = 0;                             // rounds of the caller's sink this process has taken part in
	const uint64_t pre_bytes = ranged && range->header && woff > 0 ? (uint64_t)woff : 0;   // (the header, already in the file)
	std::thread writer([&]() {
		for (uint64_t k = 0;; ++k) {
			Slot *sl = slot_of(k);
			{
				std::unique_lock<std::mutex> l(m);
				cv.wait(l, [&] { return rc != 0 || (sl->state == SL_MAPPED && sl->seq == k) || (n_batches != ~0ULL && k >= n_batches); });
				if (rc != 0 || (n_batches != ~0ULL && k >= n_batches)) return;
			}
			const double t0 = now_s();
			if (ranged) {   // where this batch goes: the sizes of the round's batches of all processes, exchanged by the sink
				uint64_t nb = 0; if (!sl->chunks.empty()) for (auto &c : sl->chunks) nb += c.size(); else nb = sl->S.sam_bytes;
				const long long off = range->sink->offset_of(range->sink->ctx, k, nb, k == 0 ? pre_bytes : 0, 1); ++sink_rounds;
				if (off < 0 || woff < 0) { fail(-4); return; }
				woff = off;
			}
			const std::function<bool(const char *, size_t)> put = [&](const char *p, size_t n) -> bool {
				if (woff >= 0) {
					std::atomic<bool> okw{true}; const long long base = woff;
					al_parallel_for(n >= ((size_t)4 << 20) ? n_wr : 1, n, [&](size_t lo, size_t hi, int) { while (lo < hi) { const ssize_t w = pwrite(ofd, p + lo, hi - lo, (off_t)(base + (long long)lo)); if (w <= 0) { okw = false; return; } lo += (size_t)w; } });
					woff += (long long)n;
					return okw.load();
				}
				while (n) { const ssize_t w = write(ofd, p, n); if (w <= 0) return false; p += w; n -= (size_t)w; }
				return true;
			};
			bool ok = true;
			if (!sl->chunks.empty()) { for (auto &c : sl->chunks) { ok = ok && put(c.data(), c.size()); bytes_out += c.size(); } sl->chunks.clear(); }
			else { const int dr = drain_sam(sl->S, wring, put); ok = dr == 0; if (dr == -1) { fail(-1); return; } bytes_out += sl->S.sam_bytes; }
			recs_out += sl->S.sam_records;
			t_write += now_s() - t0;
			if (!ok) { perror("[airlift] writing the SAM output failed"); fail(-3); return; }
			{ std::lock_guard<std::mutex> l(m); sl->state = SL_FREE; }
			cv.notify_all();
		}
	});

	// ---- ingest (this thread) ---------------------------------------------------------------------------------------------------
	double t_wait_slot = 0, t_load = 0, t_parse = 0, t_wait_read = 0; uint64_t n_reads_total = 0, n_frag_total = 0;
	{
		const int rd_threads = std::max(1, std::min(4, n_threads / (2 * n_fn)));
		std::unique_ptr<PieceReader> rd[2];
		bool open_ok = true;
		for (int i = 0; i < n_fn; ++i) {
			rd[i].reset(new PieceReader());
			if (ranged ? !rd[i]->open_ranges(fn[i], range->rstart[i], range->rend[i], range->n_ranges, PIECE, 2 * rd_threads + 2, rd_threads)
			           : !rd[i]->open(fn[i], range->start[i], range->end[i], PIECE, 2 * rd_threads + 2, rd_threads)) open_ok = false;
		}
		if (!open_ok) fail(-1);
		if (ranged) {
			// every range of the list is ONE batch: all of its text goes to the slot, the parser must take all of it (the ranges start and end at records
			// and hold the same records of both files: the caller counted lines, the grammar is checked here)
			uint64_t kk = 0;
			for (int j = 0; j < range->n_ranges && open_ok; ++j) {
				Slot *sl = slot_of(kk);
				{
					const double t0 = now_s();
					std::unique_lock<std::mutex> l(m);
					cv.wait(l, [&] { return rc != 0 || sl->state == SL_FREE; });
					t_wait_slot += now_s() - t0;
					if (rc != 0) break;
				}
				al_acct() = &sl->held;
				AlStreamSlot &S = sl->S;
				int err = 0;
				const double t1 = now_s();
				for (int i = 0; i < n_fn && !err; ++i) {
					const long long nb = range->rend[i][j] - range->rstart[i][j];
					const size_t n_pc = (size_t)((nb + (long long)PIECE - 1) / (long long)PIECE);
					if (al_stream_begin_text(S, i, n_pc * (PIECE + 16) + 64)) { err = -1; break; }
					for (;;) {
						Piece pc; const double tw = now_s();
						if (!rd[i]->next(pc)) { fprintf(stderr, "[airlift] reading '%s' failed\n", fn[i]); err = -1; break; }
						t_wait_read += now_s() - tw;
						if (al_stream_append_text(S, i, pc.p, pc.n)) err = -1;
						const bool last = pc.last;
						rd[i]->release(pc);
						if (err || last) break;
					}
				}
				const double t2 = now_s(); t_load += t2 - t1;
				AlIngestResult res; const bool ends[2] = {true, true};
				if (!err && al_stream_parse(S, ends, 1 << 30, &res)) err = al_nomem_flag() ? AL_ERR_NOMEM : -1;
				if (!err) {
					bool whole = res.n_frag > 0;
					for (int i = 0; i < n_fn; ++i) if (res.consumed[i] != S.txt_n[i] || res.first_bad[i] != ~0ULL) whole = false;
					if (!whole) { fprintf(stderr, "[airlift] range %d of the input (bytes %lld ... of '%s') is not whole four-line FASTQ records pairing up between the files: not supported in a multi-process run\n", j, range->rstart[0][j], fn[0]); err = -1; }
				}
				t_parse += now_s() - t2;
				al_acct() = nullptr;
				if (err) { fail(err); break; }
				n_reads_total += (uint64_t)res.n_reads; n_frag_total += (uint64_t)res.n_frag;
				{ std::lock_guard<std::mutex> l(m); sl->res = res; sl->seq = kk; sl->state = SL_READY; }
				cv.notify_all();
				++kk;
			}
			{ std::lock_guard<std::mutex> l(m); n_batches = kk; }
			cv.notify_all();
		} else {
		std::vector<char> carry[2]; long long base_off[2] = {range->start[0], range->start[1]}; bool eof[2] = {false, false};
		for (int i = 0; i < n_fn; ++i) if (rd[i] && rd[i]->file_size() == 0) eof[i] = true;
		double bytes_per_read = 360.0;                      // refined from every batch
		uint64_t k = 0; bool stop = false;
		while (!stop) {
			Slot *sl = slot_of(k);
			{
				const double t0 = now_s();
				std::unique_lock<std::mutex> l(m);
				cv.wait(l, [&] { return rc != 0 || sl->state == SL_FREE; });
				t_wait_slot += now_s() - t0;
				if (rc != 0) break;
			}
			al_acct() = &sl->held;
			AlStreamSlot &S = sl->S;
			// The second batch is the size the run keeps unless the input is long (growing later re-obtains every workspace): 4 x the first
			// probe, 8 x when the input has at least 6 M reads and this process has been getting device memory fast so far (index, slots).
			if (k == 0 && !sized.load() && !getenv("AL_PROBE_READS") && !getenv("AL_TWO_PROBES")) {
				double fs = 0; for (int i = 0; i < n_fn; ++i) fs += (double)rd[i]->file_size();
				const double est0 = fs / bytes_per_read / (double)NL;                 // (360 bytes per read until a batch has been seen)
				if (est0 >= 1.0e6) {
					const AlAllocStat &as = al_alloc_stat(); const double ns = (double)as.dev_ns.load(), by = (double)as.dev_bytes.load();
					probe_mult = getenv("AL_PROBE_MULT") ? std::max(1, atoi(getenv("AL_PROBE_MULT"))) : est0 >= 6.0e6 && ns > 0 && by / (ns * 1e-9) >= 100e9 ? 8 : 4;
					single_probe = true;
					int64_t first = (int64_t)probe_mult * probe_reads;
					if (long_input && !getenv("AL_PROBE_MULT")) {   // at least three batches per context, at most 2^20 reads, within 60 % of the free memory at ~80 KB per read
						size_t free_b = 0, total_b = 0;
						double b = std::min(al_long_batch_cap(est0 * (double)NL), std::max(262144.0, est0 / (3.0 * n_ctx_lane)));   // (524 288 reads; 2^20 for very long inputs: al_runtime.hip)
						if (hipSetDevice(mappers[0]->device) == hipSuccess && al_dev_mem_info(&free_b, &total_b) == hipSuccess) b = std::min(b, 0.6 * (double)free_b / ((double)n_ctx_lane * 81920.0));
						first = std::max<int64_t>(first, (int64_t)b);
					}
					max_reads = (int)std::max<int64_t>(2, std::min<int64_t>(first, k_bases / 64));
				}
			}
			if (k == 1 && probe_mult == 0) {
				probe_mult = 4;
				if (getenv("AL_PROBE_MULT")) probe_mult = std::max(1, atoi(getenv("AL_PROBE_MULT")));
				else { const AlAllocStat &as = al_alloc_stat(); const double ns = (double)as.dev_ns.load(), by = (double)as.dev_bytes.load();
				       if (est_total_reads.load() / NL >= 6000000 && ns > 0 && by / (ns * 1e-9) >= 100e9) probe_mult = 8; }
			}
			int mr = (!sized.load() && k > 0) ? std::min((probe_mult ? probe_mult : 4) * probe_reads, reads_cap_k.load()) : max_reads.load();
			{   // a slot's text block of one file stays below 2^31 bytes (al_stream_begin_text refuses more): long reads / huge batches get fewer reads per batch
				size_t cmax = 0; for (int i = 0; i < n_fn; ++i) cmax = std::max(cmax, carry[i].size());
				const double room = 1.9e9 - (double)cmax - 2.0 * (double)(PIECE + 16);
				const double lim = room > 0 ? room / (bytes_per_read * 1.02) * (double)n_fn : 2.0;
				if ((double)mr > lim) mr = (int)std::max(2.0, lim);
			}
			int err = 0;
			const double t1 = now_s();
			for (int i = 0; i < n_fn && !err; ++i) {
				// enough text for this file's share of the batch; at least one more piece unless the file is exhausted
				const size_t share = (size_t)((double)(mr / n_fn) * bytes_per_read * 1.02) + 4096;
				const size_t want = share > carry[i].size() ? share - carry[i].size() : 0;
				const size_t n_pc = eof[i] ? 0 : (want + PIECE - 1) / PIECE;
				if (al_stream_begin_text(S, i, carry[i].size() + n_pc * (PIECE + 16) + 64)) { err = -1; break; }
				if (!carry[i].empty() && al_stream_append_text(S, i, carry[i].data(), carry[i].size())) { err = -1; break; }
				for (size_t j = 0; j < n_pc && !eof[i]; ++j) {
					Piece pc; const double tw = now_s();
					if (!rd[i]->next(pc)) { if (rd[i]->has_failed()) { fprintf(stderr, "[airlift] reading '%s' failed\n", fn[i]); err = -1; } eof[i] = true; break; }
					t_wait_read += now_s() - tw;
					if (al_stream_append_text(S, i, pc.p, pc.n)) err = -1;
					if (pc.last) eof[i] = true;
					rd[i]->release(pc);
					if (err) break;
				}
			}
			const double t2 = now_s(); t_load += t2 - t1;
			AlIngestResult res;
			if (!err && al_stream_parse(S, eof, mr, &res)) err = al_nomem_flag() ? AL_ERR_NOMEM : -1;
			if (err) { al_acct() = nullptr; fail(err); break; }
			// the text this batch does not take: back to the host, in front of the next batch's pieces
			bool bad = false, all_eof = true; size_t left = 0;
			for (int i = 0; i < n_fn; ++i) {
				const uint64_t c = res.consumed[i], tn = S.txt_n[i];
				carry[i].resize((size_t)(tn - c));
				if (tn > c && al_stream_fetch_text(S, i, c, tn - c, carry[i].data())) err = -1;
				base_off[i] += (long long)c; left += carry[i].size();
				if (res.first_bad[i] != ~0ULL && res.first_bad[i] <= (n_fn == 2 ? (uint64_t)res.n_frag : (uint64_t)res.n_reads)) bad = true;
				if (!eof[i]) all_eof = false;
			}
			t_parse += now_s() - t2;
			al_acct() = nullptr;
			if (err) { fail(err); break; }
			if (res.n_reads > 0) {
				double used = 0; for (int i = 0; i < n_fn; ++i) used += (double)res.consumed[i];
				bytes_per_read = used / (double)res.n_reads;
				if (est_total_reads.load() == 0) { double fs = 0; for (int i = 0; i < n_fn; ++i) fs += (double)rd[i]->file_size(); est_total_reads = (long long)(fs / bytes_per_read); }
				if (reads_cap_k.load() == (1 << 30)) {          // -K bases as reads: a record is about 2 L + name + 6 bytes
					const double L = std::max(1.0, (bytes_per_read - 16.0) / 2.0);
					reads_cap_k = (int)std::max(2.0, std::min(1.0e9, (double)k_bases / L));
					if (max_reads.load() > reads_cap_k.load()) max_reads = reads_cap_k.load();
				}
			}
			// No fragment in the text: a record the strict grammar does not take is at its front, or one file has ended, or what is left at the
			// end of the input forms no fragment (blank lines, the longer file's extra records): the general reader continues there.
			if (res.n_frag == 0) { stop = true; if (left > 0 || !all_eof) { rs->resume = true; for (int i = 0; i < n_fn; ++i) rs->off[i] = base_off[i]; } }
			else if (all_eof && left == 0) stop = true;
			(void)bad;
			if (res.n_frag > 0) {
				n_reads_total += (uint64_t)res.n_reads; n_frag_total += (uint64_t)res.n_frag;
				{ std::lock_guard<std::mutex> l(m); sl->res = res; sl->seq = k; sl->state = SL_READY; }
				cv.notify_all();
				++k;
			}
		}
		{ std::lock_guard<std::mutex> l(m); n_batches = k; }
		cv.notify_all();
		}
	}
	writer.join();
	if (ranged) {   // the rounds this process has no batch in (or did not get to): its peers wait for every process in every round
		int okr; { std::lock_guard<std::mutex> l(m); okr = rc == 0 ? 1 : 0; }
		for (; sink_rounds < range->sink->n_rounds; ++sink_rounds) {
			const long long off = range->sink->offset_of(range->sink->ctx, sink_rounds, 0, sink_rounds == 0 ? pre_bytes : 0, okr);
			if (off < 0) { if (okr) fail(-4); break; }
			if (!okr) break;                                 // (one round with ok = 0 tells everybody)
		}
	}
	if (woff >= 0) (void)lseek(ofd, (off_t)woff, SEEK_SET);           // whoever writes next (the host driver taking over, the caller) continues behind the text
	for (auto &mp : mappers) mp->th.join();
	const double T2 = now_s();
	if (timing) {
		double ts = 0, tr = 0, tm = 0; for (auto &mp : mappers) { ts += mp->t_setup; tr += mp->t_run; tm += mp->t_sam; }
		fprintf(stderr, "[airlift] stream pipeline: %d lane(s) x (%d context(s), %d slots); init %.3f s; %llu fragments, %llu reads, %llu records, %.1f MB of SAM in %.3f s (%.2f M reads/s); ingest: wait-slot %.3f load %.3f (wait-read %.3f) parse+carry %.3f; mappers (sum): setup %.3f run %.3f sam %.3f; writer %.3f; total %.3f s\n",
		        NL, n_ctx_lane, n_slots_lane, T1 - T0, (unsigned long long)n_frag_total, (unsigned long long)n_reads_total, (unsigned long long)recs_out, bytes_out / 1e6, T2 - T1, n_reads_total / std::max(1e-9, T2 - T1) / 1e6,
		        t_wait_slot, t_load, t_wait_read, t_parse, ts, tr, tm, t_write, T2 - T0);
		for (auto &mp : mappers) fprintf(stderr, "[airlift] pipeline lane %d (device %d): context %d: %d batches, wait %.3f setup %.3f run %.3f sam %.3f; workspaces %.1f MB; total %.3f s\n", mp->lane, mp->device, mp->idx, mp->n_batch, mp->t_wait, mp->t_setup, mp->t_run, mp->t_sam, mp->held.load() / 1e6, T2 - T1);
		{ AlAllocStat &a = al_alloc_stat(); fprintf(stderr, "[airlift] allocation calls of the process so far: device %lld calls, %.1f GB, %.3f s (hipMalloc + hipFree); page-locked host %lld calls, %.1f MB, %.3f s\n", (long long)a.dev_calls, a.dev_bytes / 1e9, a.dev_ns / 1e9, (long long)a.host_calls, a.host_bytes / 1e6, a.host_ns / 1e9); }
		size_t sh = 0; for (auto &sp : slots) sh += sp->held.load();
		fprintf(stderr, "[airlift] stream pipeline: text / SAM buffers of the %d slots: %.1f MB\n", NS, sh / 1e6);
	}
	ring_free(wring); for (auto &r : mrings) ring_free(r);
	destroy_all();
	(void)NM;
	return rc;
}
