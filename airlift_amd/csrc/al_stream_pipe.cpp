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
			const long long off = start + pc.idx * (long long)piece;
			size_t want = (size_t)std::min<long long>((long long)piece, size - off), got = 0;
			while (got < want) { const ssize_t k = pread(fd, pc.p + got, want - got, (off_t)(off + (long long)got)); if (k <= 0) break; got += (size_t)k; }
			pc.n = got; pc.last = pc.idx == n_pieces - 1;
			if (pc.last && got > 0 && pc.p[got - 1] != '\n') pc.p[pc.n++] = '\n';     // an unterminated last line still counts (kseq.h)
			{ std::lock_guard<std::mutex> l(m); if (got < want) failed = true; done.push_back(pc); }
			cv.notify_all();
		}
	}
public:
	bool open(const char *fn, long long start_off, size_t piece_bytes, int n_buffers, int n_threads)
	{
		fd = ::open(fn, O_RDONLY);
		struct stat sb;
		if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return false;
		size = (long long)sb.st_size; start = start_off; piece = piece_bytes; n_buf = n_buffers;
		n_pieces = size > start ? (size - start + (long long)piece - 1) / (long long)piece : 0;
		for (int i = 0; i < n_buf; ++i) { char *p = nullptr; if (hipHostMalloc((void **)&p, piece + 16, hipHostMallocDefault) != hipSuccess) return false; bufs.push_back(p); free_bufs.push_back(i); }
		for (int t = 0; t < n_threads; ++t) workers.emplace_back(&PieceReader::work, this);
		return true;
	}
	long long file_size() const { return size; }
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
struct Slot {
	AlStreamSlot S; int lane = 0, device = 0;
	int state = SL_FREE; uint64_t seq = 0;
	AlIngestResult res;
	std::atomic<size_t> held{0};                        // device bytes this slot's context and buffers hold
	std::vector<std::vector<char>> chunks;              // SAM text of a batch that had to be cut (else it is in S.h_sam)
	std::thread mapper;
	double t_setup = 0, t_run = 0, t_sam = 0; int n_batch = 0;
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
                        const int *devices, int n_dev, AlStreamResume *rs)
{
	if (getenv("AL_HOST_IO") || n_fn < 1 || n_fn > 2 || n_dev < 1) return AL_STREAM_NA;
	for (int i = 0; i < n_fn; ++i) if (!eligible_file(fn[i])) return AL_STREAM_NA;
	const bool timing = getenv("AL_TIMING") != nullptr, trace = getenv("AL_TRACE") != nullptr;
	const double T0 = now_s();
	const int n_slots_lane = std::max(1, std::min(8, getenv("AL_SLOTS") ? atoi(getenv("AL_SLOTS")) : 3));
	const int NL = n_dev, NS = NL * n_slots_lane;
	const size_t PIECE = (size_t)(getenv("AL_PIECE_MB") ? std::max(1, atoi(getenv("AL_PIECE_MB"))) : 8) << 20;

	std::mutex m; std::condition_variable cv; int rc = 0;
	auto fail = [&](int code) { { std::lock_guard<std::mutex> l(m); if (rc == 0) rc = code ? code : -1; } cv.notify_all(); };

	std::vector<std::unique_ptr<Slot>> slots;
	for (int i = 0; i < NS; ++i) {          // batch k goes to slot k % NS: lane (k % NL), so consecutive batches sit on different GPUs
		std::unique_ptr<Slot> sl(new Slot()); sl->lane = i % NL; sl->device = devices[i % NL];
		al_acct() = &sl->held;
		const int e = al_stream_slot_init(sl->S, mi, opt, sl->device, n_fn);
		al_acct() = nullptr;
		if (e) { for (auto &s : slots) al_stream_slot_destroy(s->S); al_stream_slot_destroy(sl->S); return -2; }
		slots.push_back(std::move(sl));
	}
	const double T1 = now_s();
	char rg_id[256]; rg_id[0] = 0;
	if (rg != (const char *)-1) al_write_sam_hdr(out, mi, rg, rg_id);
	fflush(out);
	const int ofd = fileno(out);
	memcpy(rs->rg_id, rg_id, 256);

	// batch size in reads.  Upper bound: mini_batch_size bases (-K); the first batches are small probes, then what they held per read
	// decides how many reads fill the free device memory.
	const int64_t k_bases = getenv("AL_AUTO_BATCH") ? (int64_t)1 << 40 : opt->mini_batch_size > 0 ? (int64_t)opt->mini_batch_size : 50000000;
	std::atomic<int> max_reads{0}; std::atomic<bool> sized{false};
	const int probe_reads = getenv("AL_PROBE_READS") ? std::max(2, atoi(getenv("AL_PROBE_READS"))) : 65536;
	std::atomic<int> reads_cap_k{1 << 30};                 // mini_batch_size in reads, once a read length is known
	max_reads = probe_reads;
	if (getenv("AL_BATCH_READS")) { max_reads = std::max(2, atoi(getenv("AL_BATCH_READS"))); sized = true; }    // tests / tuning: fixed batches

	// ---- mappers -------------------------------------------------------------------------------------------------------------
	uint64_t n_batches = ~0ULL;                            // set by the ingest thread when the input is exhausted
	for (auto &sp : slots) {
		Slot *sl = sp.get();
		sl->mapper = std::thread([&, sl]() {
			al_acct() = &sl->held;
			AlStreamSlot &S = sl->S;
			for (;;) {
				{
					std::unique_lock<std::mutex> l(m);
					cv.wait(l, [&] { return rc != 0 || sl->state == SL_READY || n_batches != ~0ULL; });
					if (rc != 0) return;
					if (sl->state != SL_READY) return;               // the input is exhausted and every batch dealt to this slot is done (batches are marked READY before n_batches is set)
				}
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
					int r = al_stream_setup(S, rlo, rhi, flo, fhi);
					if (r == 0) { const double t1 = now_s(); sl->t_setup += t1 - t0; r = al_batch_run(S.ctx); sl->t_run += now_s() - t1; }
					else if (al_nomem_flag()) r = AL_ERR_NOMEM;
					if (r == AL_ERR_NOMEM && fhi - flo > 1) {
						const uint32_t mid = flo + (fhi - flo) / 2;
						fprintf(stderr, "[airlift] batch of %u fragments does not fit the device workspaces: running it as %u + %u\n", fhi - flo, mid - flo, fhi - mid);
						if (n_fn == 1 && fstart.empty() && al_stream_frag_starts(S, res, fstart)) return -1;
						cut = true;
						{ int mr = max_reads.load(); const int half = std::max(2, (int)((uint64_t)(n_fn == 2 ? 2 : 1) * (fhi - flo) / 2)); if (half < mr) max_reads = half; }
						if ((r = process(flo, mid)) != 0) return r;
						return process(mid, fhi);
					}
					if (r != 0) return r;
					const double t2 = now_s();
					if ((r = al_stream_sam(S, rg_id)) != 0) return r;
					sl->t_sam += now_s() - t2;
					if (cut) sl->chunks.emplace_back(S.h_sam.data(), S.h_sam.data() + S.sam_bytes);
					return 0;
				};
				const int r = process(0, (uint32_t)res.n_frag);
				if (r != 0) { fail(r); return; }
				++sl->n_batch;
				if (!sized.load() && res.n_reads >= std::min(probe_reads, 1024)) {
					// what this batch held per read -> reads that fill the device's free memory, shared by the lane's slots
					size_t free_b = 0, total_b = 0;
					if (hipSetDevice(sl->device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
						size_t held_dev = 0; for (auto &o : slots) if (o->device == sl->device) held_dev += o->held.load();
						const double per_read = (double)sl->held.load() / (double)res.n_reads;
						const double budget = ((double)free_b + (double)held_dev) * 0.90 / (double)n_slots_lane;
						double mr = budget / (per_read * 1.30);         // arenas grow in steps of 25 %; seed hits per read vary between batches
						mr = std::min(mr, (double)reads_cap_k.load()); mr = std::min(mr, 4.0e6); mr = std::max(mr, (double)std::min(probe_reads, res.n_reads));
						bool exp = false;
						if (sized.compare_exchange_strong(exp, true)) {
							max_reads = (int)mr;
							if (timing || trace) fprintf(stderr, "[airlift] stream driver: probe batch of %d reads held %.1f MB (%.0f bytes per read); %.1f GB free on device %d -> batches of %d reads on %d slots\n",
							                             res.n_reads, sl->held.load() / 1e6, per_read, free_b / 1e9, sl->device, (int)mr, n_slots_lane);
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
	std::thread writer([&]() {
		for (uint64_t k = 0;; ++k) {
			Slot *sl = slots[k % NS].get();
			{
				std::unique_lock<std::mutex> l(m);
				cv.wait(l, [&] { return rc != 0 || (sl->state == SL_MAPPED && sl->seq == k) || (n_batches != ~0ULL && k >= n_batches); });
				if (rc != 0 || (n_batches != ~0ULL && k >= n_batches)) return;
			}
			const double t0 = now_s();
			auto put = [&](const char *p, size_t n) -> bool { while (n) { const ssize_t w = write(ofd, p, n); if (w <= 0) return false; p += w; n -= (size_t)w; } return true; };
			bool ok = true;
			if (!sl->chunks.empty()) { for (auto &c : sl->chunks) { ok = ok && put(c.data(), c.size()); bytes_out += c.size(); } sl->chunks.clear(); }
			else { ok = put(sl->S.h_sam.data(), sl->S.sam_bytes); bytes_out += sl->S.sam_bytes; }
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
		for (int i = 0; i < n_fn; ++i) { rd[i].reset(new PieceReader()); if (!rd[i]->open(fn[i], 0, PIECE, 2 * rd_threads + 2, rd_threads)) open_ok = false; }
		if (!open_ok) fail(-1);
		std::vector<char> carry[2]; long long base_off[2] = {0, 0}; bool eof[2] = {false, false};
		for (int i = 0; i < n_fn; ++i) if (rd[i] && rd[i]->file_size() == 0) eof[i] = true;
		double bytes_per_read = 360.0;                      // refined from every batch
		uint64_t k = 0; bool stop = false;
		while (!stop) {
			Slot *sl = slots[k % NS].get();
			{
				const double t0 = now_s();
				std::unique_lock<std::mutex> l(m);
				cv.wait(l, [&] { return rc != 0 || sl->state == SL_FREE; });
				t_wait_slot += now_s() - t0;
				if (rc != 0) break;
			}
			al_acct() = &sl->held;
			AlStreamSlot &S = sl->S;
			const int mr = max_reads.load();
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
	writer.join();
	for (auto &sp : slots) sp->mapper.join();
	const double T2 = now_s();
	if (timing) {
		double ts = 0, tr = 0, tm = 0; for (auto &sp : slots) { ts += sp->t_setup; tr += sp->t_run; tm += sp->t_sam; }
		fprintf(stderr, "[airlift] stream pipeline: %d lane(s) x %d slots; slot init %.3f s; %llu fragments, %llu reads, %llu records, %.1f MB of SAM in %.3f s (%.2f M reads/s); ingest: wait-slot %.3f load %.3f (wait-read %.3f) parse+carry %.3f; mappers (sum): setup %.3f run %.3f sam %.3f; writer %.3f; total %.3f s\n",
		        NL, n_slots_lane, T1 - T0, (unsigned long long)n_frag_total, (unsigned long long)n_reads_total, (unsigned long long)recs_out, bytes_out / 1e6, T2 - T1, n_reads_total / std::max(1e-9, T2 - T1) / 1e6,
		        t_wait_slot, t_load, t_wait_read, t_parse, ts, tr, tm, t_write, T2 - T0);
		for (auto &sp : slots) fprintf(stderr, "[airlift] pipeline lane %d (device %d): slot: %d batches, setup %.3f run %.3f sam %.3f; held %.1f MB; total %.3f s\n", sp->lane, sp->device, sp->n_batch, sp->t_setup, sp->t_run, sp->t_sam, sp->held.load() / 1e6, T2 - T1);
	}
	for (auto &sp : slots) { al_acct() = nullptr; al_stream_slot_destroy(sp->S); }
	return rc;
}
