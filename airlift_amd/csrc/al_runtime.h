// al_runtime.h -- per-context device state of the re-alignment pipeline (product code)
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string>
#include <thread>
#include <chrono>
#include "al_internal.h"

// Set when a device allocation fails; al_batch_run reports AL_ERR_NOMEM then (the context stays usable: every DevBuf is
// either at its old size or empty, and the caller may upload a smaller batch).
inline bool &al_nomem_flag() { static thread_local bool f = false; return f; }

// Device ranges of the grow-only arenas and the index builder's temporaries, behind one pair of functions so that the test switches
// AL_TEST_POISON / AL_TEST_GUARD (al_runtime.hip) see every range.  With a reserve (al_device_reserve: chunks obtained by a background thread) they are
// served from it, best fit, no driver call; without one, or for a request larger than a chunk, by hipMalloc / hipFree.
// al_dev_free does NOT synchronise the device (hipFree did): a range goes back to the reserve at once and may be handed to another context's thread in the
// next microsecond, so the caller frees a range only when no kernel or copy that uses it can still be in flight -- DevBuf::ensure(!keep) / release() are
// called between a context's batches, on the thread that has just synchronised its streams; ctx_release_buffers drains every stream first.
hipError_t al_dev_malloc(void **p, size_t bytes);
// Accounting: while a thread has set al_acct() to a counter, the bytes of every range it allocates are added to it (and taken off
// again when the range is freed, by whichever thread): the stream driver sizes its batches from what one batch held.
#include <atomic>
std::atomic<size_t> *&al_acct();
// time this process has spent in device / page-locked host allocation calls (AL_TIMING report)
struct AlAllocSite { const char *file; int line; };
AlAllocSite &al_alloc_site();      // who asked (AL_TRACE_ALLOC lists the large allocations with their call site)
struct AlAllocStat { std::atomic<long long> dev_ns{0}, dev_bytes{0}, dev_calls{0}, host_ns{0}, host_bytes{0}, host_calls{0}; };
AlAllocStat &al_alloc_stat();
void al_dev_free(void *p);
double al_long_batch_cap(double reads); // the stream driver's bound on a long input's batches (reads per batch) for an input of about `reads` reads: AL_LONG_BATCH, else 524 288, 2^20 from 200 M reads (al_runtime.hip)
long long al_dev_reserve_room();      // bytes the process's reserve still holds free (or is yet to obtain) on the current device; -1 = no reserve there
hipError_t al_dev_mem_info(size_t *free_b, size_t *total_b);   // hipMemGetInfo plus what the process's reserve (al_device_reserve) holds free
int al_dev_guard_check();            // AL_TEST_GUARD=1: number of live ranges whose guard zones were written (messages on stderr)

template <typename T> struct DevBuf {       // grow-only device array
	T *p = nullptr; size_t cap = 0;
	int ensure(size_t n, bool keep = false, hipStream_t s = 0, int line = __builtin_LINE(), const char *file = __builtin_FILE())
	{
		if (n <= cap) return 0;
		al_alloc_site() = AlAllocSite{file, line};
		static const size_t big_div = getenv("AL_GROW_DIV") ? (size_t)std::max(1, atoi(getenv("AL_GROW_DIV"))) : 8;
		const size_t ncap = n + (n * sizeof(T) >= ((size_t)256 << 20) ? n / big_div : n / 4) + 64;   // headroom against regrowing: an eighth for the large arrays (batches of one run differ by a few per cent), a quarter for the small
		T *np = nullptr;
		if (!keep && p) { al_dev_free(p); p = nullptr; cap = 0; }         // contents not needed: release first, so the peak is one copy
		if (al_dev_malloc((void **)&np, ncap * sizeof(T)) != hipSuccess) {
			(void)hipGetLastError();                                       // not sticky: the next launch check must not see it
			fprintf(stderr, "[airlift] device allocation of %zu bytes failed\n", ncap * sizeof(T)); al_nomem_flag() = true; return -1;
		}
		if (keep && p && cap) { if (hipMemcpyAsync(np, p, cap * sizeof(T), hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -1; }
		if (p) al_dev_free(p);
		p = np; cap = ncap;
		return 0;
	}
	void release() { if (p) al_dev_free(p); p = nullptr; cap = 0; }
};

struct AlRegOut {            // per-read result header copied back to the host
	uint32_t n_regs, reg_off;   // reg_off: index into the AlReg output array
};

struct al_ctx_s {
	const al_idx_t *mi = nullptr;
	al_mapopt_t opt;
	AlParams P;
	int device = 0;
	int n_threads = 1;                    // host worker threads for packing (al_ctx_set_threads)
	hipStream_t stream = nullptr, side = nullptr;   // side: exact (serial) handling of the few fragments with equal-x anchors, next to the main pipeline
	hipEvent_t ev_fj[2] = {};             // fork / join of the side stream inside a stage (chain_post classes, DP job classes)
	hipStream_t aux[3] = {};              // more streams for stages made of independent latency-bound launches over disjoint fragments (heap merge classes, k_regs_heavy tiles)
	hipEvent_t ev_aux[3] = {};            // ... their join events
	hipStream_t ovl[3] = {};              // stages of the seed pass that run beside the main stream's: the device-wide anchor sort next to the block sorts, the lane chaining kernels next to the tile kernel
	hipEvent_t ev_ovl[5] = {};            // ... fork and join events of the two
	bool ovl_pending = false;             // chaining kernels in flight on ovl[1]: chain_tiles joins them before it reuses their scratch
	hipEvent_t ev_side[4] = {};           // [0],[1]: start / end of the side stream's work in the first pass, [2],[3]: in the re-chain pass
	// the equal-x merge of giant fragments started ahead of the re-chain pass (al_kernels_seed.hip: k_spec_build): its stream, events, slots
	hipStream_t spec = nullptr, spec2 = nullptr; hipEvent_t ev_spec[3] = {}; uint32_t n_spec = 0, spec_idle = 0, spec_batch = 0; bool spec_pending = false, spec_busy = false;   // spec_idle: batches in a row whose re-chain pass took none of the merges made ahead
	DevBuf<AlMatch> spec_match; DevBuf<uint32_t> spec_meta, spec_cnt, spec_use, spec_v32, spec_na2; DevBuf<uint64_t> spec_v64; DevBuf<AlAnchor> spec_anchors;
	float ms_side = 0;
	AlDevIndex di;
	hipEvent_t ev[ST_N + 1] = {};
	float ms_stage[ST_N] = {};
	float ms_total = 0;

	// resident batch (host mirrors kept for fetch / taps)
	int n_frag = 0, n_reads = 0;
	uint64_t n_bases = 0, mini_total = 0, seq_words = 0;
	std::vector<uint32_t> h_rd_len, h_frag_first, h_frag_hash;
	std::vector<uint64_t> h_rd_off, h_mini_off;
	std::vector<uint32_t> h_rd_seq;
	std::vector<uint8_t> h_flip;          // read was reverse-complemented for mapping (worker_for, map.c:468)

	DevBuf<uint32_t> rd_seq, rd_len, frag_first, frag_hash, mini_cnt, frag_nm, frag_na, frag_nu, rechain_list, rechain_sorted, tmp_u32;
	DevBuf<uint64_t> rd_off, mini_off, a_off, u, ws_u64, tmp_u64, tmp_u64b;
	DevBuf<uint32_t> uo;                   // per chain list entry: offset of the chain's first anchor in the fragment's range of chained[]
	DevBuf<uint64_t> big_k0, big_k1;       // key double buffer of the device-wide anchor sort (fragments above the register tiles)
	// compact copy of the fragments the tile chaining kernel hands back (al_runtime.hip: chain_fallback): a small virtual batch for the segment-wise kernels
	DevBuf<AlAnchor> v_anchors, v_chained; DevBuf<uint64_t> v_u, v_a_off, v_first64; DevBuf<uint32_t> v_na, v_nseg, v_first, v_rd_len, v_order, v_nu, fbk_list;
	DevBuf<uint64_t> d_uslot; DevBuf<uint32_t> d_rel, d_fragid, cmp_list, ctie, frag_meta, chain_cls;   // deferred segments of the tile kernel (ChainSeg direct mode), fragments whose chain lists have gaps, per-fragment tie flags
	DevBuf<int32_t> frag_rep, ws_i32;
	DevBuf<AlAnchor> mini, heap_ws, anchors, chained;
	DevBuf<AlMatch> match;
	DevBuf<unsigned long long> counters;   // [0] heap fallbacks, [1] sort-tie flags, [2] alser total, [3] n_rechain, [4..] stage specific
	DevBuf<uint8_t> scan_tmp;
	DevBuf<uint8_t> big_tmp;               // rocprim scratch of the device-wide anchor sort (it runs on ovl[0], beside users of scan_tmp)
	DevBuf<uint32_t> chain_key, chain_idx, chain_idx2, tie_list, lb_buf;
	// segment-wise chaining of large fragments (al_runtime.hip: chain_by_segments) and the device-wide sort of their anchors
	DevBuf<AlAnchor> chain_tmp; DevBuf<uint64_t> u_tmp, okey_tmp, seg_first, seg_first0, vs_off, big_off, big_toff;
	DevBuf<uint64_t> vs_res;                 // two words per segment: ChainSeg::res
	DevBuf<uint32_t> seg_cnt, seg_cnt0, seg_t1, vs_na, vs_meta, vs_cls, seg_key, seg_idx, seg_ord, fb_list, fb2_list, fb3_list, big_na, big_nt, big_tent, big_cuts, tie_frags, tie_sorted, heap_cnt;
	uint64_t n_chain_fallback = 0;
	int max_qlen_sum = 0;                 // longest fragment of the resident batch
	int max_rd_len = 0;                   // longest read of the resident batch
	bool attr_chain_order = false, attr_regs_heavy = false;   // > 64 KB dynamic-LDS opt-in of two kernels: per device (hipFuncSetAttribute acts on the current device), kept per context
	bool no_taps = false;                 // the drivers' contexts: nobody reads the sorted anchors after chaining (al_dbg_* taps, candidate counting), so the alignment stage's per-mate anchor arrays take their place (16 bytes per seed hit less)
	bool dev_batch = false;               // the batch was parsed and packed on the device (al_stream.hip): no host mirrors of the read arrays
	DevBuf<uint64_t> a_off_p1; DevBuf<uint32_t> frag_na_p1; DevBuf<int32_t> frag_rep_p1;   // pass-1 snapshots when a re-chain pass ran (taps)
	uint64_t n_anchor_total = 0, n_anchor_pass1 = 0;
	double anchor_grow_hw = 1.3;          // (starts at what a repeat-rich genome needs: C4 1.26; a batch without re-seeded fragments leaves it unused) largest (anchors after the re-seeding pass) / (anchors of the first pass) this context has seen: the first pass reserves for it, so that the second does not reallocate
	uint32_t n_rechain = 0;

	// alignment stage (al_kernels_align.hip)
	DevBuf<AlReg> regs0, regs;             // fragment-level chains -> per-mate regions
	DevBuf<uint32_t> reg_cnt, cigar;
	DevBuf<uint64_t> reg_off, cig_off;
	DevBuf<uint8_t> align_ws;
	DevBuf<AlAnchor> seg_a;
	DevBuf<uint64_t> seg_u;
	uint64_t n_regs_cap_total = 0, n_cigar_cap_total = 0;
	bool ran = false;
	uint64_t stat_bytes_in = 0;
	uint32_t stat_n_slow = 0;
	unsigned long long stat_dp_jobs[10] = {}, stat_dp_tbases[10] = {};   // extension DP: jobs and target (reference window) bases per job class of the last run
	al_batch_stat_t stat;
};

int al_upload_index(const al_idx_t *mi, int device, AlDevIndex *out);
void al_ctx_no_taps(al_ctx_t *c);         // see al_ctx_s::no_taps
int al_run_align_stage(al_ctx_t *c);      // al_kernels_align.hip: KA (regs) + K5 (extension, MAPQ, pairing)
int al_fetch_align(al_ctx_t *c, int *n_regs, al_reg1_t **regs, int *rep_len);
void al_align_grow_arena(al_ctx_t *c);

template <typename T> struct PinnedVec {    // grow-only page-locked host array (fast, asynchronous-capable D2H target)
	T *p = nullptr; size_t n = 0, cap = 0;
	PinnedVec() {}
	PinnedVec(const PinnedVec &) = delete; PinnedVec &operator=(const PinnedVec &) = delete;
	~PinnedVec() { if (p) (void)hipHostFree(p); }
	int resize(size_t m)
	{
		if (m > cap) {
			const size_t ncap = m + m / 4 + 1024;
			T *np = nullptr;
			const auto t0__ = std::chrono::steady_clock::now();
			const hipError_t e__ = hipHostMalloc((void **)&np, ncap * sizeof(T), hipHostMallocDefault);
			{ AlAllocStat &a = al_alloc_stat(); a.host_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0__).count(); a.host_bytes += (long long)(ncap * sizeof(T)); ++a.host_calls; }
			if (e__ != hipSuccess) { fprintf(stderr, "[airlift] hipHostMalloc of %zu bytes failed\n", ncap * sizeof(T)); return -1; }
			if (p) (void)hipHostFree(p);
			p = np; cap = ncap;
		}
		n = m; return 0;
	}
	T *data() { return p; } const T *data() const { return p; }
	size_t size() const { return n; }
	T &operator[](size_t i) { return p[i]; } const T &operator[](size_t i) const { return p[i]; }
};

struct AlRawResult {         // flat host copy of one batch's results (no per-record allocation); reusable across batches
	PinnedVec<uint64_t> off;              // per read: first record in out[]; off[n_reads] = total
	PinnedVec<AlReg> out;
	PinnedVec<uint32_t> arena;            // CIGAR words of records with more than 4 operations
	PinnedVec<int32_t> rep;               // per fragment repeat length
	std::vector<uint8_t> flip;            // read was mapped reverse-complemented
	std::vector<uint32_t> rd_len;
};
int  al_fetch_raw(al_ctx_t *c, AlRawResult &R);
void al_reg_from_raw(const AlRawResult &R, int read, int k, al_reg1_t &q);

// host worker pool helper: fn(lo, hi, thread) over [0, n) split into contiguous ranges
template <class F> static inline void al_parallel_for(int n_threads, size_t n, F fn)
{
	if (n_threads <= 1 || n < 2) { fn((size_t)0, n, 0); return; }
	const size_t nt = (size_t)n_threads < n ? (size_t)n_threads : n;
	std::vector<std::thread> th; th.reserve(nt);
	for (size_t t = 0; t < nt; ++t) th.emplace_back(fn, n * t / nt, n * (t + 1) / nt, (int)t);
	for (auto &x : th) x.join();
}
