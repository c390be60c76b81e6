// al_runtime.hip -- host driver of the device pipeline: index upload, batch packing, stage launches
// on one HIP stream with per-stage events, and the stage taps used by the parity tests.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include "al_internal.h"
#include "al_device.h"
#include <mutex>
#include <condition_variable>
#include <thread>
#include <atomic>
#include <map>
#include <vector>
#include "al_runtime.h"
#include "al_dev_sort.h"
#include "al_io.h"

// kernels (al_kernels_seed.hip)
extern "C" __global__ void k_sketch(const uint32_t *, const uint64_t *, const uint32_t *, const uint64_t *, AlAnchor *, uint32_t *, int, int, int, int);
extern "C" __global__ void k_seed(const uint64_t *, int, const uint32_t *, const uint32_t *, const uint64_t *, AlAnchor *, const uint32_t *, AlMatch *, uint32_t *, uint32_t *, int32_t *, const uint32_t *, int, int, int, uint32_t *);
#ifndef AL_SPEC_CAP
#define AL_SPEC_CAP 512            // slots of the merges made ahead of the re-chain pass
#endif
struct SpecOut { AlMatch *match; uint32_t *meta; uint32_t *cnt; uint64_t *cand; uint32_t cap, per, cand_cap; };
struct SpecView { uint32_t *first, *rdlen, *nm, *na, *tie, *list, *n_list; uint64_t *moff, *aoff; };
extern "C" __global__ void k_spec_count(const uint64_t *, int, const uint32_t *, const uint32_t *, const uint64_t *, const AlAnchor *, const uint32_t *, const int32_t *, int, int, uint32_t, SpecOut, int, const uint32_t *);
extern "C" __global__ void k_spec_pick(const uint64_t *, int, const uint32_t *, const uint32_t *, const uint64_t *, const AlAnchor *, const uint32_t *, int, SpecOut);
__global__ void k_spec_layout(const uint32_t *, uint32_t, uint32_t, SpecView);
__global__ void k_spec_mark(const uint32_t *, uint32_t, const uint32_t *, const uint32_t *, uint32_t *, uint32_t *, uint32_t *);
__global__ void k_spec_apply(const uint32_t *, const uint32_t *, const uint64_t *, const AlAnchor *, const uint64_t *, AlAnchor *);
extern "C" __global__ void k_alser_count(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, int, int, unsigned long long *);
extern "C" __global__ void k_rechain_test(const AlAnchor *, const uint64_t *, const uint64_t *, const uint32_t *, const uint32_t *, const int32_t *, const uint32_t *, int, uint32_t *, uint32_t *);
// al_kernels_chain.hip
struct TileSched { uint32_t n_items; uint32_t ent[7]; uint32_t item[7]; };
struct CtDefer { uint64_t *off, *uslot; uint32_t *na, *meta, *rel, *fragid, *cls; uint32_t *cnt; uint32_t cap; uint32_t *cmp_list, *cmp_cnt; uint32_t *ctie; };
__global__ void k_frag_meta(const uint32_t *, const uint32_t *, int, uint32_t *);
__global__ void k_chain_tile6(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, const TileSched, const uint32_t *, AlAnchor *, uint64_t *, uint32_t *, uint32_t *, uint32_t *, uint32_t *, const AlParams, const int, unsigned long long *, const int, const CtDefer);
template <int CAP> __global__ void k_chain_coop(const AlAnchor *, AlAnchor *, uint64_t *, uint32_t *, const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const uint32_t *, const uint32_t *, uint32_t *, const uint32_t *, int, const AlParams, unsigned long long *);
__global__ void k_u_compact(const uint32_t *, const uint32_t *, const uint64_t *, uint32_t *, uint64_t *, uint32_t *, const uint32_t *, uint32_t *, uint32_t *);
__global__ void k_uo_fill(const uint32_t *, int, const uint64_t *, const uint32_t *, const uint64_t *, uint32_t *, const uint32_t *);
__global__ void k_fb_meta(const uint32_t *, int, const uint32_t *, const uint32_t *, uint32_t *, uint32_t *);
__global__ void k_fb_reads(const uint32_t *, int, const uint32_t *, const uint32_t *, const uint64_t *, uint32_t *, uint32_t *, uint32_t *);
__global__ void k_fb_copy_in(const uint32_t *, int, const uint64_t *, const uint32_t *, const AlAnchor *, const uint64_t *, AlAnchor *);
__global__ void k_fb_copy_out(const uint32_t *, int, const uint64_t *, const uint64_t *, const uint32_t *, const uint64_t *, const AlAnchor *, uint32_t *, uint64_t *, uint32_t *, AlAnchor *);
template <int CAP> __global__ void k_anchor_sort(const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const AlMatch *, const uint32_t *, const uint32_t *, const uint64_t *, AlAnchor *, uint32_t *, unsigned int *, const uint32_t *, int, unsigned long long *, int);
__global__ void k_anchor_sort_small(const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const AlMatch *, const uint32_t *, const uint32_t *, const uint64_t *, AlAnchor *, uint32_t *, unsigned int *, const uint32_t *, int, unsigned long long *, int);
template <int HCAP, int LANES> __global__ void k_anchor_heap(const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const AlMatch *, const uint32_t *, const uint32_t *, const uint64_t *, AlAnchor *, AlAnchor *, const uint32_t *, const uint32_t *, int, int, unsigned long long *, int, const uint32_t *, uint32_t);
template <int MCAPH, int RING> __global__ void k_anchor_heap_wave(const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const AlMatch *, const uint32_t *, const uint32_t *, const uint64_t *, AlAnchor *, const uint32_t *, const uint32_t *, const uint32_t *, uint32_t, unsigned long long *, int);
template <int NSET, int RING> __global__ void k_anchor_heap_lanes(const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const AlMatch *, const uint32_t *, const uint32_t *, const uint64_t *, AlAnchor *, const uint32_t *, const uint32_t *, const uint32_t *, int, unsigned long long *, int);
template <int CAP> __global__ void k_chain(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, AlAnchor *, uint64_t *, uint32_t *, int32_t *, uint64_t *, const uint32_t *, int, AlParams, unsigned long long *, ChainSeg);
template <int CAPL, int LANES> __global__ void k_chain_lds(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, AlAnchor *, uint64_t *, uint32_t *, uint64_t *, const uint32_t *, int, int, AlParams, unsigned long long *, ChainSeg, uint32_t *, int);
template <int PER, int NW, int MCAP> __global__ void k_anchor_sort_reg(const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const AlMatch *, const uint32_t *, const uint32_t *, const uint64_t *, AlAnchor *, uint32_t *, const uint32_t *, int, int, int);
__global__ void k_anchor_big_expand(const uint64_t *, const uint64_t *, const uint32_t *, const AlMatch *, const uint32_t *, const uint32_t *, int, const uint64_t *, uint64_t *, int, int);
struct RunMergeOut { const uint64_t *a_off, *mini_off; const uint32_t *frag_first, *rd_len; const AlMatch *match; AlAnchor *anchors; uint32_t *tie_list; int rid_bits, mini_span; };
__global__ void k_big_tiles(const uint32_t *, int, uint32_t *, unsigned int *, uint32_t);
template <int PER, int NW, int MCAP> __global__ void k_anchor_run_sort(const uint64_t *, const uint32_t *, const uint64_t *, const AlMatch *, const uint32_t *, const uint32_t *, const uint32_t *, int, const uint64_t *, const uint32_t *, const uint64_t *, uint64_t *, uint32_t *, int, uint32_t, uint32_t);
__global__ void k_big_tile_ent(const uint64_t *, int, uint32_t, uint32_t *);
__global__ void k_anchor_run_cuts(const uint64_t *, const uint32_t *, const uint32_t *, uint32_t, const uint64_t *, const uint64_t *, const uint32_t *, const uint32_t *, int, uint32_t, uint32_t, uint32_t *);
__global__ void k_anchor_run_merge(const uint64_t *, uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint64_t *, const uint64_t *, const uint32_t *, const uint32_t *, int, RunMergeOut, uint32_t, uint32_t);
__global__ void k_anchor_big_scatter(const uint64_t *, const uint32_t *, int, const uint64_t *, const uint64_t *, const uint64_t *, const uint32_t *, const uint32_t *, const AlMatch *, const uint32_t *, AlAnchor *, uint32_t *, int, int, int);
template <int NW> __global__ void k_seg_scan(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, int, AlParams, int, int, const uint64_t *, const uint64_t *, uint32_t *, uint32_t *, uint64_t *, uint32_t *, uint32_t *, const uint32_t *, uint32_t *, uint32_t *, uint32_t *, int, int);
template <int NW> __global__ void k_seg_merge(const uint32_t *, int, const uint64_t *, const uint64_t *, const uint4 *, const uint64_t *, const AlAnchor *, const uint64_t *, uint64_t *, AlAnchor *, uint32_t *, uint32_t *, uint32_t *, const uint32_t *, const uint64_t *, uint64_t *, int, int);
template <int NWV> __global__ void k_chain_order_t(const uint32_t *, int, const uint64_t *, const uint32_t *, uint64_t *, AlAnchor *, const uint64_t *, uint64_t *, AlAnchor *, uint32_t *, uint32_t *, int32_t *, int, int);
__global__ void k_lower_bounds(const uint32_t *, uint32_t, LbThr, uint32_t *);
__global__ void k_collect_flagged(const uint32_t *, int, const uint32_t *, uint32_t *, uint32_t *);
__global__ void k_collect_flagged_blk(const uint32_t *, int, const uint32_t *, uint32_t *, uint32_t *);

// Test switches of the allocator.  AL_TEST_POISON=<byte>: every new range is filled with that byte, so that a kernel that reads
// what nobody wrote shows up (the driver's fresh ranges are zero, which hides such reads);
// AL_TEST_POISON_ONLY=<n> restricts it to the n-th allocation of the process, AL_TEST_POISON_LOG prints sequence numbers and sizes.
// AL_TEST_GUARD=1: 4 KB of 0xCD on either side of every range, checked when the range is freed and by al_dev_guard_check():
// the driver pads ranges to pages, so a kernel that writes a few bytes past its buffer goes unnoticed otherwise.
static hipError_t al_dev_malloc_raw(void **p, size_t bytes);
static void al_dev_free_raw(void *p);
namespace { struct GuardRec { size_t bytes; int seq; }; std::mutex g_guard_m; std::map<void *, GuardRec> g_guard; std::atomic<int> g_alloc_seq(0); const size_t GUARD = 4096; }
static int guard_check_one(void *user, const GuardRec &r)
{
	std::vector<unsigned char> h(2 * GUARD);
	if (hipDeviceSynchronize() != hipSuccess) return 0;
	if (hipMemcpy(h.data(), (char *)user - GUARD, GUARD, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(h.data() + GUARD, (char *)user + r.bytes, GUARD, hipMemcpyDeviceToHost) != hipSuccess) return 0;
	int bad = 0;
	for (size_t i = 0; i < GUARD; ++i) if (h[i] != 0xCD) { fprintf(stderr, "[airlift] GUARD: alloc #%d (%zu bytes): byte %zd BEFORE the range overwritten (0x%02x)\n", r.seq, r.bytes, (ssize_t)i - (ssize_t)GUARD, h[i]); ++bad; break; }
	for (size_t i = 0; i < GUARD; ++i) if (h[GUARD + i] != 0xCD) { fprintf(stderr, "[airlift] GUARD: alloc #%d (%zu bytes): byte +%zu AFTER the range overwritten (0x%02x)\n", r.seq, r.bytes, i, h[GUARD + i]); ++bad; break; }
	return bad;
}
int al_dev_guard_check()
{
	std::lock_guard<std::mutex> l(g_guard_m);
	int bad = 0;
	for (auto &kv : g_guard) bad += guard_check_one(kv.first, kv.second);
	return bad;
}
hipError_t al_dev_malloc(void **p, size_t bytes)
{
	static const char *poison = getenv("AL_TEST_POISON"), *only = getenv("AL_TEST_POISON_ONLY"), *plog = getenv("AL_TEST_POISON_LOG"), *guard = getenv("AL_TEST_GUARD");
	if (guard) {
		void *raw = nullptr;
		const hipError_t e = al_dev_malloc_raw(&raw, bytes + 2 * GUARD);
		if (e != hipSuccess) return e;
		const int n = g_alloc_seq.fetch_add(1);
		(void)hipMemset(raw, 0xCD, bytes + 2 * GUARD); (void)hipMemset((char *)raw + GUARD, poison ? atoi(poison) : 0, bytes); (void)hipDeviceSynchronize();
		*p = (char *)raw + GUARD;
		if (plog) fprintf(stderr, "[airlift] alloc #%d: %zu bytes\n", n, bytes);
		std::lock_guard<std::mutex> l(g_guard_m); g_guard[*p] = GuardRec{bytes, n};
		return hipSuccess;
	}
	const hipError_t e = al_dev_malloc_raw(p, bytes);
	if (e == hipSuccess && poison) {
		const int n = g_alloc_seq.fetch_add(1);
		if (plog) fprintf(stderr, "[airlift] alloc #%d: %zu bytes\n", n, bytes);
		if (!only || atoi(only) == n) { (void)hipMemset(*p, atoi(poison), bytes); (void)hipDeviceSynchronize(); }
	}
	return e;
}
void al_dev_free(void *p)
{
	if (!p) return;
	static const char *guard = getenv("AL_TEST_GUARD");
	if (guard) {
		GuardRec r{0, -1}; bool found = false;
		{ std::lock_guard<std::mutex> l(g_guard_m); auto it = g_guard.find(p); if (it != g_guard.end()) { r = it->second; found = true; g_guard.erase(it); } }
		if (found) { (void)guard_check_one(p, r); al_dev_free_raw((char *)p - GUARD); return; }
	}
	al_dev_free_raw(p);
}
std::atomic<size_t> *&al_acct() { static thread_local std::atomic<size_t> *a = nullptr; return a; }
namespace { struct OwnRec { size_t bytes; std::atomic<size_t> *owner; }; std::mutex g_own_m; std::map<void *, OwnRec> g_own; }
AlAllocStat &al_alloc_stat() { static AlAllocStat s; return s; }
AlAllocSite &al_alloc_site() { static thread_local AlAllocSite s{"", 0}; return s; }
// ---- device memory reserve (round 5) ----------------------------------------------------------------------------------------------------
// What a process pays for device memory on this platform is the driver mapping (and scrubbing) it: tenths of a second to seconds per 100 GB, inside
// the first batches of a file-to-file run.  al_device_reserve() starts a thread that obtains the run's memory in a few large chunks WHILE the
// reference is loaded and the index is built; al_dev_malloc / al_dev_free then serve every range -- index arrays, the index builder's temporaries,
// the contexts' grow-only workspaces, the slots' text buffers -- from those chunks (best fit, neighbours coalesced), without a driver call.
// A request that finds no room waits for the filler while it is still at work and falls back to hipMalloc otherwise (also: other devices, ranges
// larger than a chunk).  Peak use is reported (AL_TIMING) -- the workspace figure of the run.
namespace {
struct DevPool {
	int device = -1; std::mutex m; std::condition_variable cv;
	std::map<char *, size_t> free_;                               // free blocks by address
	std::map<char *, size_t> used_;                               // handed-out blocks
	std::vector<std::pair<char *, size_t>> regions;
	bool filling = false, started = false; size_t target = 0, obtained = 0, chunk = 0, in_use = 0, peak = 0, n_served = 0, n_missed = 0; double t_fill = 0, t_first = 0;
	std::thread filler;
	const std::pair<char *, size_t> *region_of(char *p) const { for (const auto &r : regions) if (p >= r.first && p < r.first + r.second) return &r; return nullptr; }
};
DevPool &pool() { static DevPool *P = new DevPool(); return *P; }   // (never destroyed: the filler may outlive main())
const size_t POOL_ALIGN = 4096;
void *pool_alloc(size_t bytes)
{
	DevPool &P = pool();
	if (!P.started) return nullptr;
	int dev = -1; if (hipGetDevice(&dev) != hipSuccess || dev != P.device) return nullptr;
	const size_t need = (bytes + POOL_ALIGN - 1) / POOL_ALIGN * POOL_ALIGN;
	std::unique_lock<std::mutex> l(P.m);
	for (;;) {
		auto best = P.free_.end();
		for (auto it = P.free_.begin(); it != P.free_.end(); ++it) if (it->second >= need && (best == P.free_.end() || it->second < best->second)) best = it;
		if (best != P.free_.end()) {
			char *p = best->first; const size_t sz = best->second;
			P.free_.erase(best);
			if (sz > need) P.free_[p + need] = sz - need;
			P.used_[p] = need; P.in_use += need; if (P.in_use > P.peak) P.peak = P.in_use; ++P.n_served;
			return p;
		}
		if (!P.filling || need > P.chunk) { ++P.n_missed; return nullptr; }
		P.cv.wait(l);
	}
}
bool pool_free(void *ptr)
{
	DevPool &P = pool();
	if (!P.started) return false;
	std::lock_guard<std::mutex> l(P.m);
	auto it = P.used_.find((char *)ptr);
	if (it == P.used_.end()) return false;
	char *p = it->first; size_t sz = it->second;
	P.used_.erase(it); P.in_use -= sz;
	const auto *rg = P.region_of(p);
	auto nx = P.free_.find(p + sz);                               // coalesce with the free neighbours inside the same chunk
	if (nx != P.free_.end() && rg && nx->first < rg->first + rg->second) { sz += nx->second; P.free_.erase(nx); }
	auto pv = P.free_.lower_bound(p);
	if (pv != P.free_.begin()) { --pv; if (pv->first + pv->second == p && rg && pv->first >= rg->first) { p = pv->first; sz += pv->second; P.free_.erase(pv); } }
	P.free_[p] = sz;
	return true;
}
// chunks nothing is using go back to the driver (and the filler stops): the request that did not fit the reserve gets a chance at hipMalloc
size_t pool_release_free(void)
{
	DevPool &P = pool();
	if (!P.started) return 0;
	std::vector<char *> rel; size_t bytes = 0;
	{
		std::lock_guard<std::mutex> l(P.m);
		P.target = P.obtained;                                       // (the filler asks for no further chunk)
		for (size_t i = 0; i < P.regions.size(); ) {
			auto it = P.free_.find(P.regions[i].first);
			if (it != P.free_.end() && it->second == P.regions[i].second) { rel.push_back(P.regions[i].first); bytes += P.regions[i].second; P.obtained -= P.regions[i].second; P.target = P.obtained; P.free_.erase(it); P.regions.erase(P.regions.begin() + (long)i); }
			else ++i;
		}
	}
	for (char *q : rel) (void)hipFree(q);
	if (bytes && getenv("AL_TIMING")) fprintf(stderr, "[airlift] device memory reserve: %.1f GB in %zu unused chunk(s) given back to the driver for a request the reserve could not serve\n", bytes / 1e9, rel.size());
	return bytes;
}
}
static size_t pool_release_free_chunks() { return pool_release_free(); }
extern "C" int al_device_reserve(int device, uint64_t bytes)
{
	DevPool &P = pool();
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return -1;
	if (device < 0) { const char *lr = getenv("LOCAL_RANK"); device = lr ? atoi(lr) % n_dev : 0; }
	if (device >= n_dev || bytes == 0) return -1;
	{
		std::lock_guard<std::mutex> l(P.m);
		if (P.started) return P.device == device ? 0 : -1;          // one reserve per process
		size_t free_b = 0, total_b = 0;
		if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) return -1;
		P.device = device; P.target = (size_t)std::min<uint64_t>(bytes, (uint64_t)((double)free_b * 0.92));
		static const double chunk_gb = getenv("AL_POOL_CHUNK_GB") ? atof(getenv("AL_POOL_CHUNK_GB")) : 0.0;
		P.chunk = chunk_gb > 0 ? (size_t)(chunk_gb * 1e9) : std::min<size_t>(std::max<size_t>(P.target / 6, (size_t)2 << 30), (size_t)24 << 30);
		P.chunk = P.chunk / POOL_ALIGN * POOL_ALIGN;
		P.started = true; P.filling = true;
	}
	P.filler = std::thread([&P]() {
		const auto t0 = std::chrono::steady_clock::now();
		(void)hipSetDevice(P.device);
		for (;;) {
			size_t want;
			{ std::lock_guard<std::mutex> l(P.m); want = P.target > P.obtained ? std::min(P.chunk, P.target - P.obtained) : 0; }
			if (want < ((size_t)64 << 20)) break;
			void *p = nullptr;
			if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); break; }
			std::lock_guard<std::mutex> l(P.m);
			P.regions.emplace_back((char *)p, want); P.free_[(char *)p] = want; P.obtained += want;
			if (P.regions.size() == 1) P.t_first = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			P.cv.notify_all();
		}
		std::lock_guard<std::mutex> l(P.m);
		P.filling = false; P.t_fill = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		P.cv.notify_all();
	});
	P.filler.detach();
	return 0;
}
// The reserve a file-to-file run needs, from its file sizes alone (the stream driver's own arithmetic, al_stream_pipe.cpp: ~360 bytes of FASTQ per
// read; inputs of >= 3 M reads run 2 contexts x 524288-read batches, shorter ones 3 x 131072): the index (4-bit sequence, 8 bytes per minimizer
// occurrence at one minimizer per ~5.7 bases, a table of 16-byte entries at load <= 0.5) + the larger of the index builder's temporaries and the
// contexts' workspaces + the slots' text.  Workspace bytes per read of a batch follow the reference's size -- seed hits per read grow with its repeat
// content: ~78 KB per read measured on a 3.1 Gbp reference with 45 % repeats, ~5 KB on a yeast-sized one; AL_RESERVE_KB_PER_READ overrides.
// Reads per batch of a long input.  Measured on 12.5 M reads of C4 (round 6): batches of 2^20 reads map 11.7 M reads/s against 10.95 M with 524 288 (a batch's serial tails weigh
// half as much), but the run then holds 183 GB instead of 106 GB, and what a process asks for beyond ~128 GB the driver hands out at 35 - 70 GB/s (it clears the pages): start-up
// + 1.5 ... 2.9 s, whole process 2.7 -> 4.9 - 5.3 s.  So the large batch is for inputs whose pipeline runs long enough to pay for that: from AL_LONG_BATCH_BIG_FROM reads
// (default 200 M: ~20 s of pipeline, 7 % of which is 1.4 s).
double al_long_batch_cap(double reads)
{
	static const double fixed = getenv("AL_LONG_BATCH") ? atof(getenv("AL_LONG_BATCH")) : 0.0;
	static const double big_from = getenv("AL_LONG_BATCH_BIG_FROM") ? atof(getenv("AL_LONG_BATCH_BIG_FROM")) : 2.0e8;
	return fixed > 0 ? fixed : reads >= big_from ? 1048576.0 : 524288.0;
}
extern "C" int64_t al_device_reserve_for_run(int device, const char *ref_fn, int n_fn, const char *const *fn)
{
	if (getenv("AL_NO_RESERVE")) return 0;
	struct stat sb;
	if (!ref_fn || stat(ref_fn, &sb) != 0 || !S_ISREG(sb.st_mode)) return 0;
	const double G = (double)sb.st_size;
	double fq = 0;
	for (int i = 0; i < n_fn; ++i) { if (!fn[i] || stat(fn[i], &sb) != 0 || !S_ISREG(sb.st_mode)) return 0; const size_t L = strlen(fn[i]); if (L > 3 && !strcmp(fn[i] + L - 3, ".gz")) return 0; fq += (double)sb.st_size; }
	const double reads = fq / 360.0;
	if (reads < 2.0e5) return 0;                                   // a short run: the driver's own pace is no issue
	const double n_min = G / 5.7;
	double tab = 32.0; while (tab < 2.0 * 0.9 * n_min + 2.0) tab *= 2.0; tab *= 16.0;
	const double index = 0.5 * G + 8.0 * n_min + tab, build_tmp = G + 4.0 * 8.0 * n_min + 4.0 * n_min;
	const double kb = getenv("AL_RESERVE_KB_PER_READ") ? atof(getenv("AL_RESERVE_KB_PER_READ")) : G >= 1.0e9 ? 85.0 : G >= 2.0e8 ? 40.0 : 12.0;
	const bool long_input = reads >= 3.0e6;
	// (a long input's batch by the stream driver's own rule: at least three batches per context)
	const double batch = long_input ? std::min(al_long_batch_cap(reads), std::max(262144.0, reads / 6.0)) : std::min(131072.0, reads), n_ctx = long_input ? 2.0 : 3.0, n_slots = long_input ? 4.0 : 5.0;
	const double ws = n_ctx * batch * kb * 1024.0 + n_slots * batch * 2300.0 + 1.5e9;
	const double need = index + std::max(build_tmp, ws) * 1.08;
	if (al_device_reserve(device, (uint64_t)need) != 0) return -1;
	return (int64_t)need;
}
// free / total device memory as the batch sizing sees it: what the driver reports plus what the reserve holds free (and is still to obtain)
hipError_t al_dev_mem_info(size_t *free_b, size_t *total_b)
{
	const hipError_t e = hipMemGetInfo(free_b, total_b);
	DevPool &P = pool();
	int dev = -1;
	if (e == hipSuccess && P.started && hipGetDevice(&dev) == hipSuccess && dev == P.device) {
		std::lock_guard<std::mutex> l(P.m);
		size_t f = 0; for (const auto &kv : P.free_) f += kv.second;
		// (memory the filler has yet to obtain is already part of the driver's free figure)
		*free_b += f;
	}
	return e;
}
// bytes the reserve holds free on the current device (and still has to obtain); -1 without a reserve there
long long al_dev_reserve_room()
{
	DevPool &P = pool(); int dev = -1;
	if (!P.started || hipGetDevice(&dev) != hipSuccess || dev != P.device) return -1;
	std::lock_guard<std::mutex> l(P.m);
	size_t f = 0; for (const auto &kv : P.free_) f += kv.second;
	return (long long)(f + (P.filling && P.target > P.obtained ? P.target - P.obtained : 0));
}
extern "C" void al_device_reserve_report(FILE *fp)
{
	DevPool &P = pool();
	if (!P.started) return;
	std::lock_guard<std::mutex> l(P.m);
	fprintf(fp, "[airlift] device memory reserve: %.1f of %.1f GB obtained in %zu chunk(s) of %.1f GB by the background thread in %.3f s%s (first chunk after %.3f s); peak in use %.2f GB, %zu ranges served, %zu requests passed on to hipMalloc\n",
	        P.obtained / 1e9, P.target / 1e9, P.regions.size(), P.chunk / 1e9, P.t_fill, P.filling ? " (still filling)" : "", P.t_first, P.peak / 1e9, P.n_served, P.n_missed);
}
extern "C" uint64_t al_device_reserve_peak(void) { DevPool &P = pool(); if (!P.started) return 0; std::lock_guard<std::mutex> l(P.m); return (uint64_t)P.peak; }
extern "C" void al_device_reserve_reset_peak(void) { DevPool &P = pool(); if (!P.started) return; std::lock_guard<std::mutex> l(P.m); P.peak = P.in_use; }

// hipMalloc that leaves the runtime room (round 6).  The HIP runtime obtains device memory of its own while kernels run -- the scratch of a hardware queue the
// first time a kernel with private memory is dispatched on it (k_align: 272 bytes per lane, x 64 lanes x the wave slots of 256 CUs = 142 MB per queue, and a
// context spreads its streams over up to 16 queues), code objects, signals -- and when it cannot, the process dies with HSA_STATUS_ERROR_OUT_OF_RESOURCES: no
// error a caller can catch.  So a request that would leave less than a margin free is refused HERE, as out of memory (AL_ERR_NOMEM upstream: the drivers halve
// the batch).  AL_HBM_MARGIN_MB, default 2048.  With a reserve, its free chunks go back to the driver first and the request is tried again -- also for a request
// larger than a chunk, which the reserve can never serve itself.
static size_t pool_release_free_chunks();
static hipError_t al_hip_malloc_margin(void **p, size_t bytes)
{
	static const size_t margin = (size_t)(getenv("AL_HBM_MARGIN_MB") ? std::max(0, atoi(getenv("AL_HBM_MARGIN_MB"))) : 2048) << 20;
	for (int attempt = 0; attempt < 2; ++attempt) {
		size_t fr = 0, tot = 0;
		const bool known = hipMemGetInfo(&fr, &tot) == hipSuccess;
		if (!known) (void)hipGetLastError();
		if (!known || fr >= bytes + margin) {
			const hipError_t e = hipMalloc(p, bytes);
			if (e == hipSuccess) return e;
			(void)hipGetLastError();
		}
		if (attempt == 0 && pool_release_free_chunks() == 0) break;
	}
	*p = nullptr;
	return hipErrorOutOfMemory;
}
static hipError_t al_dev_malloc_raw(void **p, size_t bytes)
{
	const auto t0 = std::chrono::steady_clock::now();
	hipError_t e = hipSuccess;
	if (void *q = pool_alloc(bytes)) *p = q; else e = al_hip_malloc_margin(p, bytes);
	{ AlAllocStat &a = al_alloc_stat(); const long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); a.dev_ns += ns; a.dev_bytes += (long long)bytes; ++a.dev_calls;
	  static const bool tr = getenv("AL_TRACE_ALLOC") != nullptr;
	  if (tr && bytes >= (32u << 20)) { const AlAllocSite &w = al_alloc_site(); const char *b = strrchr(w.file, '/'); fprintf(stderr, "[airlift] alloc: %8.1f MB in %7.1f ms for %s:%d\n", bytes / 1e6, ns / 1e6, b ? b + 1 : w.file, w.line); } }
	al_alloc_site() = AlAllocSite{"", 0};
	std::atomic<size_t> *a = al_acct();
	if (e == hipSuccess && a) { a->fetch_add(bytes); std::lock_guard<std::mutex> l(g_own_m); g_own[*p] = OwnRec{bytes, a}; }
	return e;
}
static void al_dev_free_raw(void *p)
{
	if (!p) return;
	{ std::lock_guard<std::mutex> l(g_own_m); auto it = g_own.find(p); if (it != g_own.end()) { it->second.owner->fetch_sub(it->second.bytes); g_own.erase(it); } }
	const auto t0 = std::chrono::steady_clock::now();
	if (!pool_free(p)) (void)hipFree(p);
	al_alloc_stat().dev_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
}

static const char *g_stage_names[ST_N] = { "sketch", "seed_lookup", "scan", "size_order", "anchor_sort_small", "anchor_sort", "anchor_sort_blk", "anchor_sort_big", "anchor_heap",
                                           "chain_lds32", "chain_lds48", "chain_lds64", "chain_lds128", "chain_tile", "chain_deferred", "chain_fallback", "chain_ties", "rechain",
                                           "regs", "ext_prep", "ext_sort", "ext_dp_lane", "ext_dp_g4", "ext_dp_g8", "ext_dp_g12", "ext_dp_g16", "ext_dp_g22", "ext_finish", "compact" };
// kernel behind each interval (what rocprofv3 --kernel-trace lists); "" = several launches
// the kernel of an interval that is exactly one launch of one kernel ("" otherwise: several kernels or several launches)
static const char *g_stage_kernels[ST_N] = { "k_sketch", "k_seed", "", "", "k_anchor_sort_small", "", "", "", "",
                                             "", "", "", "", "", "", "", "",  "",
                                             "", "k_ext_prep", "", "", "", "k_ext_dp<8, 256, 128, true>", "k_ext_dp<12, 256, 192, true>", "k_ext_dp<16, 256, 256, true>", "k_ext_dp<22, 256, 352, true>", "k_ext_finish", "k_compact" };   // (the DP instances of reads up to 256 bases with the default scores: two cells per lane; bench.py resolves the name against the committed profile)
extern "C" const char *al_stage_kernel(int i) { return i >= 0 && i < ST_N ? g_stage_kernels[i] : ""; }
extern "C" const char *al_stage_name(int i) { return i >= 0 && i < ST_N ? g_stage_names[i] : ""; }

// ---------------------------------------------------------------------------------------------
int al_upload_index(const al_idx_t *mi, int device, AlDevIndex *out)
{
	std::lock_guard<std::mutex> lk(mi->dev_mtx);
	auto it = mi->dev.find(device);
	if (it != mi->dev.end()) { *out = it->second; return 0; }
	AlDevIndex d;
	AL_HIP_CHECK(hipSetDevice(device));
	std::vector<uint64_t> so(mi->seq.size()); std::vector<uint32_t> sl(mi->seq.size());
	for (size_t i = 0; i < mi->seq.size(); ++i) so[i] = mi->seq[i].offset, sl[i] = mi->seq[i].len;
	AL_HIP_CHECK(al_dev_malloc((void **)&d.seq_off, so.size() * 8));
	AL_HIP_CHECK(hipMemcpy(d.seq_off, so.data(), so.size() * 8, hipMemcpyHostToDevice));
	AL_HIP_CHECK(al_dev_malloc((void **)&d.seq_len, sl.size() * 4));
	AL_HIP_CHECK(hipMemcpy(d.seq_len, sl.data(), sl.size() * 4, hipMemcpyHostToDevice));
	if (mi->built_on >= 0) {                     // index lives on another GPU only: device-to-device copy over xGMI
		auto src = mi->dev.find(mi->built_on);
		if (src == mi->dev.end()) { fprintf(stderr, "[airlift] index was built on device %d but is no longer resident there\n", mi->built_on); return -1; }
		const size_t nS4 = ((mi->tot_len + 7) / 8 + 8) * 4, nTab = ((size_t)2 << mi->tab_bits) * 8, nPos = (mi->n_pos ? mi->n_pos : 1) * 8;
		AL_HIP_CHECK(al_dev_malloc((void **)&d.S4, nS4)); AL_HIP_CHECK(hipMemcpyPeer(d.S4, device, src->second.S4, mi->built_on, nS4));
		AL_HIP_CHECK(al_dev_malloc((void **)&d.tab, nTab)); AL_HIP_CHECK(hipMemcpyPeer(d.tab, device, src->second.tab, mi->built_on, nTab));
		AL_HIP_CHECK(al_dev_malloc((void **)&d.pos, nPos)); AL_HIP_CHECK(hipMemcpyPeer(d.pos, device, src->second.pos, mi->built_on, nPos));
	} else {
		AL_HIP_CHECK(al_dev_malloc((void **)&d.S4, mi->S4.size() * 4));
		AL_HIP_CHECK(hipMemcpy(d.S4, mi->S4.data(), mi->S4.size() * 4, hipMemcpyHostToDevice));
		AL_HIP_CHECK(al_dev_malloc((void **)&d.tab, mi->tab.size() * 8));
		AL_HIP_CHECK(hipMemcpy(d.tab, mi->tab.data(), mi->tab.size() * 8, hipMemcpyHostToDevice));
		AL_HIP_CHECK(al_dev_malloc((void **)&d.pos, mi->pos.size() * 8));
		AL_HIP_CHECK(hipMemcpy(d.pos, mi->pos.data(), mi->pos.size() * 8, hipMemcpyHostToDevice));
	}
	d.tab_bits = mi->tab_bits; d.n_seq = (uint32_t)mi->seq.size();
	mi->dev[device] = d; *out = d;
	return 0;
}

void al_idx_free_device(al_idx_t *mi)
{
	std::lock_guard<std::mutex> lk(mi->dev_mtx);
	for (auto &kv : mi->dev) {
		if (hipSetDevice(kv.first) != hipSuccess) continue;
		al_dev_free(kv.second.S4); al_dev_free(kv.second.tab); al_dev_free(kv.second.pos); al_dev_free(kv.second.seq_off); al_dev_free(kv.second.seq_len);
	}
	mi->dev.clear();
}

extern "C" al_ctx_t *al_ctx_init(const al_idx_t *mi, const al_mapopt_t *opt, int device)
{
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
		fprintf(stderr, "[airlift] FATAL: no HIP device available -- this library has no CPU path\n");
		return nullptr;
	}
	if (device < 0) { const char *lr = getenv("LOCAL_RANK"); device = lr ? atoi(lr) % n_dev : 0; }
	if (device >= n_dev) { fprintf(stderr, "[airlift] FATAL: device %d out of range (%d devices)\n", device, n_dev); return nullptr; }
	if (mi->w > 32 || mi->k > AL_MAX_K) { fprintf(stderr, "[airlift] FATAL: device sketch supports w <= 32, k <= %d\n", AL_MAX_K); return nullptr; }
	al_ctx_t *c = new al_ctx_t();
	c->mi = mi; c->opt = *opt; c->device = device;
	if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
	for (int i = 0; i <= ST_N; ++i) if (hipEventCreate(&c->ev[i]) != hipSuccess) { delete c; return nullptr; }
	// AL_SIDE_PRIO=1 (experiment): the side streams at the highest stream priority.  Their thin classes then get CU slots ahead of the small
	// blocks of the kernels beside them, but the main stream's kernel does not start before they are all placed and the hardware queues are
	// shared out differently: measured 3 ms slower per C4 step than equal priorities.
	int prio_lo = 0, prio_hi = 0; (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
	static const bool side_hi = getenv("AL_SIDE_PRIO") && atoi(getenv("AL_SIDE_PRIO")) == 1;
	const int sp = side_hi ? prio_hi : 0;
	if (hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, sp) != hipSuccess || hipEventCreate(&c->ev_side[0]) != hipSuccess || hipEventCreate(&c->ev_side[1]) != hipSuccess || hipEventCreate(&c->ev_side[2]) != hipSuccess || hipEventCreate(&c->ev_side[3]) != hipSuccess ||
	    hipEventCreateWithFlags(&c->ev_fj[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_fj[1], hipEventDisableTiming) != hipSuccess) { delete c; return nullptr; }
	for (int i = 0; i < 3; ++i) if (hipStreamCreateWithPriority(&c->aux[i], hipStreamNonBlocking, sp) != hipSuccess || hipEventCreateWithFlags(&c->ev_aux[i], hipEventDisableTiming) != hipSuccess) { delete c; return nullptr; }
	for (int i = 0; i < 3; ++i) if (hipStreamCreateWithFlags(&c->ovl[i], hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
	if (hipStreamCreateWithFlags(&c->spec, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&c->spec2, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
	for (int i = 0; i < 3; ++i) if (hipEventCreateWithFlags(&c->ev_spec[i], hipEventDisableTiming) != hipSuccess) { delete c; return nullptr; }
	for (int i = 0; i < 5; ++i) if (hipEventCreateWithFlags(&c->ev_ovl[i], hipEventDisableTiming) != hipSuccess) { delete c; return nullptr; }
	if (al_upload_index(mi, device, &c->di) != 0) { delete c; return nullptr; }
	AlParams &P = c->P;
	P.k = mi->k; P.w = mi->w; P.seed = opt->seed; P.bw = opt->bw; P.max_gap = opt->max_gap; P.max_gap_ref = opt->max_gap_ref; P.max_frag_len = opt->max_frag_len;
	P.max_chain_skip = opt->max_chain_skip; P.max_chain_iter = opt->max_chain_iter; P.min_cnt = opt->min_cnt; P.min_chain_score = opt->min_chain_score;
	P.mask_level = opt->mask_level; P.pri_ratio = opt->pri_ratio; P.max_clip_ratio = opt->max_clip_ratio; P.best_n = opt->best_n;
	P.a = opt->a; P.b = opt->b; P.q = opt->q; P.e = opt->e; P.q2 = opt->q2; P.e2 = opt->e2; P.sc_ambi = opt->sc_ambi; P.zdrop = opt->zdrop; P.zdrop_inv = opt->zdrop_inv;
	P.end_bonus = opt->end_bonus; P.min_dp_max = opt->min_dp_max; P.pe_ori = opt->pe_ori; P.pe_bonus = opt->pe_bonus; P.mid_occ = opt->mid_occ; P.max_occ = opt->max_occ;
	{ const char *d = getenv("AL_DBG"); P.dbg = d ? atoi(d) : 0; if (P.dbg) fprintf(stderr, "[airlift] AL_DBG=%d: timing experiment, results are NOT valid\n", P.dbg); }
	{ const char *d = getenv("AL_DBG2"); P.dbg2 = d ? atoi(d) : 0; if (P.dbg2) fprintf(stderr, "[airlift] AL_DBG2=%d: timing experiment, results are NOT valid\n", P.dbg2); }
	memset(&c->stat, 0, sizeof(c->stat));
	return c;
}

void al_align_state_free(al_ctx_t *c);
static void ctx_release_buffers(al_ctx_t *c)
{   // every grow-only batch buffer (the index stays); each is re-ensured before its next use
	// A range given back to the reserve is handed out again at once, to any context's thread, with no device synchronisation in between (hipFree had one):
	// every stream of this context is drained BEFORE the first range goes back -- this is reached from al_batch_run's out-of-memory exit with the
	// alignment stage's kernels possibly still in flight.  (al_dev_free: the caller guarantees that nothing on the device still uses the range.)
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->side) (void)hipStreamSynchronize(c->side);
	for (int i = 0; i < 3; ++i) if (c->aux[i]) (void)hipStreamSynchronize(c->aux[i]);
	for (int i = 0; i < 3; ++i) if (c->ovl[i]) (void)hipStreamSynchronize(c->ovl[i]);
	if (c->spec) (void)hipStreamSynchronize(c->spec);
	if (c->spec2) (void)hipStreamSynchronize(c->spec2);
	(void)hipGetLastError();
	al_align_state_free(c);
	if (c->side) (void)hipStreamSynchronize(c->side);
	for (int i = 0; i < 3; ++i) if (c->aux[i]) (void)hipStreamSynchronize(c->aux[i]);
	for (int i = 0; i < 3; ++i) if (c->ovl[i]) (void)hipStreamSynchronize(c->ovl[i]);
	if (c->spec) (void)hipStreamSynchronize(c->spec);
	if (c->spec2) (void)hipStreamSynchronize(c->spec2);
	c->spec_busy = false; c->spec_pending = false; c->n_spec = 0;
	c->spec_match.release(); c->spec_meta.release(); c->spec_cnt.release(); c->spec_use.release(); c->spec_v32.release(); c->spec_v64.release(); c->spec_anchors.release(); c->spec_na2.release();
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	c->rd_seq.release(); c->rd_len.release(); c->frag_first.release(); c->frag_hash.release(); c->mini_cnt.release(); c->frag_nm.release(); c->frag_na.release();
	c->frag_nu.release(); c->rechain_list.release(); c->rechain_sorted.release(); c->tmp_u32.release(); c->rd_off.release(); c->mini_off.release(); c->a_off.release(); c->u.release();
	c->ws_u64.release(); c->tmp_u64.release(); c->frag_rep.release(); c->ws_i32.release(); c->mini.release(); c->heap_ws.release(); c->anchors.release();
	c->chained.release(); c->match.release(); c->counters.release(); c->scan_tmp.release(); c->regs0.release(); c->regs.release(); c->reg_cnt.release();
	c->big_tmp.release(); c->chain_key.release(); c->chain_idx.release(); c->chain_idx2.release(); c->tie_list.release(); c->lb_buf.release(); c->tmp_u64b.release();
	c->chain_tmp.release(); c->u_tmp.release(); c->okey_tmp.release(); c->fb2_list.release(); c->fb3_list.release(); c->seg_cnt.release(); c->seg_first.release(); c->seg_cnt0.release(); c->seg_first0.release(); c->seg_t1.release(); c->vs_off.release(); c->vs_na.release(); c->vs_meta.release(); c->vs_res.release(); c->vs_cls.release(); c->seg_key.release(); c->seg_idx.release(); c->seg_ord.release(); c->fb_list.release(); c->tie_frags.release(); c->heap_cnt.release(); c->tie_sorted.release(); c->big_na.release(); c->big_off.release();
	c->uo.release(); c->big_k0.release(); c->big_k1.release(); c->v_anchors.release(); c->v_chained.release(); c->v_u.release(); c->v_a_off.release(); c->v_first64.release(); c->v_na.release(); c->v_nseg.release(); c->v_first.release(); c->v_rd_len.release(); c->v_order.release(); c->v_nu.release(); c->fbk_list.release(); c->d_uslot.release(); c->d_rel.release(); c->chain_cls.release(); c->d_fragid.release(); c->cmp_list.release(); c->ctie.release(); c->frag_meta.release();
	c->big_nt.release(); c->big_toff.release(); c->big_tent.release(); c->big_cuts.release();
	c->a_off_p1.release(); c->frag_na_p1.release(); c->frag_rep_p1.release(); c->cigar.release(); c->reg_off.release(); c->cig_off.release(); c->align_ws.release(); c->seg_a.release(); c->seg_u.release();
}
extern "C" void al_ctx_destroy(al_ctx_t *c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	ctx_release_buffers(c);
	for (int i = 0; i <= ST_N; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
	for (int i = 0; i < 4; ++i) if (c->ev_side[i]) (void)hipEventDestroy(c->ev_side[i]);
	for (int i = 0; i < 2; ++i) if (c->ev_fj[i]) (void)hipEventDestroy(c->ev_fj[i]);
	if (c->side) (void)hipStreamDestroy(c->side);
	for (int i = 0; i < 3; ++i) { if (c->aux[i]) (void)hipStreamDestroy(c->aux[i]); if (c->ev_aux[i]) (void)hipEventDestroy(c->ev_aux[i]); }
	for (int i = 0; i < 3; ++i) if (c->ovl[i]) (void)hipStreamDestroy(c->ovl[i]);
	for (int i = 0; i < 5; ++i) if (c->ev_ovl[i]) (void)hipEventDestroy(c->ev_ovl[i]);
	if (c->spec) (void)hipStreamDestroy(c->spec);
	if (c->spec2) (void)hipStreamDestroy(c->spec2);
	for (int i = 0; i < 3; ++i) if (c->ev_spec[i]) (void)hipEventDestroy(c->ev_spec[i]);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

// ---------------------------------------------------------------------------------------------
// qname hash (map.c:291-293; khash.h:383-409)
static inline uint32_t wang_hash(uint32_t key)
{
	key += ~(key << 15); key ^= (key >> 10); key += (key << 3); key ^= (key >> 6); key += ~(key << 11); key ^= (key >> 16);
	return key;
}
static uint32_t qname_hash(const char *qname, int qlen_sum, int seed)
{
	uint32_t h = 0;
	if (qname) { const char *s = qname; h = (uint32_t)*s; if (h) for (++s; *s; ++s) h = (h << 5) - h + (uint32_t)*s; }
	h ^= wang_hash((uint32_t)qlen_sum) + wang_hash((uint32_t)seed);
	return wang_hash(h);
}

void al_ctx_no_taps(al_ctx_t *c) { if (c) c->no_taps = true; }
extern "C" void al_ctx_set_no_taps(al_ctx_t *c, int on) { if (c) c->no_taps = on != 0; }
extern "C" void al_ctx_set_threads(al_ctx_t *c, int n_threads) { if (c) c->n_threads = n_threads > 1 ? n_threads : 1; }

extern "C" int al_batch_upload(al_ctx_t *c, int n_frag, const int *n_segs, const int *qlens, const char *const *seqs, const char *const *qnames)
{
	if (!c || n_frag < 0) return -1;
	AL_HIP_CHECK(hipSetDevice(c->device));
	const unsigned char *nt4 = al_nt4();
	int n_reads = 0;
	for (int f = 0; f < n_frag; ++f) { if (n_segs[f] < 1 || n_segs[f] > 2) { fprintf(stderr, "[airlift] fragments must have 1 or 2 segments\n"); return -2; } n_reads += n_segs[f]; }
	c->n_frag = n_frag; c->n_reads = n_reads; c->ran = false;
	c->h_rd_len.resize(n_reads + 1); c->h_rd_off.resize(n_reads + 1); c->h_mini_off.resize(n_reads + 1); c->h_flip.assign(n_reads, 0);
	c->h_frag_first.resize(n_frag + 1); c->h_frag_hash.resize(n_frag + 1);
	uint64_t words = 0, mtot = 0, bases = 0; int r = 0, max_qsum = 0;
	std::vector<int> qsums(n_frag);
	const int k = c->mi->k, pe_ori = c->opt.pe_ori;
	for (int f = 0; f < n_frag; ++f) {
		c->h_frag_first[f] = r; int qsum = 0;
		for (int j = 0; j < n_segs[f]; ++j, ++r) {
			const int L = qlens[r];
			c->h_rd_len[r] = L; c->h_rd_off[r] = words; c->h_mini_off[r] = mtot;
			words += (uint64_t)(L + 7) / 8 + 1; mtot += (uint64_t)(L >= k ? L - k + 1 : 0) + 1; bases += L; qsum += L;
			if (n_segs[f] == 2 && ((j == 0 && (pe_ori >> 1 & 1)) || (j == 1 && (pe_ori & 1)))) c->h_flip[r] = 1;
		}
		qsums[f] = qsum; if (qsum > max_qsum) max_qsum = qsum;
	}
	c->max_qlen_sum = max_qsum; c->dev_batch = false;
	{ int lm = 0; for (int i = 0; i < n_reads; ++i) lm = std::max(lm, (int)c->h_rd_len[i]); c->max_rd_len = lm; }
	{   // longest read the device sketch can index (position bits of its packed window entries) and the extension kernels can hold
		const int lim = AL_MAX_READ_LEN;                                         // (the packed window entries of the sketch hold fewer position bits at large k: k_sketch then keeps two words per slot)
		for (int i = 0; i < n_reads; ++i) if ((int)c->h_rd_len[i] >= lim) { fprintf(stderr, "[airlift] a read of %u bases exceeds the limit of the GPU path (%d bases at k = %d)\n", c->h_rd_len[i], lim - 1, k); return -3; }
	}
	al_parallel_for(c->n_threads, (size_t)n_frag, [&](size_t lo, size_t hi, int) {
		for (size_t f = lo; f < hi; ++f) c->h_frag_hash[f] = qname_hash(qnames ? qnames[c->h_frag_first[f]] : nullptr, qsums[f], c->opt.seed);
	});
	c->h_frag_first[n_frag] = r; c->h_rd_off[n_reads] = words; c->h_mini_off[n_reads] = mtot; c->h_rd_len[n_reads] = 0;
	c->n_bases = bases; c->mini_total = mtot; c->seq_words = words;
	{ uint64_t b = 0; for (int i = 0; i < n_reads; ++i) b += (c->h_rd_len[i] * 3 + 7) / 8; c->stat_bytes_in = b; }   // 2-bit base + 1-bit N mask (SURVEY 8d B_in)
	c->h_rd_seq.resize(words + 1); c->h_rd_seq[words] = 0;
	al_parallel_for(c->n_threads, (size_t)n_reads, [&](size_t lo, size_t hi, int) {   // 4-bit packing in mapping orientation (mate 2 reverse-complemented: map.c:468, bseq.h:46-58)
		for (size_t i = lo; i < hi; ++i) {
			uint32_t *w = c->h_rd_seq.data() + c->h_rd_off[i]; const char *s = seqs[i]; const int L = qlens[i], nw = (L + 7) / 8 + 1;
			for (int b = 0; b < nw; ++b) {
				uint32_t v = 0; const int j0 = b * 8, j1 = j0 + 8 < L ? j0 + 8 : L;
				if (!c->h_flip[i]) for (int j = j0; j < j1; ++j) v |= (uint32_t)nt4[(unsigned char)s[j]] << ((j & 7) << 2);
				else for (int j = j0; j < j1; ++j) { const unsigned cd = nt4[(unsigned char)s[L - 1 - j]]; v |= (uint32_t)(cd < 4 ? 3 - cd : 4) << ((j & 7) << 2); }
				w[b] = v;
			}
		}
	});
	hipStream_t s = c->stream;
	if (c->rd_seq.ensure(words + 1) || c->rd_off.ensure(n_reads + 1) || c->rd_len.ensure(n_reads + 1) || c->frag_first.ensure(n_frag + 1) || c->frag_hash.ensure(n_frag + 1) ||
	    c->mini_off.ensure(n_reads + 1) || c->mini.ensure(mtot + 1) || c->mini_cnt.ensure(n_reads + 1) || c->match.ensure(mtot + 1) || c->heap_ws.ensure(mtot + 1) ||
	    c->frag_nm.ensure(n_frag + 1) || c->frag_na.ensure(n_frag + 1) || c->frag_rep.ensure(n_frag + 1) || c->frag_nu.ensure(n_frag + 1) || c->a_off.ensure(n_frag + 2) ||
	    c->rechain_list.ensure(n_frag + 1) || c->tmp_u32.ensure(n_frag + 2) || c->tmp_u64.ensure(n_frag + 2) || c->counters.ensure(32)) return -1;
	AL_HIP_CHECK(hipMemcpyAsync(c->rd_seq.p, c->h_rd_seq.data(), (words + 1) * 4, hipMemcpyHostToDevice, s));
	AL_HIP_CHECK(hipMemcpyAsync(c->rd_off.p, c->h_rd_off.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, s));
	AL_HIP_CHECK(hipMemcpyAsync(c->rd_len.p, c->h_rd_len.data(), (n_reads + 1) * 4, hipMemcpyHostToDevice, s));
	AL_HIP_CHECK(hipMemcpyAsync(c->frag_first.p, c->h_frag_first.data(), (n_frag + 1) * 4, hipMemcpyHostToDevice, s));
	AL_HIP_CHECK(hipMemcpyAsync(c->frag_hash.p, c->h_frag_hash.data(), (n_frag + 1) * 4, hipMemcpyHostToDevice, s));
	AL_HIP_CHECK(hipMemcpyAsync(c->mini_off.p, c->h_mini_off.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, s));
	AL_HIP_CHECK(hipStreamSynchronize(s));
	return 0;
}

// ---------------------------------------------------------------------------------------------
struct CastU64 { __host__ __device__ uint64_t operator()(const uint32_t &v) const { return (uint64_t)v; } };

static int scan_u32_to_u64(al_ctx_t *c, const uint32_t *in, uint64_t *out, int n, hipStream_t st = nullptr, DevBuf<unsigned char> *tmp = nullptr)
{
	hipStream_t s_ = st ? st : c->stream; DevBuf<unsigned char> &T_ = tmp ? *tmp : c->scan_tmp;   // out[0..n] = exclusive prefix sums (n+1 entries; in[n] must be readable: callers pad with 0)
	auto it = rocprim::make_transform_iterator((const uint32_t *)in, CastU64());
	size_t bytes = 0;
	AL_HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, it, out, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), s_));
	if (T_.ensure(bytes + 16)) return -1;
	AL_HIP_CHECK(rocprim::exclusive_scan(T_.p, bytes, it, out, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), s_));
	return 0;
}

struct SizeThr { uint32_t v[28]; int n; };
__global__ void k_size_class(const uint32_t *__restrict__ na, int n, SizeThr T, uint32_t *__restrict__ cls)
{   // class of a fragment = number of thresholds <= its anchor count
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t v = na[i]; uint32_t k = 0;
	for (int j = 0; j < T.n; ++j) k += T.v[j] <= v ? 1u : 0u;
	cls[i] = k;
}
__global__ void k_iota_u32(uint32_t *a, uint32_t n) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = i; }
__global__ void k_flag_list(const uint32_t *list, int n, uint32_t *flag)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) flag[list[i]] = 1u;
}
__global__ void k_gather_na(const uint32_t *frag_na, const uint32_t *list, int n, uint32_t *out)
{
	int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < n) out[t] = frag_na[list[t]]; else if (t == n) out[t] = 0;
}
__global__ void k_scatter_off(const uint64_t *off, const uint32_t *list, int n, uint64_t base, uint64_t *a_off)
{
	int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < n) a_off[list[t]] = base + off[t];
}

static int ensure_anchor_space(al_ctx_t *c, uint64_t total, bool keep)
{
	const uint64_t nf = c->n_frag;
	// per seed hit: the sorted anchor 16 B, its place in chained[] 16 B, a chain list slot (u 8 B + uo 4 B).  The chaining scratch of the
	// segment-wise kernels (56 B per seed hit) is sized for the few fragments that still take them (chain_fallback / chain_legacy).
	if (c->anchors.ensure(total + 1, keep, c->stream) || c->chained.ensure(total + 1, keep, c->stream) || c->u.ensure(total + nf + 2, keep, c->stream) ||
	    c->uo.ensure(total + nf + 2, keep, c->stream)) return -1;
	return 0;
}

// first index with keys[i] >= thr[k] in an ascending device array, for up to 16 thresholds (one kernel, one copy back)
static int lower_bounds(al_ctx_t *c, const uint32_t *keys, uint32_t n, const uint32_t *thr, int nt, uint32_t *out)
{
	if (c->lb_buf.ensure(16)) return -1;
	uint32_t init[16]; LbThr T; T.n = nt;
	for (int i = 0; i < 16; ++i) { init[i] = n; T.v[i] = i < nt ? thr[i] : 0xffffffffu; }
	AL_HIP_CHECK(hipMemcpyAsync(c->lb_buf.p, init, 64, hipMemcpyHostToDevice, c->stream));
	if (n > 0) hipLaunchKernelGGL(k_lower_bounds, dim3((n + 255) / 256), dim3(256), 0, c->stream, keys, n, T, c->lb_buf.p);
	AL_HIP_CHECK(hipMemcpyAsync(out, c->lb_buf.p, 4 * nt, hipMemcpyDeviceToHost, c->stream));
	AL_HIP_CHECK(hipStreamSynchronize(c->stream));
	return 0;
}
static int sort_u32_pairs(al_ctx_t *c, const uint32_t *k_in, uint32_t *k_out, const uint32_t *v_in, uint32_t *v_out, int n)
{
	size_t bytes = 0;
	AL_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, k_in, k_out, v_in, v_out, n, 0, 32, c->stream));
	if (c->scan_tmp.ensure(bytes + 16)) return -1;
	AL_HIP_CHECK(rocprim::radix_sort_pairs(c->scan_tmp.p, bytes, k_in, k_out, v_in, v_out, n, 0, 32, c->stream));
	return 0;
}

#define LCH(C, L, LO, AOFF, NA, CH, UO, NU, LIST, N, SEG, UOFF, WS, STRIDE) do { const int n__ = (N); if (n__ > 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain_lds<C, L>), dim3((n__ + L - 1) / L), dim3(64), 0, s, c->anchors.p, AOFF, NA, c->frag_first.p, c->rd_len.p, CH, UO, NU, WS, LIST, n__, LO, c->P, c->counters.p, SEG, UOFF, STRIDE); } while (0)

// Chaining of the fragments order[0 .. n) (more than 128 anchors each, or any size when the compact LDS rows cannot hold the
// options in force) through their segments: cut -> order the segments by length -> the lane-per-entry LDS kernels for segments
// of up to 128 anchors, the wavefront kernel above -> k_seg_merge; fragments the merge hands back (equal-x chain starts among
// more than 64 chains) are chained whole by the wavefront kernel.
// with_keys: the per-chain processing keys (ChainSeg::okey) are written as well, and fragments whose chain starts tie get their order
// from k_chain_order.  The hot call runs without them (8 scattered bytes per chain less to write and to read back) and hands the few
// fragments that turn out to need them (ties among more than 64 chains) to a second call, on those fragments only, with keys.
static int chain_by_segments(al_ctx_t *c, const uint32_t *order, int n, bool lds_ok, bool first, const uint32_t *skip_flag, bool with_keys = false, int big_from = 0 /* entries before this one have too few anchors for AL_SEGM_BIG segments */, int big8_from = 0 /* ... at most AL_SEGS_BIG anchors */)
{
	hipStream_t s = c->stream;
	auto ev = [&](int st) -> int { if (first) AL_HIP_CHECK(hipEventRecord(c->ev[st + 1], s)); return 0; };
	if (n <= 0) { if (ev(ST_SEG_FIND) || ev(ST_SEG_CHAIN_LDS) || ev(ST_SEG_CHAIN_WAVE) || ev(ST_SEG_MERGE)) return -1; return 0; }
	// a segment of fewer than lmin anchors cannot hold a chain: min_cnt anchors, and min_chain_score at <= k + 1 per anchor (chain.c:60-73,118-124)
	int lmin = c->opt.min_cnt > 1 ? c->opt.min_cnt : 1;
	{ const int per = c->mi->k + 1, need = (c->opt.min_chain_score + per - 1) / per; if (need > lmin) lmin = need; }
	if (c->seg_cnt.ensure((size_t)n + 2) || c->seg_first.ensure((size_t)n + 2) || c->seg_cnt0.ensure((size_t)n + 2) || c->seg_first0.ensure((size_t)n + 2)) return -1;
	// (fragments of more than AL_SEGS_BIG anchors: eight wavefronts each, launched first; big8_from: the entries before it have at most 8192 anchors)
	// (tests lower the two sizes so that ordinary fragments take the eight-wavefront forms; the list positions are then no bound any more)
	static const int segs_big = getenv("AL_TEST_SEG_BIG") ? atoi(getenv("AL_TEST_SEG_BIG")) : 8192, segm_big = getenv("AL_TEST_SEG_BIG") ? std::max(1, atoi(getenv("AL_TEST_SEG_BIG")) / 8) : 1024;
	if (getenv("AL_TEST_SEG_BIG")) { big_from = 0; big8_from = 0; }
	if (big8_from < 0 || big8_from > n) big8_from = 0;
	if (n > big8_from) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_seg_scan<8>), dim3(n - big8_from), dim3(512), 0, s, c->anchors.p, c->a_off.p, c->frag_na.p, c->frag_first.p, c->rd_len.p, order + big8_from, n - big8_from, c->P, lmin, 0,
	                   (const uint64_t *)nullptr, (const uint64_t *)nullptr, c->seg_cnt.p + big8_from, c->seg_cnt0.p + big8_from, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, skip_flag,
	                   (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, big8_from, segs_big);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_seg_scan<1>), dim3(n), dim3(64), 0, s, c->anchors.p, c->a_off.p, c->frag_na.p, c->frag_first.p, c->rd_len.p, order, n, c->P, lmin, 0,
	                   (const uint64_t *)nullptr, (const uint64_t *)nullptr, c->seg_cnt.p, c->seg_cnt0.p, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, skip_flag,
	                   (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, big8_from, segs_big);
	AL_HIP_CHECK(hipMemsetAsync(c->seg_cnt.p + n, 0, 4, s)); AL_HIP_CHECK(hipMemsetAsync(c->seg_cnt0.p + n, 0, 4, s));
	if (scan_u32_to_u64(c, c->seg_cnt.p, c->seg_first.p, n) || scan_u32_to_u64(c, c->seg_cnt0.p, c->seg_first0.p, n)) return -1;
	uint64_t ns64 = 0, ns0_64 = 0;
	AL_HIP_CHECK(hipMemcpyAsync(&ns64, c->seg_first.p + n, 8, hipMemcpyDeviceToHost, s));
	AL_HIP_CHECK(hipMemcpyAsync(&ns0_64, c->seg_first0.p + n, 8, hipMemcpyDeviceToHost, s));
	AL_HIP_CHECK(hipStreamSynchronize(s));
	if (ns64 >= (1ULL << 31)) { fprintf(stderr, "[airlift] %llu chaining segments in one batch: upload fewer fragments\n", (unsigned long long)ns64); al_nomem_flag() = true; return -1; }
	// ns segments, ns0 of them of class 0 (at most 16 anchors: listed by the fill pass in memory order), n1 others (ordered by class below)
	const int ns = (int)ns64, ns0 = (int)ns0_64, n1 = ns - ns0;
	if (c->vs_off.ensure((size_t)ns + 1) || c->vs_na.ensure((size_t)ns + 1) || c->vs_meta.ensure((size_t)ns + 1) || c->vs_res.ensure(2 * ((size_t)ns + 1)) ||
	    c->seg_idx.ensure((size_t)ns0 + 1) || c->vs_cls.ensure((size_t)n1 + 1) || c->seg_t1.ensure((size_t)n1 + 1) || c->seg_key.ensure((size_t)n1 + 1) || c->seg_ord.ensure((size_t)n1 + 1) || c->fb_list.ensure((size_t)n + 2)) return -1;
	if (n > big8_from) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_seg_scan<8>), dim3(n - big8_from), dim3(512), 0, s, c->anchors.p, c->a_off.p, c->frag_na.p, c->frag_first.p, c->rd_len.p, order + big8_from, n - big8_from, c->P, lmin, 1,
	                   (const uint64_t *)c->seg_first.p + big8_from, (const uint64_t *)c->seg_first0.p + big8_from, (uint32_t *)nullptr, (uint32_t *)nullptr, c->vs_off.p, c->vs_na.p, c->vs_meta.p, skip_flag,
	                   c->seg_idx.p, c->seg_t1.p, c->vs_cls.p, big8_from, segs_big);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_seg_scan<1>), dim3(n), dim3(64), 0, s, c->anchors.p, c->a_off.p, c->frag_na.p, c->frag_first.p, c->rd_len.p, order, n, c->P, lmin, 1,
	                   (const uint64_t *)c->seg_first.p, (const uint64_t *)c->seg_first0.p, (uint32_t *)nullptr, (uint32_t *)nullptr, c->vs_off.p, c->vs_na.p, c->vs_meta.p, skip_flag,
	                   c->seg_idx.p, c->seg_t1.p, c->vs_cls.p, big8_from, segs_big);
	uint32_t lb[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                // starts of classes 2 .. 9 in the class-ordered list of the n1 others
	if (n1 > 0) {
		{   // stable sort by size class only (4 bits: one radix pass)
			size_t bytes = 0;
			AL_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t *)c->vs_cls.p, c->seg_key.p, (const uint32_t *)c->seg_t1.p, c->seg_ord.p, n1, 0, 4, s));
			if (c->scan_tmp.ensure(bytes + 16)) return -1;
			AL_HIP_CHECK(rocprim::radix_sort_pairs(c->scan_tmp.p, bytes, (const uint32_t *)c->vs_cls.p, c->seg_key.p, (const uint32_t *)c->seg_t1.p, c->seg_ord.p, n1, 0, 4, s));
		}
		static const uint32_t thr[8] = {2, 3, 4, 5, 6, 7, 8, 9};
		if (lower_bounds(c, c->seg_key.p, (uint32_t)n1, thr, 8, lb)) return -1;
	}
	{ static const bool tr = getenv("AL_TRACE") != nullptr;
	  if (tr && first) fprintf(stderr, "[airlift] trace: segments: %d in %d fragments; by size <=16:%d <=24:%u <=32:%u <=48:%u <=64:%u <=128:%u more:%u\n", ns, n,
	                          ns0, lb[0], lb[1] - lb[0], lb[3] - lb[1], lb[4] - lb[3], lb[7] - lb[4], (uint32_t)n1 - lb[7]); }
	if (ev(ST_SEG_FIND)) return -1;
	const bool keys_possible = c->opt.min_cnt >= 2;                           // a chain has >= 2 anchors: the keys of a segment fit half of its range
	const bool keep_keys = with_keys && keys_possible;
	const ChainSeg sg{c->vs_meta.p, (uint32_t *)c->vs_res.p, nullptr, 0, keep_keys ? c->okey_tmp.p : nullptr};
	bool thin[4] = {false, false, false, false};
	static const uint32_t wave_max = getenv("AL_CHAIN_WAVE_MAX") ? (uint32_t)atoi(getenv("AL_CHAIN_WAVE_MAX")) : 8192u;   // (tests: 0 = never, a large value = always)
	if (ns > 0) {
		const uint32_t *so0 = c->seg_idx.p, *so1 = c->seg_ord.p;
		if (lds_ok) {
#define LSEG(C, L, LIST, A, B) LCH(C, L, -1, c->vs_off.p, c->vs_na.p, c->chain_tmp.p, c->u_tmp.p, (uint32_t *)nullptr, (LIST) + (A), (int)((B) - (A)), sg, (uint32_t *)nullptr, c->ws_u64.p, 0)
			LSEG(16, 64, so0, 0u, (uint32_t)ns0);
			LSEG(24, 64, so1, 0u, lb[0]); LSEG(32, 64, so1, lb[0], lb[1]); LSEG(40, 64, so1, lb[1], lb[2]); LSEG(48, 64, so1, lb[2], lb[3]);
			// A lane walks a 49 ... 128-anchor entry for 0.5 - 1.3 ms whatever the launch holds; the wavefront kernel takes ~1 us per anchor of an
			// entry.  A class with few entries (a small batch, the re-seeding pass) is over sooner through the latter: marked here, launched below.
			for (int k = 3; k < 7; ++k) thin[k - 3] = lb[k + 1] - lb[k] > 0 && lb[k + 1] - lb[k] < wave_max;
			if (!thin[0]) LSEG(64, 64, so1, lb[3], lb[4]);
			if (!thin[1]) LSEG(80, 64, so1, lb[4], lb[5]);
			if (!thin[2]) LSEG(96, 64, so1, lb[5], lb[6]);
			if (!thin[3]) LSEG(128, 32, so1, lb[6], lb[7]);
#undef LSEG
		}
		if (ev(ST_SEG_CHAIN_LDS)) return -1;
		// the wavefront kernel: segments of more than 128 anchors, or (options outside the compact rows' range) all of them
#define LWAVE(LIST, N) do { const int nw__ = (N); if (nw__ > 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain<AL_CHAIN_CAP>), dim3(nw__), dim3(64), 0, s, c->anchors.p, c->vs_off.p, c->vs_na.p, c->frag_first.p, c->rd_len.p, c->chain_tmp.p, c->u_tmp.p, (uint32_t *)nullptr, \
		                               c->ws_i32.p, c->ws_u64.p, (LIST), nw__, c->P, c->counters.p, sg); } while (0)
		const int nw = lds_ok ? n1 - (int)lb[7] : ns;
		{ static const bool tr = getenv("AL_TRACE") != nullptr; if (tr && nw > 0) fprintf(stderr, "[airlift] trace: %d segments to the wavefront kernel (of %d segments in %d fragments)\n", nw, ns, n); }
		if (lds_ok) { LWAVE(so1 + lb[7], n1 - (int)lb[7]); for (int k = 3; k < 7; ++k) if (thin[k - 3]) LWAVE(so1 + lb[k], (int)(lb[k + 1] - lb[k])); } else { LWAVE(so0, ns0); LWAVE(so1, n1); }
#undef LWAVE
		if (ev(ST_SEG_CHAIN_WAVE)) return -1;
	} else { if (ev(ST_SEG_CHAIN_LDS) || ev(ST_SEG_CHAIN_WAVE)) return -1; }
	uint32_t *fb_cnt = (uint32_t *)(c->counters.p + 15);
	AL_HIP_CHECK(hipMemsetAsync(fb_cnt, 0, 8, s));
	// (the fragments with many segments first: eight wavefronts each, next to the one-wavefront launch of the others)
	if (big_from < 0 || big_from > n) big_from = 0;
	if (n > big_from) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_seg_merge<8>), dim3(n - big_from), dim3(512), 0, s, order + big_from, n - big_from, c->seg_first.p + big_from, c->vs_off.p, (const uint4 *)c->vs_res.p, c->u_tmp.p, c->chain_tmp.p, c->a_off.p,
	                   c->u.p, c->chained.p, c->frag_nu.p, c->fb_list.p, fb_cnt, skip_flag, keep_keys ? (const uint64_t *)c->okey_tmp.p : nullptr, c->ws_u64.p, big_from, segm_big);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_seg_merge<1>), dim3(n), dim3(64), 0, s, order, n, c->seg_first.p, c->vs_off.p, (const uint4 *)c->vs_res.p, c->u_tmp.p, c->chain_tmp.p, c->a_off.p,
	                   c->u.p, c->chained.p, c->frag_nu.p, c->fb_list.p, fb_cnt, skip_flag, keep_keys ? (const uint64_t *)c->okey_tmp.p : nullptr, c->ws_u64.p, big_from, segm_big);
	uint32_t n_fb = 0;
	AL_HIP_CHECK(hipMemcpyAsync(&n_fb, fb_cnt, 4, hipMemcpyDeviceToHost, s));
	AL_HIP_CHECK(hipStreamSynchronize(s));
	if (n_fb > 0 && keys_possible && !with_keys) {   // the same fragments once more, with keys (the segment arrays are free again; the list moves out of the way)
		if (c->fb3_list.ensure((size_t)n_fb + 2)) return -1;
		AL_HIP_CHECK(hipMemcpyAsync(c->fb3_list.p, c->fb_list.p, (size_t)n_fb * 4, hipMemcpyDeviceToDevice, s));
		if (chain_by_segments(c, c->fb3_list.p, (int)n_fb, lds_ok, false, skip_flag, true)) return -1;
		if (ev(ST_SEG_MERGE)) return -1;
		return 0;
	}
	c->n_chain_fallback += n_fb;
	const uint32_t *fb = c->fb_list.p;
	if (n_fb > 0 && keep_keys) {   // order restated from the merged chains and their processing keys; only what does not fit its tile is chained again
		const size_t lds = (size_t)AL_ORD_CAP * (8 + 4 + 2) + 64, lds16 = (size_t)AL_ORD_CAP2 * (8 + 2) + 64;   // (lds16 >= lds: the block form keeps the one-wavefront layout up to AL_ORD_CAP chains)
		static const int nu_block = getenv("AL_ORDER_BLOCK") ? atoi(getenv("AL_ORDER_BLOCK")) : 128;             // chains from which a block of 16 wavefronts takes the fragment
		if (!c->attr_chain_order) {
			AL_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_order_t<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
			AL_HIP_CHECK(hipFuncSetAttribute((const void *)k_chain_order_t<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
			c->attr_chain_order = true;
		}
		if (c->fb2_list.ensure((size_t)n_fb + 2)) return -1;
		AL_HIP_CHECK(hipMemsetAsync(fb_cnt, 0, 8, s));
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain_order_t<16>), dim3(n_fb), dim3(1024), lds16, s, c->fb_list.p, (int)n_fb, c->a_off.p, c->frag_nu.p, c->u.p, c->chained.p, (const uint64_t *)c->ws_u64.p, c->u_tmp.p, c->chain_tmp.p, c->fb2_list.p, fb_cnt, c->ws_i32.p, nu_block, 1 << 30);
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain_order_t<1>), dim3(n_fb), dim3(64), lds, s, c->fb_list.p, (int)n_fb, c->a_off.p, c->frag_nu.p, c->u.p, c->chained.p, (const uint64_t *)c->ws_u64.p, c->u_tmp.p, c->chain_tmp.p, c->fb2_list.p, fb_cnt, c->ws_i32.p, 0, nu_block);
		AL_HIP_CHECK(hipMemcpyAsync(&n_fb, fb_cnt, 4, hipMemcpyDeviceToHost, s));
		AL_HIP_CHECK(hipStreamSynchronize(s));
		fb = c->fb2_list.p;
	}
	const ChainSeg whole{nullptr, nullptr, nullptr, 0, nullptr};
	{ static const bool tr = getenv("AL_TRACE") != nullptr; if (tr) fprintf(stderr, "[airlift] trace: chain order: %llu fragments with tied chain starts among > 64 chains (%s keys), %u of them chained whole by the wavefront kernel\n", (unsigned long long)c->n_chain_fallback, keep_keys ? "with" : "without", n_fb); }
	if (n_fb > 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain<AL_CHAIN_CAP>), dim3(n_fb), dim3(64), 0, s, c->anchors.p, c->a_off.p, c->frag_na.p, c->frag_first.p, c->rd_len.p, c->chained.p, c->u.p, c->frag_nu.p,
	                                 c->ws_i32.p, c->ws_u64.p, fb, (int)n_fb, c->P, c->counters.p, whole);
	if (ev(ST_SEG_MERGE)) return -1;
	return 0;
}

// The segment-wise kernels on fragments list[0 .. n) of the batch arrays (option values outside the tile kernel's range), then uo[].
static int chain_legacy(al_ctx_t *c, const uint32_t *list, int n, bool lds_ok, const uint32_t *skip_flag)
{
	if (n <= 0) return 0;
	const uint64_t total = c->n_anchor_total;
	if (c->ws_i32.ensure(total * 4 + 4, false, c->stream) || c->ws_u64.ensure(total + 1, false, c->stream) ||
	    c->chain_tmp.ensure(total + 1, false, c->stream) || c->u_tmp.ensure(total + 1, false, c->stream) || c->okey_tmp.ensure(total + 2, false, c->stream)) return -1;
	if (chain_by_segments(c, list, n, lds_ok, false, skip_flag)) return -1;
	hipLaunchKernelGGL(k_uo_fill, dim3(n), dim3(64), 0, c->stream, list, n, c->a_off.p, c->frag_nu.p, c->u.p, c->uo.p, skip_flag);
	return 0;
}

// Fragments the tile kernel handed back (a segment longer than its tile, more chain ends in a segment than its serial sorts take, tied
// chain starts among more than 64 chains): a compact copy of their anchors becomes a small virtual batch -- fragment v of the list is
// fragment v there -- which the segment-wise kernels chain (their per-anchor scratch is sized for this copy, not for the batch); the
// results go back to the fragments' places with their uo[].
static int chain_fallback(al_ctx_t *c, const uint32_t *fb, int n_fb, bool lds_ok)
{
	hipStream_t s = c->stream;
	if (n_fb <= 0) return 0;
	if (c->v_na.ensure((size_t)n_fb + 2) || c->v_nseg.ensure((size_t)n_fb + 2) || c->v_a_off.ensure((size_t)n_fb + 2) || c->v_first64.ensure((size_t)n_fb + 2) ||
	    c->v_first.ensure((size_t)n_fb + 2) || c->v_rd_len.ensure(2 * (size_t)n_fb + 2) || c->v_order.ensure((size_t)n_fb + 2) || c->v_nu.ensure((size_t)n_fb + 2)) return -1;
	hipLaunchKernelGGL(k_fb_meta, dim3((n_fb + 256) / 256), dim3(256), 0, s, fb, n_fb, c->frag_na.p, c->frag_first.p, c->v_na.p, c->v_nseg.p);
	if (scan_u32_to_u64(c, c->v_na.p, c->v_a_off.p, n_fb) || scan_u32_to_u64(c, c->v_nseg.p, c->v_first64.p, n_fb)) return -1;
	uint64_t vt = 0;
	AL_HIP_CHECK(hipMemcpyAsync(&vt, c->v_a_off.p + n_fb, 8, hipMemcpyDeviceToHost, s));
	AL_HIP_CHECK(hipStreamSynchronize(s));
	{ static const bool tr = getenv("AL_TRACE") != nullptr; if (tr) fprintf(stderr, "[airlift] trace: chain fallback: %d fragments, %llu anchors, through the segment-wise kernels\n", n_fb, (unsigned long long)vt); }
	if (c->v_anchors.ensure(vt + 1, false, s) || c->v_chained.ensure(vt + 1, false, s) || c->v_u.ensure(vt + (uint64_t)n_fb + 2, false, s) ||
	    c->ws_i32.ensure(vt * 4 + 4, false, s) || c->ws_u64.ensure(vt + 1, false, s) || c->chain_tmp.ensure(vt + 1, false, s) || c->u_tmp.ensure(vt + 1, false, s) || c->okey_tmp.ensure(vt + 2, false, s)) return -1;
	hipLaunchKernelGGL(k_fb_reads, dim3((n_fb + 256) / 256), dim3(256), 0, s, fb, n_fb, c->frag_first.p, c->rd_len.p, c->v_first64.p, c->v_first.p, c->v_rd_len.p, c->v_order.p);
	hipLaunchKernelGGL(k_fb_copy_in, dim3(n_fb), dim3(64), 0, s, fb, n_fb, c->a_off.p, c->frag_na.p, c->anchors.p, c->v_a_off.p, c->v_anchors.p);
	// the virtual batch stands in for the batch arrays while the segment-wise kernels run
	AlAnchor *const sv_anchors = c->anchors.p, *const sv_chained = c->chained.p; uint64_t *const sv_a_off = c->a_off.p, *const sv_u = c->u.p;
	uint32_t *const sv_na = c->frag_na.p, *const sv_first = c->frag_first.p, *const sv_rd_len = c->rd_len.p, *const sv_nu = c->frag_nu.p;
	c->anchors.p = c->v_anchors.p; c->chained.p = c->v_chained.p; c->a_off.p = c->v_a_off.p; c->u.p = c->v_u.p;
	c->frag_na.p = c->v_na.p; c->frag_first.p = c->v_first.p; c->rd_len.p = c->v_rd_len.p; c->frag_nu.p = c->v_nu.p;
	const int rc = chain_by_segments(c, c->v_order.p, n_fb, lds_ok, false, nullptr, true);   // (with the processing keys at once: most of these fragments were handed back for tied chain starts, which need them -- one pass instead of two)
	c->anchors.p = sv_anchors; c->chained.p = sv_chained; c->a_off.p = sv_a_off; c->u.p = sv_u;
	c->frag_na.p = sv_na; c->frag_first.p = sv_first; c->rd_len.p = sv_rd_len; c->frag_nu.p = sv_nu;
	if (rc) return -1;
	hipLaunchKernelGGL(k_fb_copy_out, dim3(n_fb), dim3(64), 0, s, fb, n_fb, c->a_off.p, c->v_a_off.p, c->v_nu.p, c->v_u.p, c->v_chained.p, c->frag_nu.p, c->u.p, c->uo.p, c->chained.p);
	return 0;
}

// The tile kernel over the items of S (list entries grouped by size class): it chains the segments of up to 8 anchors itself and defers the
// longer ones, which the lane-per-segment kernels take by size class across all fragments (64 equally long segments per wavefront) and write
// straight to the fragments' arrays; k_u_compact closes the unused reserve in the chain lists; then the fallback for what was handed back.
static int chain_tiles(al_ctx_t *c, const uint32_t *list, const TileSched &S, const uint32_t *skip_flag, int lmin, bool first, bool lds_ok)
{
	hipStream_t s = c->stream;
	auto ev = [&](int st) -> int { if (first) AL_HIP_CHECK(hipEventRecord(c->ev[st + 1], s)); return 0; };
	uint32_t n_fb = 0;
	if (S.n_items > 0) {
		uint32_t *cnts = (uint32_t *)(c->counters.p + 16);                          // (counters[16..17]) [0] handed back, [1] deferred segments, [2] fragments to compact
		static const int force_fb = getenv("AL_TEST_TILE_FB") ? 1 : 0;
		const size_t cap = (size_t)(c->n_anchor_total / 9) + 64;                    // a deferred segment has at least 9 anchors
		if (cap >= (1ULL << 31)) { fprintf(stderr, "[airlift] too many anchors in one batch for the deferred-segment list: upload fewer fragments\n"); al_nomem_flag() = true; return -1; }
		if (c->vs_off.ensure(cap) || c->d_uslot.ensure(cap) || c->vs_na.ensure(cap) || c->vs_meta.ensure(cap) || c->d_rel.ensure(cap) || c->d_fragid.ensure(cap) || c->vs_cls.ensure(cap) ||
		    c->cmp_list.ensure((size_t)c->n_frag + 2) || c->ctie.ensure((size_t)c->n_frag + 2)) return -1;
		AL_HIP_CHECK(hipMemsetAsync(cnts, 0, 16, s));
		AL_HIP_CHECK(hipMemsetAsync(c->ctie.p, 0, ((size_t)c->n_frag + 1) * 4, s));
		CtDefer D; D.off = c->vs_off.p; D.uslot = c->d_uslot.p; D.na = c->vs_na.p; D.meta = c->vs_meta.p; D.rel = c->d_rel.p; D.fragid = c->d_fragid.p; D.cls = c->vs_cls.p; D.cnt = cnts + 1; D.cap = (uint32_t)cap;
		D.cmp_list = c->cmp_list.p; D.cmp_cnt = cnts + 2; D.ctie = c->ctie.p;
		hipLaunchKernelGGL(k_chain_tile6, dim3(S.n_items), dim3(256), 0, s, c->anchors.p, c->a_off.p, c->frag_na.p, c->frag_meta.p, list, S, skip_flag,
		                   c->chained.p, c->u.p, c->uo.p, c->frag_nu.p, c->fb_list.p, cnts, c->P, lmin, c->counters.p, force_fb, D);
		if (ev(ST_SEG_FIND)) return -1;
		if (c->ovl_pending) { AL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_ovl[3], 0)); c->ovl_pending = false; }   // the lane kernels of the small fragments (ovl[1]): done before their scratch is used again
		{ static const bool tr = getenv("AL_TRACE") != nullptr; if (tr) { const hipError_t e = hipStreamSynchronize(s); fprintf(stderr, "[airlift] trace: tile kernel (%u items, first pass %d) -> %s\n", S.n_items, (int)first, hipGetErrorName(e)); } }
		uint32_t h[3] = {0, 0, 0};
		AL_HIP_CHECK(hipMemcpyAsync(h, cnts, 12, hipMemcpyDeviceToHost, s));
		AL_HIP_CHECK(hipStreamSynchronize(s));
		const uint32_t n_def = h[1], n_cmp = h[2];
		if (n_def > cap) { fprintf(stderr, "[airlift] deferred-segment list overflow (%u > %zu)\n", n_def, cap); return -1; }
		if (n_def > 0) {
			// the deferred segments by size class (stable: inside a class they keep the tiles' order, i.e. memory order inside a tile)
			if (c->seg_key.ensure((size_t)n_def + 1) || c->seg_ord.ensure((size_t)n_def + 1) || c->seg_t1.ensure((size_t)n_def + 1)) return -1;
			hipLaunchKernelGGL(k_iota_u32, dim3((n_def + 255) / 256), dim3(256), 0, s, c->seg_t1.p, n_def);
			size_t bytes = 0;
			AL_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t *)c->vs_cls.p, c->seg_key.p, (const uint32_t *)c->seg_t1.p, c->seg_ord.p, (int)n_def, 0, 4, s));
			if (c->scan_tmp.ensure(bytes + 16)) return -1;
			AL_HIP_CHECK(rocprim::radix_sort_pairs(c->scan_tmp.p, bytes, (const uint32_t *)c->vs_cls.p, c->seg_key.p, (const uint32_t *)c->seg_t1.p, c->seg_ord.p, (int)n_def, 0, 4, s));
			static const uint32_t thr[8] = {1, 2, 3, 4, 5, 6, 7, 8};
			uint32_t lb[8];
			if (lower_bounds(c, c->seg_key.p, n_def, thr, 8, lb)) return -1;
			const uint32_t cb[10] = {0, lb[0], lb[1], lb[2], lb[3], lb[4], lb[5], lb[6], lb[7], n_def};   // class k: entries [cb[k], cb[k + 1])
			static const uint32_t capl0[9] = {16, 24, 32, 40, 48, 64, 80, 96, 128};
			// the long classes hold few segments, and a lane walks a 49 ... 128-anchor segment for half a millisecond whatever the launch holds: as long
			// as they are thin they share ONE launch of the 128-anchor kernel (the list is ordered by class: their entries are one range)
			// A class with many segments: the lane-per-segment kernel (64 equally long segments per wavefront, rows in LDS).  A lane walks a 17 ... 128-anchor
			// segment for 0.1 - 0.5 ms whatever the launch holds, so a THIN class (a small batch, the tie rounds, the long classes) goes to k_chain_coop
			// instead -- sixteen lanes per segment, over sooner, and neighbouring thin classes share a launch.  (Measured on the full classes of a
			// 1 M-pair batch the sixteen-lane form is the slower one: 12.7 against 9.8 ms.)  AL_CHAIN_COOP = 0 / 1 (tests): never / always.
			static const int coop_env = getenv("AL_CHAIN_COOP") ? atoi(getenv("AL_CHAIN_COOP")) : -1;
			bool thin[9]; thin[0] = false;
			for (int k = 1; k < 9; ++k) thin[k] = coop_env == 1 || (coop_env != 0 && cb[k + 1] - cb[k] < 4096u);
			uint32_t capl[9]; for (int k = 0; k < 9; ++k) capl[k] = thin[k] ? 0u : capl0[k];
			size_t wsb[10]; wsb[0] = 0; for (int k = 0; k < 9; ++k) wsb[k + 1] = wsb[k] + (size_t)(cb[k + 1] - cb[k]) * capl[k];   // chain-end scratch: CAPL words per entry, by list position
			if (c->ws_u64.ensure(wsb[9] + 64, false, s)) return -1;
			{ static const bool tr = getenv("AL_TRACE") != nullptr;
			  if (tr && first) fprintf(stderr, "[airlift] trace: tile chaining: %u items, %u deferred segments (<=16:%u <=24:%u <=32:%u <=40:%u <=48:%u <=64:%u <=80:%u <=96:%u <=128:%u), %u fragments to compact\n", S.n_items, n_def,
			                          cb[1] - cb[0], cb[2] - cb[1], cb[3] - cb[2], cb[4] - cb[3], cb[5] - cb[4], cb[6] - cb[5], cb[7] - cb[6], cb[8] - cb[7], cb[9] - cb[8], n_cmp); }
			ChainSeg sg{c->vs_meta.p, nullptr, nullptr, 0, nullptr, c->d_uslot.p, c->d_rel.p, c->d_fragid.p, c->ctie.p};
#define LDEF(C, L, K) do { if (!thin[K]) LCH(C, L, -1, c->vs_off.p, c->vs_na.p, c->chained.p, c->u.p, (uint32_t *)nullptr, c->seg_ord.p + cb[K], (int)(cb[K + 1] - cb[K]), sg, c->uo.p, c->ws_u64.p + wsb[K], C); } while (0)
			// (round 5) the classes on two streams in turn, like the lane kernels of the small fragments (ovl[2] is idle here: they were joined above); AL_CHAIN_OVL2=0: all on this one
			static const bool two_env = !(getenv("AL_CHAIN_OVL2") && atoi(getenv("AL_CHAIN_OVL2")) == 0);
			const bool two = two_env && c->n_frag < 400000;                          // (a 1 M-pair batch's classes fill the chip: no gain there, C5 slightly slower)
			hipStream_t const s_cls[2] = {s, two ? c->ovl[2] : s};
			if (two) { AL_HIP_CHECK(hipEventRecord(c->ev_ovl[4], s)); AL_HIP_CHECK(hipStreamWaitEvent(c->ovl[2], c->ev_ovl[4], 0)); }
			{ int turn = 0;
#define LDEF2(C, L, K) do { if (!thin[K] && cb[K + 1] > cb[K]) { hipStream_t const s = s_cls[turn++ & 1]; LDEF(C, L, K); } } while (0)
			LDEF2(16, 64, 0); LDEF2(24, 64, 1); LDEF2(32, 64, 2); LDEF2(40, 64, 3); LDEF2(48, 64, 4); LDEF2(64, 64, 5); LDEF2(80, 64, 6); LDEF2(96, 64, 7); LDEF2(128, 32, 8);
#undef LDEF2
			}
			if (two) { AL_HIP_CHECK(hipEventRecord(c->ev_ovl[4], c->ovl[2])); AL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_ovl[4], 0)); }
			for (int k = 1; k < 9; ) {   // runs of neighbouring thin classes: one launch each
				if (!thin[k]) { ++k; continue; }
				int k1 = k; while (k1 + 1 < 9 && thin[k1 + 1]) ++k1;
				const int nn = (int)(cb[k1 + 1] - cb[k]);
				if (nn > 0) {
					if (k1 <= 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain_coop<48>), dim3((nn + 3) / 4), dim3(64), 0, s, c->anchors.p, c->chained.p, c->u.p, c->uo.p, c->vs_off.p, c->vs_na.p, c->vs_meta.p, c->d_uslot.p, c->d_rel.p, c->d_fragid.p, c->ctie.p,
					                                (const uint32_t *)c->seg_ord.p + cb[k], nn, c->P, c->counters.p);
					else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain_coop<128>), dim3((nn + 3) / 4), dim3(64), 0, s, c->anchors.p, c->chained.p, c->u.p, c->uo.p, c->vs_off.p, c->vs_na.p, c->vs_meta.p, c->d_uslot.p, c->d_rel.p, c->d_fragid.p, c->ctie.p,
					                        (const uint32_t *)c->seg_ord.p + cb[k], nn, c->P, c->counters.p);
				}
				k = k1 + 1;
			}
			{ static const bool tr = getenv("AL_TRACE") != nullptr; if (tr) { const hipError_t e = hipStreamSynchronize(s); fprintf(stderr, "[airlift] trace: deferred segments (%u, first pass %d) -> %s\n", n_def, (int)first, hipGetErrorName(e)); } }
#undef LDEF
		}
		if (n_cmp > 0) hipLaunchKernelGGL(k_u_compact, dim3(std::min<uint32_t>(n_cmp, 8192u)), dim3(64), 0, s, (const uint32_t *)c->cmp_list.p, (const uint32_t *)(cnts + 2), c->a_off.p, c->frag_nu.p, c->u.p, c->uo.p, (const uint32_t *)c->ctie.p, c->fb_list.p, cnts);
		if (ev(ST_SEG_CHAIN_LDS)) return -1;
		{ static const bool tr = getenv("AL_TRACE") != nullptr; if (tr) { const hipError_t e = hipStreamSynchronize(s); fprintf(stderr, "[airlift] trace: compact (%u fragments) -> %s\n", n_cmp, hipGetErrorName(e)); } }
		if (n_cmp > 0) { AL_HIP_CHECK(hipMemcpyAsync(&n_fb, cnts, 4, hipMemcpyDeviceToHost, s)); AL_HIP_CHECK(hipStreamSynchronize(s)); } else n_fb = h[0];
	} else if (ev(ST_SEG_FIND) || ev(ST_SEG_CHAIN_LDS)) return -1;
	if (c->ovl_pending) { AL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_ovl[3], 0)); c->ovl_pending = false; }
	if (n_fb > 0) {
		c->n_chain_fallback += n_fb;
		// (the list the kernels appended to atomically, in ascending fragment order: the same virtual batch on every run)
		size_t bytes = 0;
		if (c->fbk_list.ensure((size_t)n_fb + 2)) return -1;
		AL_HIP_CHECK(rocprim::radix_sort_keys(nullptr, bytes, (const uint32_t *)c->fb_list.p, c->fbk_list.p, (int)n_fb, 0, 32, s));
		if (c->scan_tmp.ensure(bytes + 16)) return -1;
		AL_HIP_CHECK(rocprim::radix_sort_keys(c->scan_tmp.p, bytes, (const uint32_t *)c->fb_list.p, c->fbk_list.p, (int)n_fb, 0, 32, s));
		if (chain_fallback(c, c->fbk_list.p, (int)n_fb, lds_ok)) return -1;
	}
	if (ev(ST_SEG_CHAIN_WAVE)) return -1;
	return 0;
}

static int run_seed_chain(al_ctx_t *c, const uint32_t *list, int n_list, int max_occ, uint64_t base_off, bool first)
{
	hipStream_t s = c->stream;
	const int nl = n_list;
	if (nl == 0) return 0;
	auto ev = [&](int st) -> int { if (first) AL_HIP_CHECK(hipEventRecord(c->ev[st + 1], s)); return 0; };
	static const bool spec_on0 = !(getenv("AL_SPEC_MERGE") && atoi(getenv("AL_SPEC_MERGE")) == 0) && !(getenv("AL_HEAP_OLD") && atoi(getenv("AL_HEAP_OLD")) == 1);
	// (the merges made ahead cost a few milliseconds of the first pass even when the re-chain pass ends up taking none -- a genome whose re-seeded fragments are
	//  small: after two such batches in a row a context makes them for every eighth batch only, until one is taken again)
	if (first) ++c->spec_batch;
	const bool spec_now = first && spec_on0 && c->opt.max_occ > c->opt.mid_occ && (c->spec_idle < 2 || c->spec_batch % 8 == 0 || getenv("AL_SPEC_MIN") != nullptr);
	if (spec_now && c->spec_na2.ensure((size_t)c->n_frag + 1)) return -1;
	hipLaunchKernelGGL(k_seed, dim3((nl + 255) / 256), dim3(256), 0, s, c->di.tab, c->di.tab_bits, c->frag_first.p, c->rd_len.p, c->mini_off.p, c->mini.p, c->mini_cnt.p,
	                   c->match.p, c->frag_nm.p, c->frag_na.p, c->frag_rep.p, list, nl, max_occ, c->opt.max_occ, spec_now ? c->spec_na2.p : (uint32_t *)nullptr);
	if (ev(ST_SEED)) return -1;
	if (c->chain_key.ensure(nl + 1) || c->chain_idx.ensure(nl + 1) || c->chain_idx2.ensure(nl + 1)) return -1;
	uint64_t total = 0;
	// The exact merge of GIANT fragments the re-chain pass may ask for, started now (k_spec_build, al_kernels_seed.hip): AL_SPEC_MERGE=0 turns it off,
	// AL_SPEC_MIN sets the smallest max_occ anchor count that gets a slot (tests lower it so that ordinary fragments take this path).
	constexpr uint32_t SPEC_CAP = AL_SPEC_CAP, SPEC_PER = 126;
	static const uint32_t spec_min = getenv("AL_SPEC_MIN") ? (uint32_t)atoi(getenv("AL_SPEC_MIN")) : 49152u;
	uint32_t h_spec[2 + 4 * AL_SPEC_CAP] = {0, 0};
	if (first) {
		if (c->spec_busy) { AL_HIP_CHECK(hipStreamSynchronize(c->spec)); AL_HIP_CHECK(hipStreamSynchronize(c->spec2)); c->spec_busy = false; }   // (the previous batch's slots: free again)
		c->n_spec = 0; c->spec_pending = false;
		if (spec_now) {
			const uint32_t SPEC_CAND = 16384;
			if (c->spec_match.ensure((size_t)SPEC_CAP * SPEC_PER + 1) || c->spec_meta.ensure(4 * SPEC_CAP + 4) || c->spec_cnt.ensure(4) || c->spec_use.ensure(SPEC_CAP + 1) || c->spec_v32.ensure(7 * (SPEC_CAP + 2)) || c->spec_v64.ensure(2 * (SPEC_CAP + 2) + SPEC_CAND)) return -1;
			AL_HIP_CHECK(hipMemsetAsync(c->spec_cnt.p, 0, 16, s));
			const SpecOut S{c->spec_match.p, c->spec_meta.p, c->spec_cnt.p, c->spec_v64.p + 2 * (SPEC_CAP + 2), SPEC_CAP, SPEC_PER, SPEC_CAND};
			hipLaunchKernelGGL(k_spec_count, dim3((c->n_frag + 255) / 256), dim3(256), 0, s, c->di.tab, c->di.tab_bits, c->frag_first.p, c->rd_len.p, c->mini_off.p, (const AlAnchor *)c->mini.p, (const uint32_t *)c->mini_cnt.p,
			                   (const int32_t *)c->frag_rep.p, c->n_frag, c->opt.max_occ, spec_min, S, c->opt.mid_occ, (const uint32_t *)c->spec_na2.p);
			hipLaunchKernelGGL(k_spec_pick, dim3(64), dim3(256), 0, s, c->di.tab, c->di.tab_bits, c->frag_first.p, c->rd_len.p, c->mini_off.p, (const AlAnchor *)c->mini.p, (const uint32_t *)c->mini_cnt.p, c->opt.max_occ, S);
			AL_HIP_CHECK(hipMemcpyAsync(h_spec, c->spec_cnt.p, 8, hipMemcpyDeviceToHost, s));
			AL_HIP_CHECK(hipMemcpyAsync(h_spec + 2, c->spec_meta.p, 16 * SPEC_CAP, hipMemcpyDeviceToHost, s));
		}
		AL_HIP_CHECK(hipMemsetAsync(c->frag_na.p + c->n_frag, 0, 4, s));
		if (scan_u32_to_u64(c, c->frag_na.p, c->a_off.p, c->n_frag)) return -1;
		AL_HIP_CHECK(hipMemcpyAsync(&total, c->a_off.p + c->n_frag, 8, hipMemcpyDeviceToHost, s));
		AL_HIP_CHECK(hipStreamSynchronize(s));
		if (spec_now && h_spec[0] == 0) ++c->spec_idle;                        // no candidate at all
		if (h_spec[0] > 0) {
			const uint32_t ns = std::min(h_spec[0], SPEC_CAP); uint64_t na = 0;
			for (uint32_t i = 0; i < ns; ++i) na += h_spec[2 + 4 * i + 2];
			if (c->spec_anchors.ensure(na + 1) == 0) {
				uint32_t *v = c->spec_v32.p; const uint32_t st = SPEC_CAP + 2;
				const SpecView V{v, v + st, v + 2 * st, v + 3 * st, v + 4 * st, v + 5 * st, v + 6 * st, c->spec_v64.p, c->spec_v64.p + st};
				hipLaunchKernelGGL(k_spec_layout, dim3(1), dim3(64), 0, c->spec, (const uint32_t *)c->spec_meta.p, ns, SPEC_PER, V);
				AL_HIP_CHECK(hipEventRecord(c->ev_spec[2], c->spec)); AL_HIP_CHECK(hipStreamWaitEvent(c->spec2, c->ev_spec[2], 0));   // (the two list-count classes side by side)
#define LSPEC(NS, LO) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_anchor_heap_lanes<NS, 16>), dim3(ns), dim3(64), 0, NS == 2 ? c->spec : c->spec2, c->di.pos, (const uint32_t *)V.first, (const uint32_t *)V.rdlen, (const uint64_t *)V.moff, (const AlMatch *)c->spec_match.p, \
				                                     (const uint32_t *)V.nm, (const uint32_t *)V.na, (const uint64_t *)V.aoff, c->spec_anchors.p, (const uint32_t *)V.tie, (const uint32_t *)V.list, (const uint32_t *)V.n_list, LO, c->counters.p + 23, c->mi->k)
				LSPEC(2, 63); LSPEC(1, -1);
#undef LSPEC
				c->n_spec = ns; c->spec_busy = true;
				{ static const bool tr = getenv("AL_TRACE") != nullptr; if (tr) { uint32_t mx = 0, mn = 0xffffffffu; for (uint32_t i = 0; i < ns; ++i) { mx = std::max(mx, h_spec[2 + 4 * i + 2]); mn = std::min(mn, h_spec[2 + 4 * i + 2]); }
				  fprintf(stderr, "[airlift] trace: %u giant fragment(s) of %u candidates (%llu anchors with max_occ, %u ... %u each) merged ahead of the re-chain pass\n", ns, h_spec[1], (unsigned long long)na, mn, mx); } }
			} else al_nomem_flag() = false;                                  // (no room: the re-chain pass merges them itself)
		}
		c->n_anchor_pass1 = total; c->n_anchor_total = total;
		if (ensure_anchor_space(c, (uint64_t)((double)total * c->anchor_grow_hw) + 1, false)) return -1;
	} else {
		hipLaunchKernelGGL(k_gather_na, dim3((nl + 256) / 256), dim3(256), 0, s, c->frag_na.p, list, nl, c->tmp_u32.p);
		if (scan_u32_to_u64(c, c->tmp_u32.p, c->tmp_u64.p, nl)) return -1;
		AL_HIP_CHECK(hipMemcpyAsync(&total, c->tmp_u64.p + nl, 8, hipMemcpyDeviceToHost, s));
		AL_HIP_CHECK(hipStreamSynchronize(s));
		if (c->n_anchor_pass1 > 0) { const double g = 1.03 * (double)(base_off + total) / (double)c->n_anchor_pass1; if (g > c->anchor_grow_hw) c->anchor_grow_hw = g < 3.0 ? g : 3.0; }
		if (ensure_anchor_space(c, base_off + total, true)) return -1;
		hipLaunchKernelGGL(k_scatter_off, dim3((nl + 255) / 256), dim3(256), 0, s, c->tmp_u64.p, list, nl, base_off, c->a_off.p);
		c->n_anchor_total = base_off + total;
	}
	if (ev(ST_SCAN)) return -1;
	// Fragments ordered by anchor count: the sort kernels and the chaining kernels are chosen per size class, and the lanes of one
	// wavefront get equal trip counts.  Class boundaries = lower bounds in the sorted counts.
	// The LDS chaining kernels keep 16-bit window-relative positions and 8-bit row indices: they need max_dist_x < 2^15 and
	// max_chain_iter >= 128 (true for the short-read preset); other option values go through the wave-per-entry kernel.
	const int mdx = std::max(std::max(c->opt.max_gap_ref, c->opt.max_frag_len), c->opt.max_gap);
	const bool lds_ok = mdx <= 0x7fff && c->opt.max_chain_iter >= 128 && !((c->P.dbg >> 27) & 1) && c->max_qlen_sum <= 0xfff;   // 12-bit query positions in the compact rows
	if (first) hipLaunchKernelGGL(k_iota_u32, dim3((nl + 255) / 256), dim3(256), 0, s, c->chain_idx.p, (uint32_t)nl);
	// The list by SIZE CLASS, not by exact count: the class boundaries are all the kernels below ask for, and inside a class the fragments stay in
	// memory order (the sort is stable) -- the blocks that run side by side then work on neighbouring ranges of the anchor arrays instead of
	// ranges scattered over tens of gigabytes (a TLB miss per tile cost the tile kernel a third of its time), and the lanes of the lane-per-fragment
	// kernels read neighbouring fragments.  One 5-bit radix pass instead of four 8-bit ones.
	uint32_t lb[15];
	{   // AL_TEST_SORT_BLK / AL_TEST_SORT_BIG (tests): smallest anchor count that goes to the block / device-wide sort
		static const char *e1 = getenv("AL_TEST_SORT_BLK"), *e2 = getenv("AL_TEST_SORT_BIG");
		uint32_t t_blk = e1 ? (uint32_t)atoi(e1) : 1025u, t_big = e2 ? (uint32_t)atoi(e2) : 8193u;   // above 8192 anchors: device-wide radix sort
		if (t_blk < 65u) t_blk = 65u; if (t_blk > 1025u) t_blk = 1025u; if (t_big < t_blk) t_big = t_blk; if (t_big > 8193u) t_big = 8193u;
		{ int rb = 1; while ((1ULL << rb) < c->mi->seq.size()) ++rb; if (33 + rb + 16 > 64) t_big = t_blk; }   // compact keys of the block sort: strand | contig | position | list in 64 bits
		const uint32_t thr[15] = {65, 81, 97, 129, t_blk, t_big, std::min(std::max(t_blk, 2049u), t_big), std::min(std::max(t_blk, 4097u), t_big), std::min(std::max(t_blk, 8193u), t_big),
		                          std::min(257u, t_blk), std::min(513u, t_blk), 1, 33, 257, 513};
		std::vector<uint32_t> T(thr, thr + 15);
		for (uint32_t t : {17u, 25u, 41u, 49u, 1025u, 2049u, 4097u, 8193u}) T.push_back(t);   // the lane kernels' row sizes; the large classes, so that the heaviest items start first
		std::sort(T.begin(), T.end()); T.erase(std::unique(T.begin(), T.end()), T.end());
		SizeThr ST; ST.n = (int)T.size();
		if (ST.n > 28) { fprintf(stderr, "[airlift] internal: too many size classes\n"); return -1; }
		for (int i = 0; i < 28; ++i) ST.v[i] = i < ST.n ? T[i] : 0xffffffffu;
		if (c->chain_cls.ensure((size_t)nl + 1)) return -1;
		hipLaunchKernelGGL(k_size_class, dim3((nl + 255) / 256), dim3(256), 0, s, first ? (const uint32_t *)c->frag_na.p : (const uint32_t *)c->tmp_u32.p, nl, ST, c->chain_cls.p);
		size_t bytes = 0;
		AL_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t *)c->chain_cls.p, c->chain_key.p, first ? (const uint32_t *)c->chain_idx.p : list, c->chain_idx2.p, nl, 0, 5, s));
		if (c->scan_tmp.ensure(bytes + 16)) return -1;
		AL_HIP_CHECK(rocprim::radix_sort_pairs(c->scan_tmp.p, bytes, (const uint32_t *)c->chain_cls.p, c->chain_key.p, first ? (const uint32_t *)c->chain_idx.p : list, c->chain_idx2.p, nl, 0, 5, s));
		uint32_t thr_cls[15];                                                 // count >= thr[k]  <=>  class >= (rank of thr[k] among the thresholds) + 1
		for (int k = 0; k < 15; ++k) thr_cls[k] = (uint32_t)(std::lower_bound(T.begin(), T.end(), thr[k]) - T.begin()) + 1u;
		if (lower_bounds(c, c->chain_key.p, (uint32_t)nl, thr_cls, 15, lb)) return -1;
	}
	const uint32_t *order = c->chain_idx2.p;
	const uint32_t lb1 = lb[11], lb33 = lb[12], lb257x = lb[13], lb513x = lb[14];
	const uint32_t lb65 = lb[0], lb81 = lb[1], lb97 = lb[2], lb129 = lb[3], lb1025 = lb[4], lb_big = lb[5], lb2049 = lb[6], lb4097 = lb[7], lb8193 = lb[8], lb257 = lb[9], lb513 = lb[10];
	// Fragments of more than 128 anchors: the tile kernel (al_kernels_chain.hip).  Its compact rows need what the lane kernels need, and a
	// segment that can hold a chain must have >= 2 anchors (min_cnt anchors, min_chain_score at <= k + 1 per anchor: chain.c:60-73,118-124);
	// other option values: the segment-wise kernels (chain_legacy).  AL_TEST_TILE_ALL (tests): every fragment through the tile kernel.
	int lmin = c->opt.min_cnt > 1 ? c->opt.min_cnt : 1;
	{ const int per = c->mi->k + 1, need = (c->opt.min_chain_score + per - 1) / per; if (need > lmin) lmin = need; }
	static const bool tile_all = getenv("AL_TEST_TILE_ALL") != nullptr;
	const bool tiles_ok = lds_ok && lmin >= 2 && c->opt.max_chain_skip >= 7 && !((c->P.dbg >> 28) & 1);   // (at most 7 predecessors in the segments the tile kernel chains itself: no skip rule)
	const uint32_t tile_from = !tiles_ok ? (uint32_t)nl : tile_all ? lb1 : lb129;
	if (c->fb_list.ensure((size_t)nl + 2)) return -1;
	if (ev(ST_ORDER)) return -1;
	auto chain_small = [&]() -> int {   // fragments of up to 128 anchors: the lane-per-fragment kernels
		const ChainSeg nosg{nullptr, nullptr, (const uint32_t *)c->tie_list.p, 1, nullptr};
		// chain-end scratch of the whole-fragment lane kernels: 64 words per entry of the <= 64-anchor classes, 128 above, by list position
		const uint32_t n_lo = std::min(lb65, tile_from), n_mid_end = std::min(lb129, tile_from);
		if (lds_ok && c->ws_u64.ensure((size_t)n_lo * 64 + (size_t)(n_mid_end > lb65 ? n_mid_end - lb65 : 0u) * 128 + 64, false, s)) return -1;
		// The lane-per-fragment kernels (fragments of up to 128 anchors, memory latency) run on a stream of their own: beside the tile sorts
		// of the large fragments and beside the tile kernel, which takes the rest of the list.
		hipStream_t const s_main = s;
		static const bool use_ovl = !(getenv("AL_CHAIN_OVL") && atoi(getenv("AL_CHAIN_OVL")) == 0);   // (AL_CHAIN_OVL=0: on the main stream, after the sorts)
		// (round 5) ... on TWO streams, the classes in turn (AL_CHAIN_OVL2=0: one): a class ends in the tail of its slowest wavefronts, and the next one's start fills it
		static const bool two = !(getenv("AL_CHAIN_OVL2") && atoi(getenv("AL_CHAIN_OVL2")) == 0);
		if (use_ovl) { AL_HIP_CHECK(hipStreamWaitEvent(c->ovl[1], c->ev_ovl[2], 0)); if (two) AL_HIP_CHECK(hipStreamWaitEvent(c->ovl[2], c->ev_ovl[2], 0)); }
		{ hipStream_t s = use_ovl ? c->ovl[1] : s_main; int turn = 0;
#define NEXT_S() do { if (use_ovl && two) s = c->ovl[1 + (++turn & 1)]; } while (0)
#define LFR(C, L, A, B) LCH(C, L, -1, c->a_off.p, c->frag_na.p, c->chained.p, c->u.p, c->frag_nu.p, order + (A), (int)((B) - (A)), nosg, c->uo.p, c->ws_u64.p + (size_t)n_lo * 64 + (size_t)((A) - lb65) * 128, 128)
#define LFRLO(C, L, LO) LCH(C, L, LO, c->a_off.p, c->frag_na.p, c->chained.p, c->u.p, c->frag_nu.p, order, (int)n_lo, nosg, c->uo.p, c->ws_u64.p, 64)
		if (lds_ok) { LFRLO(16, 64, -1); NEXT_S(); LFRLO(24, 64, 16); NEXT_S(); LFRLO(32, 64, 24); NEXT_S(); }
		if (lds_ok) { LFRLO(40, 64, 32); NEXT_S(); LFRLO(48, 64, 40); NEXT_S(); }
		if (lds_ok) { LFRLO(64, 64, 48); NEXT_S(); }
		if (lds_ok && n_mid_end > lb65) {   // exact ranges of the size-ordered list (lane counts differ between these classes)
			// 64-lane wavefronts hold more fragments per CU but need enough of them to cover the chip; a thin class runs on half waves
			const uint32_t fill = 64u * 3u * 256u * 2u;
			static const uint32_t wave_max = getenv("AL_CHAIN_WAVE_MAX") ? (uint32_t)atoi(getenv("AL_CHAIN_WAVE_MAX")) : 8192u;
			uint32_t from = lb65;
			if (lb129 - lb65 < wave_max) {   // few fragments of 65 ... 128 anchors (a small batch): a wavefront each is over sooner than a lane each
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain<AL_CHAIN_CAP>), dim3(lb129 - lb65), dim3(64), 0, s, c->anchors.p, c->a_off.p, c->frag_na.p, c->frag_first.p, c->rd_len.p, c->chained.p, c->u.p, c->frag_nu.p,
				                   (int32_t *)nullptr, (uint64_t *)nullptr, order + lb65, (int)(lb129 - lb65), c->P, c->counters.p, nosg);   // (<= 128 anchors: its rows are in LDS, no scratch)
				hipLaunchKernelGGL(k_uo_fill, dim3(lb129 - lb65), dim3(64), 0, s, order + lb65, (int)(lb129 - lb65), c->a_off.p, c->frag_nu.p, c->u.p, c->uo.p, (const uint32_t *)c->tie_list.p);
			} else {
				if (lb81 - lb65 >= fill) { LFR(80, 64, lb65, lb81); from = lb81; NEXT_S(); }
				if (from == lb81 && lb97 - lb81 >= fill) { LFR(96, 64, lb81, lb97); from = lb97; NEXT_S(); }
				LFR(128, 32, from, lb129);
			}
		}
		if (use_ovl && two) { AL_HIP_CHECK(hipEventRecord(c->ev_ovl[4], c->ovl[2])); AL_HIP_CHECK(hipStreamWaitEvent(c->ovl[1], c->ev_ovl[4], 0)); s = c->ovl[1]; }   // (joined on ovl[1]: one event for the main stream)
		if (use_ovl) { AL_HIP_CHECK(hipEventRecord(c->ev_ovl[3], s)); c->ovl_pending = true; } }
#undef NEXT_S
#undef LFR
#undef LFRLO
		return 0;
	};
	{
		if (c->tie_list.ensure((size_t)c->n_frag + 2)) return -1;         // one flag per fragment id
		unsigned int *tie_cnt = nullptr;
		AL_HIP_CHECK(hipMemsetAsync(c->tie_list.p, 0, ((size_t)c->n_frag + 1) * 4, s));
		int rid_bits = 1; while ((1ULL << rid_bits) < c->mi->seq.size()) ++rid_bits;
		// above the register tiles: composite-key device radix sort, a chunk of fragments at a time so that rank + x + list fit 64 bits.
		// Bandwidth-bound passes over a few hundred fragments' keys: on a stream of its own, beside the tile sorts (instruction-bound) of everything else.
		hipStream_t sb = c->ovl[0];
		AL_HIP_CHECK(hipEventRecord(c->ev_ovl[0], s)); AL_HIP_CHECK(hipStreamWaitEvent(sb, c->ev_ovl[0], 0));
		int pos_bits = 1; { uint32_t mx = 1; for (const AlSeq &sq : c->mi->seq) mx = std::max(mx, sq.len); while (pos_bits < 31 && (1ULL << pos_bits) < mx) ++pos_bits; }
		const int rank_bits = 64 - 16 - 1 - rid_bits - pos_bits;
		static const char *e3 = getenv("AL_TEST_BIG_CHUNK");                // tests: fragments per device-wide sort
		uint32_t chunk_max = rank_bits >= 31 ? 0x7fffffffu : rank_bits >= 1 ? (1u << rank_bits) : 1u;
		if (e3 && atoi(e3) > 0) chunk_max = std::min<uint32_t>(chunk_max, (uint32_t)atoi(e3));
		if (rank_bits < 0 && lb_big < (uint32_t)nl) {                       // (no such index in practice: > 2^46 contig-id x position range) exact merge for all of them
			hipLaunchKernelGGL(k_flag_list, dim3(((uint32_t)nl - lb_big + 255) / 256), dim3(256), 0, sb, order + lb_big, (int)((uint32_t)nl - lb_big), c->tie_list.p);
		}
		// (round 5) runs of 8192 anchors sorted by the register network, then merged pairwise (k_anchor_run_sort / k_anchor_run_merge): one to six passes of 8 bytes
		// per anchor by the fragment's size instead of expansion + seven radix passes + scatter.  AL_BIG_MERGE=0: the device-wide sort (tests run both).
		static const int big_merge = getenv("AL_BIG_MERGE") ? atoi(getenv("AL_BIG_MERGE")) : 1;        // 0: radix everywhere, 1: run merge everywhere, 2: run merge in the re-chain pass only
		const bool use_merge = (big_merge == 1 || (big_merge == 2 && !first)) && 33 + rid_bits + 16 <= 64 && lb_big < (uint32_t)nl;
		// AL_TEST_RUN=<run>,<tile> (tests): shorter runs and merge tiles (powers of two, tile <= run <= 8192, tile <= 2048) so that the golden sets' fragments take several passes
		static uint32_t run_len = 8192, tile = 2048;
		{ static bool once = false; if (!once) { once = true; const char *e = getenv("AL_TEST_RUN"); unsigned a = 0, b = 0;
		  if (e && sscanf(e, "%u,%u", &a, &b) == 2 && a && b && !(a & (a - 1)) && !(b & (b - 1)) && b <= a && a <= 8192 && b <= 2048) { run_len = a; tile = b; } } }
		if (use_merge) {
			const uint32_t nb = (uint32_t)nl - lb_big;
			unsigned int *const mx_d = (unsigned int *)(c->counters.p + 18);
			if (c->big_na.ensure(nb + 2) || c->big_off.ensure(nb + 2) || c->big_nt.ensure(nb + 2) || c->big_toff.ensure(nb + 2)) return -1;
			AL_HIP_CHECK(hipMemsetAsync(mx_d, 0, 8, sb));
			hipLaunchKernelGGL(k_gather_na, dim3((nb + 256) / 256), dim3(256), 0, sb, c->frag_na.p, order + lb_big, (int)nb, c->big_na.p);
			hipLaunchKernelGGL(k_big_tiles, dim3((nb + 256) / 256), dim3(256), 0, sb, (const uint32_t *)c->big_na.p, (int)nb, c->big_nt.p, mx_d, tile);
			if (scan_u32_to_u64(c, c->big_na.p, c->big_off.p, (int)nb, sb, &c->big_tmp) || scan_u32_to_u64(c, c->big_nt.p, c->big_toff.p, (int)nb, sb, &c->big_tmp)) return -1;
			uint64_t nbig = 0, ntile = 0; unsigned int mx = 0;
			AL_HIP_CHECK(hipMemcpyAsync(&nbig, c->big_off.p + nb, 8, hipMemcpyDeviceToHost, sb));
			AL_HIP_CHECK(hipMemcpyAsync(&ntile, c->big_toff.p + nb, 8, hipMemcpyDeviceToHost, sb));
			AL_HIP_CHECK(hipMemcpyAsync(&mx, mx_d, 4, hipMemcpyDeviceToHost, sb));
			AL_HIP_CHECK(hipStreamSynchronize(sb));
			if (ntile >= (1ULL << 31)) { fprintf(stderr, "[airlift] too many anchors in one batch for the run merge: upload fewer fragments\n"); al_nomem_flag() = true; return -1; }
			if (c->big_k0.ensure((size_t)nbig + 1, false, sb) || c->big_k1.ensure((size_t)nbig + 1, false, sb) || c->big_tent.ensure((size_t)ntile + 1) || c->big_cuts.ensure((size_t)ntile + 2)) return -1;
			const uint32_t nt32 = (uint32_t)ntile;
			hipLaunchKernelGGL(k_big_tile_ent, dim3((nt32 + 255) / 256), dim3(256), 0, sb, (const uint64_t *)c->big_toff.p, (int)nb, nt32, c->big_tent.p);
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_anchor_run_sort<16, 8, 1024>), dim3(nt32), dim3(512), 0, sb, c->di.pos, c->frag_first.p, c->mini_off.p, c->match.p, c->frag_nm.p, c->frag_na.p,
			                   order + lb_big, (int)nb, (const uint64_t *)c->big_toff.p, (const uint32_t *)c->big_tent.p, (const uint64_t *)c->big_off.p, c->big_k0.p, c->tie_list.p, rid_bits, run_len, tile);
			int n_pass = 1; while (((uint64_t)run_len << n_pass) < mx) ++n_pass;
			const RunMergeOut O{c->a_off.p, c->mini_off.p, c->frag_first.p, c->rd_len.p, c->match.p, c->anchors.p, c->tie_list.p, rid_bits, c->mi->k};
			for (int p = 0; p < n_pass; ++p) {
				const uint64_t *kin = (p & 1) ? c->big_k1.p : c->big_k0.p; uint64_t *kout = (p & 1) ? c->big_k0.p : c->big_k1.p;
				hipLaunchKernelGGL(k_anchor_run_cuts, dim3((nt32 + 255) / 256), dim3(256), 0, sb, kin, order + lb_big, (const uint32_t *)c->big_tent.p, nt32, (const uint64_t *)c->big_toff.p, (const uint64_t *)c->big_off.p,
				                   (const uint32_t *)c->frag_na.p, (const uint32_t *)c->frag_nm.p, p, run_len, tile, c->big_cuts.p);
				hipLaunchKernelGGL(k_anchor_run_merge, dim3(nt32), dim3(256), 0, sb, kin, kout, order + lb_big, (const uint32_t *)c->big_tent.p, (const uint32_t *)c->big_cuts.p, (const uint64_t *)c->big_toff.p,
				                   (const uint64_t *)c->big_off.p, (const uint32_t *)c->frag_na.p, (const uint32_t *)c->frag_nm.p, p, O, run_len, tile);
			}
		}
		for (uint32_t b0 = lb_big; !use_merge && rank_bits >= 0 && b0 < (uint32_t)nl; ) {
			const uint32_t nb = std::min<uint32_t>((uint32_t)nl - b0, chunk_max);
			if (c->big_na.ensure(nb + 2) || c->big_off.ensure(nb + 2)) return -1;
			hipLaunchKernelGGL(k_gather_na, dim3((nb + 256) / 256), dim3(256), 0, sb, c->frag_na.p, order + b0, (int)nb, c->big_na.p);
			if (scan_u32_to_u64(c, c->big_na.p, c->big_off.p, (int)nb, sb, &c->big_tmp)) return -1;
			uint64_t nbig = 0;
			AL_HIP_CHECK(hipMemcpyAsync(&nbig, c->big_off.p + nb, 8, hipMemcpyDeviceToHost, sb));
			AL_HIP_CHECK(hipStreamSynchronize(sb));
			if (c->big_k0.ensure((size_t)nbig + 1, false, sb) || c->big_k1.ensure((size_t)nbig + 1, false, sb)) return -1;
			uint64_t *ka = c->big_k0.p, *kbuf = c->big_k1.p;                   // key double buffer
			hipLaunchKernelGGL(k_anchor_big_expand, dim3(nb), dim3(256), 0, sb, c->di.pos, c->mini_off.p, c->frag_first.p, c->match.p, c->frag_nm.p,
			                   order + b0, (int)nb, c->big_off.p, ka, rid_bits, pos_bits);
			int rbits = 0; while ((1ULL << rbits) < nb) ++rbits;
			rocprim::double_buffer<uint64_t> dk(ka, kbuf);
			const unsigned end_bit = (unsigned)(16 + 1 + rid_bits + pos_bits + rbits);
			size_t bytes = 0;
			AL_HIP_CHECK(rocprim::radix_sort_keys(nullptr, bytes, dk, (size_t)nbig, 16u, end_bit, sb));
			if (c->big_tmp.ensure(bytes + 16)) return -1;
			AL_HIP_CHECK(rocprim::radix_sort_keys(c->big_tmp.p, bytes, dk, (size_t)nbig, 16u, end_bit, sb));
			hipLaunchKernelGGL(k_anchor_big_scatter, dim3(nb), dim3(256), 0, sb, dk.current(), order + b0, (int)nb, c->big_off.p, c->a_off.p,
			                   c->mini_off.p, c->frag_first.p, c->rd_len.p, c->match.p, c->frag_nm.p, c->anchors.p, c->tie_list.p, rid_bits, pos_bits, c->mi->k);
			b0 += nb;
		}
		AL_HIP_CHECK(hipEventRecord(c->ev_ovl[1], sb));
		if (lb65 > 0) hipLaunchKernelGGL(k_anchor_sort_small, dim3(lb65), dim3(64), 0, s, c->di.pos, c->frag_first.p, c->rd_len.p, c->mini_off.p, c->match.p, c->frag_nm.p, c->frag_na.p,
		                                 c->a_off.p, c->anchors.p, c->tie_list.p, tie_cnt, order, (int)lb65, c->counters.p, c->mi->k);
		if (ev(ST_ANCHOR_SORT_S)) return -1;
		const bool compact = 33 + rid_bits + 16 <= 64;                   // strand | contig | position | list in one 64-bit key
#define LREG(P, W, M, A, B) do { if ((B) > (A)) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_anchor_sort_reg<P, W, M>), dim3((B) - (A)), dim3(64 * W), 0, s, c->di.pos, c->frag_first.p, c->rd_len.p, c->mini_off.p, c->match.p, c->frag_nm.p, c->frag_na.p, \
		                                        c->a_off.p, c->anchors.p, c->tie_list.p, order + (A), (int)((B) - (A)), c->mi->k, rid_bits); } while (0)
		if (compact) { LREG(2, 1, 128, lb65, std::min(lb129, lb1025)); LREG(4, 1, 256, lb129, lb257); LREG(8, 1, 512, lb257, lb513); LREG(16, 1, 512, lb513, lb1025); }
		else if (lb1025 > lb65) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_anchor_sort<1024>), dim3(lb1025 - lb65), dim3(64), 0, s, c->di.pos, c->frag_first.p, c->rd_len.p, c->mini_off.p, c->match.p, c->frag_nm.p, c->frag_na.p,
		                                           c->a_off.p, c->anchors.p, c->tie_list.p, tie_cnt, order + lb65, (int)(lb1025 - lb65), c->counters.p, c->mi->k);
		if (ev(ST_ANCHOR_SORT)) return -1;
		// fragments of up to 128 anchors are in order: their chaining (ovl[1], below) may start -- unless a test lowered the block sorts' bound
		// below 129 anchors (AL_TEST_SORT_BLK): then some of them are sorted by the block kernels below, and the event follows those
		const bool small_by_blk = lb1025 < lb129;
		if (!small_by_blk) AL_HIP_CHECK(hipEventRecord(c->ev_ovl[2], s));
		LREG(8, 4, 1024, lb1025, lb2049); LREG(16, 4, 1024, lb2049, lb4097); LREG(16, 8, 1024, lb4097, lb_big);   // (non-compact keys: t_big == t_blk, empty ranges)
		if (small_by_blk) AL_HIP_CHECK(hipEventRecord(c->ev_ovl[2], s));
#undef LREG
		// (enqueued before the merge kernels of the side streams below: whichever hardware queue ovl[1] shares, these do not wait behind one of those)
		if (chain_small()) return -1;
		if (ev(ST_ANCHOR_SORT_BLK)) return -1;
		AL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_ovl[1], 0));                 // the device-wide sort (ovl[0], started before the tile sorts)
		if (ev(ST_ANCHOR_SORT_BIG)) return -1;
		// Fragments the sort kernels flagged (equal x: overlapping mates, tandem repeats): the reference's order among equal heads is
		// that of its binary heap, which only a serial emulation reproduces (one lane per fragment) -- a latency-bound tail on a few
		// hundred fragments, so it runs on a side stream next to the chaining of everything else; the flagged fragments are chained
		// (by segments, like the others) in a second, small round once the side stream is done.
		hipEvent_t *const evs = c->ev_side + (first ? 0 : 2);
		AL_HIP_CHECK(hipEventRecord(evs[0], s));
		AL_HIP_CHECK(hipStreamWaitEvent(c->side, evs[0], 0));
		// the flagged fragments are compacted first (on the side stream, count on the device): a wavefront of the merge kernels then holds 32 / 64
		// of them instead of the few that happen to sit among 64 neighbours of the size-ordered list
		if (c->tie_frags.ensure((size_t)nl + 2) || c->heap_cnt.ensure(4)) return -1;
		uint32_t *const n_heap_d = c->heap_cnt.p + (first ? 0 : 1);
		AL_HIP_CHECK(hipMemsetAsync(n_heap_d, 0, 4, c->side));
		if (!first && c->n_spec > 0) {   // slots of the merge that was started after the first seeding: taken (flag 2) before the merge kernels' list is made, copied in on their own stream
			const uint32_t st = AL_SPEC_CAP + 2;
			AL_HIP_CHECK(hipEventRecord(c->ev_spec[2], c->spec2)); AL_HIP_CHECK(hipStreamWaitEvent(c->spec, c->ev_spec[2], 0));   // (both merge kernels done before the copy)
			hipLaunchKernelGGL(k_spec_mark, dim3((AL_SPEC_CAP + 255) / 256), dim3(256), 0, c->side, (const uint32_t *)c->spec_meta.p, c->n_spec, (const uint32_t *)c->frag_nm.p, (const uint32_t *)c->frag_na.p, c->tie_list.p, c->spec_use.p, c->spec_cnt.p + 2);
			AL_HIP_CHECK(hipEventRecord(c->ev_spec[0], c->side)); AL_HIP_CHECK(hipStreamWaitEvent(c->spec, c->ev_spec[0], 0));
			hipLaunchKernelGGL(k_spec_apply, dim3(64, c->n_spec), dim3(256), 0, c->spec, (const uint32_t *)c->spec_meta.p, (const uint32_t *)c->spec_use.p, (const uint64_t *)(c->spec_v64.p + st), (const AlAnchor *)c->spec_anchors.p, (const uint64_t *)c->a_off.p, c->anchors.p);
			AL_HIP_CHECK(hipEventRecord(c->ev_spec[1], c->spec));
			c->spec_pending = true;
			{ static const bool tr = getenv("AL_TRACE") != nullptr; if (tr) { std::vector<uint32_t> u(c->n_spec); AL_HIP_CHECK(hipStreamSynchronize(c->side)); AL_HIP_CHECK(hipMemcpy(u.data(), c->spec_use.p, (size_t)c->n_spec * 4, hipMemcpyDeviceToHost));
			  uint32_t k = 0; for (uint32_t x : u) k += x; fprintf(stderr, "[airlift] trace: re-chain pass takes %u of the %u merges made ahead\n", k, c->n_spec); } }
		}
		hipLaunchKernelGGL(k_collect_flagged_blk, dim3((nl + 255) / 256), dim3(256), 0, c->side, order, nl, (const uint32_t *)c->tie_list.p, c->tie_frags.p, n_heap_d);
		// the four merge kernels take disjoint fragments and each waits for its slowest one: side by side, on a stream each
		AL_HIP_CHECK(hipEventRecord(c->ev_fj[0], c->side));
		for (int i = 0; i < 3; ++i) AL_HIP_CHECK(hipStreamWaitEvent(c->aux[i], c->ev_fj[0], 0));
#define LHEAP(H, LN, LO, ST) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_anchor_heap<H, LN>), dim3((nl + LN - 1) / LN), dim3(64), 0, ST, c->di.pos, c->frag_first.p, c->rd_len.p, c->mini_off.p, c->match.p, c->frag_nm.p, c->frag_na.p, \
		                                  c->a_off.p, c->anchors.p, c->heap_ws.p, c->tie_list.p, (const uint32_t *)c->tie_frags.p, nl, LO, c->counters.p, c->mi->k, (const uint32_t *)n_heap_d, wave_na_min)
		// A lane of the lane kernels pops ~1 us per anchor with its 63 neighbours busy, the wavefront kernel 0.8 us with a wavefront to itself: in a
		// small batch the lane kernels' longest fragment is what the main stream ends up waiting for (131 k pairs of C4: 18 -> 11 ms with the bound
		// at 8192 anchors), in a large one the wavefront kernel's share is (C5, 500 k pairs: 652 vs 672 ms): the bound follows the batch.
		static const int wave_env = getenv("AL_TEST_HEAP_WAVE") ? atoi(getenv("AL_TEST_HEAP_WAVE")) : -1;                           // (tests lower it so that small fragments take the wavefront form)
		const uint32_t wave_na_min = wave_env >= 0 ? (uint32_t)wave_env : c->n_frag >= 400000 ? 16384u : 8192u;
		// Round 5: the heap in the lanes of a wavefront (k_anchor_heap_lanes: a fixed number of wave-wide instructions per pop instead of an LDS round trip
		// per sift level) for every flagged fragment of up to 126 lists; the serial forms remain for more lists and behind AL_HEAP_OLD=1 (tests, A/B).
		static const bool heap_old = getenv("AL_HEAP_OLD") && atoi(getenv("AL_HEAP_OLD")) == 1;
		if (!heap_old) {
#define LHL(NS, LO, ST) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_anchor_heap_lanes<NS, 16>), dim3(std::min(nl, 65536)), dim3(64), 0, ST, c->di.pos, c->frag_first.p, c->rd_len.p, c->mini_off.p, c->match.p, c->frag_nm.p, c->frag_na.p, \
		                                    c->a_off.p, c->anchors.p, (const uint32_t *)c->tie_list.p, (const uint32_t *)c->tie_frags.p, (const uint32_t *)n_heap_d, LO, c->counters.p, c->mi->k)
			LHL(2, 63, c->side); LHL(1, -1, c->aux[0]);
#undef LHL
			{ const uint32_t wave_na_min = 0xffffffffu; LHEAP(0, 64, 126, c->aux[2]); }
		} else {
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_anchor_heap_wave<128, 32>), dim3(std::min(nl, 65536)), dim3(64), 0, c->side, c->di.pos, c->frag_first.p, c->rd_len.p, c->mini_off.p, c->match.p, c->frag_nm.p, c->frag_na.p,
		                   c->a_off.p, c->anchors.p, (const uint32_t *)c->tie_list.p, (const uint32_t *)c->tie_frags.p, (const uint32_t *)n_heap_d, wave_na_min, c->counters.p, c->mi->k);
		LHEAP(48, 64, -1, c->aux[0]); LHEAP(96, 32, 48, c->aux[1]); LHEAP(0, 64, 96, c->aux[2]);
		}
#undef LHEAP
		for (int i = 0; i < 3; ++i) { AL_HIP_CHECK(hipEventRecord(c->ev_aux[i], c->aux[i])); AL_HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_aux[i], 0)); }
		AL_HIP_CHECK(hipEventRecord(evs[1], c->side));
		if (ev(ST_ANCHOR_HEAP)) return -1;
	}
	{
		// (the four chain_lds intervals are empty since the lane-per-fragment kernels of the <= 128-anchor fragments moved to ovl[1], beside the block
		//  sorts and the tile kernel: their time is inside anchor_sort_blk / chain_tile; the names stay so that the stage table keeps its columns)
		if (ev(ST_CHAIN_LDS32) || ev(ST_CHAIN_LDS48) || ev(ST_CHAIN_LDS64) || ev(ST_CHAIN_LDS128)) return -1;
		if (tiles_ok) {
			TileSched S; const uint32_t b[6] = {tile_from, std::max(tile_from, lb33), std::max(tile_from, lb65), std::max(tile_from, lb129), std::max(tile_from, lb257x), std::max(tile_from, lb513x)};
			for (int k = 0; k < 6; ++k) S.ent[k] = b[k]; S.ent[6] = (uint32_t)nl;
			S.item[0] = 0; for (int k = 0; k < 6; ++k) { const uint32_t per = 32u >> k; S.item[k + 1] = S.item[k] + (S.ent[k + 1] - S.ent[k] + per - 1) / per; }
			S.n_items = S.item[6];
			if (chain_tiles(c, order, S, (const uint32_t *)c->tie_list.p, lmin, first, lds_ok)) return -1;
		} else {
			if (ev(ST_SEG_FIND) || ev(ST_SEG_CHAIN_LDS)) return -1;
			if (c->ovl_pending) { AL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_ovl[3], 0)); c->ovl_pending = false; }
			const uint32_t tail = lds_ok ? lb129 : 0u;
			if (chain_legacy(c, order + tail, nl - (int)tail, lds_ok, (const uint32_t *)c->tie_list.p)) return -1;
			if (ev(ST_SEG_CHAIN_WAVE)) return -1;
		}
	}
	{   // second round: the fragments whose anchors the side stream merged, any size
		AL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_side[first ? 1 : 3], 0));
		if (!first && c->spec_pending) { AL_HIP_CHECK(hipStreamWaitEvent(s, c->ev_spec[1], 0)); c->spec_pending = false; }
		if (c->tie_frags.ensure((size_t)nl + 2)) return -1;
		uint32_t *cnt = (uint32_t *)(c->counters.p + 15);
		AL_HIP_CHECK(hipMemsetAsync(cnt, 0, 8, s));
		hipLaunchKernelGGL(k_collect_flagged, dim3((nl + 255) / 256), dim3(256), 0, s, order, nl, (const uint32_t *)c->tie_list.p, c->tie_frags.p, cnt);
		uint32_t n_tie = 0, n_spec_used = 0;
		AL_HIP_CHECK(hipMemcpyAsync(&n_tie, cnt, 4, hipMemcpyDeviceToHost, s));
		if (!first && c->n_spec > 0) AL_HIP_CHECK(hipMemcpyAsync(&n_spec_used, c->spec_cnt.p + 2, 4, hipMemcpyDeviceToHost, s));
		AL_HIP_CHECK(hipStreamSynchronize(s));
		if (!first && c->n_spec > 0) c->spec_idle = n_spec_used ? 0u : c->spec_idle + 1u;
		if (n_tie > 0) {
			size_t bytes = 0;    // ascending fragment ids: a deterministic order (the collection above appends atomically)
			if (c->tie_sorted.ensure((size_t)n_tie + 2)) return -1;
			AL_HIP_CHECK(rocprim::radix_sort_keys(nullptr, bytes, (const uint32_t *)c->tie_frags.p, c->tie_sorted.p, (int)n_tie, 0, 32, s));
			if (c->scan_tmp.ensure(bytes + 16)) return -1;
			AL_HIP_CHECK(rocprim::radix_sort_keys(c->scan_tmp.p, bytes, (const uint32_t *)c->tie_frags.p, c->tie_sorted.p, (int)n_tie, 0, 32, s));
			if (tiles_ok) {
				TileSched S; for (int k = 0; k < 6; ++k) { S.ent[k] = 0; S.item[k] = 0; } S.ent[6] = n_tie; S.item[6] = n_tie; S.n_items = n_tie;   // a fragment per item, any size
				if (chain_tiles(c, c->tie_sorted.p, S, nullptr, lmin, false, lds_ok)) return -1;
			} else if (chain_legacy(c, c->tie_sorted.p, (int)n_tie, lds_ok, nullptr)) return -1;
		}
		if (ev(ST_SEG_MERGE)) return -1;
	}
	AL_HIP_CHECK(hipGetLastError());
	return 0;
}
#undef LCH

int al_run_seed_stages(al_ctx_t *c)
{
	hipStream_t s = c->stream;
	AL_HIP_CHECK(hipSetDevice(c->device));
	AL_HIP_CHECK(hipMemsetAsync(c->counters.p, 0, 32 * sizeof(unsigned long long), s));
	AL_HIP_CHECK(hipEventRecord(c->ev[0], s));
	const int nr = c->n_reads, w = c->mi->w, k = c->mi->k;
	const bool sk_wide = al_sketch_wide(k, c->max_rd_len);                       // two words per window slot: k of 26 ... 28, reads beyond the packed entry's position bits
	if (nr > 0) hipLaunchKernelGGL(k_sketch, dim3((nr + 63) / 64), dim3(64), (size_t)w * 64 * (sk_wide ? 12 : 8), s, c->rd_seq.p, c->rd_off.p, c->rd_len.p, c->mini_off.p, c->mini.p, c->mini_cnt.p, nr, w, k, sk_wide ? -1 : al_sketch_pos_bits(k));
	AL_HIP_CHECK(hipEventRecord(c->ev[ST_SKETCH + 1], s));
	if (c->n_frag == 0) { for (int i = ST_SEED; i < ST_N; ++i) AL_HIP_CHECK(hipEventRecord(c->ev[i + 1], s)); return 0; }
	if (c->frag_meta.ensure((size_t)c->n_frag + 1)) return -1;
	hipLaunchKernelGGL(k_frag_meta, dim3((c->n_frag + 255) / 256), dim3(256), 0, s, c->frag_first.p, c->rd_len.p, c->n_frag, c->frag_meta.p);
	if (run_seed_chain(c, nullptr, c->n_frag, c->opt.mid_occ, 0, true)) return -1;
	// re-chain with max_occ for fragments whose best chain misses a mate (map.c:353-375)
	c->n_rechain = 0;
	if (c->opt.max_occ > c->opt.mid_occ) {
		uint32_t *cnt = (uint32_t *)(c->counters.p + 3);
		hipLaunchKernelGGL(k_rechain_test, dim3((c->n_frag + 255) / 256), dim3(256), 0, s, c->chained.p, c->a_off.p, c->u.p, c->uo.p, c->frag_nu.p, c->frag_rep.p, c->frag_first.p, c->n_frag, c->rechain_list.p, cnt);
		uint32_t n = 0;
		AL_HIP_CHECK(hipMemcpyAsync(&n, cnt, 4, hipMemcpyDeviceToHost, s));
		AL_HIP_CHECK(hipStreamSynchronize(s));
		c->n_rechain = n;
		if (n == 0 && c->n_spec > 0) ++c->spec_idle;                             // (no re-chain pass: nothing taken)
		if (n > 0) {
			if (c->a_off_p1.ensure(c->n_frag + 2) || c->frag_na_p1.ensure(c->n_frag + 1) || c->frag_rep_p1.ensure(c->n_frag + 1)) return -1;
			AL_HIP_CHECK(hipMemcpyAsync(c->a_off_p1.p, c->a_off.p, (size_t)(c->n_frag + 1) * 8, hipMemcpyDeviceToDevice, s));
			AL_HIP_CHECK(hipMemcpyAsync(c->frag_na_p1.p, c->frag_na.p, (size_t)c->n_frag * 4, hipMemcpyDeviceToDevice, s));
			AL_HIP_CHECK(hipMemcpyAsync(c->frag_rep_p1.p, c->frag_rep.p, (size_t)c->n_frag * 4, hipMemcpyDeviceToDevice, s));
			// The chain list u[] of fragment f lives at a_off[f] + f, which keeps the regions of two fragments apart only if their
			// anchor offsets grow with f.  k_rechain_test appends in arbitrary (atomic) order, so put the list in ascending
			// fragment order first -- that also makes the second-pass layout the same on every run -- and start the second pass
			// n_frag slots further so that its u[] entries cannot land on those of the last first-pass fragments.
			{
				size_t bytes = 0;
				if (c->rechain_sorted.ensure(n + 1)) return -1;
				AL_HIP_CHECK(rocprim::radix_sort_keys(nullptr, bytes, (const uint32_t *)c->rechain_list.p, c->rechain_sorted.p, (int)n, 0, 32, s));
				if (c->scan_tmp.ensure(bytes + 16)) return -1;
				AL_HIP_CHECK(rocprim::radix_sort_keys(c->scan_tmp.p, bytes, (const uint32_t *)c->rechain_list.p, c->rechain_sorted.p, (int)n, 0, 32, s));
			}
			if (run_seed_chain(c, c->rechain_sorted.p, (int)n, c->opt.max_occ, c->n_anchor_pass1 + (uint64_t)c->n_frag, false)) return -1;
		}
	}
	AL_HIP_CHECK(hipEventRecord(c->ev[ST_RECHAIN + 1], s));
	return 0;
}

extern "C" int al_batch_run(al_ctx_t *c)
{
	if (!c) return -1;
	al_nomem_flag() = false;
	// out of device memory: the grow-only buffers are what fills it, so give all of them back (the batch has to be uploaded
	// again, smaller) and report it as such
	auto failed = [&]() -> int { if (!al_nomem_flag()) return -1; ctx_release_buffers(c); c->n_frag = c->n_reads = 0; c->ran = false; return AL_ERR_NOMEM; };
	{   // test hook: behave as if batches above a size did not fit (tests/test_gpu_sam.py drives the halving of the file driver with it)
		static const char *lim = getenv("AL_TEST_NOMEM_ABOVE");
		if (lim && c->n_frag > atoi(lim)) { fprintf(stderr, "[airlift] AL_TEST_NOMEM_ABOVE: pretending %d fragments do not fit\n", c->n_frag); al_nomem_flag() = true; return failed(); }
	}
	if (al_run_seed_stages(c)) return failed();
	for (int attempt = 0; ; ++attempt) {
		if ((c->P.dbg2 >> 20) & 1) { for (int i = ST_REGS; i < ST_N; ++i) AL_HIP_CHECK(hipEventRecord(c->ev[i + 1], c->stream)); AL_HIP_CHECK(hipStreamSynchronize(c->stream)); break; }   // (AL_DBG2 bit 20, timing experiments: seed / sort / chain stages only)
		if (al_run_align_stage(c)) return failed();
		AL_HIP_CHECK(hipEventRecord(c->ev[ST_COMPACT + 1], c->stream));
		AL_HIP_CHECK(hipStreamSynchronize(c->stream));
		unsigned long long ovf = 0;
		AL_HIP_CHECK(hipMemcpy(&ovf, c->counters.p + 9, 8, hipMemcpyDeviceToHost));
		if (ovf == 0 || attempt >= 8) break;
		// long CIGARs (repeat-rich or indel-rich batches) did not fit the arena: the alignment stage only reads the chaining
		// results, so it is simply run again with twice the arena
		if (getenv("AL_TRACE")) fprintf(stderr, "[airlift] trace: CIGAR arena overflow (%llu), re-running the alignment stage with twice the arena\n", ovf);
		al_align_grow_arena(c);
		AL_HIP_CHECK(hipMemsetAsync(c->counters.p + 4, 0, 8 * sizeof(unsigned long long), c->stream));     // [4..11]: stage statistics, error words, arena cursor
		AL_HIP_CHECK(hipMemsetAsync(c->counters.p + 14, 0, sizeof(unsigned long long), c->stream));
	}
	for (int i = 0; i < ST_N; ++i) { float ms = 0; if (hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]) != hipSuccess) ms = 0; c->ms_stage[i] = ms; }
	float tot = 0; (void)hipEventElapsedTime(&tot, c->ev[0], c->ev[ST_N]); c->ms_total = tot;
	{ float a = 0, b = 0; if (hipEventElapsedTime(&a, c->ev_side[0], c->ev_side[1]) != hipSuccess) a = 0; if (c->n_rechain == 0 || hipEventElapsedTime(&b, c->ev_side[2], c->ev_side[3]) != hipSuccess) b = 0; c->ms_side = a + b; (void)hipGetLastError(); }
	c->ran = true;
	{ static const bool guard = getenv("AL_TEST_GUARD") != nullptr; if (guard && al_dev_guard_check()) fprintf(stderr, "[airlift] GUARD: violations after al_batch_run\n"); }
	// counters + algorithmic bytes
	unsigned long long h[16]; AL_HIP_CHECK(hipMemcpy(h, c->counters.p, sizeof(h), hipMemcpyDeviceToHost));
	if ((c->P.dbg >> 24) & 1) { unsigned long long t[8]; AL_HIP_CHECK(hipMemcpy(t, c->counters.p + 24, sizeof(t), hipMemcpyDeviceToHost));
		fprintf(stderr, "[airlift] tile kernel profile (cycles of thread 0, all blocks, both passes): setup %llu load %llu cut %llu dp %llu emit %llu copy %llu; tiles %llu\n", t[0], t[1], t[2], t[3], t[4], t[5], t[6]); }
	if ((c->P.dbg2 >> 3) & 1) { unsigned long long t[8]; AL_HIP_CHECK(hipMemcpy(t, c->counters.p + 24, sizeof(t), hipMemcpyDeviceToHost));
		fprintf(stderr, "[airlift] lane chaining kernels (cycles per wavefront, all launches of the batch): setup %llu load %llu recurrence %llu ends/backtrack/order %llu copy %llu; wavefronts %llu\n", t[5] ? t[0] / t[5] : 0, t[5] ? t[1] / t[5] : 0, t[5] ? t[2] / t[5] : 0, t[5] ? t[3] / t[5] : 0, t[5] ? t[4] / t[5] : 0, t[5]); }
	if ((c->P.dbg2 >> 4) & 1) { unsigned long long t[8]; AL_HIP_CHECK(hipMemcpy(t, c->counters.p + 24, sizeof(t), hipMemcpyDeviceToHost));
		fprintf(stderr, "[airlift] k_regs_heavy (lane 0 = mate 0, cycles summed over fragments): gen_regs %llu set_parent %llu squeeze %llu; hits %llu; whole blocks %llu over %llu fragments\n", t[0], t[1], t[2], t[3], t[4], t[5]); }
	if (getenv("AL_TRACE")) { fprintf(stderr, "[airlift] trace: counters"); for (int i = 0; i < 16; ++i) fprintf(stderr, " [%d]=%llu", i, h[i]); fprintf(stderr, " rechain=%u\n", c->n_rechain); }
	al_batch_stat_t &st = c->stat; memset(&st, 0, sizeof(st));
	st.n_frag = c->n_frag; st.n_reads = c->n_reads; st.n_bases = c->n_bases;
	st.n_mini = ~0ULL; st.n_chain = ~0ULL;   // filled lazily by al_batch_stat()
	st.n_anchor = c->n_anchor_total; st.n_rechain = c->n_rechain; st.n_heap_fallback = h[0]; st.n_sort_tie_flag = h[1] + h[10];
	st.n_regs_aln = h[4]; st.n_refbases = h[5]; st.n_cigar = h[6];
	st.bytes_in = c->stat_bytes_in; st.bytes_out = 48 * st.n_regs_aln + 4 * st.n_cigar;
	st.ms_total = c->ms_total; st.n_stage = ST_N; st.ms_side_stream = c->ms_side; st.n_chain_fallback = c->n_chain_fallback; c->n_chain_fallback = 0;
	for (int i = 0; i < 10; ++i) { st.dp_jobs[i] = c->stat_dp_jobs[i]; st.dp_target_bases[i] = c->stat_dp_tbases[i]; }
	for (int i = 0; i < ST_N; ++i) st.ms_kernel[i] = c->ms_stage[i];
	return 0;
}

__global__ void k_sum_u32(const uint32_t *a, int n, unsigned long long *out)
{
	unsigned long long v = 0;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v += a[i];
	for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
	if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}
extern "C" void al_batch_stat(const al_ctx_t *cc, al_batch_stat_t *out)
{
	al_ctx_t *c = const_cast<al_ctx_t *>(cc);
	al_batch_stat_t &st = c->stat;
	if (c->ran && st.n_mini == ~0ULL && hipSetDevice(c->device) == hipSuccess) {
		unsigned long long h[2] = {0, 0};
		(void)hipMemsetAsync(c->counters.p + 12, 0, 16, c->stream);
		if (c->n_reads) hipLaunchKernelGGL(k_sum_u32, dim3(256), dim3(256), 0, c->stream, c->mini_cnt.p, c->n_reads, c->counters.p + 12);
		if (c->n_frag) hipLaunchKernelGGL(k_sum_u32, dim3(256), dim3(256), 0, c->stream, c->frag_nu.p, c->n_frag, c->counters.p + 13);
		(void)hipMemcpyAsync(h, c->counters.p + 12, 16, hipMemcpyDeviceToHost, c->stream);
		(void)hipStreamSynchronize(c->stream);
		st.n_mini = h[0]; st.n_chain = h[1];
		// B = B_in + M*16 + A*8 + 2*A*16 + W/2 + B_out
		st.algorithmic_bytes = (double)st.bytes_in + 16.0 * st.n_mini + 8.0 * st.n_anchor + 32.0 * st.n_anchor + 0.5 * st.n_refbases + (double)st.bytes_out;
	}
	*out = st;
}

// ---------------------------------------------------------------------------------------------
// stage taps
extern "C" int al_dbg_minimizers(al_ctx_t *c, int r, uint64_t *xy, int cap)
{
	if (!c || r < 0 || r >= c->n_reads) return -1;
	uint32_t n = 0;
	AL_HIP_CHECK(hipSetDevice(c->device));
	AL_HIP_CHECK(hipMemcpy(&n, c->mini_cnt.p + r, 4, hipMemcpyDeviceToHost));
	int m = (int)n < cap ? (int)n : cap;
	uint64_t mo = 0;
	if (c->dev_batch) AL_HIP_CHECK(hipMemcpy(&mo, c->mini_off.p + r, 8, hipMemcpyDeviceToHost)); else mo = c->h_mini_off[r];
	if (m > 0) AL_HIP_CHECK(hipMemcpy(xy, c->mini.p + mo, (size_t)m * 16, hipMemcpyDeviceToHost));
	return (int)n;
}

extern "C" int al_dbg_anchors(al_ctx_t *c, int f, uint64_t *xy, int cap, int *rep_len)
{
	if (!c || f < 0 || f >= c->n_frag) return -1;
	uint32_t n = 0; uint64_t off = 0; int32_t rep = 0;
	AL_HIP_CHECK(hipSetDevice(c->device));
	const bool p1 = c->n_rechain > 0;   // "SD"/"RS" taps are printed before the re-chain pass (map.c:333-338)
	AL_HIP_CHECK(hipMemcpy(&n, (p1 ? c->frag_na_p1.p : c->frag_na.p) + f, 4, hipMemcpyDeviceToHost));
	AL_HIP_CHECK(hipMemcpy(&off, (p1 ? c->a_off_p1.p : c->a_off.p) + f, 8, hipMemcpyDeviceToHost));
	AL_HIP_CHECK(hipMemcpy(&rep, (p1 ? c->frag_rep_p1.p : c->frag_rep.p) + f, 4, hipMemcpyDeviceToHost));
	if (rep_len) *rep_len = rep;
	int m = (int)n < cap ? (int)n : cap;
	if (m > 0) AL_HIP_CHECK(hipMemcpy(xy, c->anchors.p + off, (size_t)m * 16, hipMemcpyDeviceToHost));
	return (int)n;
}

extern "C" int al_dbg_chains(al_ctx_t *c, int f, uint64_t *u, int cap_u, uint64_t *xy, int cap_a)
{
	if (!c || f < 0 || f >= c->n_frag) return -1;
	uint32_t nu = 0; uint64_t off = 0;
	AL_HIP_CHECK(hipSetDevice(c->device));
	AL_HIP_CHECK(hipMemcpy(&nu, c->frag_nu.p + f, 4, hipMemcpyDeviceToHost));
	AL_HIP_CHECK(hipMemcpy(&off, c->a_off.p + f, 8, hipMemcpyDeviceToHost));
	int m = (int)nu < cap_u ? (int)nu : cap_u;
	if (m > 0) AL_HIP_CHECK(hipMemcpy(u, c->u.p + off + f, (size_t)m * 8, hipMemcpyDeviceToHost));
	// the chains' anchors lie at their segments' places (uo[]): gathered here into the reference's back-to-back order
	std::vector<uint32_t> uo((size_t)(m > 0 ? m : 1));
	if (m > 0) AL_HIP_CHECK(hipMemcpy(uo.data(), c->uo.p + off + f, (size_t)m * 4, hipMemcpyDeviceToHost));
	uint32_t fna = 0; AL_HIP_CHECK(hipMemcpy(&fna, c->frag_na.p + f, 4, hipMemcpyDeviceToHost));
	std::vector<uint64_t> all((size_t)fna * 2 + 2);
	if (fna > 0 && m > 0) AL_HIP_CHECK(hipMemcpy(all.data(), c->chained.p + off, (size_t)fna * 16, hipMemcpyDeviceToHost));
	int64_t na = 0;
	for (int i = 0; i < m; ++i) {
		const uint32_t cnt = (uint32_t)u[i];
		for (uint32_t j = 0; j < cnt && na < cap_a; ++j, ++na) { if ((uint64_t)uo[i] + j >= fna) return -1; xy[2 * na] = all[2 * ((size_t)uo[i] + j)]; xy[2 * na + 1] = all[2 * ((size_t)uo[i] + j) + 1]; }
	}
	return (int)nu;
}

static int alser_count_resident(al_ctx_t *c, int64_t *total);
extern "C" int al_dbg_alser_count(al_ctx_t *c, int64_t *total)
{   // expects a resident batch of single-segment fragments on which al_batch_run() has been called
	if (!c || !c->ran) return -1;
	return alser_count_resident(c, total);
}

// a8 as a product entry point: seed stages only (no extension) on the resident batch of single-segment fragments, then the
// as-shipped fork's candidate counter (map.c:299-312)
extern "C" int al_batch_count_candidates(al_ctx_t *c, int64_t *total)
{
	if (!c || !total) return -1;
	if (al_run_seed_stages(c)) return -1;
	return alser_count_resident(c, total);
}

static int alser_count_resident(al_ctx_t *c, int64_t *total)
{
	AL_HIP_CHECK(hipSetDevice(c->device));
	AL_HIP_CHECK(hipMemsetAsync(c->counters.p + 2, 0, 8, c->stream));
	if (c->n_frag) hipLaunchKernelGGL(k_alser_count, dim3((c->n_frag + 255) / 256), dim3(256), 0, c->stream, c->anchors.p, c->n_rechain ? c->a_off_p1.p : c->a_off.p, c->n_rechain ? c->frag_na_p1.p : c->frag_na.p, c->frag_first.p, c->rd_len.p, c->n_frag, c->opt.min_cnt, c->counters.p + 2);
	unsigned long long v = 0;
	AL_HIP_CHECK(hipMemcpyAsync(&v, c->counters.p + 2, 8, hipMemcpyDeviceToHost, c->stream));
	AL_HIP_CHECK(hipStreamSynchronize(c->stream));
	*total = (int64_t)v;
	return 0;
}

// bulk tap: copies a whole named device array to the host (tests); returns bytes copied or -1
extern "C" int64_t al_dbg_copy(al_ctx_t *c, const char *name, void *dst, int64_t max_bytes)
{
	if (!c) return -1;
	const void *src = nullptr; int64_t bytes = 0; const bool p1 = c->n_rechain > 0;
	const int64_t nf = c->n_frag, nr = c->n_reads;
	if (!strcmp(name, "mini")) src = c->mini.p, bytes = (int64_t)c->mini_total * 16;
	else if (!strcmp(name, "mini_cnt")) src = c->mini_cnt.p, bytes = nr * 4;
	else if (!strcmp(name, "frag_na")) src = c->frag_na.p, bytes = nf * 4;
	else if (!strcmp(name, "frag_na_p1")) src = p1 ? c->frag_na_p1.p : c->frag_na.p, bytes = nf * 4;
	else if (!strcmp(name, "frag_rep")) src = c->frag_rep.p, bytes = nf * 4;
	else if (!strcmp(name, "frag_rep_p1")) src = p1 ? c->frag_rep_p1.p : c->frag_rep.p, bytes = nf * 4;
	else if (!strcmp(name, "frag_nu")) src = c->frag_nu.p, bytes = nf * 4;
	else if (!strcmp(name, "a_off")) src = c->a_off.p, bytes = (nf + 1) * 8;
	else if (!strcmp(name, "a_off_p1")) src = p1 ? c->a_off_p1.p : c->a_off.p, bytes = (nf + 1) * 8;
	else if (!strcmp(name, "anchors")) src = c->anchors.p, bytes = (int64_t)c->n_anchor_total * 16;
	else if (!strcmp(name, "chained")) src = c->chained.p, bytes = (int64_t)c->n_anchor_total * 16;
	else if (!strcmp(name, "u")) src = c->u.p, bytes = (int64_t)(c->n_anchor_total + nf + 1) * 8;
	else if (!strcmp(name, "uo")) src = c->uo.p, bytes = (int64_t)(c->n_anchor_total + nf + 1) * 4;
	else if (!strcmp(name, "mini_off") && !c->dev_batch) { bytes = (nr + 1) * 8; if (bytes > max_bytes) bytes = max_bytes; memcpy(dst, c->h_mini_off.data(), bytes); return bytes; }
	else if (!strcmp(name, "mini_off")) src = c->mini_off.p, bytes = (nr + 1) * 8;
	else return -1;
	if (bytes > max_bytes) bytes = max_bytes;
	if (hipSetDevice(c->device) != hipSuccess) return -1;
	if (bytes > 0 && hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
	return bytes;
}

// Flat upload for synthetic / benchmark batches: sequences concatenated in read order, names = "<prefix><fragment index>"
// for every read of a fragment (what BBMap rename.sh produces upstream: extract_sequence.sh:18).
extern "C" int al_batch_upload_flat(al_ctx_t *c, int n_frag, const int *n_segs, const int *qlens, const char *seq_concat, const char *name_prefix, int64_t first_index)
{
	int nr = 0; for (int f = 0; f < n_frag; ++f) nr += n_segs[f];
	std::vector<const char *> seqs(nr), names(nr); std::vector<std::string> nm(n_frag);
	const char *p = seq_concat; int r = 0;
	for (int f = 0; f < n_frag; ++f) {
		nm[f] = std::string(name_prefix ? name_prefix : "") + std::to_string(first_index + f);
		for (int j = 0; j < n_segs[f]; ++j, ++r) { seqs[r] = p; p += qlens[r]; names[r] = nm[f].c_str(); }
	}
	return al_batch_upload(c, n_frag, n_segs, qlens, seqs.data(), names.data());
}

// ---------------------------------------------------------------------------------------------
// coordinate order for sorted BAM output: stable LSD radix sort of (rid<<32|pos, record index) on the device
#include "al_bam.h"
int al_sort_keys(al_ctx_t *c, const uint64_t *keys, uint32_t *perm, size_t n)
{
	if (!c) return -1;
	if (n == 0) return 0;
	if (n >= (1ULL << 31)) { fprintf(stderr, "[airlift] al_sort_keys: too many records for one sort\n"); return -1; }
	AL_HIP_CHECK(hipSetDevice(c->device));
	uint64_t *d_k = nullptr, *d_k2 = nullptr; uint32_t *d_v = nullptr, *d_v2 = nullptr; void *tmp = nullptr; size_t bytes = 0;
	int rc = -1;
	if (hipMalloc((void **)&d_k, n * 8) == hipSuccess && hipMalloc((void **)&d_k2, n * 8) == hipSuccess && hipMalloc((void **)&d_v, n * 4) == hipSuccess && hipMalloc((void **)&d_v2, n * 4) == hipSuccess &&
	    hipMemcpyAsync(d_k, keys, n * 8, hipMemcpyHostToDevice, c->stream) == hipSuccess) {
		hipLaunchKernelGGL(k_iota_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_v, (uint32_t)n);
		if (rocprim::radix_sort_pairs(nullptr, bytes, d_k, d_k2, d_v, d_v2, (int)n, 0, 64, c->stream) == hipSuccess && hipMalloc(&tmp, bytes + 16) == hipSuccess &&
		    rocprim::radix_sort_pairs(tmp, bytes, d_k, d_k2, d_v, d_v2, (int)n, 0, 64, c->stream) == hipSuccess &&
		    hipMemcpyAsync(perm, d_v2, n * 4, hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess) rc = 0;
	}
	(void)hipFree(d_k); (void)hipFree(d_k2); (void)hipFree(d_v); (void)hipFree(d_v2); (void)hipFree(tmp);
	if (rc) { (void)hipGetLastError(); fprintf(stderr, "[airlift] al_sort_keys: device sort failed\n"); }      // (not sticky: the caller falls back to a host sort)
	return rc;
}

// ---------------------------------------------------------------------------------------------
// Token batches: the reads of a batch are fixed-length windows of longer source sequences (AirLift tiles the
// updated regions into read-sized tokens, gaps_to_fasta.py:31-36, then aligns them single-end, align_gaps.sh:14-15).  The
// source stays packed in HBM (4 bit/base); a kernel cuts the windows, so the host neither writes nor parses the
// read_size/skip-fold larger token FASTA.
struct Nt4Tab { uint8_t t[256]; };
__global__ void k_pack_ref(const uint8_t *, uint64_t, uint32_t *, uint64_t, Nt4Tab);       // al_index_dev.hip
struct al_winsrc_s { uint32_t *S4 = nullptr; uint64_t n_bases = 0; int device = 0; };

extern "C" al_winsrc_t *al_winsrc_create(al_ctx_t *c, const char *ascii, uint64_t n_bases)
{
	if (!c) return nullptr;
	if (hipSetDevice(c->device) != hipSuccess) return nullptr;
	al_winsrc_t *w = new al_winsrc_t(); w->n_bases = n_bases; w->device = c->device;
	const uint64_t n_words = (n_bases + 7) / 8 + 8;
	uint8_t *d_ascii = nullptr;
	Nt4Tab T; memcpy(T.t, al_nt4(), 256);
	if (hipMalloc((void **)&w->S4, n_words * 4) != hipSuccess || hipMalloc((void **)&d_ascii, n_bases + 16) != hipSuccess ||
	    hipMemcpyAsync(d_ascii, ascii, n_bases, hipMemcpyHostToDevice, c->stream) != hipSuccess) { (void)hipFree(w->S4); (void)hipFree(d_ascii); delete w; return nullptr; }
	hipLaunchKernelGGL(k_pack_ref, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, c->stream, d_ascii, n_bases, w->S4, n_words, T);
	(void)hipStreamSynchronize(c->stream);
	(void)hipFree(d_ascii);
	return w;
}
extern "C" void al_winsrc_destroy(al_winsrc_t *w) { if (!w) return; (void)hipSetDevice(w->device); (void)hipFree(w->S4); delete w; }

__global__ void __launch_bounds__(256)
k_make_windows(const uint32_t *__restrict__ S4, const uint64_t *__restrict__ start, int n_tok, int len, int wpr, uint32_t *__restrict__ rd_seq)
{   // thread per (window, output word): eight bases from bit offset 4 * (start & 7) of two source words; bases past the window are zero
	const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= (uint64_t)n_tok * wpr) return;
	const int j = (int)(g / wpr), w = (int)(g % wpr);
	uint32_t v = 0;
	if (8 * w < len) {
		const uint64_t src = start[j] + 8ULL * w; const int sh = (int)(src & 7) << 2;
		v = (uint32_t)((((uint64_t)S4[(src >> 3) + 1] << 32) | S4[src >> 3]) >> sh);
		const int left = len - 8 * w;
		if (left < 8) v &= (1u << (4 * left)) - 1u;
	}
	rd_seq[(uint64_t)j * wpr + w] = v;
}

extern "C" int al_batch_upload_windows(al_ctx_t *c, const al_winsrc_t *src, int n_tok, const uint64_t *start, int read_len, const char *const *qnames)
{
	if (!c || !src || n_tok < 0 || read_len <= 0 || src->device != c->device) return -1;
	AL_HIP_CHECK(hipSetDevice(c->device));
	const int k = c->mi->k, wpr = (read_len + 7) / 8 + 1;
	c->n_frag = n_tok; c->n_reads = n_tok; c->ran = false; c->max_qlen_sum = read_len; c->max_rd_len = n_tok ? read_len : 0; c->dev_batch = false;
	if (read_len >= AL_MAX_READ_LEN) { fprintf(stderr, "[airlift] tokens of %d bases exceed the limit of the GPU path\n", read_len); return -3; }
	c->h_rd_len.assign(n_tok + 1, (uint32_t)read_len); c->h_rd_len[n_tok] = 0;
	c->h_rd_off.resize(n_tok + 1); c->h_mini_off.resize(n_tok + 1); c->h_flip.assign(n_tok, 0);
	c->h_frag_first.resize(n_tok + 1); c->h_frag_hash.resize(n_tok + 1);
	const uint64_t mper = (uint64_t)(read_len >= k ? read_len - k + 1 : 0) + 1;
	for (int i = 0; i <= n_tok; ++i) { c->h_rd_off[i] = (uint64_t)i * wpr; c->h_mini_off[i] = (uint64_t)i * mper; c->h_frag_first[i] = (uint32_t)i; }
	al_parallel_for(c->n_threads, (size_t)n_tok, [&](size_t lo, size_t hi, int) {
		for (size_t i = lo; i < hi; ++i) c->h_frag_hash[i] = qname_hash(qnames ? qnames[i] : nullptr, read_len, c->opt.seed);
	});
	for (int i = 0; i < n_tok; ++i) if (start[i] + (uint64_t)read_len > src->n_bases) { fprintf(stderr, "[airlift] al_batch_upload_windows: window %d leaves the source\n", i); return -2; }
	const uint64_t words = (uint64_t)n_tok * wpr, mtot = (uint64_t)n_tok * mper, bases = (uint64_t)n_tok * read_len;
	c->n_bases = bases; c->mini_total = mtot; c->seq_words = words;
	c->stat_bytes_in = (uint64_t)n_tok * (((uint64_t)read_len * 3 + 7) / 8);
	hipStream_t s = c->stream; const int n_frag = n_tok, n_reads = n_tok;
	if (c->rd_seq.ensure(words + 1) || c->rd_off.ensure(n_reads + 1) || c->rd_len.ensure(n_reads + 1) || c->frag_first.ensure(n_frag + 1) || c->frag_hash.ensure(n_frag + 1) ||
	    c->mini_off.ensure(n_reads + 1) || c->mini.ensure(mtot + 1) || c->mini_cnt.ensure(n_reads + 1) || c->match.ensure(mtot + 1) || c->heap_ws.ensure(mtot + 1) ||
	    c->frag_nm.ensure(n_frag + 1) || c->frag_na.ensure(n_frag + 1) || c->frag_rep.ensure(n_frag + 1) || c->frag_nu.ensure(n_frag + 1) || c->a_off.ensure(n_frag + 2) ||
	    c->rechain_list.ensure(n_frag + 1) || c->tmp_u32.ensure(n_frag + 2) || c->tmp_u64.ensure(n_frag + 2) || c->counters.ensure(32) || c->tmp_u64b.ensure(n_tok + 1)) return -1;
	AL_HIP_CHECK(hipMemcpyAsync(c->tmp_u64b.p, start, (size_t)n_tok * 8, hipMemcpyHostToDevice, s));
	if (words) hipLaunchKernelGGL(k_make_windows, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, src->S4, c->tmp_u64b.p, n_tok, read_len, wpr, c->rd_seq.p);
	AL_HIP_CHECK(hipMemsetAsync(c->rd_seq.p + words, 0, 4, s));
	AL_HIP_CHECK(hipMemcpyAsync(c->rd_off.p, c->h_rd_off.data(), (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, s));
	AL_HIP_CHECK(hipMemcpyAsync(c->rd_len.p, c->h_rd_len.data(), (size_t)(n_reads + 1) * 4, hipMemcpyHostToDevice, s));
	AL_HIP_CHECK(hipMemcpyAsync(c->frag_first.p, c->h_frag_first.data(), (size_t)(n_frag + 1) * 4, hipMemcpyHostToDevice, s));
	AL_HIP_CHECK(hipMemcpyAsync(c->frag_hash.p, c->h_frag_hash.data(), (size_t)(n_frag + 1) * 4, hipMemcpyHostToDevice, s));
	AL_HIP_CHECK(hipMemcpyAsync(c->mini_off.p, c->h_mini_off.data(), (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, s));
	AL_HIP_CHECK(hipStreamSynchronize(s));
	return 0;
}
