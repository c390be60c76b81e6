// al_kernels_seed.hip -- CDNA4 (gfx950) kernels K1..K4 of the re-alignment path:
//   K1 k_sketch       minimizer sketch, one lane per read            (mm_sketch, sketch.c:77-143)
//   K2 k_seed         hash probes + occurrence filter, lane/fragment  (collect_matches, map.c:90-123; mm_idx_get, index.c:81-98)
//   K3 k_anchor_sort_small / k_anchor_sort / k_anchor_heap   anchor expansion + x-sort, wave/fragment; exact heap merge of
//                     equal-key fragments, lane/fragment          (collect_seed_hits_heap, map.c:149-213)
//   K4 k_chain_lds    chaining DP + backtrack, lane/fragment, rows in LDS; k_chain: wave/fragment for > 128 anchors
//                                                                  (mm_chain_dp, chain.c:22-162)
// Integer / byte work, HBM- and latency-bound: no MFMA.  64-wide wavefronts throughout.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "al_internal.h"
#include "al_device.h"
#include "al_dev_sort.h"
#include <rocprim/warp/warp_scan.hpp>
typedef rocprim::warp_scan<int32_t, 64> ChainWarpScan;

// =============================================================================================
// K1: sketch.  One lane per read runs the reference's streaming window algorithm verbatim (so ties,
// N resets and the l-counter quirks are reproduced); the w-entry ring lives in LDS laid out
// [slot][lane] so that a wave's accesses hit 64 consecutive banks.
// =============================================================================================
__device__ __forceinline__ uint64_t d_hash64m(uint64_t key, uint64_t mask)
{   // sketch.c:28-38
	key = (~key + (key << 21)) & mask; key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & mask; key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & mask; key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

// W: the window as a constant (the loops over the ring unroll, their LDS reads go out back to back), 0 = the argument
template <int W>
__device__ __forceinline__ void
d_sketch(const uint32_t *__restrict__ rd_seq, const uint64_t *__restrict__ rd_off, const uint32_t *__restrict__ rd_len,
         const uint64_t *__restrict__ mini_off, AlAnchor *__restrict__ mini, uint32_t *__restrict__ mini_cnt,
         int n_reads, const int w_arg, int k, int pb)
{
	const int w = W ? W : w_arg;
	// One 64-bit ring entry per window slot: hash << pb | pos << 1 | strand (a valid entry always has span == k; pb = min(22,
	// 64 - 2k) bits hold position and strand: reads shorter than 2^(pb-1) bases -- 2 Mbases for k <= 21, 8192 for k = 25; the
	// upload checks it), UINT64_MAX = no k-mer.  The reference's comparisons are on x = hash<<8|span, i.e. on the
	// hash part only (E >> pb); "same x, different y" is "same hash part, different entry".
	extern __shared__ uint64_t lds[];           // ring[w][64]
	const int lane = threadIdx.x;
	const int r = blockIdx.x * 64 + lane;
	uint64_t *ring = lds + lane;
	if (r >= n_reads) return;
	const uint32_t len = rd_len[r];
	const uint32_t *seq = rd_seq + rd_off[r];
	AlAnchor *out = mini + mini_off[r];
	uint32_t cnt = 0;
	const uint64_t shift1 = 2 * (k - 1), mask = (1ULL << 2 * k) - 1;
	uint64_t kmer0 = 0, kmer1 = 0, mn = UINT64_MAX;
	int l = 0, buf_pos = 0, min_pos = 0;
	for (int j = 0; j < w; ++j) ring[j * 64] = UINT64_MAX;
#define HX(e) ((e) >> pb)
	const uint64_t ymask = (1ULL << pb) - 1ULL;
#define EMIT(e) do { const uint64_t e__ = (e); out[cnt].x = HX(e__) << 8 | (uint64_t)k; out[cnt].y = e__ & ymask; ++cnt; } while (0)
	// (round 6) a word of eight bases per outer step, the next word asked for at the top of the step (no branch around the load: `if ((i & 7) == 0) word = ...` made
	// every eighth base wait for its load and for the minimizers stored since)
	const uint32_t nw = (len + 7) >> 3;
	uint32_t wnext = nw ? seq[0] : 0u;
	for (uint32_t wi = 0; wi < nw; ++wi) {
	const uint32_t word = wnext;
	wnext = seq[wi + 1 < nw ? wi + 1 : nw - 1];
	const uint32_t iend = (wi + 1) * 8 < len ? (wi + 1) * 8 : len;
	for (uint32_t i = wi * 8; i < iend; ++i) {
		const int c = (word >> ((i & 7) << 2)) & 0xf;
		uint64_t info = UINT64_MAX;
		if (c < 4) {
			kmer0 = (kmer0 << 2 | (uint64_t)c) & mask;
			kmer1 = (kmer1 >> 2) | (3ULL ^ (uint64_t)c) << shift1;
			if (kmer0 == kmer1) continue;                                   // sketch.c:108
			const int z = kmer0 < kmer1 ? 0 : 1;
			++l;
			if (l >= k) info = d_hash64m(z ? kmer1 : kmer0, mask) << pb | (uint64_t)i << 1 | (uint64_t)z;
		} else l = 0;
		ring[buf_pos * 64] = info;
		const uint64_t mx = HX(mn);
		if (l == w + k - 1 && mn != UINT64_MAX) {                           // sketch.c:117-122
			for (int j = buf_pos + 1; j < w; ++j) { const uint64_t e = ring[j * 64]; if (mx == HX(e) && e != mn) EMIT(e); }
			for (int j = 0; j < buf_pos; ++j)     { const uint64_t e = ring[j * 64]; if (mx == HX(e) && e != mn) EMIT(e); }
		}
		if (HX(info) <= mx) {                                               // sketch.c:123-125 (UINT64_MAX >> pb is the largest hash part)
			if (l >= w + k && mn != UINT64_MAX) EMIT(mn);
			mn = info; min_pos = buf_pos;
		} else if (buf_pos == min_pos) {                                    // sketch.c:126-138
			if (l >= w + k - 1 && mn != UINT64_MAX) EMIT(mn);
			mn = UINT64_MAX;
			// (dup: another entry of the window has the new minimum's hash -- the only case in which the two loops below emit anything.  Some lane of the wavefront
			//  is in this branch at nearly every base, so what it costs is paid per base by all 64: the second pair of loops only where it has something to write)
			bool dup = false;
			// (slots buf_pos + 1 ... w - 1, then 0 ... buf_pos -- oldest to newest -- as ONE loop of w steps: it unrolls when the window is a constant)
#pragma unroll
			for (int t = 0; t < w; ++t) { int j = buf_pos + 1 + t; j = j >= w ? j - w : j; const uint64_t e = ring[j * 64]; if (HX(mn) >= HX(e)) { dup = HX(mn) == HX(e); mn = e, min_pos = j; } }
			if (dup && l >= w + k - 1 && mn != UINT64_MAX) {
				const uint64_t m2 = HX(mn);
				for (int j = buf_pos + 1; j < w; ++j) { const uint64_t e = ring[j * 64]; if (m2 == HX(e) && mn != e) EMIT(e); }
				for (int j = 0; j <= buf_pos; ++j)    { const uint64_t e = ring[j * 64]; if (m2 == HX(e) && mn != e) EMIT(e); }
			}
		}
		if (++buf_pos == w) buf_pos = 0;
	}
	}
	if (mn != UINT64_MAX) EMIT(mn);
#undef HX
#undef EMIT
	mini_cnt[r] = cnt;
}
// The same with TWO words per ring slot -- the hash, and position << 1 | strand -- for what the packed entry cannot hold: k of 26 ... 28 (56 hash bits leave 8 for position
// and strand) and reads beyond 2^(pb - 1) bases.  pb < 0 selects it.  Not the hot path: plain loops.
__device__ __forceinline__ void
d_sketch_wide(const uint32_t *__restrict__ rd_seq, const uint64_t *__restrict__ rd_off, const uint32_t *__restrict__ rd_len,
              const uint64_t *__restrict__ mini_off, AlAnchor *__restrict__ mini, uint32_t *__restrict__ mini_cnt, int n_reads, const int w, int k)
{
	extern __shared__ uint64_t lds[];           // rh[w][64] hashes, then rp[w][64] positions (32-bit)
	const int lane = threadIdx.x;
	const int r = blockIdx.x * 64 + lane;
	uint64_t *rh = lds + lane; uint32_t *rp = reinterpret_cast<uint32_t *>(lds + (size_t)w * 64) + lane;
	if (r >= n_reads) return;
	const uint32_t len = rd_len[r];
	const uint32_t *seq = rd_seq + rd_off[r];
	AlAnchor *out = mini + mini_off[r];
	uint32_t cnt = 0;
	const uint64_t shift1 = 2 * (k - 1), mask = (1ULL << 2 * k) - 1;
	uint64_t kmer0 = 0, kmer1 = 0, mh = UINT64_MAX; uint32_t mp = 0;
	int l = 0, buf_pos = 0, min_pos = 0;
	for (int j = 0; j < w; ++j) rh[j * 64] = UINT64_MAX, rp[j * 64] = 0;
#define EMITW(h, q) do { out[cnt].x = (h) << 8 | (uint64_t)k; out[cnt].y = (uint64_t)(q); ++cnt; } while (0)
	uint32_t word = 0;
	for (uint32_t i = 0; i < len; ++i) {
		if ((i & 7) == 0) word = seq[i >> 3];
		const int c = (word >> ((i & 7) << 2)) & 0xf;
		uint64_t ih = UINT64_MAX; uint32_t ip = 0;
		if (c < 4) {
			kmer0 = (kmer0 << 2 | (uint64_t)c) & mask;
			kmer1 = (kmer1 >> 2) | (3ULL ^ (uint64_t)c) << shift1;
			if (kmer0 == kmer1) continue;                                   // sketch.c:108
			const int z = kmer0 < kmer1 ? 0 : 1;
			++l;
			if (l >= k) { ih = d_hash64m(z ? kmer1 : kmer0, mask); ip = i << 1 | (uint32_t)z; }
		} else l = 0;
		rh[buf_pos * 64] = ih; rp[buf_pos * 64] = ip;
		if (l == w + k - 1 && mh != UINT64_MAX) {                           // sketch.c:117-122
			for (int j = buf_pos + 1; j < w; ++j) { const uint64_t h = rh[j * 64]; const uint32_t q = rp[j * 64]; if (mh == h && q != mp) EMITW(h, q); }
			for (int j = 0; j < buf_pos; ++j)     { const uint64_t h = rh[j * 64]; const uint32_t q = rp[j * 64]; if (mh == h && q != mp) EMITW(h, q); }
		}
		if (ih <= mh) {                                                     // sketch.c:123-125
			if (l >= w + k && mh != UINT64_MAX) EMITW(mh, mp);
			mh = ih; mp = ip; min_pos = buf_pos;
		} else if (buf_pos == min_pos) {                                    // sketch.c:126-138
			if (l >= w + k - 1 && mh != UINT64_MAX) EMITW(mh, mp);
			mh = UINT64_MAX;
			for (int j = buf_pos + 1; j < w; ++j) { const uint64_t h = rh[j * 64]; if (mh >= h) mh = h, mp = rp[j * 64], min_pos = j; }
			for (int j = 0; j <= buf_pos; ++j)    { const uint64_t h = rh[j * 64]; if (mh >= h) mh = h, mp = rp[j * 64], min_pos = j; }
			if (l >= w + k - 1 && mh != UINT64_MAX) {
				for (int j = buf_pos + 1; j < w; ++j) { const uint64_t h = rh[j * 64]; const uint32_t q = rp[j * 64]; if (mh == h && mp != q) EMITW(h, q); }
				for (int j = 0; j <= buf_pos; ++j)    { const uint64_t h = rh[j * 64]; const uint32_t q = rp[j * 64]; if (mh == h && mp != q) EMITW(h, q); }
			}
		}
		if (++buf_pos == w) buf_pos = 0;
	}
	if (mh != UINT64_MAX) EMITW(mh, mp);
#undef EMITW
	mini_cnt[r] = cnt;
}
extern "C" __global__ void __launch_bounds__(64)
k_sketch(const uint32_t *__restrict__ rd_seq, const uint64_t *__restrict__ rd_off, const uint32_t *__restrict__ rd_len,
         const uint64_t *__restrict__ mini_off, AlAnchor *__restrict__ mini, uint32_t *__restrict__ mini_cnt,
         int n_reads, int w, int k, int pb)
{
	if (pb < 0) d_sketch_wide(rd_seq, rd_off, rd_len, mini_off, mini, mini_cnt, n_reads, w, k);        // (k of 26 ... 28, or reads too long for the packed entry)
	else if (w == 11) d_sketch<11>(rd_seq, rd_off, rd_len, mini_off, mini, mini_cnt, n_reads, w, k, pb);    // (the short-read preset's window)
	else d_sketch<0>(rd_seq, rd_off, rd_len, mini_off, mini, mini_cnt, n_reads, w, k, pb);
}

// =============================================================================================
// K2: seed lookup.  One lane per fragment walks the fragment's minimizers in order (mate 1 then mate 2,
// positions of mate 2 offset by len(mate 1): collect_minimizers, map.c:64-77), probes the 16-byte-entry
// open-addressing table and applies the occurrence filter / repeat-length bookkeeping of collect_matches.
// =============================================================================================
__device__ __forceinline__ uint64_t d_idx_get(const uint64_t *__restrict__ tab, int tab_bits, uint64_t hash, bool &single)
{   // returns off<<32|n (single == false), the position word of a once-occurring minimizer (single == true), or 0 if absent
	const uint64_t tmask = (1ULL << tab_bits) - 1;
	uint64_t s = (hash * 0x9E3779B97F4A7C15ULL) >> (64 - tab_bits);
	single = false;
	for (;;) {
		const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(tab + 2 * s);   // one 16-B entry
		if ((e.x & ~AL_TAB_SINGLE) == hash + 1) { single = (e.x & AL_TAB_SINGLE) != 0; return e.y; }
		if (e.x == 0) return 0;
		s = (s + 1) & tmask;
	}
}
// position word k of a match's occurrence list
__device__ __forceinline__ uint64_t d_match_pos(const uint64_t *__restrict__ pos, uint32_t off_lo, uint32_t flags, uint32_t k)
{
	const uint64_t w = (uint64_t)off_lo | (uint64_t)(flags >> 16) << 32;
	return (flags >> 9 & 1u) ? w : pos[w + k];
}
// ... as an address and a select: the load is unconditional (a once-occurring minimizer reads pos[0] and ignores it), so that a thread's loads can be in flight together
__device__ __forceinline__ const uint64_t *d_match_pos_addr(const uint64_t *__restrict__ pos, uint32_t off_lo, uint32_t flags, uint32_t k)
{
	const uint64_t w = (uint64_t)off_lo | (uint64_t)(flags >> 16) << 32;
	return (flags >> 9 & 1u) ? pos : pos + (w + k);
}
__device__ __forceinline__ uint64_t d_match_pos_pick(uint64_t loaded, uint32_t off_lo, uint32_t flags)
{
	return (flags >> 9 & 1u) ? ((uint64_t)off_lo | (uint64_t)(flags >> 16) << 32) : loaded;
}

// collect_matches (map.c:90-123) of one fragment: its minimizers looked up with `max_occ`; mo == nullptr: counts only
__device__ __forceinline__ void d_seed_frag(const uint64_t *__restrict__ tab, int tab_bits, const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len,
                                            const uint64_t *__restrict__ mini_off, const AlAnchor *__restrict__ mini, const uint32_t *__restrict__ mini_cnt,
                                            const uint32_t f, const int max_occ, AlMatch *__restrict__ mo, uint32_t &n_m_out, uint32_t &n_a_out, int &rep_out, int &qlen_out,
                                            const int max_occ2 = 0, uint32_t *n_a2_out = nullptr /* anchors there would be with the larger bound max_occ2 (the re-seeding pass's) */)
{
	uint32_t n_a2 = 0;
	const uint32_t r0 = frag_first[f], r1 = frag_first[f + 1];
	int rep_st = 0, rep_en = 0, rep_len = 0; uint32_t n_m = 0, n_a = 0, sum = 0;
	uint64_t prev_hash = ~0ULL; int have_prev = 0;
	AlMatch *last = nullptr; uint64_t last_hash = 0; int last_valid = 0;
	const uint64_t tmask = (1ULL << tab_bits) - 1;
	for (uint32_t r = r0; r < r1; ++r) {
		const AlAnchor *mv = mini + mini_off[r];
		const uint32_t n = mini_cnt[r], seg = r - r0;
		// (round 6) four minimizers at a time: their records, then the first table slot of each, loaded without a branch around the loads (a lane past the end reads
		// the last minimizer again), so that four probes are in flight per lane instead of one; a probe that finds another key in its slot walks on as before.
		// The bookkeeping of collect_matches stays in the list's order.
		uint64_t xn[4], yn[4];                                                  // the next four's records: asked for beside this four's table slots
		if (n) {
#pragma unroll
			for (int j = 0; j < 4; ++j) { const uint32_t i = (uint32_t)j < n ? (uint32_t)j : n - 1u; const AlAnchor a = mv[i]; xn[j] = a.x; yn[j] = a.y; }
		}
		for (uint32_t i0 = 0; i0 < n; i0 += 4) {
			uint64_t xs[4], ys[4], ex[4], ey[4], sl[4];
#pragma unroll
			for (int j = 0; j < 4; ++j) { xs[j] = xn[j]; ys[j] = yn[j]; }
#pragma unroll
			for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(xs[j]), "+v"(ys[j]));
#pragma unroll
			for (int j = 0; j < 4; ++j) { sl[j] = ((xs[j] >> 8) * 0x9E3779B97F4A7C15ULL) >> (64 - tab_bits); const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(tab + 2 * sl[j]); ex[j] = e.x; ey[j] = e.y; }
#pragma unroll
			for (int j = 0; j < 4; ++j) { const uint32_t i = i0 + 4 + j < n ? i0 + 4 + j : n - 1u; const AlAnchor a = mv[i]; xn[j] = a.x; yn[j] = a.y; }
#pragma unroll
			for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(ex[j]), "+v"(ey[j]));
			// the four's matches stay in registers and are stored after the four (a store inside the walk would be waited for at the next branch that holds a load)
			AlMatch mk[4]; bool made[4]; uint32_t at[4];
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				made[j] = false; at[j] = 0; mk[j].off_lo = mk[j].n = mk[j].q_pos = mk[j].flags = 0;
				if (i0 + j < n) {
					const uint64_t x = xs[j], hash = x >> 8;
					const uint32_t q_pos = (uint32_t)ys[j] + (sum << 1), q_span = (uint32_t)(x & 0xff);
					uint64_t kx = ex[j], v = ey[j];
					if ((kx & ~AL_TAB_SINGLE) != hash + 1 && kx != 0) {          // another key in the slot: walk on (d_idx_get)
						uint64_t sp = sl[j];
						do { sp = (sp + 1) & tmask; const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(tab + 2 * sp); kx = e.x; v = e.y; } while ((kx & ~AL_TAB_SINGLE) != hash + 1 && kx != 0);
					}
					const bool found = kx != 0, single = found && (kx & AL_TAB_SINGLE) != 0;
					if (!found) v = 0;
					const uint32_t occ = single ? 1u : (uint32_t)v;
					// is_tandem (map.c:115-116): equal hash with the previous / next minimizer of the whole list
					const int same_prev = have_prev && prev_hash == hash;
					if (mo && same_prev && last_valid && last_hash == hash) {      // previous gets "next is same"
						if (j == 0) last->flags |= 1u << 8;                          // (the previous four's last match: stored already)
						else mk[j > 0 ? j - 1 : 0].flags |= 1u << 8;                 // (last_valid: the minimizer before this one made a match)
					}
					last_valid = 0;
					if (occ > 0 && (int)occ < max_occ2) n_a2 += occ;
					if ((int)occ >= max_occ) {                                       // map.c:105-111
						const int en = (int)(q_pos >> 1) + 1, st = en - (int)q_span;
						if (st > rep_en) { rep_len += rep_en - rep_st; rep_st = st, rep_en = en; }
						else rep_en = en;
					} else if (occ > 0) {
						if (mo) {
							mk[j].off_lo = single ? (uint32_t)v : (uint32_t)(v >> 32); mk[j].n = occ; mk[j].q_pos = q_pos;
							mk[j].flags = seg | (same_prev ? 1u << 8 : 0u) | (single ? (1u << 9 | (uint32_t)(v >> 32) << 16) : 0u);
							made[j] = true; at[j] = n_m; last = &mo[n_m]; last_hash = hash; last_valid = 1;
						}
						++n_m; n_a += occ;
					}
					prev_hash = hash; have_prev = 1;
				}
			}
#pragma unroll
			for (int j = 0; j < 4; ++j) if (made[j]) mo[at[j]] = mk[j];
		}
		sum += rd_len[r];
	}
	rep_len += rep_en - rep_st;
	n_m_out = n_m; n_a_out = n_a; rep_out = rep_len; qlen_out = (int)sum;
	if (n_a2_out) *n_a2_out = n_a2;
}

extern "C" __global__ void __launch_bounds__(256)
k_seed(const uint64_t *__restrict__ tab, int tab_bits,
       const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len,
       const uint64_t *__restrict__ mini_off, AlAnchor *__restrict__ mini, const uint32_t *__restrict__ mini_cnt,
       AlMatch *__restrict__ match, uint32_t *__restrict__ frag_nm, uint32_t *__restrict__ frag_na, int32_t *__restrict__ frag_rep,
       const uint32_t *__restrict__ frag_list, int n_list, int max_occ, int max_occ2, uint32_t *__restrict__ frag_na2 /* (first pass, may be null) anchors with the re-seeding pass's bound */)
{
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n_list) return;
	const uint32_t f = frag_list ? frag_list[t] : (uint32_t)t;
	uint32_t n_m, n_a, n_a2; int rep_len, qlen;
	d_seed_frag(tab, tab_bits, frag_first, rd_len, mini_off, mini, mini_cnt, f, max_occ, match + mini_off[frag_first[f]], n_m, n_a, rep_len, qlen, frag_na2 ? max_occ2 : 0, &n_a2);
	frag_nm[f] = n_m; frag_na[f] = n_a; frag_rep[f] = rep_len;
	if (frag_na2) frag_na2[f] = n_a2;
}

// ---- the equal-x merge of the few GIANT re-seeded fragments, started ahead of time (round 5) ------------------------------------------------
// A pair inside a high-copy family that gets re-seeded with max_occ (map.c:353-375) has 10^5 anchors, and if a query k-mer occurs twice its
// anchors need the serial heap emulation: 0.3-0.45 us per pop, > 100 ms for ONE fragment, which the re-chain pass used to wait for.  Whether a
// fragment is re-seeded is known only after its first chaining -- but what its anchors would be is known right after seeding.  So: every fragment
// with a repeat-masked minimizer (rep_len > 0: the necessary condition) whose max_occ anchors would number at least `thr` gets a slot here, its
// match lists are written to the slot, and the merge runs on a stream of its own beside the whole first pass.  If the fragment is re-seeded and
// flagged in the re-chain pass, its anchors are copied from the slot (k_spec_mark / k_spec_apply); if not, the work was wasted on a few fragments.
struct SpecOut { AlMatch *match; uint32_t *meta /* per slot: fragment, lists, anchors, fragment length */; uint32_t *cnt /* [0] slots taken, [1] candidates */; uint64_t *cand; uint32_t cap, per, cand_cap; };
// candidates: rep_len > 0, at least `thr` anchors with max_occ in at most `per` lists, and a k-mer that occurs twice among those lists (no equal x
// otherwise: the sort kernels' order is the heap's) -> key (anchors << 32 | fragment) appended to S.cand
extern "C" __global__ void __launch_bounds__(256)
k_spec_count(const uint64_t *__restrict__ tab, int tab_bits, const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len,
             const uint64_t *__restrict__ mini_off, const AlAnchor *__restrict__ mini, const uint32_t *__restrict__ mini_cnt,
             const int32_t *__restrict__ frag_rep, int n_frag, int max_occ, uint32_t thr, SpecOut S, int mid_occ, const uint32_t *__restrict__ frag_na2)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= n_frag || frag_rep[f] <= 0 || frag_na2[f] < thr) return;       // (k_seed counted the max_occ anchors on its way: the lookups below run for the few candidates only)
	const uint32_t r0 = frag_first[f], r1 = frag_first[f + 1];
	if (r1 - r0 > 2) return;
	uint32_t n_mid[2] = {0, 0};                                             // per mate: minimizers the FIRST pass keeps (a mate without any has no chain there: the usual reason for the re-seeding)
	const uint32_t c0 = mini_cnt[r0], c1 = r1 - r0 == 2 ? mini_cnt[r0 + 1] : 0u, M = c0 + c1;
	if (M > 256u) return;
	const AlAnchor *m0 = mini + mini_off[r0], *m1 = r1 - r0 == 2 ? mini + mini_off[r0 + 1] : m0;
	uint64_t mk[4] = {0, 0, 0, 0}; uint32_t n_m = 0, n_a = 0;
	for (uint32_t i = 0; i < M; ++i) {
		const uint64_t hash = (i < c0 ? m0[i].x : m1[i - c0].x) >> 8;
		bool single; const uint64_t v = d_idx_get(tab, tab_bits, hash, single);
		const uint32_t occ = single ? 1u : (uint32_t)v;
		if (occ > 0 && (int)occ < max_occ) { mk[i >> 6] |= 1ULL << (i & 63); ++n_m; n_a += occ; }
		if (occ > 0 && (int)occ < mid_occ) ++n_mid[i < c0 ? 0 : 1];
	}
	if (n_a < thr || n_m > S.per || n_m < 2) return;
	bool dup = false;
	for (uint32_t i = 1; i < M && !dup; ++i) {
		if (!(mk[i >> 6] >> (i & 63) & 1ULL)) continue;
		const uint64_t hi = (i < c0 ? m0[i].x : m1[i - c0].x) >> 8;
		for (uint32_t j = 0; j < i; ++j) if ((mk[j >> 6] >> (j & 63) & 1ULL) && ((j < c0 ? m0[j].x : m1[j - c0].x) >> 8) == hi) { dup = true; break; }
	}
	if (!dup) return;
	const uint32_t k = atomicAdd(&S.cnt[1], 1u);
	const bool likely = n_mid[0] < 2u || (r1 - r0 == 2 && n_mid[1] < 2u);
	if (k < S.cand_cap) S.cand[k] = (likely ? 1ULL << 63 : 0ULL) | (uint64_t)n_a << 32 | (uint32_t)f;   // (slots go to the likely ones first, by size)
}
// the S.cap largest candidates get the slots (rank among the keys: the same choice on every run), their lists are written by a lane each
extern "C" __global__ void __launch_bounds__(256)
k_spec_pick(const uint64_t *__restrict__ tab, int tab_bits, const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len,
            const uint64_t *__restrict__ mini_off, const AlAnchor *__restrict__ mini, const uint32_t *__restrict__ mini_cnt, int max_occ, SpecOut S)
{
	const uint32_t n = S.cnt[1] < S.cand_cap ? S.cnt[1] : S.cand_cap;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint64_t k = S.cand[i]; uint32_t rank = 0;
		for (uint32_t j = 0; j < n; ++j) rank += S.cand[j] > k ? 1u : 0u;
		if (rank < S.cap) {
			const uint32_t f = (uint32_t)k; uint32_t n_m, n_a; int rep, qlen;
			d_seed_frag(tab, tab_bits, frag_first, rd_len, mini_off, mini, mini_cnt, f, max_occ, S.match + (size_t)rank * S.per, n_m, n_a, rep, qlen);
			S.meta[4 * rank] = f; S.meta[4 * rank + 1] = n_m; S.meta[4 * rank + 2] = n_a; S.meta[4 * rank + 3] = (uint32_t)qlen;
		}
	}
	if (threadIdx.x == 0 && blockIdx.x == 0) S.cnt[0] = n < S.cap ? n : S.cap;
}
// the slots as a virtual batch for k_anchor_heap_lanes: slot i is "fragment" i with one "read" of the fragment's length
struct SpecView { uint32_t *first, *rdlen, *nm, *na, *tie, *list, *n_list; uint64_t *moff, *aoff; };
__global__ void k_spec_layout(const uint32_t *__restrict__ meta, uint32_t n, uint32_t per, SpecView V)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	uint64_t off = 0;
	for (uint32_t i = 0; i < n; ++i) {
		V.first[i] = i; V.rdlen[i] = meta[4 * i + 3]; V.nm[i] = meta[4 * i + 1]; V.na[i] = meta[4 * i + 2]; V.tie[i] = 1u; V.list[i] = i;
		V.moff[i] = (uint64_t)i * per; V.aoff[i] = off; off += meta[4 * i + 2];
	}
	V.first[n] = n; V.moff[n] = (uint64_t)n * per; V.aoff[n] = off; V.n_list[0] = n;
}
// re-chain pass, after its sorts: a slot whose fragment was re-seeded to the same lists and flagged for the merge is taken (flag 2: "merged already")
__global__ void k_spec_mark(const uint32_t *__restrict__ meta, uint32_t n, const uint32_t *__restrict__ frag_nm, const uint32_t *__restrict__ frag_na, uint32_t *__restrict__ tie_flag, uint32_t *__restrict__ use,
                            uint32_t *__restrict__ n_used)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t f = meta[4 * i];
	const bool ok = tie_flag[f] == 1u && frag_nm[f] == meta[4 * i + 1] && frag_na[f] == meta[4 * i + 2];
	if (ok) { tie_flag[f] = 2u; atomicAdd(n_used, 1u); }
	use[i] = ok ? 1u : 0u;
}
__global__ void __launch_bounds__(256)
k_spec_apply(const uint32_t *__restrict__ meta, const uint32_t *__restrict__ use, const uint64_t *__restrict__ v_aoff, const AlAnchor *__restrict__ src, const uint64_t *__restrict__ a_off, AlAnchor *__restrict__ anchors)
{
	const uint32_t i = blockIdx.y;
	if (!use[i]) return;
	const uint32_t f = meta[4 * i], n = meta[4 * i + 2];
	const uint4 *s4 = reinterpret_cast<const uint4 *>(src + v_aoff[i]); uint4 *d4 = reinterpret_cast<uint4 *>(anchors + a_off[f]);
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) d4[k] = s4[k];
}

// a8: ALSER candidate counter (map.c:299-312) on the sorted anchors of single-segment fragments
extern "C" __global__ void __launch_bounds__(256)
k_alser_count(const AlAnchor *__restrict__ a, const uint64_t *__restrict__ a_off, const uint32_t *__restrict__ frag_na,
              const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len, int n_frag, int min_cnt,
              unsigned long long *__restrict__ total)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= n_frag) return;
	const AlAnchor *p = a + a_off[f]; const uint32_t n = frag_na[f]; const int qlen0 = (int)rd_len[frag_first[f]];
	int seed_num = 0; unsigned cnt = 0;
	for (uint32_t i = 1; i < n; ++i) {
		if (((int32_t)p[i].x - (int32_t)p[i - 1].x) > qlen0) { if (seed_num >= min_cnt - 1) ++cnt; seed_num = 0; }
		else ++seed_num;
	}
	if (cnt) atomicAdd(total, (unsigned long long)cnt);
}

// =============================================================================================
// K3: anchor expansion + sort.  One wave per fragment.  Anchors are generated in parallel from the
// occurrence lists (coalesced 8-byte position reads), sorted by x with an LDS bitonic network, and
// written out.  Final order = forward-strand anchors by (rid,pos) then reverse-strand anchors: exactly
// the reference's k-way heap merge (x carries the strand in bit 63).  Equal x can only come from a query
// k-mer that occurs twice (overlapping mates, tandem repeats); the reference's order among those is
// heap-shape dependent, so such fragments -- and fragments too large for the LDS tile --
// are re-done by lane 0 with an exact emulation of the binary heap (ksort.h:43-59).
// =============================================================================================
template <int CAP>
__device__ void d_bitonic_sort_lds(AlAnchor *s, int npow2, int lane)
{   // ascending by x; npow2 <= CAP elements, 64 lanes
	for (int kk = 2; kk <= npow2; kk <<= 1)
		for (int j = kk >> 1; j > 0; j >>= 1) {
			for (int i = lane; i < npow2; i += 64) {
				const int ixj = i ^ j;
				if (ixj > i) {
					const AlAnchor a = s[i], b = s[ixj];
					const bool up = (i & kk) == 0;
					if ((a.x > b.x) == up) { s[i] = b; s[ixj] = a; }
				}
			}
			__syncthreads();
		}
}

// Heap entry of the k-way merge: head position x of one occurrence list, the list (match index) and the cursor in it.
struct HeapEnt { uint64_t x; uint32_t mi, off; };
struct HeapGlobal {            // heap in HBM scratch (one AlAnchor per entry: x, mi<<32|off)
	AlAnchor *h;
	__device__ __forceinline__ uint64_t x(size_t i) const { return h[i].x; }
	__device__ __forceinline__ HeapEnt get(size_t i) const { const AlAnchor e = h[i]; return HeapEnt{e.x, (uint32_t)(e.y >> 32), (uint32_t)e.y}; }
	__device__ __forceinline__ void set(size_t i, const HeapEnt &e) const { h[i].x = e.x; h[i].y = (uint64_t)e.mi << 32 | e.off; }
};
template <int LANES> struct HeapLds {   // heap in LDS, [entry][lane]; cursor and list packed in 32 bits (list < 2^16, cursor < 2^16)
	uint32_t *xlo, *xhi, *y;
	__device__ __forceinline__ uint64_t x(size_t i) const { return (uint64_t)xlo[i * LANES] | (uint64_t)xhi[i * LANES] << 32; }
	__device__ __forceinline__ HeapEnt get(size_t i) const { const uint32_t v = y[i * LANES]; return HeapEnt{x(i), v >> 16, v & 0xffffu}; }
	__device__ __forceinline__ void set(size_t i, const HeapEnt &e) const { xlo[i * LANES] = (uint32_t)e.x; xhi[i * LANES] = (uint32_t)(e.x >> 32); y[i * LANES] = e.mi << 16 | e.off; }
};

template <class H>
__device__ __forceinline__ void d_heapdown(size_t i, size_t n, const H &l)
{   // ksort.h:43-53 with heap_lt(a,b) = a.x > b.x (map.c:80)
	size_t k = i; const HeapEnt tmp = l.get(i);
	while ((k = (k << 1) + 1) < n) {
		if (k != n - 1 && l.x(k) > l.x(k + 1)) ++k;
		if (l.x(k) > tmp.x) break;
		l.set(i, l.get(k)); i = k;
	}
	l.set(i, tmp);
}

// exact emulation of collect_seed_hits_heap (map.c:149-213) by one lane
template <class H>
__device__ __forceinline__ void d_anchor_heap_merge(const uint64_t *__restrict__ pos, const AlMatch *__restrict__ m, uint32_t n_m, uint32_t n, int qlen, int mini_span,
                                                    const H &heap, AlAnchor *__restrict__ out, unsigned long long *__restrict__ counters)
{
	size_t hs = 0; uint64_t n_for = 0, n_rev = 0;
	atomicAdd(&counters[0], 1ULL);
	for (uint32_t i = 0; i < n_m; ++i) {
		heap.set(hs, HeapEnt{d_match_pos(pos, m[i].off_lo, m[i].flags, 0u), i, 0u}); ++hs;
	}
	if (hs > 1) for (size_t i = (hs >> 1) - 1; i != (size_t)-1; --i) d_heapdown(i, hs, heap);
	while (hs > 0) {
		HeapEnt top = heap.get(0);
		const AlMatch mm = m[top.mi];
		const uint64_t r = top.x; const int32_t rpos = (uint32_t)r >> 1; const uint32_t span = (uint32_t)mini_span;
		AlAnchor a;
		if ((r & 1) == (mm.q_pos & 1)) {
			a.x = (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
			a.y = (uint64_t)span << 32 | (mm.q_pos >> 1);
		} else {
			a.x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
			a.y = (uint64_t)span << 32 | (uint32_t)(qlen - ((int)(mm.q_pos >> 1) + 1 - (int)span) - 1);
		}
		a.y |= (uint64_t)(mm.flags & 0xff) << AL_SEED_SEG_SHIFT;
		if (mm.flags & (1u << 8)) a.y |= AL_SEED_TANDEM;
		if (!(a.x >> 63)) out[n_for++] = a; else out[n - (++n_rev)] = a;
		if (top.off < mm.n - 1) {
			++top.off;
			top.x = d_match_pos(pos, mm.off_lo, mm.flags, top.off);
			heap.set(0, top);
		} else { heap.set(0, heap.get(hs - 1)); --hs; }
		if (hs > 0) d_heapdown(0, hs, heap);
	}
	for (uint64_t j = 0; j < n_rev >> 1; ++j) {                          // map.c:202-207
		AlAnchor t = out[n - 1 - j]; out[n - 1 - j] = out[n - (n_rev - j)]; out[n - (n_rev - j)] = t;
	}
}

// Fragments whose anchors have equal x (a query k-mer occurring twice: overlapping mates, tandem repeats) or that do not fit
// the sort tiles are flagged by the sort kernels (a flag per fragment: no contended atomic) and merged here, one lane per fragment, with the reference's own binary
// heap (its pop order among equal heads is heap-shape dependent).  HCAP > 0: heap of <= HCAP lists in LDS,
// lanes whose fragment has n_m in (LO, HCAP]; HCAP == 0: heap in HBM scratch, n_m > LO.
template <int HCAP, int LANES>
__global__ void __launch_bounds__(64)
k_anchor_heap(const uint64_t *__restrict__ pos, const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len,
              const uint64_t *__restrict__ mini_off, const AlMatch *__restrict__ match,
              const uint32_t *__restrict__ frag_nm, const uint32_t *__restrict__ frag_na, const uint64_t *__restrict__ a_off,
              AlAnchor *__restrict__ anchors, AlAnchor *__restrict__ heap_ws,
              const uint32_t *__restrict__ tie_flag, const uint32_t *__restrict__ frag_list, int n_list, int lo_excl,
              unsigned long long *__restrict__ counters, int mini_span, const uint32_t *__restrict__ n_list_dev /* length of frag_list when it was compacted on the device (else null) */,
              uint32_t wave_na_min /* fragments of at least this many anchors (and <= 128 lists) are left to k_anchor_heap_wave */)
{
	__shared__ uint32_t s_h[HCAP > 0 ? 3 * HCAP * LANES : 1];
	const int lane = threadIdx.x;
	const uint32_t t = blockIdx.x * LANES + lane;
	if (n_list_dev) n_list = (int)*n_list_dev;
	if (lane >= LANES || t >= (uint32_t)n_list) return;
	const uint32_t f = frag_list ? frag_list[t] : t;
	if (!tie_flag[f]) return;
	const uint32_t n = frag_na[f], n_m = frag_nm[f];
	if (n >= wave_na_min && n_m <= 128u) return;                               // k_anchor_heap_wave takes it
	if ((int)n_m <= lo_excl || (HCAP > 0 && n_m > (uint32_t)HCAP) || n == 0) return;
	const uint32_t r0 = frag_first[f], r1 = frag_first[f + 1];
	int qlen = 0; for (uint32_t r = r0; r < r1; ++r) qlen += (int)rd_len[r];
	const AlMatch *m = match + mini_off[r0];
	AlAnchor *out = anchors + a_off[f];
	if (HCAP > 0) {
		bool fits = true;                                // cursor < 2^16 (occurrence counts are below max_occ) and list index < 2^16
		for (uint32_t i = 0; i < n_m; ++i) if (m[i].n > 0xffffu) fits = false;
		if (fits) { const HeapLds<LANES> h{s_h + lane, s_h + HCAP * LANES + lane, s_h + 2 * HCAP * LANES + lane}; d_anchor_heap_merge(pos, m, n_m, n, qlen, mini_span, h, out, counters); return; }
	}
	const HeapGlobal h{heap_ws + mini_off[r0]};
	d_anchor_heap_merge(pos, m, n_m, n, qlen, mini_span, h, out, counters);
}
template __global__ void k_anchor_heap<48, 64>(const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const AlMatch *, const uint32_t *, const uint32_t *, const uint64_t *, AlAnchor *, AlAnchor *, const uint32_t *, const uint32_t *, int, int, unsigned long long *, int, const uint32_t *, uint32_t);
template __global__ void k_anchor_heap<96, 32>(const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const AlMatch *, const uint32_t *, const uint32_t *, const uint64_t *, AlAnchor *, AlAnchor *, const uint32_t *, const uint32_t *, int, int, unsigned long long *, int, const uint32_t *, uint32_t);
template __global__ void k_anchor_heap<0, 64>(const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const AlMatch *, const uint32_t *, const uint32_t *, const uint64_t *, AlAnchor *, AlAnchor *, const uint32_t *, const uint32_t *, int, int, unsigned long long *, int, const uint32_t *, uint32_t);

// The same merge for the few flagged fragments with tens or hundreds of thousands of anchors (a pair inside a high-copy family, re-seeded
// with max_occ): one lane per fragment means one dependent HBM load per pop -- 2.5 us each, 0.8 s for 3 * 10^5 anchors, and the launch
// waits for it.  Here a wavefront takes one fragment: every occurrence list has a ring of RING positions in LDS that all lanes refill
// together (a lane per list, loads in flight at once), the binary heap sits in LDS as 16-byte entries, and lane 0 pops -- exactly as
// d_anchor_heap_merge -- until the list it is about to advance has nothing prefetched, which starts the next refill.  A pop then costs
// LDS round trips only: one per sift level (both children at once), the root stays in registers.
struct HeapW { uint64_t x; uint32_t mi, pad; };                               // heap entry of k_anchor_heap_wave: one 16-byte LDS word
struct ListW { uint32_t cur, have, n, qp; };                                   // per list: index of its head, elements fetched so far (the ring holds (cur, have)), length, query position word
template <int MCAPH, int RING>
__global__ void __launch_bounds__(64)
k_anchor_heap_wave(const uint64_t *__restrict__ pos, const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len,
                   const uint64_t *__restrict__ mini_off, const AlMatch *__restrict__ match,
                   const uint32_t *__restrict__ frag_nm, const uint32_t *__restrict__ frag_na, const uint64_t *__restrict__ a_off,
                   AlAnchor *__restrict__ anchors, const uint32_t *__restrict__ tie_flag, const uint32_t *__restrict__ frag_list,
                   const uint32_t *__restrict__ n_list_dev, uint32_t na_min, unsigned long long *__restrict__ counters, int mini_span)
{
	__shared__ __align__(16) HeapW h[MCAPH + 1];                               // (+1: the sift reads slot k + 1 unconditionally)
	__shared__ __align__(16) ListW ls[MCAPH];
	__shared__ uint64_t ring[MCAPH * RING];
	__shared__ uint32_t m_off[MCAPH], m_fl[MCAPH];
	__shared__ uint32_t s_hs, s_nfor, s_nrev;
	const int lane = threadIdx.x;
	const uint32_t n_list = *n_list_dev;
	for (uint32_t t = blockIdx.x; t < n_list; t += gridDim.x) {
		const uint32_t f = frag_list[t];
		const uint32_t n = frag_na[f], n_m = frag_nm[f];
		if (!tie_flag[f] || n < na_min || n_m > (uint32_t)MCAPH || n_m == 0) continue;   // the lane kernels take it
		const uint32_t r0 = frag_first[f], r1 = frag_first[f + 1];
		int qlen = 0; for (uint32_t r = r0; r < r1; ++r) qlen += (int)rd_len[r];
		const AlMatch *m = match + mini_off[r0];
		AlAnchor *out = anchors + a_off[f];
		__syncthreads();                                                       // (one wavefront: the LDS arrays of the previous fragment are done with)
		for (uint32_t i = lane; i < n_m; i += 64) {
			const AlMatch mm = m[i];
			m_off[i] = mm.off_lo; m_fl[i] = mm.flags;
			const uint32_t k = mm.n < (uint32_t)RING ? mm.n : (uint32_t)RING;    // element j of list i sits in ring[i * RING + j % RING]
			uint64_t v[RING];
#pragma unroll
			for (int j = 0; j < RING; ++j) v[j] = (uint32_t)j < k ? d_match_pos(pos, mm.off_lo, mm.flags, (uint32_t)j) : 0;
#pragma unroll
			for (int j = 0; j < RING; ++j) if ((uint32_t)j < k) ring[i * RING + j] = v[j];
			ls[i] = ListW{0u, k, mm.n, mm.q_pos};
			h[i] = HeapW{v[0], i, 0u};
		}
		if (lane == 0) h[n_m] = HeapW{UINT64_MAX, 0u, 0u};
		__syncthreads();
		if (lane == 0) {
			atomicAdd(&counters[0], 1ULL);
			// ks_heapmake (ksort.h:55-59) with heap_lt(a, b) = a.x > b.x (map.c:80)
			const uint32_t hs = n_m;
			if (hs > 1) for (uint32_t i0 = (hs >> 1) - 1; i0 != (uint32_t)-1; --i0) {
				uint32_t i = i0, k = i0; const HeapW tmp = h[i0];
				while ((k = (k << 1) + 1) < hs) {
					if (k != hs - 1 && h[k].x > h[k + 1].x) ++k;
					if (h[k].x > tmp.x) break;
					h[i] = h[k]; i = k;
				}
				h[i] = tmp;
			}
			s_hs = hs; s_nfor = 0; s_nrev = 0;
		}
		__syncthreads();
		for (;;) {
			if (lane == 0) {
				uint32_t hs = s_hs, n_for = s_nfor, n_rev = s_nrev;
				HeapW top = h[0];                                                  // kept in registers from one pop to the next
				while (hs > 0) {
					const uint32_t mi = top.mi; const ListW L = ls[mi];
					const uint32_t cur = L.cur + 1;
					if (cur < L.n && cur >= L.have) break;                          // its next position is not in LDS yet: refill first
					const uint64_t nx = ring[mi * RING + cur % RING];                // (read ahead of its use: one LDS round trip together with the match flags)
					const uint32_t fl = m_fl[mi], qp = L.qp;
					const uint64_t r = top.x;
					const int32_t rpos = (uint32_t)r >> 1; const uint32_t span = (uint32_t)mini_span;
					AlAnchor a;
					if ((r & 1) == (qp & 1)) { a.x = (r & 0xffffffff00000000ULL) | (uint32_t)rpos; a.y = (uint64_t)span << 32 | (qp >> 1); }
					else { a.x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (uint32_t)rpos; a.y = (uint64_t)span << 32 | (uint32_t)(qlen - ((int)(qp >> 1) + 1 - (int)span) - 1); }
					a.y |= (uint64_t)(fl & 0xff) << AL_SEED_SEG_SHIFT;
					if (fl & (1u << 8)) a.y |= AL_SEED_TANDEM;
					if (!(a.x >> 63)) out[n_for++] = a; else out[n - (++n_rev)] = a;
					HeapW tmp;
					if (cur < L.n) { tmp = HeapW{nx, mi, 0u}; ls[mi].cur = cur; }
					else { --hs; tmp = h[hs]; h[hs] = HeapW{UINT64_MAX, 0u, 0u}; }   // (the vacated slot reads as +inf for the unconditional k + 1 loads)
					if (hs == 0) break;
					// ks_heapdown (ksort.h:43-53) from the root; both children come with one round trip (16-byte entries), the comparison
					// that picks between them is the reference's (the right child only if strictly smaller and inside the heap)
					uint32_t i = 0, k = 0; bool first = true;
					while ((k = (k << 1) + 1) < hs) {
						HeapW c0, c1;                                                  // whole entries, two 16-byte reads in flight: one round trip per level
						{ const uint4 q0 = *reinterpret_cast<const uint4 *>(&h[k]), q1 = *reinterpret_cast<const uint4 *>(&h[k + 1]);
						  c0.x = (uint64_t)q0.x | (uint64_t)q0.y << 32; c0.mi = q0.z; c0.pad = 0; c1.x = (uint64_t)q1.x | (uint64_t)q1.y << 32; c1.mi = q1.z; c1.pad = 0; }
						const bool right = k != hs - 1 && c0.x > c1.x;
						const HeapW c = right ? c1 : c0; k += right ? 1u : 0u;
						if (c.x > tmp.x) break;
						h[i] = c; if (first) { top = c; first = false; }
						i = k;
					}
					h[i] = tmp; if (first) top = tmp;
				}
				s_hs = hs; s_nfor = n_for; s_nrev = n_rev;
			}
			__syncthreads();
			if (s_hs == 0) break;
			for (uint32_t i = lane; i < n_m; i += 64) {                          // refill: everything that fits behind each list's head
				const ListW L = ls[i]; const uint32_t off = m_off[i], fl = m_fl[i];
				const uint32_t room = (uint32_t)RING - (L.have - L.cur - 1), left = L.n - L.have, need = room < left ? room : left;
				uint64_t v[RING];                                                  // all loads of the lane in flight at once, then the LDS stores
#pragma unroll
				for (int j = 0; j < RING; ++j) v[j] = (uint32_t)j < need ? d_match_pos(pos, off, fl, L.have + (uint32_t)j) : 0;
#pragma unroll
				for (int j = 0; j < RING; ++j) if ((uint32_t)j < need) ring[i * RING + (L.have + (uint32_t)j) % RING] = v[j];
				ls[i].have = L.have + need;
			}
			__syncthreads();
		}
		__threadfence();                                                        // lane 0's stores before the other lanes read them back
		const uint32_t n_rev = s_nrev;                                          // map.c:202-207: the reverse-strand tail was written back to front
		for (uint32_t j = lane; j < n_rev >> 1; j += 64) { const AlAnchor tA = out[n - 1 - j]; out[n - 1 - j] = out[n - (n_rev - j)]; out[n - (n_rev - j)] = tA; }
	}
}
template __global__ void k_anchor_heap_wave<128, 32>(const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const AlMatch *, const uint32_t *, const uint32_t *, const uint64_t *, AlAnchor *, const uint32_t *, const uint32_t *, const uint32_t *, uint32_t, unsigned long long *, int);

// ---------------------------------------------------------------------------------------------
// The same merge with the binary heap IN THE LANES of a wavefront (round 5).  The serial forms above pay one LDS (or HBM) round trip
// per sift level and pop; here a pop costs a fixed number of wave-wide instructions whatever the depth:
//   * heap node v (1-based: children 2v, 2v + 1, siblings v ^ 1) lives in lane v & 63 of register set v >> 6 as (x, list); absent nodes
//     hold x = +inf.  Siblings are neighbouring lanes, so "which child does ks_heapdown prefer" (ksort.h:47: the right one only if
//     strictly smaller) is one DPP quad permute and two compares for ALL nodes at once -> a 64-bit mask PM of preferred children;
//   * the sift path of ks_heapdown from a node is the chain of preferred children below it, independent of the value being sifted:
//     node v is on it iff v and all its ancestors below the start are preferred -- (PM & ANC[v]) == ANC[v] with a per-lane constant;
//   * the heap order makes the x along that path non-decreasing, so the nodes that move up are exactly the path nodes with x <= tmp.x
//     (ksort.h:48 breaks on the first child that is greater): one compare + ballot -> mask LE; every node of LE (and the start) takes
//     the content of its child in LE (ds_bpermute), the last one takes tmp.
// ks_heapmake is the same step from the nodes n/2 .. 1.  Occurrence lists: list l belongs to the lane of node l + 1; its next
// position waits in a register of that lane, the RING positions behind it in that lane's LDS ring, refilled from HBM by all lanes
// together when the list about to advance has run dry.  Up to 63 (NSET 1) / 126 (NSET 2) lists; larger fragments keep the serial form.
// Checked against the serial emulation by a model of these steps (tests/test_heap_lanes_model.py) and by the tie-order goldens.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t d_sel_mask(uint64_t mask, uint32_t if_set, uint32_t if_clear)
{   // per lane: bit `lane` of a wave-uniform mask picks (one v_cndmask with the mask as its SGPR-pair condition)
	uint32_t r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask)); return r;
}
__device__ __forceinline__ uint64_t d_sel_mask64(uint64_t mask, uint64_t if_set, uint64_t if_clear)
{
	return (uint64_t)d_sel_mask(mask, (uint32_t)if_set, (uint32_t)if_clear) | (uint64_t)d_sel_mask(mask, (uint32_t)(if_set >> 32), (uint32_t)(if_clear >> 32)) << 32;
}
__device__ __forceinline__ uint64_t d_readlane64(uint64_t v, int l)
{
	return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32;
}
__device__ __forceinline__ uint64_t d_sibling64(uint64_t v)
{   // the value of lane ^ 1 (DPP quad_perm [1,0,3,2])
	return (uint64_t)(uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)v, 0xB1, 0xf, 0xf, true) | (uint64_t)(uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(v >> 32), 0xB1, 0xf, 0xf, true) << 32;
}

template <int NSET>
struct HeapLanes {
	uint64_t hx[NSET]; uint32_t hl[NSET];   // node content: head position, list
	uint64_t anc[NSET];                     // set-0 bits of the node's ancestors below the root (set 0: and of the node itself)
	// ks_heapdown (ksort.h:43-53) from node `start` (< 64) with the value (tx, tl); as0: bits of start and its ancestors, ins0 / ins1: its strict descendants
	__device__ __forceinline__ void sink(const int lane, const uint32_t start, const uint64_t as0, const uint64_t ins0, const uint64_t ins1, const uint64_t tx, const uint32_t tl)
	{
		uint64_t pm[NSET];
#pragma unroll
		for (int s = 0; s < NSET; ++s) {
			const uint64_t sib = d_sibling64(hx[s]);
			pm[s] = __ballot(hx[s] < sib) | (__ballot(hx[s] == sib) & 0x5555555555555555ULL);   // left child (even node) on a tie
		}
		const uint64_t pmi = pm[0] | as0;
		uint64_t le[NSET];
		le[0] = __ballot((pmi & anc[0]) == anc[0]) & __ballot(hx[0] <= tx) & ins0;
		if (NSET > 1) le[NSET - 1] = __ballot((pmi & anc[NSET - 1]) == anc[NSET - 1]) & __ballot(hx[NSET - 1] <= tx) & pm[NSET - 1] & ins1;
		// children of my set-0 node that move up (at most one): nodes 2 * lane, 2 * lane + 1
		uint32_t ch;
		if (NSET > 1) { const uint64_t m = lane < 32 ? le[0] : le[NSET - 1]; ch = (uint32_t)(m >> ((2 * lane) & 63)) & 3u; }
		else ch = lane < 32 ? (uint32_t)(le[0] >> (2 * lane)) & 3u : 0u;
		const int src = (((2 * lane) & 63) + (int)(ch >> 1)) << 2;
		// what a lane hands to its parent: its set-1 node when that one moves, else its set-0 node
		uint32_t dlo = (uint32_t)hx[0], dhi = (uint32_t)(hx[0] >> 32), dl = hl[0];
		if (NSET > 1) { dlo = d_sel_mask(le[NSET - 1], (uint32_t)hx[NSET - 1], dlo); dhi = d_sel_mask(le[NSET - 1], (uint32_t)(hx[NSET - 1] >> 32), dhi); dl = d_sel_mask(le[NSET - 1], hl[NSET - 1], dl); }
		uint32_t plo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)dlo), phi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)dhi), pl = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)dl);
		if (NSET > 1) {
			// Both nodes of a lane move when node j and node 64 + j lie on one path (j an ancestor of 64 + j: lanes 2, 4, 9, 21, 63): the parent of the
			// set-0 node then got the set-1 node above.  Rare, wave-uniform: a second round with the set-0 nodes for those parents.
			const uint64_t both = le[0] & le[NSET - 1];
			if (both != 0) {
				const uint32_t qlo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)hx[0]), qhi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(hx[0] >> 32)), ql = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)hl[0]);
				const bool fix = lane < 32 && ((both >> (2 * lane + (int)(ch >> 1))) & 1ULL) != 0;
				plo = fix ? qlo : plo; phi = fix ? qhi : phi; pl = fix ? ql : pl;
			}
		}
		const uint64_t in0 = le[0] | 1ULL << start;
		const bool pull = ch != 0;
		const uint32_t vlo = pull ? plo : (uint32_t)tx, vhi = pull ? phi : (uint32_t)(tx >> 32), vl = pull ? pl : tl;
		hx[0] = (uint64_t)d_sel_mask(in0, vlo, (uint32_t)hx[0]) | (uint64_t)d_sel_mask(in0, vhi, (uint32_t)(hx[0] >> 32)) << 32;
		hl[0] = d_sel_mask(in0, vl, hl[0]);
		if (NSET > 1) { hx[NSET - 1] = d_sel_mask64(le[NSET - 1], tx, hx[NSET - 1]); hl[NSET - 1] = d_sel_mask(le[NSET - 1], tl, hl[NSET - 1]); }   // (set-1 nodes are leaves: a moving one is where the value lands)
	}
};

template <int NSET, int RING>
__global__ void __launch_bounds__(64)
k_anchor_heap_lanes(const uint64_t *__restrict__ pos, const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len,
                    const uint64_t *__restrict__ mini_off, const AlMatch *__restrict__ match,
                    const uint32_t *__restrict__ frag_nm, const uint32_t *__restrict__ frag_na, const uint64_t *__restrict__ a_off,
                    AlAnchor *__restrict__ anchors, const uint32_t *__restrict__ tie_flag, const uint32_t *__restrict__ frag_list,
                    const uint32_t *__restrict__ n_list_dev, int lo_excl, unsigned long long *__restrict__ counters, int mini_span)
{
	static_assert(NSET == 1 || NSET == 2, "one or two register sets");
	static_assert((RING & (RING - 1)) == 0 && RING <= 32, "ring size");
	constexpr uint32_t NCAP = NSET == 1 ? 63u : 126u;
	__shared__ uint64_t ring[NSET * 64 * RING];                                // [node][RING]: only the node's own lane touches its ring
	const int lane = threadIdx.x;
	HeapLanes<NSET> H;
#pragma unroll
	for (int s = 0; s < NSET; ++s) {
		const uint32_t v = (uint32_t)(s * 64 + lane); uint64_t a = 0;
		for (uint32_t k = s == 0 ? v : v >> 1; k >= 2; k >>= 1) a |= 1ULL << k;
		H.anc[s] = v < 2 ? ~0ULL : a;                                          // (lanes 0 and 1 of set 0 are never on a path below the start)
	}
	const uint32_t n_list = *n_list_dev;
	for (uint32_t t = blockIdx.x; t < n_list; t += gridDim.x) {
		const uint32_t f = frag_list[t];
		const uint32_t n = frag_na[f], n_m = frag_nm[f];
		if (!tie_flag[f] || (int)n_m <= lo_excl || n_m > NCAP || n == 0) continue;
		const uint32_t r0 = frag_first[f], r1 = frag_first[f + 1];
		int qlen = 0; for (uint32_t r = r0; r < r1; ++r) qlen += (int)rd_len[r];
		const AlMatch *m = match + mini_off[r0];
		AlAnchor *out = anchors + a_off[f];
		// ---- the lists: node v = list v - 1; its first position is the node's content, the next up to RING go to the ring
		uint64_t nxt[NSET]; uint32_t l_st[NSET], l_ins[NSET], l_off[NSET], l_fl[NSET], l_info[NSET];   // l_st = (positions not yet in the heap) << 8 | (of those, in the ring)
#pragma unroll
		for (int s = 0; s < NSET; ++s) {
			const uint32_t v = (uint32_t)(s * 64 + lane);
			const bool valid = v >= 1 && v <= n_m;
			AlMatch mm{0u, 0u, 0u, 0u}; if (valid) mm = m[v - 1];
			l_off[s] = mm.off_lo; l_fl[s] = mm.flags; l_ins[s] = 1;
			l_info[s] = (mm.q_pos & 0xfffffu) | (mm.flags & 0xffu) << 20 | (mm.flags >> 8 & 1u) << 28;
			const uint32_t have = mm.n < (uint32_t)RING + 1u ? mm.n : (uint32_t)RING + 1u;
			uint64_t e[RING + 1];
#pragma unroll
			for (int j = 0; j <= RING; ++j) e[j] = (uint32_t)j < have ? d_match_pos(pos, mm.off_lo, mm.flags, (uint32_t)j) : UINT64_MAX;
#pragma unroll
			for (int j = 1; j <= RING; ++j) if ((uint32_t)j < have) ring[v * RING + (j & (RING - 1))] = e[j];
			H.hx[s] = e[0]; H.hl[s] = valid ? v - 1 : 0u;
			nxt[s] = e[1];
			l_st[s] = valid ? (mm.n - 1) << 8 | (have - 1) : 0u;
		}
		if (lane == 0) atomicAdd(&counters[0], 1ULL);
		// (every register the loops below carry comes from the loads above: used here once, so that the wait for those loads sits here and not
		//  inside the pop loop, where a vmcnt(0) would also wait for the previous pop's anchor store)
#pragma unroll
		for (int s = 0; s < NSET; ++s) asm volatile("" :: "v"(H.hx[s]), "v"(H.hl[s]), "v"(nxt[s]), "v"(l_st[s]), "v"(l_info[s]), "v"(l_off[s]), "v"(l_fl[s]));
		// ---- ks_heapmake (ksort.h:55-59)
		uint32_t hs = n_m;
		for (uint32_t node = hs >> 1; node >= 1; --node) {
			const int dn = 31 - __clz((int)node);
			uint64_t ins[NSET];
#pragma unroll
			for (int s = 0; s < NSET; ++s) {
				const uint32_t v = (uint32_t)(s * 64 + lane); const int d = (31 - __clz((int)(v | 1u))) - dn;
				ins[s] = __ballot(v >= 2 && d > 0 && (v >> d) == node);
			}
			const int dl0 = 31 - __clz(lane | 1);
			const uint64_t as0 = __ballot(lane >= 1 && dl0 <= dn && (node >> (dn - dl0)) == (uint32_t)lane);
			const uint64_t tx = d_readlane64(H.hx[0], (int)node); const uint32_t tl = (uint32_t)__builtin_amdgcn_readlane((int)H.hl[0], (int)node);
			H.sink(lane, node, as0, ins[0], ins[NSET - 1], tx, tl);
		}
		// ---- the merge (map.c:168-199).  A pop leaves (position word, list info) in lane `cnt` of three registers (v_writelane); every 64 pops the
		// lanes turn them into anchors together (map.c:176-187) and store them: forward strand from the front, reverse strand from the back.
		uint32_t n_for = 0, n_rev = 0, cnt = 0;
		uint32_t b_lo = 0, b_hi = 0, b_info = 0;
		auto flush = [&]() {
			const bool valid = (uint32_t)lane < cnt;
			const uint64_t rx = (uint64_t)b_lo | (uint64_t)b_hi << 32; const uint32_t info = b_info;
			const uint32_t qp = info & 0xfffffu, span = (uint32_t)mini_span;
			const bool fwd = (((uint32_t)rx ^ qp) & 1u) == 0;
			const uint64_t vm = __ballot(valid), fm = __ballot(fwd) & vm, rm = vm & ~fm, below = (1ULL << lane) - 1ULL;
			AlAnchor a;
			a.x = (rx & 0xffffffff00000000ULL) | ((uint32_t)rx >> 1) | (fwd ? 0ULL : 1ULL << 63);
			a.y = (uint64_t)span << 32 | (fwd ? qp >> 1 : (uint32_t)(qlen - ((int)(qp >> 1) + 1 - (int)span) - 1));
			a.y |= (uint64_t)(info >> 20 & 0xffu) << AL_SEED_SEG_SHIFT;
			if (info >> 28 & 1u) a.y |= AL_SEED_TANDEM;
			const uint32_t idx = fwd ? n_for + (uint32_t)__popcll(fm & below) : n - 1u - (n_rev + (uint32_t)__popcll(rm & below));
			if (valid) out[idx] = a;
			n_for += (uint32_t)__popcll(fm); n_rev += (uint32_t)__popcll(rm); cnt = 0;
		};
		while (hs > 0) {
			const uint32_t li = (uint32_t)__builtin_amdgcn_readlane((int)H.hl[0], 1);
			const uint64_t rx = d_readlane64(H.hx[0], 1);
			const uint32_t on = li + 1u; const int ol = (int)(on & 63u); const bool o1 = NSET > 1 && on >= 64u;
			uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)l_st[0], ol), info = (uint32_t)__builtin_amdgcn_readlane((int)l_info[0], ol);
			if (NSET > 1) { const uint32_t st1 = (uint32_t)__builtin_amdgcn_readlane((int)l_st[NSET - 1], ol), info1 = (uint32_t)__builtin_amdgcn_readlane((int)l_info[NSET - 1], ol); st = o1 ? st1 : st; info = o1 ? info1 : info; }
			if (__builtin_expect(st >= 0x100u && (st & 0xffu) == 0, 0)) {         // the list about to advance has nothing prefetched: every lane tops its ring(s) up
#pragma unroll
				for (int s = 0; s < NSET; ++s) {
					const uint32_t v = (uint32_t)(s * 64 + lane);
					const uint32_t av = l_st[s] & 0xffu, lf = l_st[s] >> 8, room = (uint32_t)RING - av, more = lf - av, need = room < more ? room : more;
					const uint32_t base = l_ins[s] + av;
					uint64_t e[RING];
#pragma unroll
					for (int j = 0; j < RING; ++j) e[j] = (uint32_t)j < need ? d_match_pos(pos, l_off[s], l_fl[s], base + (uint32_t)j) : 0;
#pragma unroll
					for (int j = 0; j < RING; ++j) if ((uint32_t)j < need) ring[v * RING + ((base + (uint32_t)j) & (RING - 1))] = e[j];
					l_st[s] += need;
					nxt[s] = ring[v * RING + (l_ins[s] & (RING - 1))];
				}
				st = (uint32_t)__builtin_amdgcn_readlane((int)l_st[0], ol);
				if (NSET > 1) { const uint32_t st1 = (uint32_t)__builtin_amdgcn_readlane((int)l_st[NSET - 1], ol); st = o1 ? st1 : st; }
			}
			const uint32_t left = st >> 8;
			// (the lane select through M0: one SGPR operand per VOP3 instruction on this target)
			asm("s_mov_b32 m0, %3\n\tv_writelane_b32 %0, %4, m0\n\tv_writelane_b32 %1, %5, m0\n\tv_writelane_b32 %2, %6, m0"
			    : "+v"(b_lo), "+v"(b_hi), "+v"(b_info) : "s"(cnt), "s"((uint32_t)rx), "s"((uint32_t)(rx >> 32)), "s"(info) : "m0");
			if (++cnt == 64u) flush();
			uint64_t tx; uint32_t tl;
			if (left > 0) {                                                     // the list's next position takes the root's place
				tx = d_readlane64(nxt[0], ol); tl = li;
				if (NSET > 1) { const uint64_t t1 = d_readlane64(nxt[NSET - 1], ol); tx = o1 ? t1 : tx; }
#pragma unroll
				for (int s = 0; s < NSET; ++s) {
					const bool own = lane == ol && (s == 1) == o1;
					l_ins[s] += own ? 1u : 0u; l_st[s] -= own ? 0x101u : 0u;
					nxt[s] = ring[(uint32_t)(s * 64 + lane) * RING + (l_ins[s] & (RING - 1))];
				}
			} else {                                                            // list exhausted: the last node moves to the root (map.c:193-196)
				const int hlane = (int)(hs & 63u); const bool h1 = NSET > 1 && hs >= 64u;
				tx = d_readlane64(H.hx[0], hlane); tl = (uint32_t)__builtin_amdgcn_readlane((int)H.hl[0], hlane);
				if (NSET > 1) { const uint64_t t1 = d_readlane64(H.hx[NSET - 1], hlane); const uint32_t l1 = (uint32_t)__builtin_amdgcn_readlane((int)H.hl[NSET - 1], hlane); tx = h1 ? t1 : tx; tl = h1 ? l1 : tl; }
#pragma unroll
				for (int s = 0; s < NSET; ++s) if (lane == hlane && (s == 1) == h1) H.hx[s] = UINT64_MAX;
				if (--hs == 0) break;
			}
			H.sink(lane, 1u, 2ULL, ~3ULL, ~0ULL, tx, tl);
		}
		if (cnt > 0) flush();
		__threadfence();                                                        // the stores above before other lanes read them back
		for (uint32_t j = lane; j < n_rev >> 1; j += 64) { const AlAnchor tA = out[n - 1 - j]; out[n - 1 - j] = out[n - (n_rev - j)]; out[n - (n_rev - j)] = tA; }   // map.c:202-207
	}
}
template __global__ void k_anchor_heap_lanes<1, 16>(const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const AlMatch *, const uint32_t *, const uint32_t *, const uint64_t *, AlAnchor *, const uint32_t *, const uint32_t *, const uint32_t *, int, unsigned long long *, int);
template __global__ void k_anchor_heap_lanes<2, 16>(const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const AlMatch *, const uint32_t *, const uint32_t *, const uint64_t *, AlAnchor *, const uint32_t *, const uint32_t *, const uint32_t *, int, unsigned long long *, int);

template <int CAP>
__global__ void __launch_bounds__(64)
k_anchor_sort(const uint64_t *__restrict__ pos, const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len,
              const uint64_t *__restrict__ mini_off, const AlMatch *__restrict__ match,
              const uint32_t *__restrict__ frag_nm, const uint32_t *__restrict__ frag_na, const uint64_t *__restrict__ a_off,
              AlAnchor *__restrict__ anchors, uint32_t *__restrict__ tie_list, unsigned int *__restrict__ tie_cnt,
              const uint32_t *__restrict__ frag_list, int n_list, unsigned long long *__restrict__ counters, int mini_span)
{
	__shared__ AlAnchor s[CAP];
	__shared__ uint32_t pre[CAP > 512 ? 513 : CAP + 1];
	__shared__ int s_flag;
	const int lane = threadIdx.x;
	if ((int)blockIdx.x >= n_list) return;
	const uint32_t f = frag_list ? frag_list[blockIdx.x] : blockIdx.x;
	const uint32_t n = frag_na[f], n_m = frag_nm[f];
	if (n == 0) return;
	const uint32_t r0 = frag_first[f], r1 = frag_first[f + 1];
	int qlen = 0; for (uint32_t r = r0; r < r1; ++r) qlen += (int)rd_len[r];
	const AlMatch *m = match + mini_off[r0];
	AlAnchor *out = anchors + a_off[f];
	bool fallback = (n > (uint32_t)CAP) || (n_m > 512);
	if (!fallback) {
		// prefix sums of occurrence counts (n_m <= 512): serial per 64-chunk scan
		if (lane == 0) s_flag = 0;
		uint32_t run = 0;
		for (uint32_t base = 0; base < n_m; base += 64) {
			const uint32_t i = base + lane;
			uint32_t v = i < n_m ? m[i].n : 0, incl = v;
			for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
			if (i < n_m) pre[i] = run + incl - v;
			run += __shfl(incl, 63);
		}
		if (lane == 0) pre[n_m] = run;
		__syncthreads();
		int npow2 = 1; while ((uint32_t)npow2 < n) npow2 <<= 1;
		for (uint32_t t = lane; t < (uint32_t)npow2; t += 64) {
			AlAnchor a; a.x = UINT64_MAX; a.y = UINT64_MAX;
			if (t < n) {
				uint32_t lo = 0, hi = n_m;                                   // last mi with pre[mi] <= t
				while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (pre[mid] <= t) lo = mid; else hi = mid; }
				const AlMatch mm = m[lo];
				const uint64_t r = d_match_pos(pos, mm.off_lo, mm.flags, t - pre[lo]);
				const uint32_t span = (uint32_t)mini_span, seg = mm.flags & 0xff;
				const int32_t rpos = (uint32_t)r >> 1;
				if ((r & 1) == (mm.q_pos & 1)) {
					a.x = (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
					a.y = (uint64_t)span << 32 | (mm.q_pos >> 1);
				} else {
					a.x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
					a.y = (uint64_t)span << 32 | (uint32_t)(qlen - ((int)(mm.q_pos >> 1) + 1 - (int)span) - 1);
				}
				a.y |= (uint64_t)seg << AL_SEED_SEG_SHIFT;
				if (mm.flags & (1u << 8)) a.y |= AL_SEED_TANDEM;
			}
			s[t] = a;
		}
		__syncthreads();
		d_bitonic_sort_lds<CAP>(s, npow2, lane);
		int tie = 0;
		for (uint32_t t = lane; t < n; t += 64) { if (t + 1 < n && s[t].x == s[t + 1].x) tie = 1; }
		if (tie) s_flag = 1;
		__syncthreads();
		fallback = s_flag != 0;
		if (!fallback) for (uint32_t t = lane; t < n; t += 64) out[t] = s[t];
	}
	if (fallback && lane == 0) tie_list[f] = 1u;                           // merged by k_anchor_heap
}

// ---------------------------------------------------------------------------------------------
// K3, register network, for fragments of 65 .. 8192 anchors (reads inside interspersed repeats).  An anchor is ONE 64-bit key
// (strand | contig | position) << 16 | list  (needs 33 + contig bits + 16 <= 64: the caller sends other indexes to k_anchor_sort /
// the device-wide sort); y is rebuilt from the list's match record when the anchors are written out; equal x -> exact heap merge,
// as in the other sort kernels.  The bitonic network runs on PER keys per thread held in registers.  Element e = thread * PER + r: compare-exchange distances below PER are register-to-register, distances inside a
// wavefront are cross-lane moves (DPP quad permutes, ds_swizzle, ds_bpermute: no LDS storage, no barrier), and only the top
// log2(NW) bits of the index go through an LDS exchange (three rounds for 4096 keys on 4 waves, against 78 LDS passes with a
// barrier each in the plain network).  All comparators ascend: the first step of a merge level pairs e with e ^ (kk - 1).
// ---------------------------------------------------------------------------------------------
#include "al_dev_net.h"
// One block of NT = 64 NW threads per fragment of at most PER * NT anchors and MCAP occurrence lists.
template <int PER, int NW, int MCAP>
__global__ void __launch_bounds__(64 * NW)
k_anchor_sort_reg(const uint64_t *__restrict__ pos, const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len,
                  const uint64_t *__restrict__ mini_off, const AlMatch *__restrict__ match,
                  const uint32_t *__restrict__ frag_nm, const uint32_t *__restrict__ frag_na, const uint64_t *__restrict__ a_off,
                  AlAnchor *__restrict__ anchors, uint32_t *__restrict__ tie_list,
                  const uint32_t *__restrict__ frag_list, int n_list, int mini_span, int rid_bits)
{
	constexpr int NT = 64 * NW, CAP = PER * NT;
	constexpr int PADW = CAP + CAP / 16 + 2;                // output transpose: one pad word per 16 keys
	__shared__ uint64_t sx[PADW];
	__shared__ uint32_t pre[MCAP + 1];
	// (the match records stay in global memory, read through L1 / L2: a copy in LDS cost the block kernels a block per CU -- block sorts 19.4 -> 17.4 ms without it)
	__shared__ uint32_t s_part[NT];
	__shared__ int s_flag;
	const int tid = threadIdx.x;
	if ((int)blockIdx.x >= n_list) return;
	const uint32_t f = frag_list[blockIdx.x];
	const uint32_t n = frag_na[f], n_m = frag_nm[f];
	if (n == 0) return;
	if (n > (uint32_t)CAP || n_m > (uint32_t)MCAP) { if (tid == 0) tie_list[f] = 1u; return; }   // not this kernel's class: exact merge
	const uint32_t r0 = frag_first[f], r1 = frag_first[f + 1];
	int qlen = 0; for (uint32_t r = r0; r < r1; ++r) qlen += (int)rd_len[r];
	const AlMatch *m = match + mini_off[r0];
	AlAnchor *out = anchors + a_off[f];
	if (tid == 0) s_flag = 0;
	{   // exclusive prefix sums of the list lengths: LPT lists per thread, block scan of the partial sums
		constexpr int LPT = (MCAP + NT - 1) / NT;
		uint32_t v[LPT], sum = 0;
#pragma unroll
		for (int j = 0; j < LPT; ++j) {
			const uint32_t i = (uint32_t)tid * LPT + j; v[j] = 0;
			if (i < n_m) v[j] = m[i].n;
			sum += v[j];
		}
		uint32_t incl = sum;                                              // wave scan, then the wave totals through LDS
		for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if ((tid & 63) >= d) incl += t; }
		if (NW > 1) {
			if ((tid & 63) == 63) s_part[tid >> 6] = incl;
			__syncthreads();
			for (int w = 0; w < (tid >> 6); ++w) incl += s_part[w];
		}
		uint32_t run = incl - sum;
#pragma unroll
		for (int j = 0; j < LPT; ++j) { const uint32_t i = (uint32_t)tid * LPT + j; if (i <= n_m) pre[i] = run; run += v[j]; }
	}
	__syncthreads();
	const int sb = 32 + rid_bits;                                           // strand bit of the compact key
	uint64_t k[PER];
	{   // expansion, element t = tid + j NT (coalesced position loads); the network does not care where a key starts
		// (round 6) per eight elements, three rounds without a branch around a load -- every element's list by a search of the prefix table with a fixed trip
		// count (a thread without an element searches for the last one), then the eight match records, then the eight positions: the loads of a round are in flight
		// together; with `if (t < n) { search; record; position }` per element the compiler waited at every join, and a thread's PER x 2 dependent round trips
		// were most of these kernels.  Keys are built per eight so that only they stay live.
#define TT(j) (((uint32_t)tid + (uint32_t)(j) * (uint32_t)NT) < n ? ((uint32_t)tid + (uint32_t)(j) * (uint32_t)NT) : n - 1u)
#define LIVE(j) ((uint32_t)tid + (uint32_t)(j) * (uint32_t)NT < n)
		int steps = 0; while ((1u << steps) < n_m) ++steps;
		constexpr int CH = PER < 8 ? PER : 8;
#pragma unroll
		for (int j0 = 0; j0 < PER; j0 += CH) {
			uint32_t mi[CH], fl[CH], ol[CH], qb[CH]; uint64_t rr[CH];
			if ((uint32_t)j0 * (uint32_t)NT >= n) {                             // (block-uniform) nobody has an element in this eight
#pragma unroll
				for (int j = 0; j < CH; ++j) k[j0 + j] = UINT64_MAX;
				continue;
			}
#pragma unroll
			for (int j = 0; j < CH; ++j) mi[j] = 0;
			for (int b = steps - 1; b >= 0; --b) {                             // mi = last list with pre[list] <= t, bit by bit (pre[0] = 0 <= t)
#pragma unroll
				for (int j = 0; j < CH; ++j) { const uint32_t c_ = mi[j] | (1u << b); if (c_ < n_m && pre[c_] <= TT(j0 + j)) mi[j] = c_; }
			}
#pragma unroll
			for (int j = 0; j < CH; ++j) { const AlMatch mm = m[mi[j]]; fl[j] = mm.flags; ol[j] = mm.off_lo; qb[j] = mm.q_pos; }
#pragma unroll
			for (int j = 0; j < CH; ++j) asm volatile("" : "+v"(fl[j]), "+v"(ol[j]), "+v"(qb[j]));
#pragma unroll
			for (int j = 0; j < CH; ++j) rr[j] = *d_match_pos_addr(pos, ol[j], fl[j], TT(j0 + j) - pre[mi[j]]);
#pragma unroll
			for (int j = 0; j < CH; ++j) asm volatile("" : "+v"(rr[j]));
#pragma unroll
			for (int j = 0; j < CH; ++j) {
				const uint64_t r = d_match_pos_pick(rr[j], ol[j], fl[j]); const bool rev = (r & 1) != (qb[j] & 1);      // map.c:176-190
				const uint64_t key = ((uint64_t)(rev ? 1 : 0) << sb | (r >> 32) << 32 | (uint32_t)((uint32_t)r >> 1)) << 16 | mi[j];
				k[j0 + j] = LIVE(j0 + j) ? key : UINT64_MAX;
			}
#pragma unroll
			for (int j = 0; j < CH; ++j) asm volatile("" : "+v"(k[j0 + j]));
		}
#undef TT
#undef LIVE
	}
	d_bt_levels<PER, NT, 2>(k, sx, tid);
	__syncthreads();
#pragma unroll
	for (int r = 0; r < PER; ++r) { const int e = tid * PER + r; sx[e + (e >> 4)] = k[r]; }
	__syncthreads();
	int tie = 0;
	for (uint32_t t = tid; t + 1 < n; t += NT) if ((sx[t + (t >> 4)] >> 16) == (sx[t + 1 + ((t + 1) >> 4)] >> 16)) tie = 1;
	if (tie) s_flag = 1;
	__syncthreads();
	if (s_flag) { if (tid == 0) tie_list[f] = 1u; return; }                // merged by k_anchor_heap
	const uint64_t lowmask = (1ULL << sb) - 1;
	constexpr int CHO = PER < 8 ? PER : 8;                                  // the records of eight anchors at a time, loads without a branch around them
#pragma unroll
	for (int j0 = 0; j0 < PER; j0 += CHO) {
		if ((uint32_t)j0 * (uint32_t)NT >= n) break;                         // (block-uniform)
		uint64_t kx[CHO]; uint32_t qp[CHO], fl[CHO];
#pragma unroll
		for (int j = 0; j < CHO; ++j) {
			const uint32_t t = (uint32_t)tid + (uint32_t)(j0 + j) * (uint32_t)NT, tc = t < n ? t : n - 1u;
			const uint64_t key = sx[tc + (tc >> 4)]; kx[j] = key >> 16;
			const AlMatch mm = m[(uint32_t)key & 0xffffu]; qp[j] = mm.q_pos; fl[j] = mm.flags;
		}
#pragma unroll
		for (int j = 0; j < CHO; ++j) asm volatile("" : "+v"(qp[j]), "+v"(fl[j]));
#pragma unroll
		for (int j = 0; j < CHO; ++j) {
			const uint32_t t = (uint32_t)tid + (uint32_t)(j0 + j) * (uint32_t)NT; const uint32_t span = (uint32_t)mini_span;
			AlAnchor a; a.x = (kx[j] & lowmask) | (kx[j] >> sb & 1) << 63;
			a.y = (a.x >> 63) ? (uint64_t)span << 32 | (uint32_t)(qlen - ((int)(qp[j] >> 1) + 1 - (int)span) - 1) : (uint64_t)span << 32 | (qp[j] >> 1);
			a.y |= (uint64_t)(fl[j] & 0xff) << AL_SEED_SEG_SHIFT;
			if (fl[j] & (1u << 8)) a.y |= AL_SEED_TANDEM;
			if (t < n) out[t] = a;
		}
	}
}
#define INST_SORT_REG(P, W, M) template __global__ void k_anchor_sort_reg<P, W, M>(const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const AlMatch *, const uint32_t *, const uint32_t *, const uint64_t *, AlAnchor *, uint32_t *, const uint32_t *, int, int, int);
INST_SORT_REG(2, 1, 128) INST_SORT_REG(4, 1, 256) INST_SORT_REG(8, 1, 512) INST_SORT_REG(16, 1, 512) INST_SORT_REG(8, 4, 1024) INST_SORT_REG(16, 4, 1024) INST_SORT_REG(16, 8, 1024)

// K3 for fragments above the LDS tiles (reads inside high-copy families, the max_occ re-chain pass: up to 42 x 5000 anchors):
// their anchors are expanded unsorted as ONE 64-bit key each,  ((rank of the fragment in the chunk) | strand | contig | position) << 16 |
// list  with contig and position at the widths this index needs, a device-wide keys-only radix sort (rocPRIM) over the bits above
// the list puts every fragment's anchors in x order at consecutive places, and k_anchor_big_scatter rebuilds (x, y) from the key and
// the list's match record into the fragment's anchor range (8 bytes per anchor and pass instead of 16).  One wavefront per 64 anchors of a fragment.
__global__ void __launch_bounds__(256)
k_anchor_big_expand(const uint64_t *__restrict__ pos, const uint64_t *__restrict__ mini_off, const uint32_t *__restrict__ frag_first, const AlMatch *__restrict__ match,
                    const uint32_t *__restrict__ frag_nm, const uint32_t *__restrict__ frag_list, int n_list, const uint64_t *__restrict__ big_off /* n_list + 1 */,
                    uint64_t *__restrict__ keys, int rid_bits, int pos_bits)
{
	// one block per fragment; lists are walked one after the other, 256 positions at a time (coalesced 8-byte reads)
	if ((int)blockIdx.x >= n_list) return;
	const uint32_t f = frag_list[blockIdx.x];
	const uint32_t n_m = frag_nm[f];
	const AlMatch *m = match + mini_off[frag_first[f]];
	const int sb = rid_bits + pos_bits;                                    // strand bit of the x part
	const uint64_t rank = (uint64_t)blockIdx.x << (sb + 1);
	uint64_t o = big_off[blockIdx.x];
	for (uint32_t i = 0; i < n_m; ++i) {
		const AlMatch mm = m[i];
		for (uint32_t t = threadIdx.x; t < mm.n; t += 256) {
			const uint64_t r = d_match_pos(pos, mm.off_lo, mm.flags, t);
			const bool rev = (r & 1) != (mm.q_pos & 1);
			keys[o + t] = (rank | (uint64_t)(rev ? 1 : 0) << sb | (r >> 32) << pos_bits | (uint32_t)((uint32_t)r >> 1)) << 16 | (i & 0xffffu);
		}
		o += mm.n;
	}
}
__global__ void __launch_bounds__(256)
k_anchor_big_scatter(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ frag_list, int n_list,
                     const uint64_t *__restrict__ big_off, const uint64_t *__restrict__ a_off,
                     const uint64_t *__restrict__ mini_off, const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len, const AlMatch *__restrict__ match,
                     const uint32_t *__restrict__ frag_nm, AlAnchor *__restrict__ anchors, uint32_t *__restrict__ tie_list, int rid_bits, int pos_bits, int mini_span)
{
	if ((int)blockIdx.x >= n_list) return;
	const uint32_t f = frag_list[blockIdx.x];
	const uint64_t b = big_off[blockIdx.x], n = big_off[blockIdx.x + 1] - b;
	const uint32_t r0 = frag_first[f], r1 = frag_first[f + 1];
	int qlen = 0; for (uint32_t r = r0; r < r1; ++r) qlen += (int)rd_len[r];
	const AlMatch *m = match + mini_off[r0];
	AlAnchor *out = anchors + a_off[f];
	const int sb = rid_bits + pos_bits;
	const uint64_t pmask = (1ULL << pos_bits) - 1, rmask = (1ULL << rid_bits) - 1;
	const uint32_t span = (uint32_t)mini_span;
	int tie = frag_nm[f] > 0x10000u ? 1 : 0;                               // list index does not fit the key: exact merge instead
	for (uint64_t t = threadIdx.x; t < n; t += 256) {
		const uint64_t k = keys[b + t], kx = k >> 16;
		const AlMatch mm = m[(uint32_t)k & 0xffffu];
		AlAnchor a; a.x = (kx >> sb & 1) << 63 | (kx >> pos_bits & rmask) << 32 | (kx & pmask);
		a.y = (a.x >> 63) ? (uint64_t)span << 32 | (uint32_t)(qlen - ((int)(mm.q_pos >> 1) + 1 - (int)span) - 1) : (uint64_t)span << 32 | (mm.q_pos >> 1);
		a.y |= (uint64_t)(mm.flags & 0xff) << AL_SEED_SEG_SHIFT;
		if (mm.flags & (1u << 8)) a.y |= AL_SEED_TANDEM;
		if (t + 1 < n && (keys[b + t + 1] >> 16) == kx) tie = 1;
		out[t] = a;
	}
	if (tie) tie_list[f] = 1u;                                             // equal x: merged again by k_anchor_heap (exact heap order)
}

// (round 5) The same fragments WITHOUT the device-wide sort: runs of AL_RUN_CAP = 8192 anchors are sorted by the register network above (a block per run, keys
// only, into the key buffer), then merged pairwise -- run length 8192, 16384, ... -- by k_anchor_run_merge: a block per AL_MERGE_TILE = 2048 output keys finds
// the tile's two input ranges by binary search (merge path), stages them in LDS, every thread merges eight outputs from its own diagonal.  A fragment of n
// anchors takes ceil(log2(ceil(n / 8192))) passes -- one for <= 16384 anchors, three for <= 65536 -- of 8 bytes in and out per anchor, instead of the
// expansion, seven radix passes over rank | x and the scatter; its last pass writes the anchors (x, y rebuilt from the list's match record) and raises the
// tie flag for equal x.  The keys carry the list in the low 16 bits, so no two are equal and the merge needs no stability rule.
// The entry of a block: largest i with off[i] <= block (off = running sum of the tiles per fragment, n + 1 entries).
#define AL_RUN_CAP 8192
#define AL_MERGE_TILE 2048
__device__ __forceinline__ uint32_t d_entry_of(const uint64_t *__restrict__ off, int n, uint64_t b)
{
	uint32_t lo = 0, hi = (uint32_t)n;
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= b) lo = mid; else hi = mid; }
	return lo;
}
__global__ void k_big_tiles(const uint32_t *__restrict__ na, int n, uint32_t *__restrict__ nt, unsigned int *__restrict__ max_na, const uint32_t tile)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i > n) return;
	const uint32_t v = i < n ? na[i] : 0u;
	nt[i] = (v + tile - 1) / tile;                                            // (nt[n] = 0: the scan's pad)
	if (i < n) atomicMax(max_na, v);
}
__global__ void k_big_tile_ent(const uint64_t *__restrict__ tile_off, int n_list, uint32_t n_tile, uint32_t *__restrict__ tile_ent)
{   // the entry of every tile, once: the blocks of the sort and of every merge pass start from it instead of a twelve-step search each
	const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b < n_tile) tile_ent[b] = d_entry_of(tile_off, n_list, b);
}
template <int PER, int NW, int MCAP>
__global__ void __launch_bounds__(64 * NW)
k_anchor_run_sort(const uint64_t *__restrict__ pos, const uint32_t *__restrict__ frag_first, const uint64_t *__restrict__ mini_off, const AlMatch *__restrict__ match,
                  const uint32_t *__restrict__ frag_nm, const uint32_t *__restrict__ frag_na, const uint32_t *__restrict__ frag_list, int n_list,
                  const uint64_t *__restrict__ tile_off /* n_list + 1 */, const uint32_t *__restrict__ tile_ent, const uint64_t *__restrict__ big_off /* n_list + 1: the entry's keys */,
                  uint64_t *__restrict__ keys, uint32_t *__restrict__ tie_list, int rid_bits, const uint32_t run_len /* <= AL_RUN_CAP (tests: shorter) */, const uint32_t tile)
{
	constexpr int NT = 64 * NW, CAP = PER * NT;
	static_assert(CAP == AL_RUN_CAP, "run length");
	constexpr int PADW = CAP + CAP / 16 + 2;
	__shared__ uint64_t sx[PADW];
	__shared__ uint32_t pre[MCAP + 1];
	__shared__ uint32_t s_part[NT];
	const int tid = threadIdx.x;
	const uint64_t blk = blockIdx.x;                                           // the grid counts merge tiles: the block of a run's first tile sorts the run, the other three leave
	const uint32_t ent = tile_ent[blk];
	const uint32_t tl = (uint32_t)(blk - tile_off[ent]);
	if (tl % (run_len / tile)) return;
	const uint32_t f = frag_list[ent];
	const uint32_t n = frag_na[f], n_m = frag_nm[f];
	if (n_m > (uint32_t)MCAP) { if (tid == 0 && tl == 0) tie_list[f] = 1u; return; }   // more lists than the prefix table: exact merge
	const uint32_t base = tl * tile, cnt = n - base < run_len ? n - base : run_len;
	const AlMatch *m = match + mini_off[frag_first[f]];
	{
		constexpr int LPT = (MCAP + NT - 1) / NT;
		uint32_t v[LPT], sum = 0;
#pragma unroll
		for (int j = 0; j < LPT; ++j) { const uint32_t i = (uint32_t)tid * LPT + j; v[j] = i < n_m ? m[i].n : 0u; sum += v[j]; }
		uint32_t incl = sum;
		for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if ((tid & 63) >= d) incl += t; }
		if ((tid & 63) == 63) s_part[tid >> 6] = incl;
		__syncthreads();
		for (int w = 0; w < (tid >> 6); ++w) incl += s_part[w];
		uint32_t run = incl - sum;
#pragma unroll
		for (int j = 0; j < LPT; ++j) { const uint32_t i = (uint32_t)tid * LPT + j; if (i <= n_m) pre[i] = run; run += v[j]; }
	}
	__syncthreads();
	const int sb = 32 + rid_bits;
	uint64_t k[PER];
	{
		// as in k_anchor_sort_reg: per eight elements the searches with a fixed trip count, then the records, then the positions, no branch around a load
#define TT(j) (base + (((uint32_t)tid + (uint32_t)(j) * (uint32_t)NT) < cnt ? ((uint32_t)tid + (uint32_t)(j) * (uint32_t)NT) : cnt - 1u))
#define LIVE(j) ((uint32_t)tid + (uint32_t)(j) * (uint32_t)NT < cnt)
		int steps = 0; while ((1u << steps) < n_m) ++steps;
		constexpr int CH = PER < 8 ? PER : 8;
#pragma unroll
		for (int j0 = 0; j0 < PER; j0 += CH) {
			uint32_t mi[CH], fl[CH], ol[CH], qb[CH]; uint64_t rr[CH];
			if ((uint32_t)j0 * (uint32_t)NT >= cnt) {
#pragma unroll
				for (int j = 0; j < CH; ++j) k[j0 + j] = UINT64_MAX;
				continue;
			}
#pragma unroll
			for (int j = 0; j < CH; ++j) mi[j] = 0;
			for (int b = steps - 1; b >= 0; --b) {
#pragma unroll
				for (int j = 0; j < CH; ++j) { const uint32_t c_ = mi[j] | (1u << b); if (c_ < n_m && pre[c_] <= TT(j0 + j)) mi[j] = c_; }
			}
#pragma unroll
			for (int j = 0; j < CH; ++j) { const AlMatch mm = m[mi[j]]; fl[j] = mm.flags; ol[j] = mm.off_lo; qb[j] = mm.q_pos; }
#pragma unroll
			for (int j = 0; j < CH; ++j) asm volatile("" : "+v"(fl[j]), "+v"(ol[j]), "+v"(qb[j]));
#pragma unroll
			for (int j = 0; j < CH; ++j) rr[j] = *d_match_pos_addr(pos, ol[j], fl[j], TT(j0 + j) - pre[mi[j]]);
#pragma unroll
			for (int j = 0; j < CH; ++j) asm volatile("" : "+v"(rr[j]));
#pragma unroll
			for (int j = 0; j < CH; ++j) {
				const uint64_t r = d_match_pos_pick(rr[j], ol[j], fl[j]); const bool rev = (r & 1) != (qb[j] & 1);      // map.c:176-190
				const uint64_t key = ((uint64_t)(rev ? 1 : 0) << sb | (r >> 32) << 32 | (uint32_t)((uint32_t)r >> 1)) << 16 | mi[j];
				k[j0 + j] = LIVE(j0 + j) ? key : UINT64_MAX;
			}
#pragma unroll
			for (int j = 0; j < CH; ++j) asm volatile("" : "+v"(k[j0 + j]));
		}
#undef TT
#undef LIVE
	}
	d_bt_levels<PER, NT, 2>(k, sx, tid);
	__syncthreads();
#pragma unroll
	for (int r = 0; r < PER; ++r) { const int e = tid * PER + r; sx[e + (e >> 4)] = k[r]; }
	__syncthreads();
	uint64_t *out = keys + big_off[ent] + base;
	for (uint32_t t = tid; t < cnt; t += NT) out[t] = sx[t + (t >> 4)];
}
template __global__ void k_anchor_run_sort<16, 8, 1024>(const uint64_t *, const uint32_t *, const uint64_t *, const AlMatch *, const uint32_t *, const uint32_t *, const uint32_t *, int, const uint64_t *, const uint32_t *, const uint64_t *, uint64_t *, uint32_t *, int, uint32_t, uint32_t);

struct RunMergeOut { const uint64_t *a_off, *mini_off; const uint32_t *frag_first, *rd_len; const AlMatch *match; AlAnchor *anchors; uint32_t *tie_list; int rid_bits, mini_span; };
struct RunGeo { uint32_t ent, f, n, o, la, lb, d0, d1; uint64_t A0; int last; bool skip; };
__device__ __forceinline__ RunGeo d_run_geo(const uint32_t blk, const uint32_t *__restrict__ tile_ent, const uint64_t *__restrict__ tile_off, const uint32_t *__restrict__ frag_list,
                                            const uint32_t *__restrict__ frag_na, const uint32_t *__restrict__ frag_nm, const int pass, const uint32_t run_len, const uint32_t tile)
{
	RunGeo g; g.ent = tile_ent[blk]; g.f = frag_list[g.ent]; g.n = frag_na[g.f];
	g.last = 0; while (((uint64_t)run_len << (g.last + 1)) < g.n) ++g.last;   // the fragment's last pass: two runs of run_len << last cover it
	g.skip = frag_nm[g.f] > 1024u || pass > g.last;                           // (k_anchor_run_sort's MCAP: flagged there)
	g.o = (uint32_t)(blk - tile_off[g.ent]) * tile;
	const uint64_t L = (uint64_t)run_len << pass, B0 = (uint64_t)g.o / (2 * L) * (2 * L) + L;
	g.A0 = B0 - L;
	g.la = (uint32_t)(g.n - g.A0 < L ? g.n - g.A0 : L); g.lb = B0 < g.n ? (uint32_t)(g.n - B0 < L ? g.n - B0 : L) : 0u;
	g.d0 = g.o - (uint32_t)g.A0; g.d1 = g.d0 + tile < g.la + g.lb ? g.d0 + tile : g.la + g.lb;
	return g;
}
// merge path, a thread per tile: how many keys of run A are among the first d0 of the pair's merged sequence (the tile's end is the next tile's start, or the end of the pair)
__global__ void __launch_bounds__(256)
k_anchor_run_cuts(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ frag_list, const uint32_t *__restrict__ tile_ent, uint32_t n_tile, const uint64_t *__restrict__ tile_off,
                  const uint64_t *__restrict__ big_off, const uint32_t *__restrict__ frag_na, const uint32_t *__restrict__ frag_nm, const int pass, const uint32_t run_len, const uint32_t tile, uint32_t *__restrict__ cuts)
{
	const uint32_t blk = blockIdx.x * blockDim.x + threadIdx.x;
	if (blk >= n_tile) return;
	const RunGeo g = d_run_geo(blk, tile_ent, tile_off, frag_list, frag_na, frag_nm, pass, run_len, tile);
	if (g.skip || g.lb == 0) { cuts[blk] = g.d0; return; }
	const uint64_t *A = kin + big_off[g.ent] + g.A0, *B = A + ((uint64_t)run_len << pass);
	const uint32_t d = g.d0;
	uint32_t lo = d > g.lb ? d - g.lb : 0u, hi = d < g.la ? d : g.la;
	while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (A[mid] < B[d - 1 - mid]) lo = mid + 1; else hi = mid; }
	cuts[blk] = lo;
}
__global__ void __launch_bounds__(256)
k_anchor_run_merge(const uint64_t *__restrict__ kin, uint64_t *__restrict__ kout, const uint32_t *__restrict__ frag_list, const uint32_t *__restrict__ tile_ent, const uint32_t *__restrict__ cuts, const uint64_t *__restrict__ tile_off,
                   const uint64_t *__restrict__ big_off, const uint32_t *__restrict__ frag_na, const uint32_t *__restrict__ frag_nm, const int pass /* input runs of run_len << pass keys */, const RunMergeOut O, const uint32_t run_len, const uint32_t tile /* <= AL_MERGE_TILE */)
{
	constexpr int NT = 256, PER = AL_MERGE_TILE / NT;
	__shared__ uint64_t sk[AL_MERGE_TILE + 1];
	const int tid = threadIdx.x;
	const uint32_t blk = blockIdx.x;
	const RunGeo g = d_run_geo(blk, tile_ent, tile_off, frag_list, frag_na, frag_nm, pass, run_len, tile);
	if (g.skip) return;
	const uint32_t ent = g.ent, f = g.f, n = g.n, o = g.o, la = g.la, lb = g.lb, d0 = g.d0, d1 = g.d1, tot = d1 - d0; const int last = g.last;
	const uint64_t L = (uint64_t)run_len << pass;
	const uint64_t *A = kin + big_off[ent] + g.A0, *B = A + L;
	if (tot == 0) return;                                                     // (block-uniform; every tile of a fragment holds a key)
	if (lb == 0 && pass < last) {   // a run without a partner in this pass: copied  (lb == 0 in the last pass: a fragment of one run -- tests lower the class bound)
		uint64_t *out = kout + big_off[ent] + o;
		for (uint32_t t = tid; t < tot; t += NT) out[t] = A[d0 + t];
		return;
	}
	const uint32_t a0 = lb ? cuts[blk] : d0, a1 = !lb ? d1 : d1 == la + lb ? la : cuts[blk + 1];   // (d1 < la + lb: the next tile is this pair's too)
	(void)n;
	const uint32_t b0 = d0 - a0, b1 = d1 - a1, na = a1 - a0, nb = b1 - b0;
	{   // (round 6) the tile's PER loads per thread in flight together: one address per key (run A's or run B's), no branch around the load -- a thread past the tile's end reads
		// its last key again -- the stores to LDS after them (`sk[t] = t < na ? A[..] : B[..]` in a loop waited for every key's load in turn: 1.25 -> ms per pass of 238 M keys)
		uint64_t kk[PER];
#pragma unroll
		for (int r = 0; r < PER; ++r) {
			const uint32_t t = (uint32_t)tid + (uint32_t)r * NT, tc = t < tot ? t : tot - 1u;
			const uint64_t *src = tc < na ? A + a0 + tc : B + b0 + (tc - na);
			kk[r] = *src;
		}
#pragma unroll
		for (int r = 0; r < PER; ++r) asm volatile("" : "+v"(kk[r]));
#pragma unroll
		for (int r = 0; r < PER; ++r) { const uint32_t t = (uint32_t)tid + (uint32_t)r * NT; if (t < tot) sk[t] = kk[r]; }
	}
	__syncthreads();
	uint64_t v[PER];
	{
		const uint64_t *sa = sk, *sb = sk + na;
		const uint32_t d = (uint32_t)tid * PER < tot ? (uint32_t)tid * PER : tot;
		uint32_t lo = d > nb ? d - nb : 0u, hi = d < na ? d : na;
		while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sa[mid] < sb[d - 1 - mid]) lo = mid + 1; else hi = mid; }
		uint32_t ia = lo, ib = d - lo;
		uint64_t ka = ia < na ? sa[ia] : UINT64_MAX, kb = ib < nb ? sb[ib] : UINT64_MAX;   // (a key is never all ones: its list index would be 0xffff with every x bit set)
#pragma unroll
		for (int r = 0; r < PER; ++r) {
			const bool ta = ka < kb;
			v[r] = ta ? ka : kb;
			if (ta) { ++ia; ka = ia < na ? sa[ia] : UINT64_MAX; } else { ++ib; kb = ib < nb ? sb[ib] : UINT64_MAX; }
		}
	}
	if (pass < last) {
		uint64_t *out = kout + big_off[ent] + o;
#pragma unroll
		for (int r = 0; r < PER; ++r) { const uint32_t t = (uint32_t)tid * PER + r; if (t < tot) out[t] = v[r]; }
		return;
	}
	// the fragment's last pass: anchors out, tie flag
	uint64_t pred = 0; bool has_pred = false;
	if (tid == 0) { if (a0) { pred = A[a0 - 1]; has_pred = true; } if (b0) { const uint64_t q = B[b0 - 1]; if (!has_pred || q > pred) pred = q; has_pred = true; } }
	__syncthreads();
#pragma unroll
	for (int r = 0; r < PER; ++r) { const uint32_t t = (uint32_t)tid * PER + r; if (t < tot) sk[t + 1] = v[r]; }
	if (tid == 0) sk[0] = has_pred ? pred : ~v[0];                             // (no predecessor: anything with another x)
	__syncthreads();
	const uint32_t r0 = O.frag_first[f], r1 = O.frag_first[f + 1];
	int qlen = 0; for (uint32_t r = r0; r < r1; ++r) qlen += (int)O.rd_len[r];
	const AlMatch *m = O.match + O.mini_off[r0];
	AlAnchor *out = O.anchors + O.a_off[f] + o;
	const int sbit = 32 + O.rid_bits; const uint64_t lowmask = (1ULL << sbit) - 1; const uint32_t span = (uint32_t)O.mini_span;
	int tie = 0;
	{   // the PER match records of a thread's keys loaded together, then the anchors stored
		uint64_t kx[PER]; uint32_t qp[PER], fl[PER];
#pragma unroll
		for (int r = 0; r < PER; ++r) {
			const uint32_t t = (uint32_t)tid + (uint32_t)r * NT, tc = t < tot ? t : tot - 1u;
			const uint64_t key = sk[tc + 1]; kx[r] = key >> 16;
			if (t < tot && (sk[tc] >> 16) == kx[r]) tie = 1;
			const AlMatch mm = m[(uint32_t)key & 0xffffu]; qp[r] = mm.q_pos; fl[r] = mm.flags;
		}
#pragma unroll
		for (int r = 0; r < PER; ++r) asm volatile("" : "+v"(qp[r]), "+v"(fl[r]));
#pragma unroll
		for (int r = 0; r < PER; ++r) {
			const uint32_t t = (uint32_t)tid + (uint32_t)r * NT;
			AlAnchor a; a.x = (kx[r] & lowmask) | (kx[r] >> sbit & 1) << 63;
			a.y = (a.x >> 63) ? (uint64_t)span << 32 | (uint32_t)(qlen - ((int)(qp[r] >> 1) + 1 - (int)span) - 1) : (uint64_t)span << 32 | (qp[r] >> 1);
			a.y |= (uint64_t)(fl[r] & 0xff) << AL_SEED_SEG_SHIFT;
			if (fl[r] & (1u << 8)) a.y |= AL_SEED_TANDEM;
			if (t < tot) out[t] = a;
		}
	}
	if (tie) O.tie_list[f] = 1u;                                              // merged again by the heap kernels (exact order among equal x)
}

// K3 for fragments with at most 64 anchors (the bulk on a low-repeat genome): nothing but registers, so 32 wavefronts per CU
// stay resident and hide the dependent HBM reads (count -> match records -> positions) that bound this stage.
// Lane t builds anchor t (its owning match is found by counting prefix sums, read with wave-uniform readlanes), the
// sort is a rank sort on x (64-bit compares against one broadcast element per step); equal x -> exact heap emulation.
__global__ void __launch_bounds__(64)
k_anchor_sort_small(const uint64_t *__restrict__ pos, const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len,
                    const uint64_t *__restrict__ mini_off, const AlMatch *__restrict__ match,
                    const uint32_t *__restrict__ frag_nm, const uint32_t *__restrict__ frag_na, const uint64_t *__restrict__ a_off,
                    AlAnchor *__restrict__ anchors, uint32_t *__restrict__ tie_list, unsigned int *__restrict__ tie_cnt,
                    const uint32_t *__restrict__ frag_list, int n_list, unsigned long long *__restrict__ counters, int mini_span)
{
	const int lane = threadIdx.x;
	if ((int)blockIdx.x >= n_list) return;
	const uint32_t f = frag_list ? frag_list[blockIdx.x] : blockIdx.x;
	const uint32_t n = frag_na[f], n_m = frag_nm[f];
	if (n == 0) return;
	const uint32_t r0 = frag_first[f], r1 = frag_first[f + 1];
	int qlen = 0; for (uint32_t r = r0; r < r1; ++r) qlen += (int)rd_len[r];
	const AlMatch *m = match + mini_off[r0];
	AlAnchor *out = anchors + a_off[f];
	if (n > 64u) { if (lane == 0) tie_list[f] = 1u; return; }             // not this kernel's class (unreachable with the caller's ordering)
	AlMatch mm; mm.off_lo = 0; mm.n = 0; mm.q_pos = 0; mm.flags = 0;
	if ((uint32_t)lane < n_m) mm = m[lane];
	uint32_t incl = mm.n;
	for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
	const uint32_t pre = incl - mm.n;
	uint32_t le = 0;                                 // matches whose first anchor index is <= lane
	for (uint32_t mi = 0; mi < n_m; ++mi) le += (uint32_t)__shfl((int)pre, (int)mi) <= (uint32_t)lane ? 1u : 0u;
	const int own = (int)le - 1;
	const uint32_t o_off = (uint32_t)__shfl((int)mm.off_lo, own), o_fl = (uint32_t)__shfl((int)mm.flags, own), o_qp = (uint32_t)__shfl((int)mm.q_pos, own), o_pre = (uint32_t)__shfl((int)pre, own);
	uint64_t x = UINT64_MAX, y = UINT64_MAX;
	if ((uint32_t)lane < n) {
		const uint64_t r = d_match_pos(pos, o_off, o_fl, (uint32_t)lane - o_pre);
		const uint32_t span = (uint32_t)mini_span, seg = o_fl & 0xff;
		const int32_t rpos = (uint32_t)r >> 1;
		if ((r & 1) == (o_qp & 1)) {                                          // map.c:176-190
			x = (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
			y = (uint64_t)span << 32 | (o_qp >> 1);
		} else {
			x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
			y = (uint64_t)span << 32 | (uint32_t)(qlen - ((int)(o_qp >> 1) + 1 - (int)span) - 1);
		}
		y |= (uint64_t)seg << AL_SEED_SEG_SHIFT;
		if (o_fl & (1u << 8)) y |= AL_SEED_TANDEM;
	}
	uint32_t rank = 0; bool tie = false;
	const int xlo = (int)(uint32_t)x, xhi = (int)(uint32_t)(x >> 32);
	for (uint32_t j = 0; j < n; ++j) {
		const uint64_t xj = (uint64_t)(uint32_t)__shfl(xlo, (int)j) | (uint64_t)(uint32_t)__shfl(xhi, (int)j) << 32;
		rank += xj < x ? 1u : 0u;
		tie = tie || (xj == x && j != (uint32_t)lane);
	}
	if (__ballot(tie && (uint32_t)lane < n)) { if (lane == 0) tie_list[f] = 1u; return; }   // merged by k_anchor_heap
	if ((uint32_t)lane < n) { AlAnchor a; a.x = x; a.y = y; out[rank] = a; }
}

// =============================================================================================
// K4: chaining.  One wave per fragment; anchors + DP arrays staged in LDS (global scratch for fragments
// larger than the tile).  The predecessor scan of anchor i is done 64 candidates at a time: every lane
// scores one j, an exclusive prefix-max (shuffles) decides which lanes would have raised max_f in the
// sequential order, the "already a predecessor" marks t[] go through LDS, and the max_skip early exit
// is replayed on the two ballot masks -- bit-identical to the scalar loop (chain.c:52-82).
// =============================================================================================
__device__ __forceinline__ int d_ilog2(uint32_t v) { return 31 - __clz((int)v); }

struct ChainArrays { uint64_t *x; int32_t *q; uint32_t *m; int32_t *f, *p, *t, *v; };   // m: span | sid<<8


// accessor for ordering chains by the x of their first anchor: elements are chain ids in T[], chain c's first anchor is
// a[V[Pp[c] + len(c) - 1]] (V = backtrack visit list, Pp = chain start offsets into V, utmp low word = chain length).
struct ChainOrderAcc {
	typedef int32_t E;
	int32_t *T; const int32_t *V, *Pp; const uint64_t *utmp, *X; const AlAnchor *a; bool in_lds;
	__device__ __forceinline__ uint64_t keyof(const int32_t &c) const { const int32_t i = V[Pp[c] + (int32_t)(uint32_t)utmp[c] - 1]; return in_lds ? X[i] : a[i].x; }
	__device__ __forceinline__ uint64_t key(int i) const { return keyof(T[i]); }
	__device__ __forceinline__ int32_t get(int i) const { return T[i]; }
	__device__ __forceinline__ void set(int i, const int32_t &c) { T[i] = c; }
};
template <int CAP>
__global__ void __launch_bounds__(64)
k_chain(const AlAnchor *__restrict__ anchors, const uint64_t *__restrict__ a_off, const uint32_t *__restrict__ frag_na,
        const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len,
        AlAnchor *__restrict__ chained, uint64_t *__restrict__ u_out, uint32_t *__restrict__ frag_nu,
        int32_t *__restrict__ ws_i32 /* 4 ints per anchor */, uint64_t *__restrict__ ws_u64 /* 1 per anchor */,
        const uint32_t *__restrict__ frag_list, int n_list, AlParams P, unsigned long long *__restrict__ counters, ChainSeg seg)
{
	__shared__ uint64_t s_all[4 * CAP];                  // one block, so that the backtrack of a large fragment can use all of it
	uint64_t *const sx = s_all, *const s_qm = s_all + CAP;   // s_qm: Q (int32) and M (u32) halves during the DP; chain list (u64) in the tail
	int32_t *const sf = (int32_t *)(s_all + 2 * CAP), *const sp = sf + CAP, *const st_ = sp + CAP, *const sv = st_ + CAP;
	__shared__ int32_t s_nu;
	__shared__ typename ChainWarpScan::storage_type s_scan;
	__shared__ uint16_t s_rs[AL_RS_SCRATCH / 2];         // work area of the > 64-chain ordering sort
	int32_t *sq = (int32_t *)s_qm; uint32_t *sm = (uint32_t *)s_qm + CAP;
	const int lane = threadIdx.x;
	if ((int)blockIdx.x >= n_list) return;
	const uint32_t f = frag_list ? frag_list[blockIdx.x] : blockIdx.x;
	if (seg.tie_mode) { const bool flagged = seg.tie_flag[f] != 0; if (flagged != (seg.tie_mode == 2)) return; }
	const int64_t n = frag_na[f];
	__shared__ int s_tie_eq;                                                   // segment mode: equal chain starts seen by the <= 64-chain ranking
	uint32_t *const rec = seg.meta ? seg.res + 4 * (size_t)f : nullptr;         // the entry's result record (ChainSeg)
	if (lane == 0) { s_tie_eq = 0; if (rec) *reinterpret_cast<uint4 *>(rec) = make_uint4(0u, 0u, 0u, 0u); else frag_nu[f] = 0; }
	if (n == 0) return;
	int n_segs, qlen_sum = 0;
	if (seg.meta) { const uint32_t mt = seg.meta[f]; qlen_sum = (int)(mt & 0x7fffffffu); n_segs = (mt >> 31) ? 2 : 1; }
	else {
		const uint32_t r0 = frag_first[f], r1 = frag_first[f + 1];
		n_segs = (int)(r1 - r0);
		for (uint32_t r = r0; r < r1; ++r) qlen_sum += (int)rd_len[r];
	}
	const AlAnchor *a = anchors + a_off[f];
	const bool in_lds = n <= CAP;
#define CHAIN_SYNC() __syncthreads()
	uint64_t *X; int32_t *Q, *F, *Pp, *T, *V; uint32_t *M;
	if (in_lds) { X = sx; Q = sq; F = sf; Pp = sp; T = st_; V = sv; M = sm; }
	else {
		int32_t *w = ws_i32 + a_off[f] * 4;
		F = w; Pp = w + n; T = w + 2 * n; V = w + 3 * n;
		X = nullptr; Q = nullptr; M = nullptr;
	}
	// chaining gaps (map.c:341-351)
	const int max_dist_y = qlen_sum > P.max_gap ? qlen_sum : P.max_gap;
	int max_dist_x;
	if (P.max_gap_ref > 0) max_dist_x = P.max_gap_ref;
	else if (P.max_frag_len > 0) { max_dist_x = P.max_frag_len - qlen_sum; if (max_dist_x < P.max_gap) max_dist_x = P.max_gap; }
	else max_dist_x = P.max_gap;
	const int bw = P.bw, max_skip = P.max_chain_skip, max_iter = P.max_chain_iter, min_cnt = P.min_cnt, min_sc = P.min_chain_score;

	uint64_t sum_qspan = 0;
	for (int64_t i = lane; i < n; i += 64) {
		const AlAnchor e = a[i];
		if (in_lds) { X[i] = e.x; Q[i] = (int32_t)e.y; M[i] = (uint32_t)(e.y >> 32 & 0xff) | (uint32_t)((e.y & AL_SEED_SEG_MASK) >> AL_SEED_SEG_SHIFT) << 8; }
		T[i] = 0;
		sum_qspan += e.y >> 32 & 0xff;
	}
	for (int d = 32; d > 0; d >>= 1) sum_qspan += __shfl_xor(sum_qspan, d);
	const float avg_qspan = (float)((double)(float)sum_qspan / (double)(float)n);   // == (float)sum/n in IEEE fp32 (chain.c:42)
	const double avg_d = (double)avg_qspan;
	CHAIN_SYNC();

	// Fragments of more than CAP anchors keep their DP arrays in global memory (the backtrack needs all of them), but a row only
	// looks back over its predecessor window [st, i): the last RING rows are mirrored in the LDS arrays (slot = row & (RING-1)),
	// and a row whose window lies inside the mirror reads and marks LDS only -- the dependent global round trips (F, P, mark,
	// re-read of the mark) that made such rows latency-bound are gone.  Rows with a longer window (tandem arrays: thousands of
	// predecessors) take the global path; every row writes F / P / V through to global memory.
	constexpr int RING = CAP >= 512 ? 512 : 256;
	static_assert(CAP >= RING, "ring lives in the LDS arrays");
	const uint32_t msk = in_lds ? 0xffffffffu : (uint32_t)(RING - 1);
#define LX(j) sx[(uint32_t)(j) & msk]
#define LQ(j) sq[(uint32_t)(j) & msk]
#define LM(j) sm[(uint32_t)(j) & msk]
#define LF(j) sf[(uint32_t)(j) & msk]
#define LP(j) sp[(uint32_t)(j) & msk]
#define LT(j) st_[(uint32_t)(j) & msk]
#define LV(j) sv[(uint32_t)(j) & msk]
#define AX(i) (lr ? LX(i) : a[i].x)
#define AQ(i) (lr ? LQ(i) : (int32_t)a[i].y)
#define AM(i) (lr ? LM(i) : ((uint32_t)(a[i].y >> 32 & 0xff) | (uint32_t)((a[i].y & AL_SEED_SEG_MASK) >> AL_SEED_SEG_SHIFT) << 8))

	int64_t st = 0;
	for (int64_t i = 0; i < ((P.dbg & 32) ? 0 : n); ++i) {
		if (!in_lds && (i & 63) == 0) {                                   // mirror rows i .. i+63 (their slots held rows i-RING .. i-RING+63)
			CHAIN_SYNC();
			const int64_t r = i + lane;
			if (r < n) { const AlAnchor e = a[r]; LX(r) = e.x; LQ(r) = (int32_t)e.y; LM(r) = (uint32_t)(e.y >> 32 & 0xff) | (uint32_t)((e.y & AL_SEED_SEG_MASK) >> AL_SEED_SEG_SHIFT) << 8; LT(r) = -1; }   // no mark yet
			CHAIN_SYNC();
		}
		const uint64_t ri = in_lds ? LX(i) : a[i].x;
		while (st < i && ri > (in_lds || st >= (i & ~63LL) + 64 - RING ? LX(st) : a[st].x) + (uint64_t)max_dist_x) ++st;
		if (i - st > max_iter) st = i - max_iter;
		const bool lr = in_lds || st >= (i & ~63LL) + 64 - RING;            // this row's window is inside the LDS arrays
		const int32_t qi = AQ(i); const uint32_t mi_ = AM(i);
		const int32_t q_span = mi_ & 0xff, sidi = mi_ >> 8;
		int32_t max_f = q_span, n_skip = 0; int64_t max_j = -1; bool broke = false;
		for (int64_t base = i - 1; base >= st && !broke; base -= 64) {
			const int64_t j = base - lane;
			bool active = j >= st; int32_t sc = INT32_MIN; int32_t pj = -1;
			if (active) {
				const int64_t dr = (int64_t)(ri - AX(j));
				const int32_t dq = qi - AQ(j); const uint32_t mj = AM(j); const int32_t sidj = mj >> 8;
				bool skip = (sidi == sidj && dr == 0) || dq <= 0;
				skip = skip || (sidi == sidj && dq > max_dist_y) || dq > max_dist_x;
				const int32_t dd = dr > dq ? (int32_t)(dr - dq) : (int32_t)(dq - dr);
				skip = skip || (sidi == sidj && dd > bw);
				skip = skip || (n_segs > 1 && sidi == sidj && dr > max_dist_y);
				if (!skip) {
					const int32_t min_d = dq < dr ? dq : (int32_t)dr;
					int32_t s0 = min_d > q_span ? q_span : min_d;
					const int32_t log_dd = dd ? d_ilog2((uint32_t)dd) : 0;
					const int32_t c_lin = (int)((double)dd * .01 * avg_d);
					if (sidi != sidj) {
						if (dr == 0) ++s0;
						else s0 -= c_lin < log_dd ? c_lin : log_dd;
					} else s0 -= c_lin + (log_dd >> 1);
					sc = s0 + (lr ? LF(j) : F[j]);
					pj = lr ? LP(j) : Pp[j];
				} else active = false;
			}
			// sequential replay over lanes 0..63 (descending j)
			int32_t excl;                                                     // exclusive prefix max over the lower lanes (DPP row shifts, not LDS permutes)
			ChainWarpScan().exclusive_scan(sc, excl, INT32_MIN, s_scan, rocprim::maximum<int32_t>());
			const int32_t before = excl > max_f ? excl : max_f;
			const bool upd = active && sc > before;
			// t[p[j]] = i (chain.c:81).  A mark below st is never tested in this row (and would alias a newer row's slot in the mirror).
			if (active && pj >= (int32_t)st) { if (lr) LT(pj) = (int32_t)i; else T[pj] = (int32_t)i; }
			CHAIN_SYNC();
			const bool marked = active && !upd && (lr ? LT(j) : T[j]) == (int32_t)i;
			unsigned long long U = __ballot(upd), K = __ballot(marked);
			unsigned long long both = U | K; int brk = 64;
			while (both) {
				const int b = __ffsll((long long)both) - 1; both &= both - 1;
				if (U >> b & 1) { if (n_skip > 0) --n_skip; }
				else if (++n_skip > max_skip) { brk = b; break; }
			}
			if (brk < 64) { broke = true; U &= (brk == 0 ? 0ULL : (~0ULL >> (64 - brk))); }
			if (U) {
				const int lastu = 63 - __clzll((long long)U);
				max_f = __shfl(sc, lastu); max_j = base - lastu;
			}
			CHAIN_SYNC();
		}
		if (lane == 0) {
			const int32_t vj = max_j < 0 ? INT32_MIN : (lr ? LV(max_j) : V[max_j]);
			const int32_t v = max_j >= 0 && vj > max_f ? vj : max_f;
			if (in_lds) { LF(i) = max_f; LP(i) = (int32_t)max_j; LV(i) = v; }
			else { F[i] = max_f; Pp[i] = (int32_t)max_j; V[i] = v; LF(i) = max_f; LP(i) = (int32_t)max_j; LV(i) = v; }
		}
		CHAIN_SYNC();
	}
#undef LX
#undef LQ
#undef LM
#undef LF
#undef LP
#undef LT
#undef LV

	// ---- chain ends, peaks, backtrack (chain.c:87-160): lane 0, O(n) -----------------------------------
	for (int64_t i = lane; i < n; i += 64) T[i] = 0;
	CHAIN_SYNC();
	for (int64_t i = lane; i < n; i += 64) if (Pp[i] >= 0) T[Pp[i]] = 1;
	CHAIN_SYNC();
	AlAnchor *b = chained + a_off[f];
	uint64_t *u = u_out + a_off[f] + (seg.meta ? 0u : f);   // capacity n + 1 (segment mode: n, a chain has at least one anchor)
	uint64_t *utmp = in_lds ? s_qm : ws_u64 + a_off[f]; // capacity n (Q/M are dead after the DP)
	uint64_t *const okf = seg.okey ? seg.okey + a_off[f] : nullptr, *const okp = okf ? okf + (n + 1) / 2 : nullptr;   // final / processing-order keys (at most n / 2 chains)
	// Peaks (chain.c:87-110) by all lanes: every chain end walks back to its peak on its own; the list of plain 64-bit keys
	// (score << 32 | anchor index) is sorted next, so the order the lanes append in does not matter.  At most n/2 entries
	// (a peak needs a predecessor to reach min_sc), which leaves the upper half of utmp[] free as a second buffer.
	int32_t n_u0 = 0;
	if (!(P.dbg & 64))
	for (int64_t i0 = 0; i0 < n; i0 += 64) {
		const int64_t i = i0 + lane;
		const bool end = i < n && T[i] == 0 && V[i] >= min_sc;
		uint64_t key = 0;
		if (end) {
			int64_t j = i;
			while (j >= 0 && F[j] < V[j]) j = Pp[j];
			if (j < 0) j = i;
			key = (uint64_t)(uint32_t)F[j] << 32 | (uint64_t)j;
		}
		const unsigned long long m = __ballot(end);
		if (end) utmp[n_u0 + __popcll(m & ((1ULL << lane) - 1ULL))] = key;
		n_u0 += __popcll(m);
	}
	CHAIN_SYNC();
	const bool rank_sorted = n_u0 > 64 && 2 * (int64_t)n_u0 <= n;
	if (rank_sorted) {   // descending order: every lane ranks its entries against all others
		uint64_t *dst = utmp + n_u0;
		for (int32_t i0 = 0; i0 < n_u0; i0 += 64) {
			const int32_t i = i0 + lane;
			if (i < n_u0) { const uint64_t x = utmp[i]; int32_t r = 0; for (int32_t j = 0; j < n_u0; ++j) { const uint64_t y = utmp[j]; r += y > x || (y == x && j < i); } dst[r] = x; }   // equal keys (two ends, one peak) take consecutive ranks
		}
		CHAIN_SYNC();
		for (int32_t i = lane; i < n_u0; i += 64) utmp[i] = dst[i];
		CHAIN_SYNC();
	}
	// Backtrack (lane 0, one dependent load per visited anchor): for a fragment whose arrays are in global memory the predecessor
	// indices and the visit marks are staged in LDS (free after the DP) when they fit: 5 bytes per anchor.
	const bool stage_bt = !in_lds && 5 * n <= (int64_t)(32 * CAP);
	int32_t *const bP = (int32_t *)s_all; uint8_t *const bT = (uint8_t *)(bP + n);
	if (n_u0 > 0) {
		if (stage_bt) { for (int64_t i = lane; i < n; i += 64) { bP[i] = Pp[i]; bT[i] = 0; } }
		else { for (int64_t i = lane; i < n; i += 64) T[i] = 0; }
	}
	CHAIN_SYNC();
	// keys / permutation of the chain-ordering sort: for fragments whose DP arrays are in global memory the LDS arrays are free now
	if (lane == 0) {
		int32_t n_u = n_u0, n_v = 0, k = 0;
		if (n_u > 0) {
			// keys are unique (distinct j): any sort; descending by insertion / heap sort
			if (n_u <= 64) {
				for (int32_t i = 1; i < n_u; ++i) { uint64_t t = utmp[i]; int32_t j = i; while (j > 0 && utmp[j - 1] < t) { utmp[j] = utmp[j - 1]; --j; } utmp[j] = t; }
			} else if (!rank_sorted) {
				for (int32_t s0 = (n_u >> 1) - 1; s0 >= 0; --s0) { int32_t i = s0; uint64_t t = utmp[i]; for (;;) { int32_t c = 2 * i + 1; if (c >= n_u) break; if (c + 1 < n_u && utmp[c + 1] > utmp[c]) ++c; if (utmp[c] <= t) break; utmp[i] = utmp[c]; i = c; } utmp[i] = t; }
				for (int32_t e = n_u - 1; e > 0; --e) { uint64_t t = utmp[e]; utmp[e] = utmp[0]; int32_t i = 0; for (;;) { int32_t c = 2 * i + 1; if (c >= e) break; if (c + 1 < e && utmp[c + 1] > utmp[c]) ++c; if (utmp[c] <= t) break; utmp[i] = utmp[c]; i = c; } utmp[i] = t; }
				for (int32_t i = 0; i < n_u >> 1; ++i) { uint64_t t = utmp[i]; utmp[i] = utmp[n_u - 1 - i]; utmp[n_u - 1 - i] = t; }
			}
			// backtrack; V[] is reused as the visit list v[] (chain.c:113-127)
			for (int32_t i = 0; i < n_u; ++i) {
				const uint64_t key0 = utmp[i];
				const int32_t n_v0 = n_v, k0 = k; int64_t j = (int32_t)utmp[i];
				if (stage_bt) { do { V[n_v++] = (int32_t)j; bT[j] = 1; j = bP[j]; } while (j >= 0 && bT[j] == 0); }
				else { do { V[n_v++] = (int32_t)j; T[j] = 1; j = Pp[j]; } while (j >= 0 && T[j] == 0); }
				if (j < 0) { if (n_v - n_v0 >= min_cnt) utmp[k++] = utmp[i] >> 32 << 32 | (uint32_t)(n_v - n_v0); }
				else if ((int32_t)(utmp[i] >> 32) - F[j] >= min_sc) { if (n_v - n_v0 >= min_cnt) utmp[k++] = (uint64_t)((utmp[i] >> 32) - (uint64_t)(uint32_t)F[j]) << 32 | (uint32_t)(n_v - n_v0); }
				if (k0 == k) n_v = n_v0; else if (okp) okp[k0] = key0;
			}
			n_u = k;
			// order chains by the x of their first anchor (chain.c:144-160): stable insertion for <= 64 (ksort.h:149).
			// first anchor of chain c = a[V[k0 + ni - 1]].  Pp[] (free now) = chain start offsets into V, T[] = permutation.
			int32_t off = 0;
			for (int32_t c = 0; c < n_u; ++c) { Pp[c] = off; off += (int32_t)(uint32_t)utmp[c]; T[c] = c; }
		}
		s_nu = n_u;
	}
	CHAIN_SYNC();
	// Chain order (chain.c:144-160).  The key of chain c is the position of its first anchor, a[V[Pp[c] + len - 1]].x: three to four
	// dependent loads, and the sorts compare keys O(n_u^2) times -- so the keys are computed once, by all lanes, into LDS, and up
	// to 64 chains (almost every fragment) are put in order by the wavefront itself: the rank of a chain among (key, index) pairs
	// is its place after the reference's stable insertion sort (ksort.h:149).  More chains: lane 0 runs the radix restatement.
	const int32_t n_u1 = s_nu;
	uint64_t *keysL = nullptr; int32_t *permL = nullptr;
	if (n_u1 > 1) {
		if (!in_lds && n_u1 <= CAP) { keysL = sx; permL = sp; }                       // DP arrays in global memory: LDS is free
		else if (in_lds && 2 * n_u1 <= CAP) { keysL = s_qm + n_u1; permL = st_; }     // behind the chain list; the permutation is T[]
	}
	const bool use_lds_order = keysL != nullptr && !in_lds;                          // per-chain records in LDS too (see the copy-out)
	// thousands of chains (fragments inside high-copy families): keys once into the free half of the chain list's work area, so that
	// the serial sort restatement reads one word per key instead of following chain -> visit list -> anchor
	uint64_t *const keysG = (!keysL && !in_lds && n_u1 > 1 && 2 * (int64_t)n_u1 <= n) ? utmp + n_u1 : nullptr;
	if (keysG) {
		for (int32_t c = lane; c < n_u1; c += 64) { const uint64_t uc = utmp[c]; keysG[c] = a[V[Pp[c] + (int32_t)(uint32_t)uc - 1]].x; }
		CHAIN_SYNC();
	}
	if (keysL) {
		for (int32_t c = lane; c < n_u1; c += 64) {
			const uint64_t uc = utmp[c]; const int32_t k0 = Pp[c], idx = V[k0 + (int32_t)(uint32_t)uc - 1];
			keysL[c] = in_lds ? X[idx] : a[idx].x;
			if (!in_lds) { permL[c] = c; s_qm[c] = uc; sf[c] = k0; }
		}
		CHAIN_SYNC();
		if (n_u1 <= 64) {
			bool eq = false;
			if (lane < n_u1) {
				const uint64_t kx = keysL[lane]; int32_t r = 0;
				for (int32_t j = 0; j < n_u1; ++j) { const uint64_t kj = keysL[j]; r += kj < kx || (kj == kx && j < lane); eq = eq || (kj == kx && j != lane); }
				permL[r] = lane;
			}
			if (seg.meta && __ballot(eq) && lane == 0) s_tie_eq = 1;
		}
	}
	CHAIN_SYNC();
	if (lane == 0) {
		const int32_t n_u = n_u1;
		if (n_u > 0) {
			bool tie = false;
			if (keysL) {
				if (n_u > 64) {   // radix_sort_128x above 64 entries (ksort.h:147-151): its order among equal keys is reproduced
					struct { typedef int32_t E; int32_t *t; const uint64_t *k;
					         __device__ __forceinline__ uint64_t keyof(const int32_t &c) const { return k[c]; }
					         __device__ __forceinline__ uint64_t key(int i) const { return k[t[i]]; }
					         __device__ __forceinline__ int32_t get(int i) const { return t[i]; }
					         __device__ __forceinline__ void set(int i, const int32_t &c) { t[i] = c; } } acc{permL, keysL};
					tie = d_rs_sort(acc, n_u, s_rs);
					if (seg.meta) for (int32_t i = 1; i < n_u; ++i) tie = tie || acc.key(i) == acc.key(i - 1);   // equal keys: the fragment-wide sort decides
				}
			} else if (keysG) {
				struct { typedef int32_t E; int32_t *t; const uint64_t *k;
				         __device__ __forceinline__ uint64_t keyof(const int32_t &c) const { return k[c]; }
				         __device__ __forceinline__ uint64_t key(int i) const { return k[t[i]]; }
				         __device__ __forceinline__ int32_t get(int i) const { return t[i]; }
				         __device__ __forceinline__ void set(int i, const int32_t &c) { t[i] = c; } } acc{T, keysG};
				tie = d_rs_sort(acc, n_u, s_rs);
				if (seg.meta) for (int32_t i = 1; i < n_u; ++i) tie = tie || acc.key(i) == acc.key(i - 1);
			} else if (n_u > 1) {
				ChainOrderAcc acc{T, V, Pp, utmp, X, a, in_lds};
				tie = d_rs_sort(acc, n_u, s_rs);
				if (seg.meta) for (int32_t i = 1; i < n_u; ++i) tie = tie || acc.key(i) == acc.key(i - 1);
			}
			int32_t o = 0;
			if (use_lds_order) { for (int32_t i = 0; i < n_u; ++i) { st_[i] = o; o += (int32_t)(uint32_t)s_qm[sp[i]]; } }                // st_[] = output offset of sorted chain i
			else { for (int32_t i = 0; i < n_u; ++i) { const int32_t c = T[i]; u[i] = utmp[c]; if (okf) okf[i] = okp[c]; F[i] = o; o += (int32_t)(uint32_t)utmp[c]; } }   // F[] = output offset of sorted chain i
			if (tie && !seg.meta) atomicAdd(&counters[1], 1ULL);
			if (rec) { rec[3] = (uint32_t)o | ((tie || s_tie_eq) ? 1u << 31 : 0u); }    // (the chain list of this kernel's entries always goes to the scratch range: words 0-1 stay 0)
		}
		if (rec) rec[2] = (uint32_t)n_u; else frag_nu[f] = (uint32_t)n_u;
	}
	CHAIN_SYNC();
	if (use_lds_order) {   // many short chains: a lane per chain
		const int32_t n_u = s_nu;
		for (int32_t i = lane; i < n_u; i += 64) {
			const int32_t c = sp[i]; const uint64_t uc = s_qm[c]; const int32_t ni = (int32_t)(uint32_t)uc, k0 = sf[c], o = st_[i];
			u[i] = uc; if (okf) okf[i] = okp[c];
			for (int32_t j = 0; j < ni; ++j) b[o + j] = a[V[k0 + (ni - j - 1)]];
		}
	} else {   // copy-out of the chained anchors by the whole wave: b[F[i] + j] = a[V[k0 + ni - 1 - j]]
		const int32_t n_u = s_nu;
		for (int32_t i = 0; i < n_u; ++i) {
			const int32_t c = T[i], ni = (int32_t)(uint32_t)utmp[c], k0 = Pp[c], o = F[i];
			for (int32_t j = lane; j < ni; j += 64) b[o + j] = a[V[k0 + (ni - j - 1)]];
		}
	}
#undef AX
#undef CHAIN_SYNC
#undef AQ
#undef AM
}

// =============================================================================================
// K4 (fragments of <= 128 anchors, LDS resident): one lane per fragment, LANES fragments per wavefront, every per-anchor
// row staged in LDS as [anchor][lane] so that nothing is re-fetched from HBM inside the O(n^2) recurrence
// (10 bytes per anchor, layout in the kernel).  CAPL anchors per fragment; the caller orders fragments by anchor count and
// a wavefront whose largest fragment does not fall in (lo_excl, CAPL] leaves the work to another instantiation.  All spans
// must equal k and max_dist_x < 2^15 (true on the short-read path); a fragment violating that is a counted error.
// Round 6: nothing in this kernel waits for HBM one access at a time any more.  By the cycle counters (AL_DBG2 bit 3) a wavefront spent 92 k
// cycles loading its entries' anchors (a dependent 16-byte load per anchor and lane), 168 k in the recurrence, 50 k on chain ends / backtrack /
// order (their scratch lay in HBM, ws_u64: a round trip per access, and the chains' first x came back from HBM too) and 105 k copying the chained
// anchors out (load, then store, one at a time).  Now: the anchors are loaded four at a time (clamped indices, no branch around a load); the chain
// ends' scratch lives in the rows' dead halves (after the recurrence bytes 0-2 of row c hold chain end / chain c); chains are ordered by
// first-anchor x without reading x back -- the anchors are sorted, so index order is x order but for equal x, which the loading loop notes in a
// bit mask per lane; the chained anchors leave four at a time.  (Measured and dropped on the way: the WAVEFRONT loading entry by entry, 64 anchors
// per load -- coalesced, but 64 entries x two phases of cross-lane work and LDS round trips at ~480 cycles each under this kernel's LDS load.)
// =============================================================================================
#define AL_CLIN_N 512
__device__ __forceinline__ uint64_t d_rl64(uint64_t v, int l) { return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32; }
template <int CAPL, int LANES>
__global__ void __launch_bounds__(64)
k_chain_lds(const AlAnchor *__restrict__ anchors, const uint64_t *__restrict__ a_off, const uint32_t *__restrict__ frag_na,
            const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len,
            AlAnchor *__restrict__ chained, uint64_t *__restrict__ u_out, uint32_t *__restrict__ frag_nu, uint64_t *__restrict__ ws_u64,
            const uint32_t *__restrict__ order, int n_list, int lo_excl, AlParams P, unsigned long long *__restrict__ counters, ChainSeg seg,
            uint32_t *__restrict__ uo_out /* whole-fragment mode: offset of every chain's first anchor in the fragment's range of chained[] */,
            int utmp_stride /* (rounds 2-5: where the chain-end scratch lay in ws_u64; unused since it moved into the rows) */)
{
	// 10 bytes per anchor: one 8-byte row  [ xlo:16 | q:12 | seg:1 | far:1 | -:2 | f:16 | p:8 | t:8 ]  + the peak score v:16.
	//  xlo = low 16 bits of the reference position: inside the predecessor window the true distance is <= max_dist_x < 2^15,
	//        so (xlo_i - xlo_j) mod 2^16 is the distance; "far" marks an anchor whose predecessor lies in another
	//        (strand, contig) block or >= 2^15 away, which is all the window start needs to know.
	constexpr int RS = LANES;
	constexpr int NR = (CAPL + 63) / 64;                                         // rounds of 64 anchors per entry
	__shared__ uint64_t srow[CAPL * RS];
	__shared__ int16_t sv[CAPL * LANES];
	__shared__ uint8_t s_pen_same[AL_CLIN_N], s_pen_diff[AL_CLIN_N];
	(void)ws_u64; (void)utmp_stride;
	const int lane = threadIdx.x;
	// the list is ordered by anchor count: blocks are issued roughly in index order, so the heaviest wavefronts go first and
	// the light ones fill the tail of the launch
	const int t0 = (int)(gridDim.x - 1 - blockIdx.x) * LANES + lane;
	const bool have = lane < LANES && t0 < n_list;
	const uint32_t f = have ? (order ? order[t0] : (uint32_t)t0) : 0;
	// the entry's record: every word fetched NOW, unconditionally (a lane without an entry reads entry 0, an absent array is stood in for by one that exists),
	// so that the loads are in flight together -- behind a branch each would be a dependent HBM round trip of its own
	const uint32_t *const p_tie = seg.tie_mode == 1 ? seg.tie_flag : frag_na, *const p_meta = seg.meta ? seg.meta : frag_na;
	const uint64_t *const p_uslot = seg.uslot ? seg.uslot : a_off; const uint32_t *const p_rel = seg.uslot ? seg.rel : frag_na, *const p_fid = seg.uslot ? seg.fragid : frag_na;
	uint32_t v_na = frag_na[f], v_tie = p_tie[f], v_meta = p_meta[f], rel_f = p_rel[f], fragid_f = p_fid[f]; uint64_t aoff = a_off[f], uslot_f = p_uslot[f];
	asm volatile("" : "+v"(v_na), "+v"(v_tie), "+v"(v_meta), "+v"(rel_f), "+v"(fragid_f), "+v"(aoff), "+v"(uslot_f));
	const bool side = have && seg.tie_mode == 1 && v_tie != 0;                 // chained on the side stream (equal-x anchors)
	const int n = have ? (int)v_na : 0;
	int nmax = n;
	for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_xor(nmax, d); nmax = o > nmax ? o : nmax; }
	if (nmax > CAPL || nmax <= lo_excl) return;                              // another instantiation / kernel owns this wavefront
	// gap costs of chain.c:64-72 for avg_d == k, tabulated with the same two double multiplications for (int)(dd * .01 * avg_d)
	// (entries of at most 16 anchors -- the short segments of repeat-rich fragments, millions per batch -- score too few pairs to
	// repay building the tables: they compute the two costs directly)
	constexpr bool USE_TAB = CAPL > 16;
	if (USE_TAB) {
		for (int d = lane; d < AL_CLIN_N; d += 64) {
			const int c_lin = (int)((double)d * .01 * (double)P.k), lg = d ? d_ilog2((uint32_t)d) : 0;
			s_pen_same[d] = (uint8_t)(c_lin + (lg >> 1)); s_pen_diff[d] = (uint8_t)(c_lin < lg ? c_lin : lg);
		}
		__syncthreads();
	}
	bool alive = have && !side;                                                // (the others: no result at all, as before)
	const bool prof = (P.dbg2 >> 3) & 1;                                        // timing experiment: cycles per phase, summed over the wavefronts (counters[24 ..]; printed by al_batch_run)
	long long tp = prof ? clock64() : 0;
#define CL_PROF(i) do { if (prof) { const long long t_ = clock64(); if (lane == 0) atomicAdd(&counters[24 + (i)], (unsigned long long)(t_ - tp)); tp = t_; } } while (0)
	const bool direct = seg.uslot != nullptr;                                  // deferred segment of the tile kernel: straight to the fragment's arrays
	uint32_t *const rec = (seg.meta && !direct) ? seg.res + 4 * (size_t)f : nullptr;         // the entry's result record (ChainSeg): written once, at the exits
#define CHAIN_EXIT0() do { if (rec) *reinterpret_cast<uint4 *>(rec) = make_uint4(0u, 0u, 0u, 0u); else if (!direct) frag_nu[f] = 0; alive = false; } while (0)
	if (alive && n == 0) CHAIN_EXIT0();
#define ROW(j) srow[(j) * RS + lane]
#define ROW32(j) (reinterpret_cast<uint32_t *>(&srow[(j) * RS + lane])[0])
#define VL(j) sv[(j) * LANES + lane]
#define TB(j) (reinterpret_cast<uint8_t *>(&srow[(j) * RS + lane])[7])
#define SRCB(j) (reinterpret_cast<uint8_t *>(&srow[(j) * RS + lane])[3])
#define R_XLO(r) ((uint32_t)(r) & 0xffffu)
#define R_Q(r) ((int32_t)((uint32_t)(r) >> 16 & 0xfffu))
#define R_SEG(r) ((int32_t)((uint32_t)(r) >> 28 & 1u))
#define R_FAR(r) ((uint32_t)(r) >> 29 & 1u)
#define R_F(r) ((int32_t)(int16_t)((r) >> 32))
#define R_P(r) ((uint32_t)((r) >> 48) & 0xffu)
#define R_T(r) ((uint32_t)((r) >> 56))
	int n_segs = 1, qlen_sum = 0;
	if (have) {
		if (seg.meta) { const uint32_t mt = v_meta; qlen_sum = (int)(mt & 0x7fffffffu); n_segs = (mt >> 31) ? 2 : 1; }
		else {
			const uint32_t r0 = frag_first[f], r1 = frag_first[f + 1];
			n_segs = (int)(r1 - r0);
			for (uint32_t r = r0; r < r1; ++r) qlen_sum += (int)rd_len[r];
		}
	}
	const AlAnchor *a = anchors + aoff;
	const int max_dist_y = qlen_sum > P.max_gap ? qlen_sum : P.max_gap;           // map.c:341-351
	int max_dist_x;
	if (P.max_gap_ref > 0) max_dist_x = P.max_gap_ref;
	else if (P.max_frag_len > 0) { max_dist_x = P.max_frag_len - qlen_sum; if (max_dist_x < P.max_gap) max_dist_x = P.max_gap; }
	else max_dist_x = P.max_gap;
	const int bw = P.bw, max_skip = P.max_chain_skip, min_cnt = P.min_cnt, min_sc = P.min_chain_score;
	const int32_t q_span = P.k;
	CL_PROF(0);
	// ---- rows: eight anchors per round, their loads in flight together ----
	uint64_t eqm[NR];                                                          // bit i of word r: anchor 64 r + i has the x of the anchor in front of it
	bool bad = max_dist_x > 0x7fff || P.max_chain_iter < CAPL;
#pragma unroll
	for (int r = 0; r < NR; ++r) eqm[r] = 0;
	if (alive) {
		uint64_t prev_x = 0;
		for (int i0 = 0; i0 < n; i0 += 8) {
			AlAnchor e[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) e[u] = a[i0 + u < n ? i0 + u : n - 1];      // (clamped: no branch around a load -- the compiler waits for everything in flight at each join)
#pragma unroll
			for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(e[u].x), "+v"(e[u].y));
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const int i = i0 + u;
				if (i < n) {
					const uint64_t x = e[u].x, y = e[u].y;
					const bool far = i == 0 || (x >> 32) != (prev_x >> 32) || (uint32_t)x - (uint32_t)prev_x > 0x7fffu;
					if (i > 0 && x == prev_x) { if (NR == 1 || i < 64) eqm[0] |= 1ULL << (i & 63); else eqm[NR - 1] |= 1ULL << (i & 63); }
					prev_x = x;
					if ((int)(y >> 32 & 0xff) != q_span || ((y & AL_SEED_SEG_MASK) >> AL_SEED_SEG_SHIFT) > 1 || (uint32_t)y > 0xfffu) bad = true;
					ROW(i) = (uint64_t)((uint32_t)x & 0xffffu) | (uint64_t)((uint32_t)y & 0xfffu) << 16 | (uint64_t)((y & AL_SEED_SEG_MASK) >> AL_SEED_SEG_SHIFT & 1) << 28
					         | (uint64_t)(far ? 1u : 0u) << 29 | (uint64_t)0xffu << 56;
				}
			}
		}
	}
	CL_PROF(1);
	if (alive && bad) { atomicAdd(&counters[7], 1ULL); CHAIN_EXIT0(); }   // not representable in the compact rows (never on the short-read path)
	if (alive) {
	const double avg_d = (double)(float)((double)(float)((uint32_t)q_span * (uint32_t)n) / (double)(float)n);   // (float)sum/n (chain.c:42)
	const bool tab_ok = USE_TAB && avg_d == (double)P.k && P.k * 0.01 * (AL_CLIN_N - 1) + 5.0 < 255.0;
	const uint32_t dr_lim = n_segs > 1 ? (uint32_t)max_dist_y : 0x7fffffffu;   // (uint32_t)(dr - 1) >= dr_lim  <=>  dr == 0 or dr > max_dist_y (paired end only)
	int st = 0; int32_t dist = 0;                                                 // dist = x_i - x_st while st..i lie in one window
	uint32_t prev_xlo = 0;
	uint64_t succ[NR], vis[NR];                                                   // per anchor, in registers: is some anchor's best predecessor / visited by the backtrack (the reference's t[] after chain.c:87)
#pragma unroll
	for (int r = 0; r < NR; ++r) { succ[r] = 0; vis[r] = 0; }
#define BIT_SET(m, j) do { if (NR == 1 || (j) < 64) m[0] |= 1ULL << ((j) & 63); else m[NR - 1] |= 1ULL << ((j) & 63); } while (0)
#define BIT_GET(m, j) (((NR == 1 || (j) < 64 ? m[0] : m[NR - 1]) >> ((j) & 63)) & 1ULL)
	for (int i = 0; i < n; ++i) {                                                 // chain.c:46-85
		const uint64_t ri = ROW(i);
		const uint32_t xi = R_XLO(ri); const int32_t qi = R_Q(ri), sidi = R_SEG(ri);
		int max_j = -1; int32_t max_f = q_span, n_skip = 0;
		// window start (chain.c:52: ri > a[st].x + max_dist_x): a far anchor has no predecessor at all; otherwise the
		// distance to st grows by this step and st advances while it exceeds max_dist_x (steps inside a window are < 2^15)
		if (R_FAR(ri)) { st = i; dist = 0; }
		else {
			dist += (int32_t)((xi - prev_xlo) & 0xffffu);
			while (dist > max_dist_x) { ++st; dist -= (int32_t)((R_XLO(ROW(st)) - R_XLO(ROW(st - 1))) & 0xffffu); }
		}
		prev_xlo = xi;
		// (round 6) TWO candidates per round: their scores are independent (row halves fetched a round ahead, two table lookups each), only the few
		// instructions of the sequential rule -- first maximum wins, marks, max_skip (chain.c:74-81) -- run one after the other.  With 1.5 - 3.5
		// wavefronts per SIMD (LDS capacity) a round's chain of dependent VALU and LDS latencies was what a wavefront waited for.
		// The body is written with selects rather than branches.  t[] marks made by this round for rows already fetched are patched in registers.
		// A row with at most max_skip candidates can never break (it takes max_skip + 1 marked ones), and marks are read only by the row that makes
		// them: such rows -- all rows of the <= 24-anchor classes -- take the loop without marks.
		auto score = [&](const uint32_t lo, const uint32_t hi, bool &skip) -> int32_t {
			const int32_t dr = (int32_t)((xi - (lo & 0xffffu)) & 0xffffu);
			const int32_t dq = qi - (int32_t)(lo >> 16 & 0xfffu);
			const bool same = (int32_t)(lo >> 28 & 1u) == sidi;
			const int32_t dd = dr > dq ? dr - dq : dq - dr;
			// chain.c:56-63 with the range tests folded into unsigned compares: dq <= 0 or dq > max_dist_x; for anchors of the same
			// mate also dr == 0 or (paired end) dr > max_dist_y, dq > max_dist_y, dd > bw
			skip = ((uint32_t)(dq - 1) >= (uint32_t)max_dist_x) | (same & (((uint32_t)(dr - 1) >= dr_lim) | (dq > max_dist_y) | (dd > bw)));
			const int32_t min_d = dq < dr ? dq : dr;
			int32_t sc = min_d > q_span ? q_span : min_d;
			// gap cost (chain.c:64-72) from two per-wavefront tables: same mate c_lin + (ilog2(dd) >> 1), other mate min(c_lin, ilog2(dd))
			const uint32_t di = (uint32_t)dd < AL_CLIN_N ? (uint32_t)dd : AL_CLIN_N - 1;
			int32_t pen_same = (int32_t)s_pen_same[di], pen_diff = (int32_t)s_pen_diff[di];
			if (__builtin_expect(!tab_ok || dd >= AL_CLIN_N, 0)) {
				const int32_t log_dd = dd ? d_ilog2((uint32_t)dd) : 0, c_lin = (int)((double)dd * .01 * avg_d);
				pen_same = c_lin + (log_dd >> 1); pen_diff = c_lin < log_dd ? c_lin : log_dd;
			}
			pen_diff = dr == 0 ? -1 : pen_diff;
			return sc - (same ? pen_same : pen_diff) + (int32_t)(int16_t)(hi & 0xffffu);
		};
#define ROWLO(j) (reinterpret_cast<const uint32_t *>(&srow[(j) * RS + lane])[0])
#define ROWHI(j) (reinterpret_cast<const uint32_t *>(&srow[(j) * RS + lane])[1])
		const bool marks = i - st > max_skip;
		if (!__any(marks)) {
			int j = i - 1;
			uint32_t l0 = ROWLO(j > 0 ? j : 0), h0 = ROWHI(j > 0 ? j : 0), l1 = ROWLO(j > 1 ? j - 1 : 0), h1 = ROWHI(j > 1 ? j - 1 : 0);
			for (; j >= st; j -= 2) {
				const uint32_t la = l0, ha = h0, lb = l1, hb = h1;
				const int ja = j > 2 ? j - 2 : 0, jb = j > 3 ? j - 3 : 0;
				l0 = ROWLO(ja); h0 = ROWHI(ja); l1 = ROWLO(jb); h1 = ROWHI(jb);
				bool skip_a, skip_b;
				const int32_t sa = score(la, ha, skip_a), sb = score(lb, hb, skip_b);
				const bool better_a = !skip_a && sa > max_f;
				max_f = better_a ? sa : max_f; max_j = better_a ? j : max_j;
				const bool better_b = j - 1 >= st && !skip_b && sb > max_f;
				max_f = better_b ? sb : max_f; max_j = better_b ? j - 1 : max_j;
			}
		} else {
			int j = i - 1;
			uint32_t l0 = ROWLO(j > 0 ? j : 0), h0 = ROWHI(j > 0 ? j : 0), l1 = ROWLO(j > 1 ? j - 1 : 0), h1 = ROWHI(j > 1 ? j - 1 : 0);
			bool done = false;
			const uint32_t im = (uint32_t)i << 24;
			for (; j >= st && !done; j -= 2) {
				const uint32_t la = l0, ha = h0, lb = l1; uint32_t hb = h1;
				const int ja = j > 2 ? j - 2 : 0, jb = j > 3 ? j - 3 : 0;
				l0 = ROWLO(ja); h0 = ROWHI(ja); l1 = ROWLO(jb); h1 = ROWHI(jb);
				bool skip_a, skip_b;
				const int32_t sa = score(la, ha, skip_a), sb = score(lb, hb, skip_b);
				// candidate j
				const uint32_t pa = ha >> 16 & 0xffu;
				const bool better_a = !skip_a && sa > max_f;
				const bool marked_a = !skip_a && !better_a && (ha >> 24) == (uint32_t)i;
				max_f = better_a ? sa : max_f; max_j = better_a ? j : max_j;
				n_skip += marked_a ? 1 : (better_a && n_skip > 0 ? -1 : 0);
				const bool done_a = marked_a && n_skip > max_skip;                       // the reference breaks before marking p[j]
				const bool mark_a = !skip_a && !done_a && pa != 0xff;
				if (mark_a) TB(pa) = (uint8_t)i;
				hb = mark_a && (int)pa == j - 1 ? ((hb & 0x00ffffffu) | im) : hb;         // (rows fetched before the mark was made)
				h0 = mark_a && (int)pa == ja ? ((h0 & 0x00ffffffu) | im) : h0;
				h1 = mark_a && (int)pa == jb ? ((h1 & 0x00ffffffu) | im) : h1;
				// candidate j - 1
				const bool act_b = j - 1 >= st && !done_a;
				const uint32_t pb = hb >> 16 & 0xffu;
				const bool better_b = act_b && !skip_b && sb > max_f;
				const bool marked_b = act_b && !skip_b && !better_b && (hb >> 24) == (uint32_t)i;
				max_f = better_b ? sb : max_f; max_j = better_b ? j - 1 : max_j;
				n_skip += marked_b ? 1 : (better_b && n_skip > 0 ? -1 : 0);
				const bool done_b = marked_b && n_skip > max_skip;
				const bool mark_b = act_b && !skip_b && !done_b && pb != 0xff;
				if (mark_b) TB(pb) = (uint8_t)i;
				h0 = mark_b && (int)pb == ja ? ((h0 & 0x00ffffffu) | im) : h0;
				h1 = mark_b && (int)pb == jb ? ((h1 & 0x00ffffffu) | im) : h1;
				done = done_a || done_b;
			}
		}
#undef ROWLO
#undef ROWHI
		const int32_t vmax = max_j >= 0 ? (int32_t)VL(max_j) : 0;
		ROW(i) = (ri & 0x00000000ffffffffULL) | (uint64_t)((uint32_t)max_f & 0xffffu) << 32 | (uint64_t)(max_j < 0 ? 0xffu : (uint32_t)max_j) << 48 | (ROW(i) & 0xff00000000000000ULL);
		VL(i) = max_j >= 0 && vmax > max_f ? (int16_t)vmax : (int16_t)max_f;
		if (max_j >= 0) BIT_SET(succ, max_j);
	}
	CL_PROF(2);
#define FL(j) R_F(ROW(j))
#define PLv(j) R_P(ROW(j))
	// chain.c:87-109.  NB: the t[] marks above use the row index i (< CAPL <= 128) with 0xff as "never"; from here t[] is a 0/1 flag.
	// From here the rows' low halves are dead too: bytes 0-2 of row c = chain end c (peak f << 8 | peak anchor), later chain c (score << 8 | anchors).
#define UT(c) (ROW32(c) & 0xffffffu)
#define SET_UT(c, v) (ROW32(c) = (ROW32(c) & 0xff000000u) | ((uint32_t)(v) & 0xffffffu))
	int32_t n_u = 0, n_v = 0, k = 0;
	for (int i = 0; i < n; ++i)
		if (!BIT_GET(succ, i) && (int32_t)VL(i) >= min_sc) {
			int j = i;
			while (j >= 0 && FL(j) < (int32_t)VL(j)) j = PLv(j) == 0xff ? -1 : (int)PLv(j);
			if (j < 0) j = i;
			const uint32_t key = (uint32_t)FL(j) << 8 | (uint32_t)j;            // (n_u <= i / 2: rows in front of i, whose low halves nothing reads any more)
			SET_UT(n_u, key); ++n_u;
		}
	if (n_u == 0) CHAIN_EXIT0();
	if (alive) {
	for (int32_t i = 1; i < n_u; ++i) { const uint32_t t = UT(i); int32_t j = i; while (j > 0 && UT(j - 1) < t) { SET_UT(j, UT(j - 1)); --j; } SET_UT(j, t); }
	uint64_t *const okf = seg.okey ? seg.okey + aoff : nullptr, *const okp = okf ? okf + (n + 1) / 2 : nullptr;
	for (int32_t i = 0; i < n_u; ++i) {                                              // chain.c:111-128; v[] reused as the visit list
		const uint32_t key0 = UT(i);
		const int32_t n_v0 = n_v, k0 = k, sc0 = (int32_t)(key0 >> 8); int j = (int32_t)(key0 & 0xffu);
		do { VL(n_v) = (int16_t)j; ++n_v; BIT_SET(vis, j); const uint32_t pj_ = PLv(j); j = pj_ == 0xff ? -1 : (int)pj_; } while (j >= 0 && !BIT_GET(vis, j));
		if (j < 0) { if (n_v - n_v0 >= min_cnt) { SET_UT(k, (uint32_t)sc0 << 8 | (uint32_t)(n_v - n_v0)); ++k; } }
		else if (sc0 - FL(j) >= min_sc) { if (n_v - n_v0 >= min_cnt) { SET_UT(k, (uint32_t)(sc0 - FL(j)) << 8 | (uint32_t)(n_v - n_v0)); ++k; } }
		if (k0 == k) n_v = n_v0; else if (okp) okp[k0] = (uint64_t)(uint32_t)sc0 << 32 | (uint64_t)(key0 & 0xffu);
	}
	n_u = k;
	// chains ordered by the x of their first anchor (chain.c:144-160); n_u <= 64: stable insertion sort (ksort.h:149).
	// The visit list v[] keeps each chain's anchors last-to-first; t[] (free now) = permutation, p-bytes = start offsets.
	// The anchors are sorted by x, so the first anchors compare as their indices do unless every anchor between them repeats its neighbour's x.
#define OFFB(c) (reinterpret_cast<uint8_t *>(&srow[(c) * RS + lane])[6])
#define UCNT(c) ((int32_t)(UT(c) & 0xffu))
#define U64OF(c) ((uint64_t)(UT(c) >> 8) << 32 | (uint64_t)(UT(c) & 0xffu))
	int32_t off = 0;
	for (int32_t c = 0; c < n_u; ++c) { OFFB(c) = (uint8_t)off; off += UCNT(c); TB(c) = (uint8_t)c; }
#define CIDX(c) ((int)VL((int)OFFB(c) + UCNT(c) - 1))
	auto same_x = [&](int lo, int hi) -> bool {                                  // anchors lo < hi: equal x <=> every anchor in (lo, hi] has the x of the one in front of it
		bool all = true;
#pragma unroll
		for (int r = 0; r < NR; ++r) {
			const int l = lo + 1 - 64 * r, h = hi - 64 * r;                          // bits [l, h] of word r, clipped
			if (h < 0 || l > 63) continue;
			const int l0 = l < 0 ? 0 : l, h0 = h > 63 ? 63 : h;
			const uint64_t m = (h0 == 63 ? ~0ULL : (1ULL << (h0 + 1)) - 1ULL) & ~((1ULL << l0) - 1ULL);
			all = all && (eqm[r] & m) == m;
		}
		return all;
	};
	bool eqx = false;
	for (int32_t i = 1; i < n_u; ++i) {
		const uint8_t ci = TB(i); const int xi = CIDX(ci); int32_t j = i;
		while (j > 0) {
			const uint8_t cj = TB(j - 1); const int xj = CIDX(cj);
			const bool eq = xi == xj || same_x(xi < xj ? xi : xj, xi < xj ? xj : xi);
			if (!eq && xi < xj) { TB(j) = cj; --j; } else { eqx = eqx || eq; break; }
		}
		TB(j) = ci;
	}
#undef CIDX
	AlAnchor *const b = chained + aoff;
	uint64_t *u = direct ? u_out + uslot_f : u_out + aoff + (seg.meta ? 0u : f);
	uint32_t *const uo = direct ? uo_out + uslot_f : (!seg.meta && uo_out) ? uo_out + aoff + f : nullptr;
	const uint32_t uo_base = direct ? rel_f : 0u;
	int32_t o = 0; uint64_t u1 = 0;
	const bool one = rec && n_u == 1;                                          // a segment with one chain (most of them): its list entry travels in the record
	for (int32_t i = 0; i < n_u; ++i) {
		const int32_t c = TB(i), ni = UCNT(c), k0 = OFFB(c);
		if (one) u1 = U64OF(c); else u[i] = U64OF(c);
		if (uo) uo[i] = uo_base + (uint32_t)o;
		if (okf) okf[i] = okp[c];
		for (int32_t j = 0; j < ni; ++j) { SRCB(o) = (uint8_t)VL(k0 + (ni - j - 1)); ++o; }   // source of chained anchor o (the visit list holds a chain last to first) in byte 3 of row o: bytes 0-2 may still be a chain's entry
	}
	for (int32_t j0 = 0; j0 < o; j0 += 8) {                                      // the chained anchors: eight loads in flight, then their stores
		AlAnchor v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) v[u] = a[(int)SRCB(j0 + u < o ? j0 + u : o - 1)];
#pragma unroll
		for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y));
#pragma unroll
		for (int u = 0; u < 8; ++u) if (j0 + u < o) b[j0 + u] = v[u];
	}
	if (rec) *reinterpret_cast<uint4 *>(rec) = make_uint4((uint32_t)u1, (uint32_t)(u1 >> 32), (uint32_t)n_u, (uint32_t)o | (eqx ? 1u << 31 : 0u));
	else if (direct) { if (eqx) seg.ctie[fragid_f] = 1u; }
	else frag_nu[f] = (uint32_t)n_u;
	}
	}
	CL_PROF(3);
	CL_PROF(4);
	if (prof && lane == 0) atomicAdd(&counters[29], 1ULL);
#undef BIT_SET
#undef BIT_GET
#undef CL_PROF
#undef CHAIN_EXIT0
#undef OFFB
#undef UCNT
#undef U64OF
#undef UT
#undef SET_UT
#undef FL
#undef PLv
#undef ROW
#undef ROW32
#undef VL
#undef TB
#undef SRCB
#undef R_XLO
#undef R_Q
#undef R_SEG
#undef R_FAR
#undef R_F
#undef R_P
#undef R_T
}
#define INST_CHAIN_LDS(C, L) template __global__ void k_chain_lds<C, L>(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, AlAnchor *, uint64_t *, uint32_t *, uint64_t *, const uint32_t *, int, int, AlParams, unsigned long long *, ChainSeg, uint32_t *, int);
INST_CHAIN_LDS(16, 64) INST_CHAIN_LDS(24, 64) INST_CHAIN_LDS(32, 64) INST_CHAIN_LDS(40, 64) INST_CHAIN_LDS(48, 64) INST_CHAIN_LDS(64, 64) INST_CHAIN_LDS(80, 64) INST_CHAIN_LDS(96, 64) INST_CHAIN_LDS(128, 32)

// explicit instantiations used by the runtime
template __global__ void k_anchor_sort<1024>(const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const AlMatch *, const uint32_t *, const uint32_t *, const uint64_t *, AlAnchor *, uint32_t *, unsigned int *, const uint32_t *, int, unsigned long long *, int);
template __global__ void k_chain<AL_CHAIN_CAP>(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, AlAnchor *, uint64_t *, uint32_t *, int32_t *, uint64_t *, const uint32_t *, int, AlParams, unsigned long long *, ChainSeg);

// =============================================================================================
// Segments (see ChainSeg, al_internal.h).  k_seg_scan: one wavefront per fragment of the list walks the sorted anchors 64 at a
// time and cuts them wherever x[i] > x[i-1] + max_dist_x (the test that moves the window start in chain.c:52 -- once it fails
// between neighbours no later anchor can reach back over the gap).  Segments that cannot hold a chain (fewer than lmin
// anchors) are dropped.  mode 0 counts the fragment's segments, mode 1 writes them at seg_first[entry] in x order.
// =============================================================================================
// size class of a segment = which chaining kernel takes it (CAPL 16, 24, 32, 40, 48, 64, 80, 96, 128, wave).  Class 0 (almost all
// segments of a repeat-rich fragment) is listed by k_seg_scan itself, in memory order; the others are ordered by class (stable: inside a
// class the segments keep their order in memory)
__device__ __forceinline__ uint32_t d_seg_class(uint32_t n) { return n <= 16 ? 0u : n <= 24 ? 1u : n <= 32 ? 2u : n <= 40 ? 3u : n <= 48 ? 4u : n <= 64 ? 5u : n <= 80 ? 6u : n <= 96 ? 7u : n <= 128 ? 8u : 9u; }
// NW wavefronts per fragment: fragments of more than AL_SEGS_BIG anchors are taken by the NW = 8 launch -- every wavefront walks its slice
// of the anchors: first for the last cut inside it (the open segment a later slice starts in begins at the last cut before it), then
// counting, then (mode 1) writing at the block prefix of the counts -- the others by the NW = 1 launch.
#define AL_SEGS_BIG 8192
template <int NW>
__global__ void __launch_bounds__(64 * NW)
k_seg_scan(const AlAnchor *__restrict__ anchors, const uint64_t *__restrict__ a_off, const uint32_t *__restrict__ frag_na,
           const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len, const uint32_t *__restrict__ frag_list, int n_list,
           AlParams P, int lmin, int mode, const uint64_t *__restrict__ seg_first, const uint64_t *__restrict__ seg_first0, uint32_t *__restrict__ seg_cnt, uint32_t *__restrict__ seg_cnt0,
           uint64_t *__restrict__ vs_off, uint32_t *__restrict__ vs_na, uint32_t *__restrict__ vs_meta, const uint32_t *__restrict__ tie_flag,
           uint32_t *__restrict__ list0 /* mode 1: the segments of class 0 (<= 16 anchors), in memory order */, uint32_t *__restrict__ list1, uint32_t *__restrict__ cls1 /* the others and their classes */,
           int big_from /* NW == 1: the NW = 8 launch covers the list entries from this one on */, int big_n /* anchors above which the NW = 8 launch takes a fragment */)
{
	__shared__ long long s_last[NW];
	__shared__ uint32_t s_cnt[NW], s_cnt0[NW];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if ((int)blockIdx.x >= n_list) return;
	const uint32_t f = frag_list[blockIdx.x];
	const int64_t n = tie_flag && tie_flag[f] ? 0 : frag_na[f];                // equal-x anchors: chained whole on the side stream
	if (NW > 1 ? n <= big_n : (n > big_n && (int)blockIdx.x >= big_from)) return;   // the other instantiation's fragment
	const uint32_t r0 = frag_first[f], r1 = frag_first[f + 1];
	int qlen_sum = 0; for (uint32_t r = r0; r < r1; ++r) qlen_sum += (int)rd_len[r];
	int max_dist_x;                                                            // map.c:341-351
	if (P.max_gap_ref > 0) max_dist_x = P.max_gap_ref;
	else if (P.max_frag_len > 0) { max_dist_x = P.max_frag_len - qlen_sum; if (max_dist_x < P.max_gap) max_dist_x = P.max_gap; }
	else max_dist_x = P.max_gap;
	const uint32_t meta = (uint32_t)qlen_sum | (r1 - r0 > 1 ? 1u << 31 : 0u);
	const uint64_t base_off = a_off[f];
	const AlAnchor *a = anchors + base_off;
	const uint64_t out = mode ? seg_first[blockIdx.x] : 0, out0 = mode ? seg_first0[blockIdx.x] : 0;   // first segment / first class-0 segment of this fragment
	const unsigned long long below = (1ULL << lane) - 1ULL;
	// a segment's record index k; class 0 goes to list0 at its running number among the class-0 segments, the rest to list1 / cls1
	auto emit = [&](uint64_t k, uint64_t k0, uint64_t k1, int64_t start, uint32_t len) {
		vs_off[k] = base_off + (uint64_t)start; vs_na[k] = len; vs_meta[k] = meta;
		const uint32_t cl = d_seg_class(len);
		if (cl == 0) list0[k0] = (uint32_t)k; else { list1[k1] = (uint32_t)k; cls1[k1] = cl; }
	};
	// this wavefront's slice of the anchors (whole windows of 64)
	int64_t i0 = 0, i1 = n;
	if (NW > 1) { const int64_t per = ((n + NW - 1) / NW + 63) / 64 * 64; i0 = (int64_t)w * per; if (i0 > n) i0 = n; i1 = i0 + per < n ? i0 + per : n; }
	const uint64_t x_before = i0 > 0 && i0 < n ? a[i0 - 1].x : 0;                // x of the anchor in front of the slice
	// the walk over [i0, i1): what == 0 finds the last cut, 1 counts, 2 writes (cnt / cnt0 then hold the slice's offsets); the segment
	// that is open at i0 starts at open_in
	auto walk = [&](int what, int64_t open_in, uint32_t &cnt, uint32_t &cnt0, int64_t &open_out) {
		int64_t open_start = open_in; uint64_t prev_last = x_before;
		for (int64_t base = i0; base < i1; base += 64) {
			const int64_t i = base + lane; const bool valid = i < i1;
			const uint64_t x = valid ? a[i].x : 0;
			uint64_t xp = (uint64_t)(uint32_t)__shfl_up((int)(uint32_t)x, 1) | (uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(x >> 32), 1) << 32;
			if (lane == 0) xp = prev_last;
			const bool brk = valid && (i == 0 || x > xp + (uint64_t)max_dist_x);
			const unsigned long long mask = __ballot(brk);
			if (what != 0) {
				bool useful = false; int64_t start = 0;
				if (brk && i > 0) {                                                 // this anchor closes the segment before it
					const unsigned long long lower = mask & below;
					start = lower ? base + (63 - __clzll((long long)lower)) : open_start;
					useful = i - start >= lmin;
				}
				const bool small = useful && d_seg_class((uint32_t)(i - start)) == 0;
				const unsigned long long um = __ballot(useful), um0 = __ballot(small);
				if (what == 2 && useful) {
					const uint32_t o = cnt + (uint32_t)__popcll(um & below), o0 = cnt0 + (uint32_t)__popcll(um0 & below);
					emit(out + o, out0 + o0, (out - out0) + (o - o0), start, (uint32_t)(i - start));
				}
				cnt += (uint32_t)__popcll(um); cnt0 += (uint32_t)__popcll(um0);
			}
			if (mask) open_start = base + (63 - __clzll((long long)mask));
			prev_last = (uint64_t)(uint32_t)__shfl((int)(uint32_t)x, 63) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)(x >> 32), 63) << 32;
		}
		open_out = open_start;
	};
	int64_t open_in = 0;
	if (NW > 1) {
		uint32_t d0 = 0, d1 = 0; int64_t last = -1;
		walk(0, -1, d0, d1, last);
		if (lane == 0) s_last[w] = last;
		__syncthreads();
		for (int j = 0; j < w; ++j) if (s_last[j] >= 0) open_in = s_last[j];
	}
	uint32_t cnt = 0, cnt0 = 0; int64_t open_end = 0;
	const bool last_wave = w == NW - 1;
	if (NW == 1) {
		walk(mode ? 2 : 1, 0, cnt, cnt0, open_end);
	} else {
		walk(1, open_in, cnt, cnt0, open_end);
		// (the fragment's last segment is the last wavefront's: it ends at n whatever the slices are)
		uint32_t tail = 0, tail0 = 0;
		if (last_wave && n > 0 && n - open_end >= lmin) { tail = 1; tail0 = d_seg_class((uint32_t)(n - open_end)) == 0 ? 1u : 0u; }
		if (lane == 0) { s_cnt[w] = cnt + tail; s_cnt0[w] = cnt0 + tail0; }
		__syncthreads();
		if (mode) {
			uint32_t b = 0, b0 = 0; for (int j = 0; j < w; ++j) { b += s_cnt[j]; b0 += s_cnt0[j]; }
			cnt = b; cnt0 = b0;
			walk(2, open_in, cnt, cnt0, open_end);
		} else if (threadIdx.x == 0) {
			uint32_t t = 0, t0 = 0; for (int j = 0; j < NW; ++j) { t += s_cnt[j]; t0 += s_cnt0[j]; }
			seg_cnt[blockIdx.x] = t; seg_cnt0[blockIdx.x] = t0;
		}
		if (!last_wave || !mode) return;
		if (n > 0 && n - open_end >= lmin && lane == 0) emit(out + cnt, out0 + cnt0, (out - out0) + (cnt - cnt0), open_end, (uint32_t)(n - open_end));
		return;
	}
	if (n > 0 && n - open_end >= lmin) {
		if (mode && lane == 0) emit(out + cnt, out0 + cnt0, (out - out0) + (cnt - cnt0), open_end, (uint32_t)(n - open_end));
		++cnt; cnt0 += d_seg_class((uint32_t)(n - open_end)) == 0 ? 1u : 0u;
	}
	if (!mode && lane == 0) { seg_cnt[blockIdx.x] = cnt; seg_cnt0[blockIdx.x] = cnt0; }
}
#define INST_SEG_SCAN(NWV) template __global__ void k_seg_scan<NWV>(const AlAnchor *, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, int, AlParams, int, int, const uint64_t *, const uint64_t *, uint32_t *, uint32_t *, uint64_t *, uint32_t *, uint32_t *, const uint32_t *, uint32_t *, uint32_t *, uint32_t *, int, int);
INST_SEG_SCAN(1) INST_SEG_SCAN(8)
#undef INST_SEG_SCAN

// k_seg_merge: the fragment's chain list from the chain lists of its segments.  chain.c:144-160 orders the chains by the x of
// their first anchor; segments are disjoint x ranges in ascending order, so the fragment's order is the segments' orders
// one after the other -- except that the reference's sort of more than 64 chains is not stable, so a fragment with more than
// 64 chains of which two start at equal x is handed to the whole-fragment kernel (fb_list), which restates that sort.
// NW wavefronts per fragment: fragments with more than AL_SEGM_BIG segments are taken by the NW = 8 launch (each wavefront sums the
// chains / chained anchors of its slice of the segments, a block prefix gives the slices their offsets, then every wavefront copies its
// slice), the others by the NW = 1 launch -- so that a launch does not wait for one wavefront walking a fragment of 10^5 anchors.
#define AL_SEGM_BIG 1024
template <int NW>
__global__ void __launch_bounds__(64 * NW)
k_seg_merge(const uint32_t *__restrict__ frag_list, int n_list, const uint64_t *__restrict__ seg_first,
            const uint64_t *__restrict__ vs_off, const uint4 *__restrict__ vs_res /* ChainSeg::res */,
            const uint64_t *__restrict__ u_tmp, const AlAnchor *__restrict__ chain_tmp, const uint64_t *__restrict__ a_off,
            uint64_t *__restrict__ u_out, AlAnchor *__restrict__ chained, uint32_t *__restrict__ frag_nu,
            uint32_t *__restrict__ fb_list, uint32_t *__restrict__ fb_cnt, const uint32_t *__restrict__ tie_flag,
            const uint64_t *__restrict__ okey_tmp, uint64_t *__restrict__ okey_out, int big_from /* NW == 1: the NW = 8 launch covers the list entries from this one on */, int big_n /* segments above which the NW = 8 launch takes a fragment */)
{
	__shared__ uint32_t s_bu_[NW][64], s_bc_[NW][64];
	__shared__ uint64_t s_off_[NW][64], s_u1_[NW][64];
	__shared__ uint64_t s_sum_u[NW], s_sum_c[NW];
	__shared__ int s_tie_any;
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if ((int)blockIdx.x >= n_list) return;
	const uint32_t f = frag_list[blockIdx.x];
	if (tie_flag && tie_flag[f]) return;                                       // the side stream writes this fragment's chains
	const uint64_t s0 = seg_first[blockIdx.x], s1 = seg_first[blockIdx.x + 1];
	if (NW > 1 ? s1 - s0 <= (uint64_t)big_n : (s1 - s0 > (uint64_t)big_n && (int)blockIdx.x >= big_from)) return;   // the other instantiation's fragment
	uint32_t *const s_bu = s_bu_[w], *const s_bc = s_bc_[w]; uint64_t *const s_off = s_off_[w], *const s_u1 = s_u1_[w];
	// this wavefront's slice of the segments (whole groups of 64)
	uint64_t w0 = s0, w1 = s1;
	if (NW > 1) { const uint64_t per = ((s1 - s0 + NW - 1) / NW + 63) / 64 * 64; w0 = s0 + (uint64_t)w * per; if (w0 > s1) w0 = s1; w1 = w0 + per < s1 ? w0 + per : s1; }
	uint64_t run_u = 0, run_c = 0; bool tie = false;
	if (NW > 1) {
		uint64_t su = 0, sc = 0;
		for (uint64_t s = w0 + lane; s < w1; s += 64) { const uint4 r = vs_res[s]; su += r.z; sc += r.w & 0x7fffffffu; }
		for (int d = 32; d > 0; d >>= 1) { su += __shfl_xor(su, d); sc += __shfl_xor(sc, d); }
		if (threadIdx.x == 0) s_tie_any = 0;
		if (lane == 0) { s_sum_u[w] = su; s_sum_c[w] = sc; }
		__syncthreads();
		for (int i = 0; i < w; ++i) { run_u += s_sum_u[i]; run_c += s_sum_c[i]; }
	}
	uint64_t *u = u_out + a_off[f] + f; AlAnchor *b = chained + a_off[f];
	for (uint64_t sb = w0; sb < w1; sb += 64) {
		const uint64_t s = sb + lane; const bool valid = s < w1;
		const uint4 r = valid ? vs_res[s] : make_uint4(0u, 0u, 0u, 0u);
		const uint32_t nu = r.z, nc = r.w & 0x7fffffffu;
		tie = tie || (r.w >> 31) != 0;
		uint32_t iu = nu, ic = nc;
		for (int d = 1; d < 64; d <<= 1) { const uint32_t tu = __shfl_up(iu, d), tc = __shfl_up(ic, d); if (lane >= d) { iu += tu; ic += tc; } }
		const uint32_t tot_u = __shfl(iu, 63), tot_c = __shfl(ic, 63);
		__threadfence_block();                                                   // (one wavefront per slice: its LDS rows of the last round are done with)
		s_bu[lane] = iu - nu; s_bc[lane] = ic - nc; s_off[lane] = valid ? vs_off[s] : 0; s_u1[lane] = nu == 1 ? ((uint64_t)r.x | (uint64_t)r.y << 32) : 0;
		__threadfence_block();
		for (uint32_t t = lane; t < tot_u; t += 64) {
			int lo = 0, hi = 64; while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_bu[mid] <= t) lo = mid; else hi = mid; }
			const uint64_t u1 = s_u1[lo];
			u[run_u + t] = u1 ? u1 : u_tmp[s_off[lo] + (t - s_bu[lo])];
			if (okey_tmp) {   // processing key with the peak anchor's index made fragment-wide
				const uint64_t k = okey_tmp[s_off[lo] + (t - s_bu[lo])];
				okey_out[a_off[f] + run_u + t] = (k & 0xffffffff00000000ULL) | (uint32_t)((uint32_t)k + (uint32_t)(s_off[lo] - a_off[f]));
			}
		}
		for (uint32_t t = lane; t < tot_c; t += 64) {
			int lo = 0, hi = 64; while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_bc[mid] <= t) lo = mid; else hi = mid; }
			b[run_c + t] = chain_tmp[s_off[lo] + (t - s_bc[lo])];
		}
		run_u += tot_u; run_c += tot_c;
	}
	bool any_tie = __ballot(tie) != 0;
	if (NW > 1) {
		if (any_tie && lane == 0) s_tie_any = 1;
		__syncthreads();
		any_tie = s_tie_any != 0;
		if (w != NW - 1) return;                                                // the last slice ends at the fragment's totals
	}
	if (lane == 0) {
		frag_nu[f] = (uint32_t)run_u;
		if (any_tie && run_u > 64) fb_list[atomicAdd(fb_cnt, 1u)] = f;
	}
}
template __global__ void k_seg_merge<1>(const uint32_t *, int, const uint64_t *, const uint64_t *, const uint4 *, const uint64_t *, const AlAnchor *, const uint64_t *, uint64_t *, AlAnchor *, uint32_t *, uint32_t *, uint32_t *, const uint32_t *, const uint64_t *, uint64_t *, int, int);
template __global__ void k_seg_merge<8>(const uint32_t *, int, const uint64_t *, const uint64_t *, const uint4 *, const uint64_t *, const AlAnchor *, const uint64_t *, uint64_t *, AlAnchor *, uint32_t *, uint32_t *, uint32_t *, const uint32_t *, const uint64_t *, uint64_t *, int, int);

// k_chain_order: the reference's order of a fragment's chains when some of them start at anchors of equal x and there are more than
// 64 of them -- its sort (radix_sort_128x, ksort.h:116-151) is not stable, so the order among the equal ones depends on where every
// chain stood before the sort: in the order the chain ends were processed (peak score, peak anchor: chain.c:111-114).
// k_seg_merge left the chains in (segment, x) order with those processing keys; here a wavefront (1) sorts chain ids by processing key,
// descending, in LDS, (2) loads the x keys in that order, (3) lane 0 restates the radix sort on the (x, id) pairs, (4) the wavefront
// writes chain list and chained anchors in the resulting order.  More chains than the LDS tile: fb2_list (whole-fragment kernel).


// (round 5) the same by a block of NWV wavefronts for the fragments with many chains: the bitonic steps and the copies are spread over the block (a
// 16 000-chain fragment of a re-seeded repeat pair took one wavefront 21 ms of dependent global-memory steps), the two prefix passes and the
// restatement of the reference's sort stay with the first wavefront.
#define CO_SYNC() do { __threadfence_block(); if (NWV > 1) __syncthreads(); } while (0)
template <int NWV>
__global__ void __launch_bounds__(64 * NWV)
k_chain_order_t(const uint32_t *__restrict__ fb_list, int n_fb, const uint64_t *__restrict__ a_off, const uint32_t *__restrict__ frag_nu,
              uint64_t *__restrict__ u_all, AlAnchor *__restrict__ chained, const uint64_t *__restrict__ okey,
              uint64_t *__restrict__ u_tmp, AlAnchor *__restrict__ chain_tmp, uint32_t *__restrict__ fb2_list, uint32_t *__restrict__ fb2_cnt,
              int32_t *__restrict__ ws_i32 /* chaining scratch, 16 bytes per anchor, free here */, int nu_lo /* fragments of nu_lo <= chains < nu_hi */, int nu_hi)
{
	extern __shared__ __align__(16) unsigned char s_raw[];
	__shared__ uint16_t s_rs[AL_RS_SCRATCH / 2];
	const int lane = threadIdx.x & 63, tid = threadIdx.x, wv = threadIdx.x >> 6; constexpr int NT = 64 * NWV;
	if ((int)blockIdx.x >= n_fb) return;
	const uint32_t f = fb_list[blockIdx.x];
	const int n_u = (int)frag_nu[f];
	if (n_u < nu_lo || n_u >= nu_hi) return;                                 // (the other instance's)
	if (n_u > 65535) { if (tid == 0) fb2_list[atomicAdd(fb2_cnt, 1u)] = f; return; }   // beyond the 16-bit chain ids of the sort restatement
	int npow2 = 1; while (npow2 < n_u) npow2 <<= 1;
	// keys, offsets and ids in LDS for up to AL_ORD_CAP chains; more (a read pair inside a high-copy family, re-seeded with max_occ: tens
	// of thousands of chains) use the fragment's range of the chaining scratch: a chain has >= 2 anchors, so 10 npow2 + 4 n_u < 12 n bytes
	// fit its 16 n (slower per step, and still milliseconds against seconds for chaining 3 * 10^5 anchors again as one problem)
	uint64_t *key; uint32_t *off; uint16_t *id;
	if (n_u <= AL_ORD_CAP) { key = (uint64_t *)s_raw; off = (uint32_t *)(key + AL_ORD_CAP); id = (uint16_t *)(off + AL_ORD_CAP); }
	else { key = (uint64_t *)(ws_i32 + 4 * a_off[f]); off = (uint32_t *)(key + npow2); id = (uint16_t *)(off + n_u); }
	uint64_t *u = u_all + a_off[f] + f; AlAnchor *b = chained + a_off[f];
	const uint64_t *ok = okey + a_off[f];
	// offsets of the chains' anchors in the merged order (running sum of the counts)
	if (wv == 0) {
		uint32_t run = 0;
		for (int c0 = 0; c0 < n_u; c0 += 64) {
			const int c = c0 + lane; const uint32_t cnt = c < n_u ? (uint32_t)u[c] : 0u;
			uint32_t incl = cnt; for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
			if (c < n_u) off[c] = run + incl - cnt;
			run += __shfl(incl, 63);
		}
	}
	for (int c = tid; c < npow2; c += NT) { key[c] = c < n_u ? ok[c] : 0; id[c] = (uint16_t)c; }
	CO_SYNC();
	for (int kk = 2; kk <= npow2; kk <<= 1)                                  // (1) descending by processing key (keys are distinct: distinct peak anchors)
		for (int j = kk >> 1; j > 0; j >>= 1) {
			for (int i = tid; i < npow2; i += NT) {
				const int ixj = i ^ j;
				if (ixj > i) { const uint64_t x = key[i], y = key[ixj]; if ((x < y) == ((i & kk) == 0)) { key[i] = y; key[ixj] = x; const uint16_t t = id[i]; id[i] = id[ixj]; id[ixj] = t; } }
			}
			CO_SYNC();
		}
	if (NWV > 1 && n_u > AL_ORD_CAP && n_u <= AL_ORD_CAP2) {                  // sorted in the scratch; (x, id) pairs fit the tile without the offsets: the serial
		uint64_t *key2 = (uint64_t *)s_raw; uint16_t *id2 = (uint16_t *)(key2 + AL_ORD_CAP2);   // permutation of (3) then takes LDS steps instead of global-memory ones
		for (int i = tid; i < n_u; i += NT) { const uint16_t c = id[i]; id2[i] = c; key2[i] = b[off[c]].x; }
		key = key2; id = id2;
	} else
	for (int i = tid; i < n_u; i += NT) key[i] = b[off[id[i]]].x;           // (2) x of the chain's first anchor
	CO_SYNC();
	if (wv == 0) {                                                            // (3) by the first wavefront: counts and small buckets in parallel, the permutation by lane 0
		struct KI { uint64_t k; uint16_t i; };
		struct { typedef KI E; uint64_t *k; uint16_t *i;
		         __device__ __forceinline__ uint64_t keyof(const KI &e) const { return e.k; }
		         __device__ __forceinline__ uint64_t key(int j) const { return k[j]; }
		         __device__ __forceinline__ KI get(int j) const { return KI{k[j], i[j]}; }
		         __device__ __forceinline__ void set(int j, const KI &e) { k[j] = e.k; i[j] = e.i; } } acc{key, id};
		(void)d_rs_sort_wave(acc, n_u, s_rs, lane);
	}
	CO_SYNC();
	// (4) new chain list and anchors into the fragment's scratch ranges, then back
	uint64_t *ut = u_tmp + a_off[f]; AlAnchor *bt = chain_tmp + a_off[f];
	__shared__ uint32_t s_run;
	if (wv == 0) {
		uint32_t run = 0;
		for (int c0 = 0; c0 < n_u; c0 += 64) {
			const int i = c0 + lane; const int c = i < n_u ? (int)id[i] : 0; const uint64_t uc = i < n_u ? u[c] : 0; const uint32_t cnt = (uint32_t)uc;
			uint32_t incl = cnt; for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
			if (i < n_u) { ut[i] = uc; key[i] = (uint64_t)(run + incl - cnt) << 32 | off[c]; }   // (the sorted keys are done with: new and old offset)
			run += __shfl(incl, 63);
		}
		if (lane == 0) s_run = run;
	}
	CO_SYNC();
	for (int i = tid; i < n_u; i += NT) {
		const uint32_t cnt = (uint32_t)ut[i], o = (uint32_t)(key[i] >> 32), so = (uint32_t)key[i];
		for (uint32_t j = 0; j < cnt; ++j) bt[o + j] = b[so + j];
	}
	CO_SYNC();
	{
		const uint32_t run = s_run;
		for (int i = tid; i < n_u; i += NT) u[i] = ut[i];
		for (uint32_t t = tid; t < run; t += NT) b[t] = bt[t];
	}
}
#undef CO_SYNC
template __global__ void k_chain_order_t<1>(const uint32_t *, int, const uint64_t *, const uint32_t *, uint64_t *, AlAnchor *, const uint64_t *, uint64_t *, AlAnchor *, uint32_t *, uint32_t *, int32_t *, int, int);
template __global__ void k_chain_order_t<16>(const uint32_t *, int, const uint64_t *, const uint32_t *, uint64_t *, AlAnchor *, const uint64_t *, uint64_t *, AlAnchor *, uint32_t *, uint32_t *, int32_t *, int, int);

// list entries whose fragment is flagged, appended to out block by block: inside a block of 256 entries the order of the list is kept
// (the list is ordered by anchor count: the lanes of a wavefront that walks `out` get fragments of similar size)
__global__ void __launch_bounds__(256)
k_collect_flagged_blk(const uint32_t *__restrict__ list, int n, const uint32_t *__restrict__ flag, uint32_t *__restrict__ out, uint32_t *__restrict__ cnt)
{
	__shared__ uint32_t s_w[4], s_base;
	const int t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint32_t f = t < n ? list[t] : 0u;
	const bool on = t < n && flag[f] == 1u;                                  // (2 = merged ahead of time, k_spec_mark: not for the merge kernels)
	const unsigned long long m = __ballot(on);
	if (lane == 0) s_w[w] = (uint32_t)__popcll(m);
	__syncthreads();
	if (threadIdx.x == 0) { const uint32_t tot = s_w[0] + s_w[1] + s_w[2] + s_w[3]; s_base = tot ? atomicAdd(cnt, tot) : 0u; }
	__syncthreads();
	if (on) { uint32_t o = s_base; for (int i = 0; i < w; ++i) o += s_w[i]; out[o + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL))] = f; }
}

// list entries whose fragment is flagged, appended to out (any order)
__global__ void __launch_bounds__(256)
k_collect_flagged(const uint32_t *__restrict__ list, int n, const uint32_t *__restrict__ flag, uint32_t *__restrict__ out, uint32_t *__restrict__ cnt)
{
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	const uint32_t f = list[t];
	if (flag[f]) out[atomicAdd(cnt, 1u)] = f;
}

// lower bounds of up to 16 thresholds in an ascending key array (out[k] pre-set to n)
__global__ void __launch_bounds__(256)
k_lower_bounds(const uint32_t *__restrict__ keys, uint32_t n, LbThr T, uint32_t *__restrict__ out)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t k = keys[i], kp = i ? keys[i - 1] : 0u;
	for (int j = 0; j < T.n; ++j) if (k >= T.v[j] && (i == 0 || kp < T.v[j])) out[j] = i;
}

// rechain decision (map.c:353-375): one lane per fragment; appends fragments that must be re-seeded with max_occ
extern "C" __global__ void __launch_bounds__(256)
k_rechain_test(const AlAnchor *__restrict__ chained, const uint64_t *__restrict__ a_off, const uint64_t *__restrict__ u_all, const uint32_t *__restrict__ uo_all,
               const uint32_t *__restrict__ frag_nu, const int32_t *__restrict__ frag_rep, const uint32_t *__restrict__ frag_first,
               int n_frag, uint32_t *__restrict__ list, uint32_t *__restrict__ n_list)
{
	// a lane per fragment; a fragment with more than 64 chains (a thread walked the 12 000 chains of one for a millisecond and a half while the rest of the
	// chip waited) by the whole wavefront, one such fragment after the other: the best chain is the FIRST of the highest score (map.c:357-358)
	const int f = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
	const bool act = f < n_frag && frag_rep[f] > 0;
	const int n_segs = act ? (int)(frag_first[f + 1] - frag_first[f]) : 0;
	const uint32_t n_u = act ? frag_nu[f] : 0u;
	int rechain = 0;
	const bool heavy = act && n_u > 64u;
	if (act && !heavy) {
		if (n_u > 0) {
			const uint64_t *u = u_all + a_off[f] + f; const uint32_t *uo = uo_all + a_off[f] + f; const AlAnchor *a = chained + a_off[f];
			int n_chained_segs = 1, max = 0, max_i = -1, max_off = -1;
			for (uint32_t i = 0; i < n_u; ++i) { if (max < (int)(u[i] >> 32)) max = (int)(u[i] >> 32), max_i = (int)i; }
			if (max_i >= 0) max_off = (int)uo[max_i];
			if (max_i >= 0) {
				for (int i = 1; i < (int32_t)(uint32_t)u[max_i]; ++i)
					if ((a[max_off + i].y & AL_SEED_SEG_MASK) != (a[max_off + i - 1].y & AL_SEED_SEG_MASK)) ++n_chained_segs;
				if (n_chained_segs < n_segs) rechain = 1;
			}
		} else rechain = 1;
	}
	for (unsigned long long hm = __ballot(heavy); hm; hm &= hm - 1) {
		const int src = __ffsll((long long)hm) - 1;
		const int ff = __shfl(f, src), nsg = __shfl(n_segs, src); const uint32_t nu = (uint32_t)__shfl((int)n_u, src);
		const uint64_t *u = u_all + a_off[ff] + ff; const uint32_t *uo = uo_all + a_off[ff] + ff; const AlAnchor *a = chained + a_off[ff];
		int bs = 0, bi = -1;
		for (uint32_t i = (uint32_t)lane; i < nu; i += 64u) { const int sc = (int)(u[i] >> 32); if (sc > bs) { bs = sc; bi = (int)i; } }
		for (int d = 32; d > 0; d >>= 1) { const int os = __shfl_xor(bs, d), oi = __shfl_xor(bi, d); if (oi >= 0 && (bi < 0 || os > bs || (os == bs && oi < bi))) { bs = os; bi = oi; } }
		int r = 0;
		if (bi >= 0) {
			const int off = (int)uo[bi], cnt = (int32_t)(uint32_t)u[bi];
			int chg = 0;
			for (int i = 1 + lane; i < cnt; i += 64) chg += (a[off + i].y & AL_SEED_SEG_MASK) != (a[off + i - 1].y & AL_SEED_SEG_MASK) ? 1 : 0;
			for (int d = 32; d > 0; d >>= 1) chg += __shfl_xor(chg, d);
			r = 1 + chg < nsg ? 1 : 0;
		}
		if (lane == src) rechain = r;
	}
	if (rechain) list[atomicAdd(n_list, 1u)] = (uint32_t)f;
}

// ---------------------------------------------------------------------------------------------
// test hook (tests/test_gpu_stages.py): the serial and the wavefront form of the radix-sort restatement on the same keys; both
// permutations come back (they must be equal -- and equal to the reference's ksort.h, which the CPU test of the serial form pins)
// ---------------------------------------------------------------------------------------------
struct DbgKI { uint64_t k; uint16_t i; };
struct DbgRsAcc { typedef DbgKI E; uint64_t *k; uint16_t *i;
	__device__ __forceinline__ uint64_t keyof(const DbgKI &e) const { return e.k; }
	__device__ __forceinline__ uint64_t key(int j) const { return k[j]; }
	__device__ __forceinline__ DbgKI get(int j) const { return DbgKI{k[j], i[j]}; }
	__device__ __forceinline__ void set(int j, const DbgKI &e) { k[j] = e.k; i[j] = e.i; } };
__global__ void __launch_bounds__(64)
k_dbg_rs_sort(uint64_t *ka, uint16_t *ia, uint64_t *kb, uint16_t *ib, int n)
{
	__shared__ uint16_t s_rs[AL_RS_SCRATCH / 2];
	const int lane = threadIdx.x;
	if (blockIdx.x == 0) { if (lane == 0) { DbgRsAcc acc{ka, ia}; (void)d_rs_sort(acc, n, s_rs); } }
	else { DbgRsAcc acc{kb, ib}; (void)d_rs_sort_wave(acc, n, s_rs, lane); }
}
extern "C" int al_dbg_rs_sort(int device, const uint64_t *keys, int n, uint16_t *order_serial, uint16_t *order_wave)
{
	if (n <= 0 || n > 65535 || hipSetDevice(device < 0 ? 0 : device) != hipSuccess) return -1;
	uint64_t *ka = nullptr, *kb = nullptr; uint16_t *ia = nullptr, *ib = nullptr; int rc = -1;
	std::vector<uint16_t> id((size_t)n); for (int i = 0; i < n; ++i) id[i] = (uint16_t)i;
	if (hipMalloc((void **)&ka, (size_t)n * 8) != hipSuccess || hipMalloc((void **)&kb, (size_t)n * 8) != hipSuccess ||
	    hipMalloc((void **)&ia, (size_t)n * 2) != hipSuccess || hipMalloc((void **)&ib, (size_t)n * 2) != hipSuccess) goto done;
	if (hipMemcpy(ka, keys, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(kb, keys, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(ia, id.data(), (size_t)n * 2, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(ib, id.data(), (size_t)n * 2, hipMemcpyHostToDevice) != hipSuccess) goto done;
	hipLaunchKernelGGL(k_dbg_rs_sort, dim3(2), dim3(64), 0, 0, ka, ia, kb, ib, n);
	if (hipDeviceSynchronize() != hipSuccess) goto done;
	if (hipMemcpy(order_serial, ia, (size_t)n * 2, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(order_wave, ib, (size_t)n * 2, hipMemcpyDeviceToHost) != hipSuccess) goto done;
	rc = 0;
done:
	(void)hipFree(ka); (void)hipFree(kb); (void)hipFree(ia); (void)hipFree(ib);
	return rc;
}
