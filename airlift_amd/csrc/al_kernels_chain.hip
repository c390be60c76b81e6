// al_kernels_chain.hip -- K4t: chaining (mm_chain_dp, chain.c:22-162) of fragments with many anchors, ONE pass per fragment.
//
// A fragment's sorted anchors fall apart into independent chaining problems ("segments") wherever two neighbours are further
// apart than max_dist_x: the test that moves the predecessor window in chain.c:52 can never reach back over such a gap.  A read
// pair inside interspersed repeats has thousands of anchors but segments of 2 - 40.  k_chain_tile gives a wavefront a TILE of
// up to CT_TILE anchors -- several small fragments, one fragment, or a slice of a large one that ends at a segment boundary --
// and does everything for it from LDS:
//   load     coalesced 16-byte loads -> compact 8-byte rows [xlo:16 | q:12 | mate:1 | cut:1 | flags:2 | f:16 | p:16]
//   cut      ballots over the cut flags -> segment list (start, length) in tile order, ordered by size class for the next steps
//   DP       segments of <= 16 anchors: a lane each (64 at a time, equal size classes together);
//            longer segments: 16 lanes each, one row at a time, the row's predecessors scored 16 at a time -- the sequential
//            max_skip rule of chain.c:74-81 replayed on two ballot masks (only rows with more than max_skip predecessors need it)
//   back     chain ends, peaks, backtrack, min_cnt / min_sc filters, order by first anchor (chain.c:87-160): a lane per segment
//   emit     chain list entries appended at the fragment's running offset (the segments' x ranges ascend, so tile order is
//            the reference's order by first-anchor x); every chain's anchors are written at the SEGMENT's own place in
//            `chained` -- a chain keeps <= the segment's anchors -- and the entry carries that offset (uo[]): nothing is
//            compacted or merged afterwards, the consumers read (u, uo) pairs.
// Handed to the caller's fallback list (the segment-wise kernels of al_kernels_seed.hip on a compact copy): fragments with a
// segment longer than the tile or with more than CT_NU_CAP chain ends, anchors outside the compact rows, and fragments with
// more than 64 chains of which two start at equal x (the reference's unstable sort decides their order, k_chain_order).
// Integer work on LDS; HBM traffic = the anchors once in, the chained anchors and 12 bytes per chain out.  No MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "al_internal.h"
#include "al_device.h"

#define CT_TILE 1024
#define CT_SEGS (CT_TILE / 2)            // a segment that can hold a chain has >= 2 anchors (the caller checks lmin >= 2)
#define CT_FRAGS 32
#define CT_NU_CAP 64                     // chain ends of one segment the serial sorts take; more: fallback
#define CT_CLIN_N 512
#define CT_NONE 0xffffu

#define R_XLO(r) ((uint32_t)(r) & 0xffffu)
#define R_Q(r) ((int32_t)((uint32_t)(r) >> 16 & 0xfffu))
#define R_SEG(r) ((int32_t)((uint32_t)(r) >> 28 & 1u))
#define R_F(r) ((int32_t)(int16_t)((r) >> 32))
#define R_P(r) ((uint32_t)((r) >> 48))
#define ROW_B30 (1u << 30)
#define ROW_B31 (1u << 31)

struct CtFrag {                 // one fragment of the tile
	uint64_t aoff;              // its anchors in the batch arrays
	uint32_t f, na;
	int32_t mdx, mdy; uint32_t drlim;
	uint32_t start;             // first row of the tile that is this fragment's
	uint32_t flags;             // 1: skipped (chained elsewhere), 2: bad row / segment beyond the kernels -> fallback, 4: tie, 8: has deferred segments
	uint32_t meta;              // ChainSeg::meta of its segments: qlen_sum | paired << 31
};

__device__ __forceinline__ int ct_ilog2(uint32_t v) { return 31 - __clz((int)v); }
__device__ __forceinline__ int ct_dpp_shr(int old, int v, const int n)
{
	switch (n) {
	case 1: return __builtin_amdgcn_update_dpp(old, v, 0x111, 0xf, 0xf, false);
	case 2: return __builtin_amdgcn_update_dpp(old, v, 0x112, 0xf, 0xf, false);
	case 4: return __builtin_amdgcn_update_dpp(old, v, 0x114, 0xf, 0xf, false);
	default: return __builtin_amdgcn_update_dpp(old, v, 0x118, 0xf, 0xf, false);
	}
}

// score of predecessor row rj for row (xi, qi, sidi): chain.c:53-73 with the range tests folded into unsigned compares.
// Returns false when the pair is skipped.
struct CtPen { const uint8_t *same, *diff; bool tab_ok; double avg_d; };
__device__ __forceinline__ bool ct_score(const uint64_t rj, const uint32_t xi, const int32_t qi, const int32_t sidi, const int32_t q_span,
                                         const int32_t mdx, const int32_t mdy, const uint32_t drlim, const int32_t bw, const CtPen &pen, int32_t &sc_out)
{
	const int32_t dr = (int32_t)((xi - R_XLO(rj)) & 0xffffu);
	const int32_t dq = qi - R_Q(rj);
	const bool same = R_SEG(rj) == sidi;
	const int32_t dd = dr > dq ? dr - dq : dq - dr;
	// dq <= 0 or dq > max_dist_x; for anchors of the same mate also dr == 0 or (paired end) dr > max_dist_y, dq > max_dist_y, dd > bw
	const bool skip = ((uint32_t)(dq - 1) >= (uint32_t)mdx) | (same & (((uint32_t)(dr - 1) >= drlim) | (dq > mdy) | (dd > bw)));
	const int32_t min_d = dq < dr ? dq : dr;
	int32_t sc = min_d > q_span ? q_span : min_d;
	const int32_t log_dd = dd ? ct_ilog2((uint32_t)dd) : 0, c_lin = (int)((double)dd * .01 * pen.avg_d);
	int32_t pen_same = c_lin + (log_dd >> 1), pen_diff = c_lin < log_dd ? c_lin : log_dd;
	pen_diff = dr == 0 ? -1 : pen_diff;                                        // other mate, same position: + 1 (chain.c:67)
	sc_out = sc - (same ? pen_same : pen_diff) + R_F(rj);
	return !skip;
}

struct TileSched {
	uint32_t n_items;
	uint32_t ent[7];      // first list entry of class c: c = 0 .. 4: 32, 16, 8, 4, 2 fragments per item (at most 32, 64, ..., 512 anchors each), 5: one; ent[6] = end
	uint32_t item[7];     // first item of class c
};

#define CT_NW 4                          // wavefronts per tile: they share the rows and take the segments of every step in turn
#define CT_NT (64 * CT_NW)
#define CT_INLINE 8                      // segments of up to this many anchors are chained here; longer ones are deferred to the lane kernels
#define CT_DEFER_MAX 128                 // ... whose rows hold up to 128 anchors; a longer segment hands the fragment to the fallback
#define CT_DEF 0x4000u                   // s_snu: the segment is deferred (the count is the number of chain list slots reserved for it)
struct CtMisc { uint32_t n_seg, n_inl, n_def, proc_end, next_pos, more, too_long, def_base; };
struct CtDefer {                         // the deferred segments (ChainSeg direct mode), appended tile by tile
	uint64_t *off, *uslot; uint32_t *na, *meta, *rel, *fragid, *cls; uint32_t *cnt; uint32_t cap;
	uint32_t *cmp_list, *cmp_cnt;        // fragments with deferred segments: their chain lists have gaps until k_u_compact has run
	uint32_t *ctie;                      // per fragment: 1 = two chains start at equal x
};
__device__ __forceinline__ uint32_t ct_defer_class(uint32_t n) { return n <= 16 ? 0u : n <= 24 ? 1u : n <= 32 ? 2u : n <= 40 ? 3u : n <= 48 ? 4u : n <= 64 ? 5u : n <= 80 ? 6u : n <= 96 ? 7u : 8u; }

// Barrier of the tile kernel: the steps hand rows over through LDS only, so a wavefront waits for ITS LDS operations and the barrier -- not, as
// __syncthreads() does on this target, for every global store it has in flight (chain list entries, chained anchors: microseconds each when the
// page is not in the TLB).
__device__ __forceinline__ void ct_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void
ct_tile_body(const AlAnchor *__restrict__ anchors, const uint64_t *__restrict__ a_off, const uint32_t *__restrict__ frag_na,
             const uint32_t *__restrict__ frag_meta /* per fragment: qlen_sum | paired << 31 (k_frag_meta) */, const uint32_t *__restrict__ list, const TileSched &S,
             const uint32_t *__restrict__ skip_flag, AlAnchor *__restrict__ chained, uint64_t *__restrict__ u_out, uint32_t *__restrict__ uo_out,
             uint32_t *__restrict__ frag_nu, uint32_t *__restrict__ fb_list, uint32_t *__restrict__ fb_cnt, const AlParams &P, const int lmin,
             unsigned long long *__restrict__ counters, const int force_fb /* tests: every fragment is handed back */, const CtDefer &D)
{
	// 14 bytes per row + 6.5 per possible segment: 19.6 KB, eight tiles per CU by LDS (six by the compiler's count, which caps the registers at 80)
	__shared__ uint64_t s_row[CT_TILE];
	__shared__ uint16_t s_u[CT_TILE];                                          // chain ends (peak score << 4 | anchor), then chains (score << 4 | anchors): at most CT_INLINE = 8 anchors per segment here
	__shared__ uint16_t s_v[CT_TILE], s_tm[CT_TILE];                           // s_tm: low byte = first visit-list entry of chain c, high byte = chain at sorted position i
	__shared__ uint16_t s_sstart[CT_SEGS], s_snu[CT_SEGS], s_proc[CT_SEGS];
	uint16_t *const s_G = s_proc;                                              // (the size-ordered list is dead when the emit step starts)
	__shared__ uint8_t s_slen[CT_SEGS], s_Dn[CT_SEGS], s_sfrag[CT_SEGS];
	__shared__ CtFrag s_tf[CT_FRAGS];
	__shared__ uint32_t s_fseg[CT_FRAGS + 1];                                  // first segment (tile order) of every fragment of the tile
	__shared__ uint32_t s_ctot[CT_SEGS / 64 + 1], s_dtot[CT_SEGS / 64 + 1];
	__shared__ CtMisc s_misc;
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const unsigned long long below = (1ULL << lane) - 1ULL;
	if (blockIdx.x >= S.n_items) return;
	const bool prof = (P.dbg >> 24) & 1;                                       // timing experiment: cycles per step of the tile loop, summed over the blocks (counters[24 ..])
	long long tp = prof ? clock64() : 0;
#define CT_PROF(i) do { if (prof && tid == 0) { const long long t_ = clock64(); atomicAdd(&counters[24 + (i)], (unsigned long long)(t_ - tp)); tp = t_; } } while (0)
	const uint32_t it = S.n_items - 1u - blockIdx.x;                            // the list ascends by anchor count: the heaviest items first
	int cl = 0; while (cl < 5 && it >= S.item[cl + 1]) ++cl;
	const uint32_t per = 32u >> cl;
	const uint32_t e0 = S.ent[cl] + (it - S.item[cl]) * per, e1 = e0 + per < S.ent[cl + 1] ? e0 + per : S.ent[cl + 1];
	const int nfr = (int)(e1 - e0);
	const int32_t q_span = P.k, bw = P.bw, max_iter = P.max_chain_iter, min_cnt = P.min_cnt, min_sc = P.min_chain_score;
	// gap costs of chain.c:64-72 computed per pair (the segments chained here score too few pairs to repay tables); every span is k (checked per
	// row): avg_qspan = (float)sum / n == k exactly (chain.c:42)
	CtPen pen; pen.same = nullptr; pen.diff = nullptr; pen.avg_d = (double)P.k; pen.tab_ok = false;
	// ---- the fragments of this item ----
	if (w == 0) {
		const bool mine = lane < nfr;
		const uint32_t f = mine ? list[e0 + lane] : 0u;
		uint32_t fl = skip_flag ? skip_flag[f] : 0u, na0 = frag_na[f], mt = frag_meta[f];   // (lanes without an entry read fragment 0: harmless, one latency for all)
		uint64_t ao = a_off[f];
		asm volatile("" : "+v"(fl), "+v"(na0), "+v"(mt), "+v"(ao));
		const bool skipped = fl != 0;
		const uint32_t na = mine && !skipped ? na0 : 0u;
		uint32_t incl = na;                                                     // rows of the fragments in front of this one
		for (int d = 1; d < 32; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
		if (mine) {
			const int qlen_sum = (int)(mt & 0x7fffffffu);
			const int mdy = qlen_sum > P.max_gap ? qlen_sum : P.max_gap;               // map.c:341-351
			int mdx;
			if (P.max_gap_ref > 0) mdx = P.max_gap_ref;
			else if (P.max_frag_len > 0) { mdx = P.max_frag_len - qlen_sum; if (mdx < P.max_gap) mdx = P.max_gap; }
			else mdx = P.max_gap;
			CtFrag t; t.aoff = ao; t.f = f; t.na = na; t.mdx = mdx; t.mdy = mdy; t.drlim = (mt >> 31) ? (uint32_t)mdy : 0x7fffffffu; t.flags = skipped ? 1u : 0u;
			t.meta = mt;
			t.start = incl - na;
			s_tf[lane] = t;
		}
	}
	ct_sync();
	const bool single = nfr == 1;
	if (!single) {                                                              // (the size classes guarantee the fit; a caller's mistake must not be silent)
		const CtFrag &tl = s_tf[nfr - 1];
		if (tl.start + tl.na > CT_TILE) { if (tid == 0) atomicAdd(&counters[7], 1ULL << 48); return; }
	}
 	CT_PROF(0);
	uint32_t *const rlo = reinterpret_cast<uint32_t *>(s_row);                   // rlo[2 t]: static half of row t, rlo[2 t + 1]: f | p << 16
	uint32_t pos = 0;                    // single: rows of the fragment in front of this tile
	uint32_t u_run = 0;                  // single: chain list slots written by the earlier tiles
	bool any_def = false;                // single: a segment of an earlier tile was deferred
	for (;;) {
		// ---- load: rows with their cut flags, by all wavefronts; a thread's four rows (a row and its left neighbour's x each) in one round of eight loads in flight
		//      together (round 6; two rounds of two until then, to stay at 80 registers: it is 72 either way at seven wavefronts per SIMD) ----
		{
			const uint32_t rows = single ? (s_tf[0].na - pos < CT_TILE ? s_tf[0].na - pos : CT_TILE) : s_tf[nfr - 1].start + s_tf[nfr - 1].na;
			static_assert(CT_TILE / CT_NT == 4, "four rows per thread");
			{
				AlAnchor e[4]; uint64_t xp[4]; int fis[4]; uint32_t is[4];
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const uint32_t r = (uint32_t)j * CT_NT + tid;
					int fi = 0;
					if (!single) while (fi + 1 < nfr && s_tf[fi + 1].start <= r) ++fi;
					// (round 6) no branch around the loads: the compiler waits for everything in flight wherever a conditionally loaded value meets its default,
					// which made the four rows' loads four round trips.  A thread without a row reads the last row the tile has (or the fragment's first
					// anchor, readable even when the tile is empty), and stores nothing below.
					const uint32_t rr = r < rows ? r : (rows ? rows - 1u : 0u);
					if (!single && r >= rows) { fi = 0; while (fi + 1 < nfr && s_tf[fi + 1].start <= rr) ++fi; }
					fis[j] = fi;
					const uint32_t i = rows ? rr - s_tf[fi].start : 0u; is[j] = i;
					const AlAnchor *src = anchors + s_tf[fi].aoff + pos;
					e[j] = src[i];
					xp[j] = src[i > 0 ? i - 1 : 0].x;
				}
				// (the compiler would sink every load to its use -- one memory latency per row: the loaded values are pinned here)
				asm volatile("" : "+v"(e[0].x), "+v"(e[0].y), "+v"(e[1].x), "+v"(e[1].y), "+v"(xp[0]), "+v"(xp[1]), "+v"(e[2].x), "+v"(e[2].y), "+v"(e[3].x), "+v"(e[3].y), "+v"(xp[2]), "+v"(xp[3]));
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const uint32_t r = (uint32_t)j * CT_NT + tid;
					if (r < rows) {
						const AlAnchor ee = e[j];
						const bool cut = is[j] == 0 || ee.x - xp[j] > (uint64_t)(int64_t)s_tf[fis[j]].mdx;
						const bool bad = (int32_t)(ee.y >> 32 & 0xff) != q_span || ((ee.y & AL_SEED_SEG_MASK) >> AL_SEED_SEG_SHIFT) > 1 || (uint32_t)ee.y > 0xfffu;
						s_row[r] = (uint64_t)((uint32_t)ee.x & 0xffffu) | (uint64_t)((uint32_t)ee.y & 0xfffu) << 16 | (uint64_t)((ee.y & AL_SEED_SEG_MASK) >> AL_SEED_SEG_SHIFT & 1) << 28
						           | (uint64_t)(cut ? 1u : 0u) << 29 | (uint64_t)CT_NONE << 48;
						if (bad) atomicOr(&s_tf[fis[j]].flags, 2u);
					}
				}
			}
		}
		ct_sync();
		CT_PROF(1);
		// ---- cut: segment list in tile order; the segments chained here ordered by size class (one wavefront walks the cut flags) ----
		if (w == 0) {
			uint32_t n_seg = 0, cnt0 = 0, cnt1 = 0, cnt2 = 0;
			uint32_t proc_end = 0, next_pos = 0; bool more = false, too_long = false;
			for (int fi = 0; fi < nfr; ++fi) {
				const CtFrag tf = s_tf[fi];
				const uint32_t base_t = tf.start;
				const uint32_t n = single ? (tf.na - pos < CT_TILE ? tf.na - pos : CT_TILE) : tf.na;
				if (lane == 0) s_fseg[fi] = n_seg;
				uint32_t open = base_t; bool over = false;
				for (uint32_t b = 0; b < n; b += 64) {
					const uint32_t i = b + lane; const bool valid = i < n;
					const bool cut = valid && (rlo[2 * (base_t + i)] >> 29 & 1u);
					const unsigned long long mask = __ballot(cut);
					bool useful = false; uint32_t start = 0, len = 0;
					if (cut && i > 0) {                                                 // this anchor closes the segment in front of it
						const unsigned long long lower = mask & below;
						start = lower ? base_t + b + (uint32_t)(63 - __clzll((long long)lower)) : open;
						len = base_t + i - start;
						useful = (int)len >= lmin;
					}
					over = over || (useful && len > CT_DEFER_MAX);
					const int sc = len <= 2 ? 0 : len <= 4 ? 1 : len <= CT_INLINE ? 2 : 3;
					const unsigned long long um = __ballot(useful);
					if (useful) { const uint32_t k = n_seg + (uint32_t)__popcll(um & below); s_sstart[k] = (uint16_t)start; s_slen[k] = (uint8_t)(len < 255u ? len : 255u); s_sfrag[k] = (uint8_t)fi; }
					n_seg += (uint32_t)__popcll(um);
					cnt0 += (uint32_t)__popcll(__ballot(useful && sc == 0)); cnt1 += (uint32_t)__popcll(__ballot(useful && sc == 1)); cnt2 += (uint32_t)__popcll(__ballot(useful && sc == 2));
					if (mask) open = base_t + b + (uint32_t)(63 - __clzll((long long)mask));
				}
				const uint32_t end_t = base_t + n;
				if (single && pos + n < tf.na) {                                        // a slice: its open segment belongs to the next tile
					more = true; proc_end = open; next_pos = pos + (open - base_t);
					if (open == base_t) too_long = true;                                // a segment longer than the tile
				} else {
					proc_end = end_t;
					const uint32_t len = end_t - open;
					if (n > 0 && (int)len >= lmin) {
						if (lane == 0) { s_sstart[n_seg] = (uint16_t)open; s_slen[n_seg] = (uint8_t)(len < 255u ? len : 255u); s_sfrag[n_seg] = (uint8_t)fi; }
						++n_seg;
						if (len <= 2) ++cnt0; else if (len <= 4) ++cnt1; else if (len <= CT_INLINE) ++cnt2;
						over = over || len > CT_DEFER_MAX;
					}
				}
				if (__ballot(over) && lane == 0) s_tf[fi].flags |= 2u;                  // beyond the lane kernels' rows: the fallback takes the fragment
			}
			__threadfence_block();
			// the inline segments by size class: s_proc[]
			{
				uint32_t b0 = 0, b1 = cnt0, b2 = cnt0 + cnt1;
				for (uint32_t kb = 0; kb < n_seg; kb += 64) {
					const uint32_t k = kb + lane; const bool v = k < n_seg;
					const uint32_t len = v ? s_slen[k] : 0u;
					const int sc = len <= 2 ? 0 : len <= 4 ? 1 : len <= CT_INLINE ? 2 : 3;
					const unsigned long long m0 = __ballot(v && sc == 0), m1 = __ballot(v && sc == 1), m2 = __ballot(v && sc == 2);
					if (v && sc < 3) {
						const uint32_t o = sc == 0 ? b0 + (uint32_t)__popcll(m0 & below) : sc == 1 ? b1 + (uint32_t)__popcll(m1 & below) : b2 + (uint32_t)__popcll(m2 & below);
						s_proc[o] = (uint16_t)k;
					}
					if (v && sc == 3) s_snu[k] = (uint16_t)(CT_DEF | (len >> 1));       // a chain has >= 2 anchors (lmin >= 2: min_cnt >= 2 or two anchors cannot reach min_chain_score ... the caller checks min_cnt >= 2)
					b0 += (uint32_t)__popcll(m0); b1 += (uint32_t)__popcll(m1); b2 += (uint32_t)__popcll(m2);
				}
			}
			if (lane == 0) {
				s_fseg[nfr] = n_seg;
				CtMisc m; m.n_seg = n_seg; m.n_inl = cnt0 + cnt1 + cnt2; m.n_def = n_seg - (cnt0 + cnt1 + cnt2); m.proc_end = proc_end; m.next_pos = next_pos; m.more = more ? 1u : 0u; m.too_long = too_long ? 1u : 0u; m.def_base = 0;
				s_misc = m;
			}
		}
		ct_sync();
		CT_PROF(2);
		const uint32_t n_seg = s_misc.n_seg, n_inl = s_misc.n_inl, proc_end = s_misc.proc_end, n_def = s_misc.n_def;
		const bool more = s_misc.more != 0;
		if (s_misc.too_long) { if (tid == 0) s_tf[0].flags |= 2u; break; }
		uint32_t def_base_r = 0;
		if (tid == 0 && n_def) def_base_r = atomicAdd(D.cnt, n_def);                      // the tile's deferred segments, one range of the list (the answer is needed at the emit step)
		// ---- DP, a lane per segment (chain.c:46-85; at most CT_INLINE - 1 predecessors: the max_skip rule of chain.c:74-80 cannot fire, the caller
		//      checks max_chain_skip); the wavefronts take every fourth segment of the size-ordered list ----
		// (every wavefront takes a contiguous quarter of the size-ordered list: the two-anchor segments -- more than half of them -- end up in
		//  wavefronts of their own, which take the closed form below and leave the issue slots to the others)
		const uint32_t q_inl = (n_inl + CT_NW - 1) / CT_NW, p_end = (w + 1) * q_inl < n_inl ? (w + 1) * q_inl : n_inl;
		for (uint32_t r0 = w * q_inl; r0 < p_end; r0 += 64) {
			const uint32_t pi = r0 + lane;
			const bool have = pi < p_end;
			const uint32_t k = have ? s_proc[pi] : 0u;
			const int n = have ? (int)s_slen[k] : 0;
			const uint32_t Sg = s_sstart[k];
			const CtFrag &tf = s_tf[s_sfrag[k]];
			const int32_t mdx = tf.mdx, mdy = tf.mdy; const uint32_t drlim = tf.drlim;
			if (!__ballot(n > 2)) {
				// two anchors: one pair to score.  A chain needs the pair (a lone anchor scores k < min_chain_score, or min_cnt >= 2 rejects it -- lmin >= 2):
				// f[1] = sc > k makes anchor 1 the only chain end and its own peak (chain.c:87-109), the backtrack visits 1, 0 (chain.c:111-128)
				if (have) {
					const uint64_t r0w = s_row[Sg], r1w = s_row[Sg + 1];
					int32_t sc;
					const bool ok = ct_score(r0w | (uint64_t)(uint32_t)q_span << 32, R_XLO(r1w), R_Q(r1w), R_SEG(r1w), q_span, mdx, mdy, drlim, bw, pen, sc);   // f[0] = k
					const bool chain = ok && sc > q_span && sc >= min_sc && 2 >= min_cnt && (int32_t)((R_XLO(r1w) - R_XLO(r0w)) & 0xffffu) <= mdx && 1 <= max_iter;
					if (chain) { s_u[Sg] = (uint16_t)((uint32_t)sc << 4 | 2u); s_v[Sg] = 1; s_v[Sg + 1] = 0; s_tm[Sg] = 0; s_snu[k] = 1; }
					else s_snu[k] = 0;
				}
				continue;
			}
			int st = 0; int32_t dist = 0; uint32_t prev_xlo = 0;
			for (int i = 0; i < n; ++i) {
				const uint64_t ri = s_row[Sg + i];
				const uint32_t xi = R_XLO(ri); const int32_t qi = R_Q(ri), sidi = R_SEG(ri);
				if (i > 0) {                                                        // window start (chain.c:52-53): x[i] - x[st] <= max_dist_x, at most max_iter rows
					dist += (int32_t)((xi - prev_xlo) & 0xffffu);
					while (dist > mdx || i - st > max_iter) { ++st; dist -= (int32_t)((R_XLO(s_row[Sg + st]) - R_XLO(s_row[Sg + st - 1])) & 0xffffu); }
				}
				prev_xlo = xi;
				int max_j = -1; int32_t max_f = q_span;
				int jn = i > 0 ? i - 1 : 0;
				uint64_t nrow = s_row[Sg + jn];
				for (int j = i - 1; j >= st; --j) {
					const uint64_t rj = nrow;
					jn = j > 0 ? j - 1 : 0;
					nrow = s_row[Sg + jn];
					int32_t sc;
					const bool ok = ct_score(rj, xi, qi, sidi, q_span, mdx, mdy, drlim, bw, pen, sc);
					const bool better = ok && sc > max_f;
					max_f = better ? sc : max_f; max_j = better ? j : max_j;
				}
				const int32_t vmax = max_j >= 0 ? (int32_t)s_v[Sg + max_j] : 0;
				s_row[Sg + i] = (ri & 0x00000000ffffffffULL) | (uint64_t)((uint32_t)max_f & 0xffffu) << 32 | (uint64_t)(max_j < 0 ? CT_NONE : (uint32_t)max_j) << 48;
				s_v[Sg + i] = (uint16_t)(max_j >= 0 && vmax > max_f ? vmax : max_f);
			}
			// ---- chain ends, peaks, backtrack, order (chain.c:87-160), same lane ----
			if (!have) continue;
			const int fi = (int)s_sfrag[k];
#define FLG(t) rlo[2 * (Sg + (t))]
#define F_(t) ((int32_t)(int16_t)(rlo[2 * (Sg + (t)) + 1] & 0xffffu))
#define P_(t) (rlo[2 * (Sg + (t)) + 1] >> 16)
			for (int i = 0; i < n; ++i) { const uint32_t p = P_(i); if (p != CT_NONE) FLG(p) |= ROW_B30; }     // has a successor
			int32_t n_u = 0;
			for (int i = 0; i < n; ++i)
				if (!(FLG(i) & ROW_B30) && (int32_t)s_v[Sg + i] >= min_sc) {
					int j = i;
					while (j >= 0 && F_(j) < (int32_t)s_v[Sg + j]) { const uint32_t p = P_(j); j = p == CT_NONE ? -1 : (int)p; }
					if (j < 0) j = i;
					s_u[Sg + n_u] = (uint16_t)((uint32_t)F_(j) << 4 | (uint32_t)j);
					++n_u;
				}
			if (n_u == 0) { s_snu[k] = 0; continue; }
			for (int32_t i = 1; i < n_u; ++i) { const uint16_t t = s_u[Sg + i]; int32_t j = i; while (j > 0 && s_u[Sg + j - 1] < t) { s_u[Sg + j] = s_u[Sg + j - 1]; --j; } s_u[Sg + j] = t; }
			int32_t n_v = 0, kk = 0;
			for (int32_t i = 0; i < n_u; ++i) {                                          // chain.c:111-128; v[] becomes the visit list
				const uint32_t key0 = s_u[Sg + i];
				const int32_t n_v0 = n_v, k0 = kk, sc_i = (int32_t)(key0 >> 4); int j = (int)(key0 & 0xfu);
				do { s_v[Sg + n_v] = (uint16_t)j; ++n_v; FLG(j) |= ROW_B31; const uint32_t p = P_(j); j = p == CT_NONE ? -1 : (int)p; } while (j >= 0 && !(FLG(j) & ROW_B31));
				if (j < 0) { if (n_v - n_v0 >= min_cnt) s_u[Sg + kk++] = (uint16_t)((uint32_t)sc_i << 4 | (uint32_t)(n_v - n_v0)); }
				else if (sc_i - F_(j) >= min_sc) { if (n_v - n_v0 >= min_cnt) s_u[Sg + kk++] = (uint16_t)((uint32_t)(sc_i - F_(j)) << 4 | (uint32_t)(n_v - n_v0)); }
				if (k0 == kk) n_v = n_v0;
			}
			n_u = kk;
			// chains by the x of their first anchor (chain.c:144-160): a stable insertion sort is the reference's order for the <= 64 chains of a
			// fragment (ksort.h:149); with more chains in the fragment the order among equal x is the fallback's business (tie flag)
			int32_t off = 0;
			for (int32_t c = 0; c < n_u; ++c) { s_tm[Sg + c] = (uint16_t)((uint32_t)off | (uint32_t)c << 8); off += (int32_t)(s_u[Sg + c] & 0xfu); }
			bool eqx = false;
			if (n_u > 1) {
				const AlAnchor *a = anchors + tf.aoff + pos + (Sg - tf.start);
#define CX(c) (a[(int)s_v[Sg + (int)(s_tm[Sg + (c)] & 0xffu) + (int32_t)(s_u[Sg + (c)] & 0xfu) - 1]].x)
#define PERM(i) (s_tm[Sg + (i)] >> 8)
#define SETPERM(i, c) (s_tm[Sg + (i)] = (uint16_t)((s_tm[Sg + (i)] & 0xffu) | (uint32_t)(c) << 8))
				for (int32_t i = 1; i < n_u; ++i) {
					const uint32_t ci = PERM(i); const uint64_t xi = CX(ci); int32_t j = i;
					while (j > 0) { const uint32_t cj = PERM(j - 1); const uint64_t xj = CX(cj); if (xi < xj) { SETPERM(j, cj); --j; } else { eqx = eqx || xi == xj; break; } }
					SETPERM(j, ci);
				}
#undef CX
#undef PERM
#undef SETPERM
			}
			s_snu[k] = (uint16_t)((uint32_t)n_u | (eqx ? 0x8000u : 0u));
#undef FLG
#undef F_
#undef P_
		}
		ct_sync();
		CT_PROF(3);
		// ---- emit: chain list entries in tile order at the fragment's running offset; chained anchors at the segment's own place ----
		for (uint32_t t = tid; t < proc_end; t += CT_NT) rlo[2 * t + 1] = 0xffffffffu;   // f / p are dead: the half becomes "source row of the chained anchor at this place"
		for (uint32_t c = w; c * 64 < n_seg; c += CT_NW) {                                // list slots / deferred segments in front of every segment: inside its group of 64 ...
			const uint32_t k = c * 64 + lane;
			const uint32_t snu = k < n_seg ? (uint32_t)s_snu[k] : 0u;
			const uint32_t nu = snu & 0x3fffu, df = (snu & CT_DEF) ? 1u : 0u;
			uint32_t incl = nu, incd = df;
			for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d), od = __shfl_up(incd, d); if (lane >= d) { incl += o; incd += od; } }
			if (k < n_seg) { s_G[k] = (uint16_t)(incl - nu); s_Dn[k] = (uint8_t)(incd - df); }
			if (lane == 63) { s_ctot[c] = incl; s_dtot[c] = incd; }
		}
		ct_sync();
		auto G_of = [&](uint32_t k) -> uint32_t {                                        // ... and in the tile (k == n_seg: all of them)
			uint32_t g = 0; const uint32_t cc = k >> 6;
			for (uint32_t c = 0; c < cc; ++c) g += s_ctot[c];
			return k < n_seg ? g + s_G[k] : ((k & 63) ? g + s_ctot[cc] : g);
		};
		auto D_of = [&](uint32_t k) -> uint32_t {
			uint32_t g = 0; const uint32_t cc = k >> 6;
			for (uint32_t c = 0; c < cc; ++c) g += s_dtot[c];
			return k < n_seg ? g + s_Dn[k] : ((k & 63) ? g + s_dtot[cc] : g);
		};
		if (tid == 0) s_misc.def_base = def_base_r;
		ct_sync();
		const uint32_t def_base = s_misc.def_base;
		const bool def_ok = n_def == 0 || def_base + n_def <= D.cap;
		if (!def_ok && tid == 0) atomicAdd(&counters[7], 1ULL << 52);                     // (the list is sized for every possible deferred segment: a caller's mistake must not be silent)
		for (uint32_t c = w; c * 64 < n_seg; c += CT_NW) {
			const uint32_t k = c * 64 + lane;
			if (k >= n_seg) continue;
			const uint32_t snu = s_snu[k], nu = snu & 0x3fffu;
			const int fi = (int)s_sfrag[k];
			const CtFrag &tf = s_tf[fi];
			if ((snu & 0x8000u) && !(snu & CT_DEF)) { atomicOr(&s_tf[fi].flags, 4u); }
			if (!nu) continue;
			const uint32_t w0 = (single ? u_run : 0u) + G_of(k) - G_of(s_fseg[fi]);
			uint64_t *const ub = u_out + tf.aoff + tf.f; uint32_t *const uob = uo_out + tf.aoff + tf.f;
			const uint32_t Sg = s_sstart[k], rel = pos + (Sg - tf.start);
			if (snu & CT_DEF) {                                                          // reserved slots, empty until the lane kernels fill them
				for (uint32_t i = 0; i < nu; ++i) ub[w0 + i] = 0ULL;
				atomicOr(&s_tf[fi].flags, 8u);
				if (def_ok) {                                                            // (a fragment the fallback takes anyway: an empty entry)
					const uint32_t d = def_base + D_of(k), len = (tf.flags & 2u) ? 0u : (uint32_t)s_slen[k];
					D.off[d] = tf.aoff + rel; D.na[d] = len; D.meta[d] = tf.meta; D.uslot[d] = tf.aoff + tf.f + w0; D.rel[d] = rel; D.fragid[d] = tf.f; D.cls[d] = ct_defer_class(len);
				}
				continue;
			}
			uint32_t o = 0;
			for (uint32_t i = 0; i < nu; ++i) {
				const uint32_t cc = s_tm[Sg + i] >> 8, e = s_u[Sg + cc], cnt = e & 0xfu, off = s_tm[Sg + cc] & 0xffu;
				ub[w0 + i] = (uint64_t)(e >> 4) << 32 | cnt; uob[w0 + i] = rel + o;
				for (uint32_t j = 0; j < cnt; ++j) rlo[2 * (Sg + o + j) + 1] = Sg + (uint32_t)s_v[Sg + off + (cnt - 1 - j)];
				o += cnt;
			}
		}
		ct_sync();
		CT_PROF(4);
		{   // the chained anchors, by all lanes: a thread's four places in two rounds of two loads in flight together, then their stores
			const uint32_t t1 = single ? proc_end : s_tf[nfr - 1].start + s_tf[nfr - 1].na;
			for (int h = 0; h < 2; ++h) {
				AlAnchor v[2]; uint64_t di[2]; bool ok[2];
#pragma unroll
				for (int j = 0; j < 2; ++j) {
					const uint32_t t = (uint32_t)(2 * h + j) * CT_NT + tid;
					const uint32_t sr = t < t1 ? rlo[2 * t + 1] : 0xffffffffu;
					ok[j] = sr != 0xffffffffu;
					int fi = 0;
					if (!single) while (fi + 1 < nfr && s_tf[fi + 1].start <= t) ++fi;
					const uint64_t base = s_tf[fi].aoff + pos; const uint32_t st0 = s_tf[fi].start;
					di[j] = base + (t - st0);
					v[j] = anchors[ok[j] ? base + (sr - st0) : s_tf[0].aoff];          // (a place without a chained anchor: any readable anchor, not stored)
				}
				asm volatile("" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[1].x), "+v"(v[1].y));
#pragma unroll
				for (int j = 0; j < 2; ++j) if (ok[j]) chained[di[j]] = v[j];
			}
		}
		if (single) { u_run += G_of(n_seg); any_def = any_def || n_def != 0; }
		else if (w == 0 && lane < nfr) {
			const CtFrag &tf = s_tf[lane];
			if (!(tf.flags & 1u)) {
				const uint32_t run = G_of(s_fseg[lane + 1]) - G_of(s_fseg[lane]);
				frag_nu[tf.f] = run;
				if (tf.flags & 4u) D.ctie[tf.f] = 1u;
				if ((tf.flags & 2u) || force_fb) fb_list[atomicAdd(fb_cnt, 1u)] = tf.f;
				else if (tf.flags & 8u) D.cmp_list[atomicAdd(D.cmp_cnt, 1u)] = tf.f;       // gaps in its list: k_u_compact decides about ties afterwards
				else if ((tf.flags & 4u) && run > 64) fb_list[atomicAdd(fb_cnt, 1u)] = tf.f;
			}
		}
		CT_PROF(5);
		if (prof && tid == 0) atomicAdd(&counters[30], 1ULL);
		if (!more) break;
		pos = s_misc.next_pos;
		ct_sync();
	}
	if (single && tid == 0) {
		const CtFrag &tf = s_tf[0];
		if (!(tf.flags & 1u)) {
			frag_nu[tf.f] = (tf.flags & 2u) ? 0u : u_run;
			if (tf.flags & 4u) D.ctie[tf.f] = 1u;
			if ((tf.flags & 2u) || force_fb) fb_list[atomicAdd(fb_cnt, 1u)] = tf.f;
			else if (any_def) D.cmp_list[atomicAdd(D.cmp_cnt, 1u)] = tf.f;
			else if ((tf.flags & 4u) && u_run > 64) fb_list[atomicAdd(fb_cnt, 1u)] = tf.f;
		}
	}
}

// registers capped at 80: six tiles per CU (the 20 KB of LDS would allow eight; at 64 registers the compiler gives up the cap altogether)
#define CT_ARGS const AlAnchor *__restrict__ anchors, const uint64_t *__restrict__ a_off, const uint32_t *__restrict__ frag_na, const uint32_t *__restrict__ frag_meta, const uint32_t *__restrict__ list, const TileSched S, \
	const uint32_t *__restrict__ skip_flag, AlAnchor *__restrict__ chained, uint64_t *__restrict__ u_out, uint32_t *__restrict__ uo_out, uint32_t *__restrict__ frag_nu, uint32_t *__restrict__ fb_list, uint32_t *__restrict__ fb_cnt, \
	const AlParams P, const int lmin, unsigned long long *__restrict__ counters, const int force_fb, const CtDefer D
#define CT_PASS anchors, a_off, frag_na, frag_meta, list, S, skip_flag, chained, u_out, uo_out, frag_nu, fb_list, fb_cnt, P, lmin, counters, force_fb, D
__global__ void __launch_bounds__(CT_NT) __attribute__((amdgpu_waves_per_eu(7))) k_chain_tile6(CT_ARGS) { ct_tile_body(CT_PASS); }

// The chain lists of the fragments with deferred segments have empty slots (the reserve the lane kernels did not need): closed here, a wavefront per
// fragment, in place and in order.  Then the fragment-wide tie rule: more than 64 chains of which two start at equal x -> fallback list.
__global__ void __launch_bounds__(64)
k_u_compact(const uint32_t *__restrict__ cmp_list, const uint32_t *__restrict__ cmp_cnt, const uint64_t *__restrict__ a_off, uint32_t *__restrict__ frag_nu,
            uint64_t *__restrict__ u_all, uint32_t *__restrict__ uo_all, const uint32_t *__restrict__ ctie, uint32_t *__restrict__ fb_list, uint32_t *__restrict__ fb_cnt)
{
	const int lane = threadIdx.x;
	const uint32_t n = *cmp_cnt;
	for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
		const uint32_t f = cmp_list[e];
		const uint32_t n_slots = frag_nu[f];
		uint64_t *u = u_all + a_off[f] + f; uint32_t *uo = uo_all + a_off[f] + f;
		uint32_t run = 0;
		for (uint32_t c0 = 0; c0 < n_slots; c0 += 64) {
			const uint32_t c = c0 + lane;
			const uint64_t uc = c < n_slots ? u[c] : 0ULL; const uint32_t oc = c < n_slots ? uo[c] : 0u;
			const unsigned long long m = __ballot(uc != 0ULL);
			if (uc != 0ULL) { const uint32_t d = run + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL)); u[d] = uc; uo[d] = oc; }   // d <= c; the wavefront has read this group before any of it is written
			run += (uint32_t)__popcll(m);
		}
		if (lane == 0) {
			frag_nu[f] = run;
			if (ctie[f] && run > 64) fb_list[atomicAdd(fb_cnt, 1u)] = f;
		}
	}
}

// =============================================================================================
// The deferred segments of 17 ... 128 anchors: SIXTEEN LANES PER SEGMENT, four segments per wavefront (ChainSeg direct mode, like the lane kernels:
// chains straight to the fragment's arrays).  The lane-per-segment kernels keep 10 bytes per anchor and LANE in LDS -- 25 KB per wavefront at 40
// anchors, six wavefronts per CU, every one of them a chain of dependent LDS round trips -- and took as long as the tile kernel itself.  Here a
// segment's rows are 18 bytes per anchor for the whole GROUP (3.5 KB per wavefront at 48 anchors: the CU is full), a row scores its predecessors 16
// at a time, and the row maximum is a 16-lane reduction; the sequential max_skip rule of chain.c:74-80 is replayed on two ballot masks for the rows
// that have more than max_skip predecessors.  Chain ends, backtrack and order (chain.c:87-160): lane 0 of the group; the chained anchors leave by
// all sixteen lanes.
// =============================================================================================
template <int CAP>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8)))
k_chain_coop(const AlAnchor *__restrict__ anchors, AlAnchor *__restrict__ chained, uint64_t *__restrict__ u_all, uint32_t *__restrict__ uo_all,
             const uint64_t *__restrict__ vs_off, const uint32_t *__restrict__ vs_na, const uint32_t *__restrict__ vs_meta, const uint64_t *__restrict__ uslot,
             const uint32_t *__restrict__ rel, const uint32_t *__restrict__ fragid, uint32_t *__restrict__ ctie, const uint32_t *__restrict__ list, int n_list,
             const AlParams P, unsigned long long *__restrict__ counters)
{
	__shared__ uint64_t s_row_[4][CAP];
	__shared__ uint32_t s_u_[4][CAP];
	__shared__ uint16_t s_v_[4][CAP], s_tm_[4][CAP], s_perm_[4][CAP];
	const int lane = threadIdx.x, g = lane >> 4, gl = lane & 15, gbase = lane & 48;
	uint64_t *const s_row = s_row_[g]; uint32_t *const s_u = s_u_[g]; uint16_t *const s_v = s_v_[g], *const s_tm = s_tm_[g], *const s_perm = s_perm_[g];
	uint32_t *const rlo = reinterpret_cast<uint32_t *>(s_row);
	const int e = (int)blockIdx.x * 4 + g;
	const bool have = e < n_list;
	const uint32_t d = have ? list[e] : 0u;
	int n = have ? (int)vs_na[d] : 0;
	if (n > CAP) { if (gl == 0) atomicAdd(&counters[7], 1ULL << 56); n = 0; }    // (the caller's classes guarantee the fit)
	const uint64_t off = have ? vs_off[d] : 0ULL;
	const uint32_t mt = have ? vs_meta[d] : 0u;
	const int32_t q_span = P.k, bw = P.bw, max_skip = P.max_chain_skip, max_iter = P.max_chain_iter, min_cnt = P.min_cnt, min_sc = P.min_chain_score;
	const int qlen_sum = (int)(mt & 0x7fffffffu);
	const int32_t mdy = qlen_sum > P.max_gap ? qlen_sum : P.max_gap;                   // map.c:341-351
	int32_t mdx;
	if (P.max_gap_ref > 0) mdx = P.max_gap_ref;
	else if (P.max_frag_len > 0) { mdx = P.max_frag_len - qlen_sum; if (mdx < P.max_gap) mdx = P.max_gap; }
	else mdx = P.max_gap;
	const uint32_t drlim = (mt >> 31) ? (uint32_t)mdy : 0x7fffffffu;
	CtPen pen; pen.same = nullptr; pen.diff = nullptr; pen.avg_d = (double)P.k; pen.tab_ok = false;
	const AlAnchor *a = anchors + off;
	// ---- rows ----
	bool bad = false;
	for (int i = gl; i < n; i += 16) {
		const AlAnchor ee = a[i];
		bad = bad || (int32_t)(ee.y >> 32 & 0xff) != q_span || ((ee.y & AL_SEED_SEG_MASK) >> AL_SEED_SEG_SHIFT) > 1 || (uint32_t)ee.y > 0xfffu;
		s_row[i] = (uint64_t)((uint32_t)ee.x & 0xffffu) | (uint64_t)((uint32_t)ee.y & 0xfffu) << 16 | (uint64_t)((ee.y & AL_SEED_SEG_MASK) >> AL_SEED_SEG_SHIFT & 1) << 28 | (uint64_t)CT_NONE << 48;
		s_tm[i] = (uint16_t)CT_NONE;
	}
	{ const unsigned long long bm = __ballot(bad); if ((uint32_t)(bm >> gbase) & 0xffffu) { if (gl == 0) atomicAdd(&counters[7], 1ULL); n = 0; } }   // not representable in the compact rows (the tile kernel checked: never)
	__threadfence_block();
	// ---- DP (chain.c:46-85) ----
	int nmax = n;
	{ int o = __shfl_xor(nmax, 16); nmax = o > nmax ? o : nmax; o = __shfl_xor(nmax, 32); nmax = o > nmax ? o : nmax; }
	int st = 0; int32_t dist = 0; uint32_t prev_xlo = 0;
	for (int i = 0; i < nmax; ++i) {
		const bool ga = i < n;
		const uint64_t ri = ga ? s_row[i] : 0ULL;
		const uint32_t xi = R_XLO(ri); const int32_t qi = R_Q(ri), sidi = R_SEG(ri);
		if (ga && i > 0) {                                                          // window start (chain.c:52-53)
			dist += (int32_t)((xi - prev_xlo) & 0xffffu);
			while (dist > mdx || i - st > max_iter) { ++st; dist -= (int32_t)((R_XLO(s_row[st]) - R_XLO(s_row[st - 1])) & 0xffffu); }
		}
		prev_xlo = ga ? xi : prev_xlo;
		int32_t max_f = q_span, n_skip = 0; int max_j = -1; bool broke = !ga;
		const bool need_marks = ga && (i - st) > max_skip;                            // fewer predecessors can never count max_skip + 1 skips: no marks, no replay
		if (!__ballot(need_marks)) {
			// the row's maximum is the first predecessor (in processing order) with the highest score: a 16-lane reduction per 16 predecessors
			for (int base = i - 1; ; base -= 16) {
				const bool work = ga && base >= st;
				if (!__ballot(work)) break;
				const int j = base - gl;
				const bool act0 = work && j >= st;
				const uint64_t rj = act0 ? s_row[j] : 0ULL;
				int32_t sc;
				const bool ok = ct_score(rj, xi, qi, sidi, q_span, mdx, mdy, drlim, bw, pen, sc);
				int32_t key = (act0 && ok) ? sc * 16 + (15 - gl) : INT32_MIN;
				{ int o = __builtin_amdgcn_update_dpp(key, key, 0xB1, 0xf, 0xf, false); key = o > key ? o : key; }
				{ int o = __builtin_amdgcn_update_dpp(key, key, 0x4E, 0xf, 0xf, false); key = o > key ? o : key; }
				{ int o = __builtin_amdgcn_update_dpp(key, key, 0x141, 0xf, 0xf, false); key = o > key ? o : key; }
				{ int o = __builtin_amdgcn_update_dpp(key, key, 0x140, 0xf, 0xf, false); key = o > key ? o : key; }
				if (key != INT32_MIN && (key >> 4) > max_f) { max_f = key >> 4; max_j = base - (15 - (key & 15)); }
			}
		} else
		for (int base = i - 1; ; base -= 16) {
			const bool work = !broke && base >= st;
			if (!__ballot(work)) break;
			const int j = base - gl;
			bool act = work && j >= st;
			const uint64_t rj = act ? s_row[j] : 0ULL;
			int32_t sc;
			const bool ok = ct_score(rj, xi, qi, sidi, q_span, mdx, mdy, drlim, bw, pen, sc);
			act = act && ok;
			const uint32_t pj = R_P(rj);
			if (need_marks && act && pj != CT_NONE) s_tm[pj] = (uint16_t)i;           // t[p[j]] = i (chain.c:81); marks of lanes behind the break are never tested
			const int32_t scm = act ? sc : INT32_MIN;
			int32_t inc = scm;                                                          // prefix maximum in processing order (lane order inside the 16-lane row)
			{ int o = ct_dpp_shr(INT32_MIN, inc, 1); inc = o > inc ? o : inc; o = ct_dpp_shr(INT32_MIN, inc, 2); inc = o > inc ? o : inc;
			  o = ct_dpp_shr(INT32_MIN, inc, 4); inc = o > inc ? o : inc; o = ct_dpp_shr(INT32_MIN, inc, 8); inc = o > inc ? o : inc; }
			const int32_t excl = ct_dpp_shr(INT32_MIN, inc, 1);
			const int32_t before = excl > max_f ? excl : max_f;
			const bool upd = act && sc > before;
			__threadfence_block();
			const bool marked = need_marks && act && !upd && s_tm[j] == (uint16_t)i;
			const unsigned long long Uw = __ballot(upd), Kw = __ballot(marked);
			uint32_t U = (uint32_t)(Uw >> gbase) & 0xffffu; const uint32_t K = (uint32_t)(Kw >> gbase) & 0xffffu;
			if (need_marks) {                                                           // chain.c:74-80 replayed in order over the two masks
				uint32_t both = U | K; int brk = 16;
				while (both) {
					const int b = __ffs((int)both) - 1; both &= both - 1;
					if (U >> b & 1) { if (n_skip > 0) --n_skip; }
					else if (++n_skip > max_skip) { brk = b; break; }
				}
				if (brk < 16) { broke = true; U &= (1u << brk) - 1u; }
			}
			const int lastu = U ? 31 - __clz((int)U) : 0;
			const int32_t scl = __shfl(sc, gbase + lastu);
			if (work && U) { max_f = scl; max_j = base - lastu; }
		}
		if (ga && gl == 0) {
			const int32_t vmax = max_j >= 0 ? (int32_t)s_v[max_j] : 0;
			s_row[i] = (ri & 0x00000000ffffffffULL) | (uint64_t)((uint32_t)max_f & 0xffffu) << 32 | (uint64_t)(max_j < 0 ? CT_NONE : (uint32_t)max_j) << 48;
			s_v[i] = (uint16_t)(max_j >= 0 && vmax > max_f ? vmax : max_f);
		}
		__threadfence_block();
	}
	// ---- chain ends, peaks, backtrack, order (chain.c:87-160): lane 0 of the group ----
	if (gl == 0 && n > 0) {
#define FLG(t) rlo[2 * (t)]
#define F_(t) ((int32_t)(int16_t)(rlo[2 * (t) + 1] & 0xffffu))
#define P_(t) (rlo[2 * (t) + 1] >> 16)
		for (int i = 0; i < n; ++i) { const uint32_t p = P_(i); if (p != CT_NONE) FLG(p) |= ROW_B30; }     // has a successor
		int32_t n_u = 0;
		for (int i = 0; i < n; ++i)
			if (!(FLG(i) & ROW_B30) && (int32_t)s_v[i] >= min_sc) {
				int j = i;
				while (j >= 0 && F_(j) < (int32_t)s_v[j]) { const uint32_t p = P_(j); j = p == CT_NONE ? -1 : (int)p; }
				if (j < 0) j = i;
				s_u[n_u++] = (uint32_t)F_(j) << 16 | (uint32_t)j;
			}
		for (int32_t i = 1; i < n_u; ++i) { const uint32_t t = s_u[i]; int32_t j = i; while (j > 0 && s_u[j - 1] < t) { s_u[j] = s_u[j - 1]; --j; } s_u[j] = t; }
		int32_t n_v = 0, kk = 0;
		for (int32_t i = 0; i < n_u; ++i) {                                          // chain.c:111-128; v[] becomes the visit list
			const uint32_t key0 = s_u[i];
			const int32_t n_v0 = n_v, k0 = kk, sc_i = (int32_t)(key0 >> 16); int j = (int)(key0 & 0xffffu);
			do { s_v[n_v] = (uint16_t)j; ++n_v; FLG(j) |= ROW_B31; const uint32_t p = P_(j); j = p == CT_NONE ? -1 : (int)p; } while (j >= 0 && !(FLG(j) & ROW_B31));
			if (j < 0) { if (n_v - n_v0 >= min_cnt) s_u[kk++] = (uint32_t)sc_i << 16 | (uint32_t)(n_v - n_v0); }
			else if (sc_i - F_(j) >= min_sc) { if (n_v - n_v0 >= min_cnt) s_u[kk++] = (uint32_t)(sc_i - F_(j)) << 16 | (uint32_t)(n_v - n_v0); }
			if (k0 == kk) n_v = n_v0;
		}
		n_u = kk;
		// chains by the x of their first anchor (chain.c:144-160): stable insertion sort (ksort.h:149 for the <= 64 chains of a fragment; with more
		// chains in the fragment the order among equal x is the fallback's business: ctie)
		int32_t offv = 0;
		for (int32_t c = 0; c < n_u; ++c) { s_tm[c] = (uint16_t)offv; offv += (int32_t)(s_u[c] & 0xffffu); s_perm[c] = (uint16_t)c; }
		bool eqx = false;
#define CX(c) (a[(int)s_v[(int)s_tm[(c)] + (int32_t)(s_u[(c)] & 0xffffu) - 1]].x)
		for (int32_t i = 1; i < n_u; ++i) {
			const uint16_t ci = s_perm[i]; const uint64_t xi = CX(ci); int32_t j = i;
			while (j > 0) { const uint16_t cj = s_perm[j - 1]; const uint64_t xj = CX(cj); if (xi < xj) { s_perm[j] = cj; --j; } else { eqx = eqx || xi == xj; break; } }
			s_perm[j] = ci;
		}
#undef CX
		for (int t = 0; t < n; ++t) rlo[2 * t + 1] = 0xffffffffu;                      // f / p are dead: "source row of the chained anchor at this place"
		uint64_t *const ub = u_all + uslot[d]; uint32_t *const uob = uo_all + uslot[d]; const uint32_t rl = rel[d];
		uint32_t o = 0;
		for (int32_t i = 0; i < n_u; ++i) {
			const uint32_t cc = s_perm[i], ee = s_u[cc], cnt = ee & 0xffffu, offc = s_tm[cc];
			ub[i] = (uint64_t)(ee >> 16) << 32 | cnt; uob[i] = rl + o;
			for (uint32_t j = 0; j < cnt; ++j) rlo[2 * (o + j) + 1] = (uint32_t)s_v[offc + (cnt - 1 - j)];
			o += cnt;
		}
		if (eqx) ctie[fragid[d]] = 1u;
#undef FLG
#undef F_
#undef P_
	}
	__threadfence_block();
	{   // the chained anchors, by the sixteen lanes
		AlAnchor *const b = chained + off;
		for (int t = gl; t < n; t += 16) { const uint32_t sr = rlo[2 * t + 1]; if (sr != 0xffffffffu) b[t] = a[sr]; }
	}
}
template __global__ void k_chain_coop<48>(const AlAnchor *, AlAnchor *, uint64_t *, uint32_t *, const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const uint32_t *, const uint32_t *, uint32_t *, const uint32_t *, int, const AlParams, unsigned long long *);
template __global__ void k_chain_coop<128>(const AlAnchor *, AlAnchor *, uint64_t *, uint32_t *, const uint64_t *, const uint32_t *, const uint32_t *, const uint64_t *, const uint32_t *, const uint32_t *, uint32_t *, const uint32_t *, int, const AlParams, unsigned long long *);

// uo[] for chain lists whose anchors were written back to back (the whole-fragment kernels and the fallback): running sum of the counts.
// One wavefront per list entry.
__global__ void __launch_bounds__(64)
k_uo_fill(const uint32_t *__restrict__ list, int n_list, const uint64_t *__restrict__ a_off, const uint32_t *__restrict__ frag_nu,
          const uint64_t *__restrict__ u_all, uint32_t *__restrict__ uo_all, const uint32_t *__restrict__ skip_flag)
{
	const int lane = threadIdx.x;
	if ((int)blockIdx.x >= n_list) return;
	const uint32_t f = list ? list[blockIdx.x] : blockIdx.x;
	if (skip_flag && skip_flag[f]) return;
	const uint32_t n_u = frag_nu[f];
	const uint64_t *u = u_all + a_off[f] + f; uint32_t *uo = uo_all + a_off[f] + f;
	uint32_t run = 0;
	for (uint32_t c0 = 0; c0 < n_u; c0 += 64) {
		const uint32_t c = c0 + lane; const uint32_t cnt = c < n_u ? (uint32_t)u[c] : 0u;
		uint32_t incl = cnt; for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
		if (c < n_u) uo[c] = run + incl - cnt;
		run += __shfl(incl, 63);
	}
}

// ---- the fallback's compact copy: fragment v of the list becomes fragment v of a small virtual batch ----
// meta: per v the anchor count and the number of mates (two scans give the virtual a_off / frag_first)
__global__ void __launch_bounds__(256)
k_fb_meta(const uint32_t *__restrict__ fb_list, int n_fb, const uint32_t *__restrict__ frag_na, const uint32_t *__restrict__ frag_first, uint32_t *__restrict__ v_na, uint32_t *__restrict__ v_nseg)
{
	const int v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v > n_fb) return;
	if (v == n_fb) { v_na[v] = 0; v_nseg[v] = 0; return; }
	const uint32_t f = fb_list[v];
	v_na[v] = frag_na[f]; v_nseg[v] = frag_first[f + 1] - frag_first[f];
}
__global__ void __launch_bounds__(256)
k_fb_reads(const uint32_t *__restrict__ fb_list, int n_fb, const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len, const uint64_t *__restrict__ v_first64,
           uint32_t *__restrict__ v_first, uint32_t *__restrict__ v_rd_len, uint32_t *__restrict__ v_order)
{
	const int v = blockIdx.x * blockDim.x + threadIdx.x;
	if (v > n_fb) return;
	v_first[v] = (uint32_t)v_first64[v];
	if (v == n_fb) return;
	v_order[v] = (uint32_t)v;
	const uint32_t f = fb_list[v], r0 = frag_first[f], r1 = frag_first[f + 1];
	for (uint32_t r = r0; r < r1; ++r) v_rd_len[(uint32_t)v_first64[v] + (r - r0)] = rd_len[r];
}
__global__ void __launch_bounds__(64)
k_fb_copy_in(const uint32_t *__restrict__ fb_list, int n_fb, const uint64_t *__restrict__ a_off, const uint32_t *__restrict__ frag_na, const AlAnchor *__restrict__ anchors,
             const uint64_t *__restrict__ v_a_off, AlAnchor *__restrict__ v_anchors)
{
	if ((int)blockIdx.x >= n_fb) return;
	const uint32_t f = fb_list[blockIdx.x]; const uint32_t n = frag_na[f];
	const AlAnchor *s = anchors + a_off[f]; AlAnchor *d = v_anchors + v_a_off[blockIdx.x];
	for (uint32_t i = threadIdx.x; i < n; i += 64) d[i] = s[i];
}
__global__ void __launch_bounds__(64)
k_fb_copy_out(const uint32_t *__restrict__ fb_list, int n_fb, const uint64_t *__restrict__ a_off, const uint64_t *__restrict__ v_a_off, const uint32_t *__restrict__ v_nu,
              const uint64_t *__restrict__ v_u, const AlAnchor *__restrict__ v_chained, uint32_t *__restrict__ frag_nu, uint64_t *__restrict__ u_all, uint32_t *__restrict__ uo_all, AlAnchor *__restrict__ chained)
{
	const int lane = threadIdx.x, v = blockIdx.x;
	if (v >= n_fb) return;
	const uint32_t f = fb_list[v]; const uint32_t n_u = v_nu[v];
	const uint64_t *su = v_u + v_a_off[v] + v; uint64_t *du = u_all + a_off[f] + f; uint32_t *duo = uo_all + a_off[f] + f;
	uint32_t run = 0;
	for (uint32_t c0 = 0; c0 < n_u; c0 += 64) {
		const uint32_t c = c0 + lane; const uint64_t uc = c < n_u ? su[c] : 0ULL; const uint32_t cnt = (uint32_t)uc;
		uint32_t incl = cnt; for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
		if (c < n_u) { du[c] = uc; duo[c] = run + incl - cnt; }
		run += __shfl(incl, 63);
	}
	const AlAnchor *sc = v_chained + v_a_off[v]; AlAnchor *dc = chained + a_off[f];
	for (uint32_t t = lane; t < run; t += 64) dc[t] = sc[t];
	if (lane == 0) frag_nu[f] = n_u;
}

// per fragment: qlen_sum | paired << 31 (what the chaining kernels derive max_dist_x / max_dist_y from, map.c:341-351)
__global__ void __launch_bounds__(256)
k_frag_meta(const uint32_t *__restrict__ frag_first, const uint32_t *__restrict__ rd_len, int n_frag, uint32_t *__restrict__ meta)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= n_frag) return;
	const uint32_t r0 = frag_first[f], r1 = frag_first[f + 1];
	uint32_t q = 0; for (uint32_t r = r0; r < r1; ++r) q += rd_len[r];
	meta[f] = q | (r1 - r0 > 1 ? 1u << 31 : 0u);
}
