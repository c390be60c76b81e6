// al_internal.h -- shared host/device declarations of the MI355X re-alignment path.
// Product code (not the oracle).  Reference citations are to /root/reference/src/minimap2-master_remapping/.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <memory>
#include <utility>
#include <map>
#include <mutex>
#include "../../include/airlift.h"

#define AL_MAX_K 28                        // sketch.c:84.  The packed ring entry of the read sketch (hash (2k bits) << pb | pos << 1 | strand in 64 bits) holds k <= 25 and reads below
                                           // 2^(pb - 1) bases; beyond either, k_sketch keeps two words per slot (al_sketch_wide)
static inline int al_sketch_pos_bits(int k) { const int b = 64 - 2 * k; return b > 22 ? 22 : b; }
static inline bool al_sketch_wide(int k, int max_read_len) { const int pb = al_sketch_pos_bits(k); return k > 25 || max_read_len >= (1 << (pb - 1)); }
#define AL_MAX_READ_LEN 32768               // longest read of the GPU path: state tiles of the long-read extension kernel (al_kernels_align.hip)
#define AL_SEED_TANDEM    (1ULL<<42)       // mmpriv.h:20
#define AL_SEED_SEG_SHIFT 48               // mmpriv.h:23
#define AL_SEED_SEG_MASK  (0xffULL<<AL_SEED_SEG_SHIFT)
#define AL_PARENT_UNSET   (-1)
#define AL_PARENT_TMP_PRI (-2)

// ---------------------------------------------------------------------------------------------
// Index.  Semantics of mm_idx_t (index.c:27-98): minimizer hash -> ascending list of
// (rid<<32 | pos<<1 | strand).  HBM layout (own design):
//   tab[2*slot+0] = hash+1 (0 = empty), tab[2*slot+1] = off<<32 | n   -- 16 B entries, open addressing,
//                   slot = (hash * GOLDEN) >> (64 - tab_bits), linear probing, load <= 0.5
//   pos[off .. off+n) = occurrence list, singletons included (one contiguous array)
//   S4[]             = reference bases, 4 bit/base, 8 bases per uint32 (same code as mm_seq4_get, mmpriv.h:29)
// ---------------------------------------------------------------------------------------------
struct AlSeq { std::string name; uint64_t offset; uint32_t len; };

struct AlDevIndex {           // device pointers (one per GPU)
	uint32_t *S4 = nullptr;
	uint64_t *tab = nullptr;
	uint64_t *pos = nullptr;
	uint64_t *seq_off = nullptr;
	uint32_t *seq_len = nullptr;
	int tab_bits = 0;
	uint32_t n_seq = 0;
};

struct al_idx_s {
	int k = 21, w = 11;
	std::vector<AlSeq> seq;
	std::vector<uint32_t> S4;
	uint64_t tot_len = 0;
	int tab_bits = 4;
	std::vector<uint64_t> tab;
	std::vector<uint64_t> pos;
	uint64_t n_keys = 0, n_pos = 0;
	int built_on = -1;                       // >= 0: built by al_idx_build_device on that GPU; the host arrays above stay empty
	mutable std::mutex dev_mtx;
	mutable std::map<int, AlDevIndex> dev;   // lazily uploaded per device
};

static inline uint64_t al_tab_slot(uint64_t hash, int bits) { return (hash * 0x9E3779B97F4A7C15ULL) >> (64 - bits); }
// A minimizer that occurs once keeps its position in the table entry itself (key word | AL_TAB_SINGLE, value = the position
// word rid<<32|pos<<1|strand), as the reference does for singletons (index.c:229-233): one random HBM access less per anchor.
// Used when contig ids fit 16 bits (the match record carries 48 bits).
#define AL_TAB_SINGLE (1ULL << 63)
#define AL_TAB_SINGLE_MAX_SEQ 65536u

// whole-file parallel FASTA loader (al_fasta.cpp); false: not a plain uncompressed FASTA file, use the block reader
// bytes whose resize() does not write them: the loader's threads touch the pages of a genome's worth of text first (a value-initialising vector zeroes -- and
// faults in -- 3 GB on one thread before the copy starts)
template <class T> struct AlNoInit : std::allocator<T> {
	template <class U> struct rebind { typedef AlNoInit<U> other; };
	AlNoInit() = default; template <class U> AlNoInit(const AlNoInit<U> &) {}
	template <class U, class... A> void construct(U *p, A &&...a) { if constexpr (sizeof...(A) == 0) ::new ((void *)p) U; else ::new ((void *)p) U(std::forward<A>(a)...); }
};
typedef std::vector<char, AlNoInit<char>> AlText;
bool al_fasta_load_parallel(const char *fn, int n_threads, std::vector<AlSeq> &seqs, AlText &ascii);
// host-side helpers (al_index.cpp)
void al_sketch_host(const uint8_t *codes, uint32_t len, int w, int k, uint32_t rid, std::vector<uint64_t> &hash_out, std::vector<uint64_t> &y_out);

// ---------------------------------------------------------------------------------------------
// Device-side record types
// ---------------------------------------------------------------------------------------------
struct AlParams {             // kernel parameter block (copy of the options the device code needs)
	int k, w;
	int seed, bw, max_gap, max_gap_ref, max_frag_len, max_chain_skip, max_chain_iter, min_cnt, min_chain_score;
	float mask_level, pri_ratio, max_clip_ratio;
	int best_n, a, b, q, e, q2, e2, sc_ambi, zdrop, zdrop_inv, end_bonus, min_dp_max;
	int pe_ori, pe_bonus, mid_occ, max_occ;
	int dbg;                  // timing experiments only (AL_DBG env): skips phases, results become wrong
	int dbg2;                 // more of the same (AL_DBG2 env): bit 3 cycle counters of the lane chaining kernels' phases (printed per batch), bit 20 the seed / sort / chain stages only
};

struct AlMatch {              // one query minimizer that passed the occurrence filter (mm_match_t, map.c:82-88)
	uint32_t off_lo;          // offset into pos[] (low 32 bits)
	uint32_t n;               // occurrences
	uint32_t q_pos;           // (pos<<1 | strand) in the concatenated fragment
	uint32_t flags;           // seg_id | is_tandem<<8 | single<<9 | off_hi<<16 (single: off_lo/off_hi hold the position word itself)
};

struct AlAnchor { uint64_t x, y; };   // mm128_t anchor (map.c:176-187)

#define AL_MAX_CIGAR_INLINE 0

struct AlReg {                // device mm_reg1_t (+ mm_extra_t scalars); 112 bytes
	int32_t id, cnt, rid, score;
	int32_t qs, qe, rs, re;
	int32_t parent, subsc, as, mlen;
	int32_t blen, n_sub, score0;
	uint32_t hash;
	uint32_t mapq, flags;     // flags: split(2) | rev<<2 | inv<<3 | sam_pri<<4 | proper<<5 | pe_thru<<6 | seg_split<<7 | seg_id<<8 | split_inv<<16 | has_p<<17
	int32_t dp_score, dp_max, dp_max2;
	uint32_t n_ambi, n_cigar, cigar_off;   // cigar_off: index into the CIGAR arena, or AL_CIG_INLINE when n_cigar <= 4
	uint32_t cig_inl[4];                   // short CIGARs are stored in the record itself (no allocation)
};
#define AL_CIG_INLINE 0xfffffffeu
#define ALR_SPLIT(f)     ((f)&3u)
#define ALR_REV          (1u<<2)
#define ALR_INV          (1u<<3)
#define ALR_SAM_PRI      (1u<<4)
#define ALR_PROPER       (1u<<5)
#define ALR_PE_THRU      (1u<<6)
#define ALR_SEG_SPLIT    (1u<<7)
#define ALR_SEG_ID(f)    (((f)>>8)&0xffu)
#define ALR_SPLIT_INV    (1u<<16)
#define ALR_HAS_P        (1u<<17)

// Segment mode of the chaining kernels.  A fragment's sorted anchors fall apart into independent sub-problems wherever two
// neighbours are more than max_dist_x apart (no predecessor window crosses such a gap, chain.c:52): reads inside
// interspersed repeats have thousands of anchors but only short runs of them, so these runs ("segments") are chained as
// list entries of their own (a_off / frag_na are then per segment) and k_seg_merge puts the fragment's chain list together.
// meta == nullptr: the list entries are whole fragments.
struct ChainSeg {
	const uint32_t *meta;     // per entry: qlen_sum | (paired ? 1 << 31 : 0) of the fragment it belongs to
	// per entry ONE 16-byte result record (one store by the kernel, one load by k_seg_merge):
	//   words 0-1: the chain list entry (score << 32 | count) when the entry has exactly one chain (0: the list is in the scratch range),
	//   word 2: number of chains, word 3: number of chained anchors written | 1 << 31 if two of its chains start at anchors of equal x
	//   (their order is the fragment-wide sort's business)
	uint32_t *res;
	// fragments whose anchors had equal x (tie_flag[f] != 0: exact heap merge, then the whole-fragment wavefront kernel) run on a
	// side stream next to everything else.  tie_mode 1: skip flagged fragments (main stream), 2: only flagged fragments (side stream)
	const uint32_t *tie_flag; int tie_mode;
	// segment mode, min_cnt >= 2: per chain (same order as the chain list) the key the reference processes chain ends by, peak score << 32 |
	// peak anchor (chain.c:111-114) -- what k_chain_order needs to restate the fragment-wide sort of equal chain starts
	uint64_t *okey;
	// direct mode (the segments the tile kernel defers, al_kernels_chain.hip): the entry's chains go straight to the fragment's arrays -- their anchors
	// at the segment's own place in chained[] (a_off[] = the segment's first anchor), the list entries at u[uslot[entry] ...] (slots the tile kernel
	// reserved and zeroed), uo = rel[entry] + offset inside the segment; equal chain starts set ctie[fragid[entry]]
	const uint64_t *uslot; const uint32_t *rel; const uint32_t *fragid; uint32_t *ctie;
};

#define AL_ORD_CAP 8192                   // chains of a fragment whose exact order k_chain_order restates in LDS (14 bytes each)
#define AL_ORD_CAP2 16128                 // (x, id) pairs of the block form of k_chain_order in LDS for the sort restatement (10 bytes each; the 160 KB of a CU less the sort's scratch)
struct LbThr { uint32_t v[16]; int n; };     // thresholds of k_lower_bounds

// one HIP-event interval per entry; where a stage is several kernels (chaining and extension DP are dispatched by size class)
// each kernel with real weight has its own interval so that bench.py's per-kernel times line up with rocprofv3's
enum AlStage { ST_SKETCH = 0, ST_SEED, ST_SCAN, ST_ORDER, ST_ANCHOR_SORT_S, ST_ANCHOR_SORT, ST_ANCHOR_SORT_BLK, ST_ANCHOR_SORT_BIG, ST_ANCHOR_HEAP,
               ST_CHAIN_LDS32, ST_CHAIN_LDS48, ST_CHAIN_LDS64, ST_CHAIN_LDS128, ST_SEG_FIND, ST_SEG_CHAIN_LDS, ST_SEG_CHAIN_WAVE, ST_SEG_MERGE, ST_RECHAIN,
               ST_REGS, ST_EXT_PREP, ST_EXT_SORT, ST_EXT_DP_LANE, ST_EXT_DP_G4, ST_EXT_DP_G8, ST_EXT_DP_G12, ST_EXT_DP_G16, ST_EXT_DP_G22, ST_EXT_FINISH, ST_COMPACT, ST_N };
#define ST_CHAIN ST_SEG_MERGE
#define ST_EXT_DP ST_EXT_DP_G22

// anchors per fragment the wavefront-per-fragment chaining kernel keeps entirely in LDS (32 bytes each); larger fragments keep
// their DP arrays in global memory and mirror the most recent rows
#ifndef AL_CHAIN_CAP
#define AL_CHAIN_CAP 384
#endif
