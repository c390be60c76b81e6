// al_index_dev.hip -- minimizer index built on the MI355X (product code).
//
// Same result as the host builder in al_index.cpp, i.e. the meaning of mm_idx_gen (index.c:353-372): sketch every
// contig (mm_sketch, sketch.c:77-143), group the minimizers by hash with the positions of each hash ascending
// (worker_post, index.c:191-243) -- but laid out for HBM and produced by kernels:
//
//   host    FASTA(.gz) -> one ASCII buffer of all contigs (block reader), H2D
//   KI1     k_pack_ref       ASCII -> 4 bit/base words S4 (mm_seq4_set, index.c:320-326); thread per output word
//   KI2     k_ref_sketch<0>  count, then k_ref_sketch<1> emit: one lane per 256-base segment of a contig.  The window
//                            algorithm is local: restarting it w+k bases (rounded up generously) before the segment
//                            reproduces the state exactly at the segment start (ring contents, current minimum and the
//                            thresholds on the run length l), so a lane replays that lead-in silently and emits only
//                            the events of its own positions; the end-of-sequence flush belongs to the last segment.
//   rocPRIM radix sort by position word then (stable) by hash  == sort by (hash, position)
//   rocPRIM run-length encode + scan -> distinct hashes, counts, offsets
//   KI3     k_tab_insert     open-addressing table of 16-byte entries {hash+1, off<<32|n}; atomicCAS claims a slot
//
// Odd k (the short-read preset: 21): a k-mer never equals its reverse complement, the `continue` of sketch.c:108 never fires and 2 (w + k) + 8 bases of
// lead-in are enough.  Even k: the lead-in is counted in iterations that move the window (k_ref_sketch).
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <thread>
#include <time.h>
#include "al_internal.h"
#include "al_device.h"
#include "al_io.h"
#include "al_seqio.h"
#include "al_runtime.h"

#define AL_ISEG 256            // reference positions per lane

__device__ __forceinline__ uint64_t di_hash64m(uint64_t key, uint64_t mask)
{   // sketch.c:28-38
	key = (~key + (key << 21)) & mask; key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & mask; key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & mask; key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

struct Nt4Tab { uint8_t t[256]; };

__global__ void __launch_bounds__(256)
k_pack_ref(const uint8_t *__restrict__ ascii, uint64_t n_bases, uint32_t *__restrict__ S4, uint64_t n_words, Nt4Tab T)
{
	__shared__ uint8_t lut[256];
	lut[threadIdx.x] = T.t[threadIdx.x];
	__syncthreads();
	const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (o >= n_words) return;
	uint32_t v = 0;
	const uint64_t b0 = o * 8;
	if (b0 + 8 <= n_bases) {
		const uint2 raw = *reinterpret_cast<const uint2 *>(ascii + b0);
		const uint32_t lo = raw.x, hi = raw.y;
#pragma unroll
		for (int j = 0; j < 4; ++j) v |= (uint32_t)lut[(lo >> (8 * j)) & 0xff] << (4 * j);
#pragma unroll
		for (int j = 0; j < 4; ++j) v |= (uint32_t)lut[(hi >> (8 * j)) & 0xff] << (4 * (j + 4));
	} else {
		for (int j = 0; j < 8 && b0 + j < n_bases; ++j) v |= (uint32_t)lut[ascii[b0 + j]] << (4 * j);
	}
	S4[o] = v;
}

// segment s of the concatenated segment list -> (contig, start): seg_first[c] = first segment of contig c (n_seq+1 entries)
__device__ __forceinline__ uint32_t d_find_contig(const uint64_t *__restrict__ seg_first, uint32_t n_seq, uint64_t s)
{
	uint32_t lo = 0, hi = n_seq;                     // last c with seg_first[c] <= s
	while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (seg_first[mid] <= s) lo = mid; else hi = mid; }
	return lo;
}

template <int EMIT>
__global__ void __launch_bounds__(64)
k_ref_sketch(const uint32_t *__restrict__ S4, const uint64_t *__restrict__ seq_off, const uint32_t *__restrict__ seq_len,
             const uint64_t *__restrict__ seg_first, uint32_t n_seq, uint64_t n_seg, int w, int k,
             uint32_t *__restrict__ cnt_out, const uint64_t *__restrict__ out_off, uint64_t *__restrict__ out_h, uint64_t *__restrict__ out_y)
{
	extern __shared__ uint64_t lds[];               // bx[w][64], by[w][64]
	const int lane = threadIdx.x;
	const uint64_t sg = (uint64_t)blockIdx.x * 64 + lane;
	uint64_t *bx = lds + lane, *by = lds + (size_t)w * 64 + lane;
	if (sg >= n_seg) return;
	const uint32_t rid = d_find_contig(seg_first, n_seq, sg);
	const uint32_t len = seq_len[rid];
	const uint64_t base = seq_off[rid];
	const uint32_t s0 = (uint32_t)(sg - seg_first[rid]) * AL_ISEG;
	const uint32_t e0 = s0 + AL_ISEG < len ? s0 + AL_ISEG : len;
	// Even k (round 6): a k-mer that is its own reverse complement is skipped WITHOUT moving the window or the run length (sketch.c:108), so the lead-in is
	// measured in iterations that DO move it, after the first k bases (the k-mer words hold the last k non-N bases -- an N does not clear them, sketch.c:116 -- and
	// are not known before): 2 (w + k) + 8 of them put the ring, the minimum and every threshold on l where the serial run has them (l is at most k off after
	// the warm-up and only compared with w + k and below; an N sets both to 0).  A lane that finds fewer -- its lead-in lies in a palindromic repeat, (AT)n -- starts
	// again from twice as far back, up to the contig's start.
	const bool even = !(k & 1);
	const uint32_t need = 2 * (uint32_t)(w + k) + 8;
	uint32_t lead = even ? need + (uint32_t)k : need;
	const uint64_t o_first = EMIT ? out_off[sg] : 0;
	const uint64_t shift1 = 2 * (k - 1), mask = (1ULL << 2 * k) - 1;
	uint64_t o = o_first; uint32_t cnt = 0;
	uint64_t minx = UINT64_MAX, miny = UINT64_MAX;
#define EMIT_XY(X, Y) do { if (i >= s0) { if (EMIT) { out_h[o] = (X) >> 8; out_y[o] = (Y); ++o; } ++cnt; } } while (0)
	for (;;) {
		const uint32_t i0 = s0 > lead ? s0 - lead : 0;
		uint64_t kmer0 = 0, kmer1 = 0; minx = UINT64_MAX; miny = UINT64_MAX;
		int l = 0, buf_pos = 0, min_pos = 0;
		uint32_t moved = 0, warm = 0;                                        // even k: window moves after the first k non-N bases of the lead-in
		bool again = false;
		for (int j = 0; j < w; ++j) bx[j * 64] = UINT64_MAX, by[j * 64] = UINT64_MAX;
		uint32_t word = 0;
		for (uint32_t i = i0; i < e0; ++i) {
			if (even && i == s0 && i0 > 0 && moved < need) { again = true; break; }
			const uint64_t gp = base + i;
			if (i == i0 || (gp & 7) == 0) word = S4[gp >> 3];
			const int c = (word >> ((gp & 7) << 2)) & 0xf;
			uint64_t ix = UINT64_MAX, iy = UINT64_MAX;
			if (c < 4) {
				const int span = l + 1 < k ? l + 1 : k;
				kmer0 = (kmer0 << 2 | (uint64_t)c) & mask;
				kmer1 = (kmer1 >> 2) | (3ULL ^ (uint64_t)c) << shift1;
				if (even) { if (warm < (uint32_t)k) ++warm; if (kmer0 == kmer1) continue; }   // sketch.c:108 (odd k: never equal)
				const int z = kmer0 < kmer1 ? 0 : 1;
				++l;
				if (l >= k) { ix = di_hash64m(z ? kmer1 : kmer0, mask) << 8 | (uint64_t)span; iy = (uint64_t)rid << 32 | (uint64_t)i << 1 | (uint64_t)z; }
			} else l = 0;
			if (even && warm >= (uint32_t)k && i < s0) ++moved;
			bx[buf_pos * 64] = ix; by[buf_pos * 64] = iy;
			if (l == w + k - 1 && minx != UINT64_MAX) {                         // sketch.c:117-122
				for (int j = buf_pos + 1; j < w; ++j) { const uint64_t x = bx[j * 64], y = by[j * 64]; if (minx == x && y != miny) EMIT_XY(x, y); }
				for (int j = 0; j < buf_pos; ++j)     { const uint64_t x = bx[j * 64], y = by[j * 64]; if (minx == x && y != miny) EMIT_XY(x, y); }
			}
			if (ix <= minx) {                                                   // sketch.c:123-125
				if (l >= w + k && minx != UINT64_MAX) EMIT_XY(minx, miny);
				minx = ix, miny = iy, min_pos = buf_pos;
			} else if (buf_pos == min_pos) {                                    // sketch.c:126-138
				if (l >= w + k - 1 && minx != UINT64_MAX) EMIT_XY(minx, miny);
				minx = UINT64_MAX;
				for (int j = buf_pos + 1; j < w; ++j) { const uint64_t x = bx[j * 64]; if (minx >= x) minx = x, miny = by[j * 64], min_pos = j; }
				for (int j = 0; j <= buf_pos; ++j)    { const uint64_t x = bx[j * 64]; if (minx >= x) minx = x, miny = by[j * 64], min_pos = j; }
				if (l >= w + k - 1 && minx != UINT64_MAX) {
					for (int j = buf_pos + 1; j < w; ++j) { const uint64_t x = bx[j * 64], y = by[j * 64]; if (minx == x && miny != y) EMIT_XY(x, y); }
					for (int j = 0; j <= buf_pos; ++j)    { const uint64_t x = bx[j * 64], y = by[j * 64]; if (minx == x && miny != y) EMIT_XY(x, y); }
				}
			}
			if (++buf_pos == w) buf_pos = 0;
		}
		if (!again) break;
		lead *= 2;
	}
	if (e0 == len && minx != UINT64_MAX) { const uint32_t i = e0; EMIT_XY(minx, miny); }   // sketch.c:141-142 (end of the contig only)
#undef EMIT_XY
	if (!EMIT) cnt_out[sg] = cnt;
}

__global__ void __launch_bounds__(256)
k_tab_insert(const uint64_t *__restrict__ uniq, const uint32_t *__restrict__ counts, const uint64_t *__restrict__ offs, uint64_t n_keys,
             const uint64_t *__restrict__ pos, int single_ok, unsigned long long *__restrict__ tab, int tab_bits)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_keys) return;
	const uint64_t h = uniq[i], tmask = (1ULL << tab_bits) - 1;
	const bool single = single_ok && counts[i] == 1;
	const unsigned long long key = (unsigned long long)(h + 1) | (single ? AL_TAB_SINGLE : 0ULL);
	const uint64_t val = single ? pos[offs[i]] : (offs[i] << 32 | (uint64_t)counts[i]);
	uint64_t s = (h * 0x9E3779B97F4A7C15ULL) >> (64 - tab_bits);
	for (;;) {
		if (atomicCAS(&tab[2 * s], 0ULL, key) == 0ULL) { tab[2 * s + 1] = val; return; }
		s = (s + 1) & tmask;
	}
}

struct CastU64I { __host__ __device__ uint64_t operator()(const uint32_t &v) const { return (uint64_t)v; } };

#define IDX_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	fprintf(stderr, "[airlift] HIP error %s at %s:%d: %s\n", hipGetErrorName(e_), __FILE__, __LINE__, hipGetErrorString(e_)); goto fail; } } while (0)

extern "C" al_idx_t *al_idx_build_device(const char *fn, const al_idxopt_t *io, int device)
{
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { fprintf(stderr, "[airlift] FATAL: al_idx_build_device: no HIP device available\n"); return nullptr; }
	if (device < 0) { const char *lr = getenv("LOCAL_RANK"); device = lr ? atoi(lr) % n_dev : 0; }
	if (device >= n_dev) { fprintf(stderr, "[airlift] FATAL: device %d out of range (%d devices)\n", device, n_dev); return nullptr; }
	const int w = io->w, k = io->k;
	if (k < 1 || k > AL_MAX_K || w > 32 || w < 1) { fprintf(stderr, "[airlift] al_idx_build_device: needs k <= %d and w <= 32 (got k=%d w=%d)\n", AL_MAX_K, k, w); return nullptr; }
	AlSeqReader rd;
	if (!rd.open(fn)) { fprintf(stderr, "[ERROR] airlift: failed to open '%s'\n", fn); return nullptr; }
	al_idx_t *mi = new al_idx_t();
	mi->k = k; mi->w = w;
	AlText ascii; uint64_t sum = 0;
	const bool timing = getenv("AL_TIMING") != nullptr;
	struct timespec tq0, tq1, tq2; clock_gettime(CLOCK_MONOTONIC, &tq0);
	auto secs = [](const struct timespec &a, const struct timespec &b) { return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec); };
	int n_host = getenv("AL_IDX_THREADS") ? atoi(getenv("AL_IDX_THREADS")) : (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
	if (al_fasta_load_parallel(fn, n_host, mi->seq, ascii)) sum = ascii.size();
	else {
		AlChunk c;
		for (;;) {
			c.text.clear(); c.recs.clear();
			if (!rd.read(c)) break;
			const AlRec &r = c.recs[0];
			AlSeq s; s.name = c.text.data() + r.name; s.len = r.len; s.offset = sum; sum += r.len; mi->seq.push_back(s);
			ascii.insert(ascii.end(), c.text.data() + r.seq, c.text.data() + r.seq + r.len);
		}
	}
	if (mi->seq.empty()) { fprintf(stderr, "[ERROR] airlift: no sequences in '%s'\n", fn); delete mi; return nullptr; }
	clock_gettime(CLOCK_MONOTONIC, &tq1);
	mi->tot_len = sum;
	const uint32_t n_seq = (uint32_t)mi->seq.size();
	const uint64_t n_words = (sum + 7) / 8 + 8;
	std::vector<uint64_t> so(n_seq), seg_first(n_seq + 1); std::vector<uint32_t> sl(n_seq);
	uint64_t n_seg = 0;
	for (uint32_t i = 0; i < n_seq; ++i) { so[i] = mi->seq[i].offset; sl[i] = mi->seq[i].len; seg_first[i] = n_seg; n_seg += ((uint64_t)sl[i] + AL_ISEG - 1) / AL_ISEG; }
	seg_first[n_seq] = n_seg;

	AlDevIndex d;
	uint8_t *d_ascii = nullptr; uint64_t *d_segf = nullptr, *d_off = nullptr, *d_h = nullptr, *d_y = nullptr, *d_h2 = nullptr, *d_y2 = nullptr, *d_uniq = nullptr, *d_koff = nullptr;
	uint32_t *d_cnt = nullptr, *d_kcnt = nullptr; uint64_t *d_nruns = nullptr; void *d_tmp = nullptr; size_t tmp_bytes = 0;
	uint64_t total = 0, n_keys = 0; int rid_bits = 1, bits = 4;
	Nt4Tab T; memcpy(T.t, al_nt4(), 256);
	hipStream_t st = nullptr;
	if (hipSetDevice(device) != hipSuccess) { delete mi; return nullptr; }
	IDX_CHECK(hipStreamCreate(&st));
	IDX_CHECK(al_dev_malloc((void **)&d_ascii, sum + 16));
	IDX_CHECK(hipMemcpyAsync(d_ascii, ascii.data(), sum, hipMemcpyHostToDevice, st));
	IDX_CHECK(al_dev_malloc((void **)&d.S4, n_words * 4));
	hipLaunchKernelGGL(k_pack_ref, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, d_ascii, sum, d.S4, n_words, T);
	IDX_CHECK(al_dev_malloc((void **)&d.seq_off, (size_t)n_seq * 8)); IDX_CHECK(al_dev_malloc((void **)&d.seq_len, (size_t)n_seq * 4)); IDX_CHECK(al_dev_malloc((void **)&d_segf, (size_t)(n_seq + 1) * 8));
	IDX_CHECK(hipMemcpyAsync(d.seq_off, so.data(), (size_t)n_seq * 8, hipMemcpyHostToDevice, st));
	IDX_CHECK(hipMemcpyAsync(d.seq_len, sl.data(), (size_t)n_seq * 4, hipMemcpyHostToDevice, st));
	IDX_CHECK(hipMemcpyAsync(d_segf, seg_first.data(), (size_t)(n_seq + 1) * 8, hipMemcpyHostToDevice, st));
	IDX_CHECK(al_dev_malloc((void **)&d_cnt, (n_seg + 1) * 4)); IDX_CHECK(al_dev_malloc((void **)&d_off, (n_seg + 2) * 8));
	IDX_CHECK(hipMemsetAsync(d_cnt, 0, (n_seg + 1) * 4, st));
	if (n_seg) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ref_sketch<0>), dim3((unsigned)((n_seg + 63) / 64)), dim3(64), (size_t)w * 64 * 16, st,
	                              d.S4, d.seq_off, d.seq_len, d_segf, n_seq, n_seg, w, k, d_cnt, (const uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr);
	{
		auto it = rocprim::make_transform_iterator((const uint32_t *)d_cnt, CastU64I());
		IDX_CHECK(rocprim::exclusive_scan(nullptr, tmp_bytes, it, d_off, (uint64_t)0, (size_t)((int)(n_seg + 1)), rocprim::plus<uint64_t>(), st));
		IDX_CHECK(al_dev_malloc(&d_tmp, tmp_bytes + 16));
		IDX_CHECK(rocprim::exclusive_scan(d_tmp, tmp_bytes, it, d_off, (uint64_t)0, (size_t)((int)(n_seg + 1)), rocprim::plus<uint64_t>(), st));
		IDX_CHECK(hipMemcpyAsync(&total, d_off + n_seg, 8, hipMemcpyDeviceToHost, st));
		IDX_CHECK(hipStreamSynchronize(st));
		al_dev_free(d_tmp); d_tmp = nullptr;
	}
	al_dev_free(d_ascii); d_ascii = nullptr;
	if (total >= (1ULL << 31)) { fprintf(stderr, "[airlift] al_idx_build_device: %llu minimizers exceed the 31-bit item count of the device sorts (references above ~12 Gbp need a multi-part index, which this path does not build)\n", (unsigned long long)total); goto fail; }
	IDX_CHECK(al_dev_malloc((void **)&d_h, (total + 1) * 8)); IDX_CHECK(al_dev_malloc((void **)&d_y, (total + 1) * 8));
	IDX_CHECK(al_dev_malloc((void **)&d_h2, (total + 1) * 8)); IDX_CHECK(al_dev_malloc((void **)&d_y2, (total + 1) * 8));
	if (n_seg) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ref_sketch<1>), dim3((unsigned)((n_seg + 63) / 64)), dim3(64), (size_t)w * 64 * 16, st,
	                              d.S4, d.seq_off, d.seq_len, d_segf, n_seq, n_seg, w, k, (uint32_t *)nullptr, d_off, d_h, d_y);
	while ((1u << rid_bits) < n_seq) ++rid_bits;
	if (total) {
		// (hash, position) order: stable LSD sorts, position word first (keys y, values h), then hash (keys h, values y)
		IDX_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_y, d_y2, d_h, d_h2, (int)total, 0, 32 + rid_bits, st));
		IDX_CHECK(al_dev_malloc(&d_tmp, tmp_bytes + 16));
		IDX_CHECK(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_y, d_y2, d_h, d_h2, (int)total, 0, 32 + rid_bits, st));
		(void)hipStreamSynchronize(st); al_dev_free(d_tmp); d_tmp = nullptr;
		IDX_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_h2, d_h, d_y2, d_y, (int)total, 0, 2 * k, st));
		IDX_CHECK(al_dev_malloc(&d_tmp, tmp_bytes + 16));
		IDX_CHECK(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_h2, d_h, d_y2, d_y, (int)total, 0, 2 * k, st));
		(void)hipStreamSynchronize(st); al_dev_free(d_tmp); d_tmp = nullptr;
		// d_h / d_y now sorted.  Distinct hashes, their counts and offsets:
		d_uniq = d_h2; d_h2 = nullptr;                                    // reuse
		IDX_CHECK(al_dev_malloc((void **)&d_kcnt, (total + 1) * 4)); IDX_CHECK(al_dev_malloc((void **)&d_nruns, 8));
		IDX_CHECK(rocprim::run_length_encode(nullptr, tmp_bytes, d_h, (unsigned int)((int)total), d_uniq, d_kcnt, d_nruns, st));
		IDX_CHECK(al_dev_malloc(&d_tmp, tmp_bytes + 16));
		IDX_CHECK(rocprim::run_length_encode(d_tmp, tmp_bytes, d_h, (unsigned int)((int)total), d_uniq, d_kcnt, d_nruns, st));
		IDX_CHECK(hipMemcpyAsync(&n_keys, d_nruns, 8, hipMemcpyDeviceToHost, st));
		IDX_CHECK(hipStreamSynchronize(st));
		al_dev_free(d_tmp); d_tmp = nullptr;
		d_koff = d_y2; d_y2 = nullptr;                                    // reuse
		{
			auto it = rocprim::make_transform_iterator((const uint32_t *)d_kcnt, CastU64I());
			IDX_CHECK(rocprim::exclusive_scan(nullptr, tmp_bytes, it, d_koff, (uint64_t)0, (size_t)((int)n_keys), rocprim::plus<uint64_t>(), st));
			IDX_CHECK(al_dev_malloc(&d_tmp, tmp_bytes + 16));
			IDX_CHECK(rocprim::exclusive_scan(d_tmp, tmp_bytes, it, d_koff, (uint64_t)0, (size_t)((int)n_keys), rocprim::plus<uint64_t>(), st));
		}
	}
	while ((1ULL << bits) < n_keys * 2 + 2) ++bits;
	IDX_CHECK(al_dev_malloc((void **)&d.tab, ((size_t)2 << bits) * 8));
	IDX_CHECK(hipMemsetAsync(d.tab, 0, ((size_t)2 << bits) * 8, st));
	if (n_keys) hipLaunchKernelGGL(k_tab_insert, dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, st, d_uniq, d_kcnt, d_koff, n_keys, d_y, n_seq <= AL_TAB_SINGLE_MAX_SEQ ? 1 : 0, (unsigned long long *)d.tab, bits);
	IDX_CHECK(hipStreamSynchronize(st));
	d.pos = d_y; d_y = nullptr;
	if (!total) IDX_CHECK(al_dev_malloc((void **)&d.pos, 8));
	d.tab_bits = bits; d.n_seq = n_seq;
	mi->tab_bits = bits; mi->n_keys = n_keys; mi->n_pos = total; mi->built_on = device;
	mi->dev[device] = d;
	al_dev_free(d_tmp); al_dev_free(d_segf); al_dev_free(d_cnt); al_dev_free(d_off); al_dev_free(d_h); al_dev_free(d_uniq); al_dev_free(d_koff); al_dev_free(d_kcnt); al_dev_free(d_nruns);
	al_dev_free(d_h2); al_dev_free(d_y2);
	(void)hipStreamDestroy(st);
	if (timing) { clock_gettime(CLOCK_MONOTONIC, &tq2); fprintf(stderr, "[airlift] index: FASTA load %.3f s (%d threads), upload + kernels %.3f s; %llu minimizers, %llu distinct; index arrays %.2f GB on the device\n", secs(tq0, tq1), n_host, secs(tq1, tq2), (unsigned long long)total, (unsigned long long)n_keys,
	                    (n_words * 4.0 + ((double)((size_t)2 << bits)) * 8.0 + (total + 1) * 8.0) / 1e9); }
	return mi;
fail:
	al_dev_free(d_tmp); al_dev_free(d_ascii); al_dev_free(d_segf); al_dev_free(d_cnt); al_dev_free(d_off); al_dev_free(d_h); al_dev_free(d_y); al_dev_free(d_h2); al_dev_free(d_y2);
	al_dev_free(d_uniq); al_dev_free(d_koff); al_dev_free(d_kcnt); al_dev_free(d_nruns);
	al_dev_free(d.S4); al_dev_free(d.tab); al_dev_free(d.seq_off); al_dev_free(d.seq_len);
	if (st) (void)hipStreamDestroy(st);
	delete mi;
	return nullptr;
}

// the host arrays of a device-built index (al_idx_dump): table, positions and packed sequence copied back once
int al_idx_to_host(al_idx_t *mi)
{
	if (!mi || mi->built_on < 0) return mi ? 0 : -1;
	std::lock_guard<std::mutex> lk(mi->dev_mtx);
	auto it = mi->dev.find(mi->built_on);
	if (it == mi->dev.end() || hipSetDevice(mi->built_on) != hipSuccess) return -1;
	const AlDevIndex &d = it->second;
	mi->tab.resize((size_t)2 << mi->tab_bits); mi->pos.resize(mi->n_pos ? mi->n_pos : 1); mi->S4.assign((size_t)((mi->tot_len + 7) / 8) + 8, 0);
	if (hipMemcpy(mi->tab.data(), d.tab, mi->tab.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
	if (mi->n_pos && hipMemcpy(mi->pos.data(), d.pos, (size_t)mi->n_pos * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
	if (mi->tot_len && hipMemcpy(mi->S4.data(), d.S4, (size_t)((mi->tot_len + 7) / 8) * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
	return 0;
}

// copy of the sorted position array (host- or device-built index); returns the number of entries, or -1
extern "C" int64_t al_idx_export_pos(const al_idx_t *mi, uint64_t *dst, int64_t cap)
{
	if (!mi) return -1;
	if (mi->built_on < 0) {
		const int64_t n = (int64_t)mi->n_pos;
		if (dst) memcpy(dst, mi->pos.data(), (size_t)(n < cap ? n : cap) * 8);
		return n;
	}
	std::lock_guard<std::mutex> lk(mi->dev_mtx);
	auto it = mi->dev.find(mi->built_on);
	if (it == mi->dev.end()) return -1;
	const int64_t n = (int64_t)mi->n_pos;
	if (dst && n) { if (hipSetDevice(mi->built_on) != hipSuccess || hipMemcpy(dst, it->second.pos, (size_t)(n < cap ? n : cap) * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1; }
	return n;
}

// ---------------------------------------------------------------------------------------------
// Occurrence threshold from the index (mm_idx_cal_max_occ, index.c:164-185): the (1-f) quantile of the per-minimizer
// occurrence counts, plus one.  Host-built index: the table is walked on the host; device-built index: one kernel
// gathers the counts of the occupied slots, a device radix sort orders them and one element comes back.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_tab_counts(const uint64_t *__restrict__ tab, uint64_t n_slots, uint32_t *__restrict__ out, unsigned long long *__restrict__ n_out)
{
	const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= n_slots) return;
	const uint64_t key = tab[2 * s];
	if (!key) return;
	const uint32_t c = (key & AL_TAB_SINGLE) ? 1u : (uint32_t)tab[2 * s + 1];
	out[atomicAdd(n_out, 1ULL)] = c;
}

extern "C" int32_t al_idx_cal_max_occ(const al_idx_t *mi, float f)
{
	if (!mi || f <= 0.f) return INT32_MAX;
	const uint64_t n = mi->n_keys;
	if (n == 0) return INT32_MAX;
	const uint64_t kth = (uint64_t)((1. - (double)f) * (double)n);           // ks_ksmall's 0-based rank (index.c:182)
	if (mi->built_on < 0) {
		std::vector<uint32_t> a; a.reserve(n);
		const uint64_t n_slots = 1ULL << mi->tab_bits;
		for (uint64_t s = 0; s < n_slots; ++s) if (mi->tab[2 * s]) a.push_back((mi->tab[2 * s] & AL_TAB_SINGLE) ? 1u : (uint32_t)mi->tab[2 * s + 1]);
		if (a.empty()) return INT32_MAX;
		const uint64_t kk = kth < a.size() ? kth : a.size() - 1;
		std::nth_element(a.begin(), a.begin() + kk, a.end());
		return (int32_t)a[kk] + 1;
	}
	std::lock_guard<std::mutex> lk(mi->dev_mtx);
	auto it = mi->dev.find(mi->built_on);
	if (it == mi->dev.end() || hipSetDevice(mi->built_on) != hipSuccess) return -1;
	const uint64_t n_slots = 1ULL << mi->tab_bits;
	uint32_t *d_c = nullptr, *d_s = nullptr; unsigned long long *d_n = nullptr; void *d_tmp = nullptr; size_t tmp_bytes = 0;
	uint32_t v = 0; int32_t ret = -1;
	const uint64_t kk = kth < n ? kth : n - 1;
	if (hipMalloc((void **)&d_c, n * 4) != hipSuccess || hipMalloc((void **)&d_s, n * 4) != hipSuccess || hipMalloc((void **)&d_n, 8) != hipSuccess) goto done;
	if (hipMemset(d_n, 0, 8) != hipSuccess) goto done;
	hipLaunchKernelGGL(k_tab_counts, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, 0, it->second.tab, n_slots, d_c, d_n);
	if (rocprim::radix_sort_keys(nullptr, tmp_bytes, d_c, d_s, (int)n) != hipSuccess) goto done;
	if (hipMalloc(&d_tmp, tmp_bytes + 16) != hipSuccess) goto done;
	if (rocprim::radix_sort_keys(d_tmp, tmp_bytes, d_c, d_s, (int)n) != hipSuccess) goto done;
	if (hipMemcpy(&v, d_s + kk, 4, hipMemcpyDeviceToHost) != hipSuccess) goto done;
	ret = (int32_t)v + 1;
done:
	(void)hipFree(d_c); (void)hipFree(d_s); (void)hipFree(d_n); (void)hipFree(d_tmp);
	if (ret < 0) fprintf(stderr, "[airlift] al_idx_cal_max_occ: HIP error: %s\n", hipGetErrorString(hipGetLastError()));
	return ret;
}

// mm_mapopt_update (options.c:51-61): fills the index-dependent fields.  With mid_occ > 0 (every preset on this path sets
// 1000 for `sr`) nothing changes.
extern "C" void al_mapopt_update(al_mapopt_t *opt, const al_idx_t *mi)
{
	if (!opt || !mi) return;
	if (opt->mid_occ <= 0) {
		const int32_t m = al_idx_cal_max_occ(mi, 2e-4f);                       // mid_occ_frac of mm_mapopt_init (options.c:25)
		if (m > 0) opt->mid_occ = m;
	}
}

// mm_idx_reader_open/read/close (minimap.h:206-232, index.c:585-648): the reference hands out one index part per read()
// call; this path builds a single part (the sr preset's batch size of 4 Gbp covers AirLift's references, larger inputs are
// refused by the builders), so the first read() returns the whole index and the second NULL.
struct al_idx_reader_s { std::string fn, fn_out; al_idxopt_t opt; int n_parts; bool is_idx; };
extern "C" int64_t al_idx_is_idx(const char *fn);
extern "C" al_idx_t *al_idx_load(const char *fn);
extern "C" int al_idx_dump(const char *fn, al_idx_t *mi);

extern "C" al_idx_reader_t *al_idx_reader_open(const char *fn, const al_idxopt_t *io, const char *fn_out)
{
	if (!fn || !io) return nullptr;
	const int64_t is_idx = al_idx_is_idx(fn);                                  // index.c:585-600: a file that starts with the magic is a prebuilt index
	if (is_idx < 0) return nullptr;                                            // minimap.h:206: NULL when the file cannot be opened
	al_idx_reader_t *r = new al_idx_reader_t();
	r->fn = fn; r->opt = *io; r->n_parts = 0; r->is_idx = is_idx > 0; if (fn_out) r->fn_out = fn_out;
	return r;
}

extern "C" al_idx_t *al_idx_reader_read(al_idx_reader_t *r, int device)
{
	if (!r || r->n_parts > 0) return nullptr;
	al_idx_t *mi = r->is_idx ? al_idx_load(r->fn.c_str()) : al_idx_build_device(r->fn.c_str(), &r->opt, device);
	if (mi) ++r->n_parts;
	if (mi && !r->is_idx && !r->fn_out.empty() && al_idx_dump(r->fn_out.c_str(), mi) != 0) { al_idx_destroy(mi); return nullptr; }   // index.c:629-630: a part is dumped as it is read
	return mi;
}

extern "C" int al_idx_reader_eof(const al_idx_reader_t *r) { return !r || r->n_parts > 0; }
extern "C" void al_idx_reader_close(al_idx_reader_t *r) { delete r; }
