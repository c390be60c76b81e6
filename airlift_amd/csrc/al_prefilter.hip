// al_prefilter.hip -- SURVEY.md 8f N4: pre-alignment filters on the candidate locations the as-shipped forks count.
//
// The bundled forks are instrumented to count candidate locations "before alignment (verification)": the minimap2 fork counts seed
// clusters (ALSER loop, map.c:299-312; row a8, k_alser_count), the mrFAST fork counts what survives its adjacency filter
// (MrFAST.c:1741-1764, mappingCnt_BeforeAlignment :1787) and carries an unused GreedySnake() pre-alignment filter
// (GreedySnake.c:52-200).  This kernel applies those two filters to the clusters of the ALSER count, on the device, between the
// anchor sort (K3) and whatever would verify the candidates: it reports how many candidates each filter keeps.  Off by default and
// outside the SAM path (the alignment records do not depend on it).
//
//   candidate     a cluster the ALSER loop counts; its location is the diagonal of the cluster's first anchor: read base j lies on
//                 reference base ref_start + j (the read reverse-complemented for a reverse-strand anchor)
//   adjacency     MrFAST.c:1741-1764 on minimizers instead of 12-mers: every other seed of the read (each query minimizer with an
//                 occurrence list, collect_matches map.c:90-123) is looked up at the position the candidate's diagonal predicts --
//                 searchKey, an exact binary search in that seed's sorted occurrence list; more than adj_e absent seeds reject it
//   GreedySnake   GreedySnake.c:52-200 statement for statement on the read and the reference window of its length
//                 (EditThreshold, KmerSize, IterationNo as given); bases as nt4 codes, positions outside the contig never match
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "al_internal.h"
#include "al_device.h"
#include "al_runtime.h"

#define N4_LMAX 512                    // longest read the filters take (GreedySnake works on one read-length window)

struct N4Par { int adj_e, snake_e, snake_k, snake_iter, min_cnt, k; };

// GreedySnake.c:52-200 (DebugMode off).  ref / rd: one code per base; returns 1 (accept) or 0 (reject).
__device__ int d_greedy_snake(int ReadLength, const uint8_t *RefSeq, const uint8_t *ReadSeq, int EditThreshold, int KmerSize, int IterationNo)
{
	int Edits = 0;
	for (int K = 0; K < ReadLength / KmerSize; ++K) {                          // :79
		const int KmerStart = K * KmerSize, KmerEnd = K < ReadLength / KmerSize - 1 ? (K + 1) * KmerSize : ReadLength;
		int index = KmerStart, roundsNo = 1;
		while (index < KmerEnd) {                                               // :89
			int GlobalCount = 0, n;
			for (n = index; n < KmerEnd; ++n) { if (ReadSeq[n] != RefSeq[n]) break; ++GlobalCount; }            // main diagonal :94-101
			if (GlobalCount == KmerEnd - KmerStart) goto LOOP;                   // :103
			for (int e = 1; e <= EditThreshold; ++e) {                          // :109
				int count = 0;
				for (n = index; n < KmerEnd; ++n) { if (n < e) break; if (ReadSeq[n - e] != RefSeq[n]) break; ++count; }    // upper diagonals :113-121
				if (count > GlobalCount) GlobalCount = count;
				if (count == KmerEnd - KmerStart) goto LOOP;
				count = 0;
				for (n = index; n < KmerEnd; ++n) { if (n > ReadLength - e - 1) break; if (ReadSeq[n + e] != RefSeq[n]) break; ++count; }   // lower diagonals :132-140
				if (count > GlobalCount) GlobalCount = count;
				if (count == KmerEnd - KmerStart) goto LOOP;
			}
			index += GlobalCount;                                               // :152
			if (index < KmerEnd) { ++Edits; ++index; }
			if (roundsNo > IterationNo) goto LOOP;
			++roundsNo;
			if (Edits > EditThreshold) return 0;
		}
		LOOP:
		if (Edits > EditThreshold) return 0;                                    // :190
	}
	return 1;
}

// searchKey (MrFAST.c): is `word` in the ascending list?
__device__ __forceinline__ bool d_search_key(const uint64_t *__restrict__ pos, const AlMatch &m, uint64_t word)
{
	if (m.flags >> 9 & 1u) return ((uint64_t)m.off_lo | (uint64_t)(m.flags >> 16) << 32) == word;     // a once-occurring minimizer keeps its position in the record
	const uint64_t *l = pos + m.off_lo; uint32_t lo = 0, hi = m.n;
	while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; const uint64_t v = l[mid]; if (v == word) return true; if (v < word) lo = mid + 1; else hi = mid; }
	return false;
}

__global__ void __launch_bounds__(64)
k_prefilter(const AlAnchor *__restrict__ a, const uint64_t *__restrict__ a_off, const uint32_t *__restrict__ frag_na, const uint32_t *__restrict__ frag_first,
            const uint32_t *__restrict__ rd_len, const uint32_t *__restrict__ rd_seq, const uint64_t *__restrict__ rd_off,
            const AlMatch *__restrict__ match, const uint64_t *__restrict__ mini_off, const uint32_t *__restrict__ frag_nm,
            const uint64_t *__restrict__ pos, const uint32_t *__restrict__ S4, const uint64_t *__restrict__ seq_off, const uint32_t *__restrict__ seq_len,
            int n_frag, N4Par P, uint8_t *__restrict__ ws /* 3 * N4_LMAX bytes per lane */, unsigned long long *__restrict__ out,
            uint64_t *__restrict__ dec /* optional: one word per candidate, fragment << 32 | first anchor of the cluster << 2 | kept by adjacency << 1 | kept by GreedySnake */, uint32_t dec_cap)
{
	const int f = blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long c_all = 0, c_adj = 0, c_snk = 0, c_both = 0;
	if (f < n_frag) {
		const uint32_t r = frag_first[f]; const int L = (int)rd_len[r];
		const AlAnchor *p = a + a_off[f]; const uint32_t n = frag_na[f];
		uint8_t *fw = ws + (size_t)f * 3 * N4_LMAX, *rv = fw + N4_LMAX, *rf = rv + N4_LMAX;
		bool have_seq = false;
		int seed_num = 0; uint32_t cs = 0;
		for (uint32_t i = 1; i < n; ++i) {
			if (((int32_t)p[i].x - (int32_t)p[i - 1].x) > L) {                   // map.c:301-308: the cluster [cs, i) ends
				if (seed_num >= P.min_cnt - 1 && L <= N4_LMAX) {
					++c_all;
					if (!have_seq) {   // the read in both orientations, one code per base
						const uint32_t *w = rd_seq + rd_off[r];
						for (int j = 0; j < L; ++j) { const uint32_t cd = w[j >> 3] >> ((j & 7) << 2) & 0xf; fw[j] = (uint8_t)cd; rv[L - 1 - j] = (uint8_t)(cd < 4 ? 3 - cd : 4); }
						have_seq = true;
					}
					const AlAnchor A = p[cs];
					const int rev = (int)(A.x >> 63), rid = (int)(A.x << 1 >> 33);
					const int64_t ref_start = (int64_t)(int32_t)A.x - (int64_t)(int32_t)A.y;    // both coordinates are k-mer end positions
					// adjacency (MrFAST.c:1741-1764)
					bool keep_adj = true;
					{
						const AlMatch *m = match + mini_off[r]; const uint32_t n_m = frag_nm[f]; int diff = 0;
						for (uint32_t ix = 0; ix < n_m && keep_adj; ++ix) {
							const int qend = (int)(m[ix].q_pos >> 1), qs = (int)(m[ix].q_pos & 1);
							const int64_t rp = rev ? ref_start + (L - (qend + 1 - P.k) - 1) : ref_start + qend;   // map.c:183: position of the k-mer on the reverse-complemented read
							bool hit = false;
							if (rp >= 0 && rp < (int64_t)seq_len[rid]) hit = d_search_key(pos, m[ix], (uint64_t)rid << 32 | (uint64_t)rp << 1 | (uint64_t)(rev ? 1 - qs : qs));
							if (!hit && ++diff > P.adj_e) keep_adj = false;
						}
					}
					// GreedySnake (GreedySnake.c:52)
					const uint64_t so = seq_off[rid]; const int64_t sl = (int64_t)seq_len[rid];
					for (int j = 0; j < L; ++j) { const int64_t g = ref_start + j; rf[j] = (g >= 0 && g < sl) ? (uint8_t)d_seq4(S4, so + (uint64_t)g) : (uint8_t)5; }
					const bool keep_snk = d_greedy_snake(L, rf, rev ? rv : fw, P.snake_e, P.snake_k, P.snake_iter) != 0;
					c_adj += keep_adj; c_snk += keep_snk; c_both += keep_adj && keep_snk;
					if (dec) { const unsigned long long at = atomicAdd(out + 4, 1ULL); if (at < dec_cap) dec[at] = (uint64_t)(uint32_t)f << 32 | (uint64_t)cs << 2 | (keep_adj ? 2u : 0u) | (keep_snk ? 1u : 0u); }
				}
				seed_num = 0; cs = i;
			} else ++seed_num;
		}
	}
	for (int d = 32; d > 0; d >>= 1) { c_all += __shfl_xor(c_all, d); c_adj += __shfl_xor(c_adj, d); c_snk += __shfl_xor(c_snk, d); c_both += __shfl_xor(c_both, d); }
	if ((threadIdx.x & 63) == 0) { if (c_all) atomicAdd(out, c_all); if (c_adj) atomicAdd(out + 1, c_adj); if (c_snk) atomicAdd(out + 2, c_snk); if (c_both) atomicAdd(out + 3, c_both); }
}

int al_run_seed_stages(al_ctx_t *c);     // al_runtime.hip

// Seed stages on the resident batch of single-segment fragments, then the filters on the ALSER candidates.
// out4: candidates (= the ALSER count), kept by the adjacency filter, kept by GreedySnake, kept by both.
extern "C" int al_batch_prefilter_decisions(al_ctx_t *c, int adj_e, int snake_e, int snake_k, int snake_iter, int64_t *out4, uint64_t *dec, int64_t dec_cap);
extern "C" int al_batch_prefilter(al_ctx_t *c, int adj_e, int snake_e, int snake_k, int snake_iter, int64_t *out4) { return al_batch_prefilter_decisions(c, adj_e, snake_e, snake_k, snake_iter, out4, nullptr, 0); }
// ... and every candidate's two decisions (tests): dec[0 .. min(out4[0], dec_cap)) = fragment << 32 | first anchor of the candidate's cluster << 2 | kept by the
// adjacency filter << 1 | kept by GreedySnake, in no particular order
extern "C" int al_batch_prefilter_decisions(al_ctx_t *c, int adj_e, int snake_e, int snake_k, int snake_iter, int64_t *out4, uint64_t *dec, int64_t dec_cap)
{
	if (!c || !out4 || snake_k < 1) return -1;
	if (c->max_rd_len > N4_LMAX) { fprintf(stderr, "[airlift] the pre-alignment filters take reads of up to %d bases\n", N4_LMAX); return -3; }
	{   // the candidates are those of the first (mid_occ) seeding pass, as the fork counts them: no max_occ re-seeding here
		const int32_t mo = c->opt.max_occ; c->opt.max_occ = c->opt.mid_occ;
		const int e = al_run_seed_stages(c);
		c->opt.max_occ = mo;
		if (e) return -1;
	}
	AL_HIP_CHECK(hipSetDevice(c->device));
	hipStream_t s = c->stream;
	const int nf = c->n_frag;
	uint8_t *ws = nullptr; unsigned long long *d_out = nullptr; unsigned long long h[4] = {0, 0, 0, 0}; uint64_t *d_dec = nullptr;
	int rc = -1;
	if (dec_cap < 0 || dec_cap > 0x7fffffff) return -1;
	if (dec && dec_cap > 0 && hipMalloc((void **)&d_dec, (size_t)dec_cap * 8) != hipSuccess) return -1;
	if (hipMalloc((void **)&ws, (size_t)(nf > 0 ? nf : 1) * 3 * N4_LMAX) == hipSuccess && hipMalloc((void **)&d_out, 64) == hipSuccess && hipMemsetAsync(d_out, 0, 64, s) == hipSuccess) {
		const N4Par P{adj_e, snake_e, snake_k, snake_iter, c->opt.min_cnt, c->mi->k};
		const bool p1 = c->n_rechain > 0;
		if (nf) hipLaunchKernelGGL(k_prefilter, dim3((nf + 63) / 64), dim3(64), 0, s, c->anchors.p, p1 ? c->a_off_p1.p : c->a_off.p, p1 ? c->frag_na_p1.p : c->frag_na.p, c->frag_first.p,
		                           c->rd_len.p, c->rd_seq.p, c->rd_off.p, c->match.p, c->mini_off.p, c->frag_nm.p, c->di.pos, c->di.S4, c->di.seq_off, c->di.seq_len, nf, P, ws, d_out, d_dec, (uint32_t)dec_cap);
		if (hipMemcpyAsync(h, d_out, 32, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess && hipGetLastError() == hipSuccess) rc = 0;
		if (rc == 0 && d_dec) { const size_t nd = (size_t)std::min<unsigned long long>(h[0], (unsigned long long)dec_cap); if (nd && hipMemcpy(dec, d_dec, nd * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = -1; }
	}
	(void)hipFree(ws); (void)hipFree(d_out); (void)hipFree(d_dec);
	for (int i = 0; i < 4; ++i) out4[i] = (int64_t)h[i];
	return rc;
}
