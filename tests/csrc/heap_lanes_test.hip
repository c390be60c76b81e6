// heap_lanes_test.hip -- k_anchor_heap_lanes<1|2, 16> against a host restatement of the reference's heap merge
// (collect_seed_hits_heap, map.c:149-213; ksort.h:43-59) on random occurrence lists with many equal positions.
// Built and run by tests/test_gpu_heap_lanes.py:  hipcc --offload-arch=gfx950 -O2 -I airlift_amd/csrc -I include ...
#include "../../airlift_amd/csrc/al_kernels_seed.hip"
#include <vector>
#include <random>
#include <algorithm>
#include <cstdio>

struct HE { uint64_t x; uint32_t mi, off; };
static void ref_merge(const uint64_t *pos, const AlMatch *m, uint32_t n_m, uint32_t n, int qlen, int span, AlAnchor *out)
{
	std::vector<HE> h(n_m);
	auto down = [&](size_t i, size_t nn) { size_t k = i; HE tmp = h[i]; while ((k = (k << 1) + 1) < nn) { if (k != nn - 1 && h[k].x > h[k + 1].x) ++k; if (h[k].x > tmp.x) break; h[i] = h[k]; i = k; } h[i] = tmp; };
	size_t hs = 0; uint64_t n_for = 0, n_rev = 0;
	for (uint32_t i = 0; i < n_m; ++i) h[hs++] = HE{pos[m[i].off_lo], i, 0};
	if (hs > 1) for (size_t i = (hs >> 1) - 1; i != (size_t)-1; --i) down(i, hs);
	while (hs > 0) {
		HE top = h[0]; const AlMatch mm = m[top.mi]; const uint64_t r = top.x; const uint32_t rpos = (uint32_t)r >> 1;
		AlAnchor a;
		if ((r & 1) == (mm.q_pos & 1)) { a.x = (r & 0xffffffff00000000ULL) | rpos; a.y = (uint64_t)span << 32 | (mm.q_pos >> 1); }
		else { a.x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | rpos; a.y = (uint64_t)span << 32 | (uint32_t)(qlen - ((int)(mm.q_pos >> 1) + 1 - span) - 1); }
		a.y |= (uint64_t)(mm.flags & 0xff) << AL_SEED_SEG_SHIFT; if (mm.flags & (1u << 8)) a.y |= AL_SEED_TANDEM;
		if (!(a.x >> 63)) out[n_for++] = a; else out[n - (++n_rev)] = a;
		if (top.off < mm.n - 1) { ++top.off; top.x = pos[mm.off_lo + top.off]; h[0] = top; } else { h[0] = h[hs - 1]; --hs; }
		if (hs > 0) down(0, hs);
	}
	for (uint64_t j = 0; j < n_rev >> 1; ++j) std::swap(out[n - 1 - j], out[n - (n_rev - j)]);
}

int main(int argc, char **argv)
{
	const int n_frag = argc > 1 ? atoi(argv[1]) : 600; const unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 7u;
	const int big_lists = argc > 3 ? atoi(argv[3]) : 0, big_len = argc > 4 ? atoi(argv[4]) : 4000;   // timing mode: every fragment has big_lists lists of big_len positions
	std::mt19937_64 rng(seed);
	std::vector<uint64_t> pos; std::vector<AlMatch> match; std::vector<uint32_t> frag_first(n_frag + 1), rd_len(n_frag + 1, 300), frag_nm(n_frag), frag_na(n_frag), tie(n_frag, 1), list(n_frag);
	std::vector<uint64_t> mini_off(n_frag + 1), a_off(n_frag + 1);
	uint64_t na_tot = 0;
	for (int f = 0; f < n_frag; ++f) {
		frag_first[f] = f; list[f] = f; mini_off[f] = match.size(); a_off[f] = na_tot;
		const uint32_t n_m = big_lists ? (uint32_t)big_lists : f % 7 == 0 ? 64 + rng() % 63 : f % 7 == 1 ? 1 + rng() % 4 : 1 + rng() % 63;
		const uint64_t range = big_lists ? 3000000000ULL : 50 + rng() % 4000; uint32_t na = 0;
		std::vector<AlMatch> ms;
		for (uint32_t i = 0; i < n_m; ++i) {
			AlMatch mm;
			if (i > 0 && rng() % (big_lists ? 16 : 4) == 0) { mm = ms[rng() % i]; }   // the same k-mer again: an identical list
			else {
				const uint32_t len = big_lists ? (uint32_t)big_len : rng() % 5 == 0 ? 18 + rng() % 120 : 1 + rng() % 12;
				std::vector<uint64_t> v(len); for (auto &x : v) x = (rng() % 3) << 32 | (rng() % range);
				std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end());
				mm.off_lo = (uint32_t)pos.size(); mm.n = (uint32_t)v.size(); pos.insert(pos.end(), v.begin(), v.end());
			}
			mm.q_pos = (uint32_t)(rng() % 280) << 1 | (uint32_t)(rng() & 1); mm.flags = (uint32_t)(rng() & 1) | (uint32_t)(rng() & 1) << 8;
			ms.push_back(mm); na += mm.n;
		}
		match.insert(match.end(), ms.begin(), ms.end());
		frag_nm[f] = n_m; frag_na[f] = na; na_tot += na;
	}
	frag_first[n_frag] = n_frag; mini_off[n_frag] = match.size(); a_off[n_frag] = na_tot;
	std::vector<AlAnchor> exp(na_tot), got(na_tot);
	for (int f = 0; f < n_frag; ++f) ref_merge(pos.data(), match.data() + mini_off[f], frag_nm[f], frag_na[f], 300, 21, exp.data() + a_off[f]);
#define UP(T, name, vec) T *d_##name = nullptr; if (hipMalloc((void **)&d_##name, (vec).size() * sizeof(T) + 64) != hipSuccess || hipMemcpy(d_##name, (vec).data(), (vec).size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "upload failed\n"); return 2; }
	UP(uint64_t, pos, pos) UP(AlMatch, match, match) UP(uint32_t, ff, frag_first) UP(uint32_t, rl, rd_len) UP(uint32_t, nm, frag_nm) UP(uint32_t, na, frag_na) UP(uint32_t, tie, tie) UP(uint32_t, list, list)
	UP(uint64_t, mo, mini_off) UP(uint64_t, ao, a_off)
	AlAnchor *d_out = nullptr; unsigned long long *d_cnt = nullptr; uint32_t *d_n = nullptr; const uint32_t nl = (uint32_t)n_frag;
	if (hipMalloc((void **)&d_out, na_tot * 16 + 64) != hipSuccess || hipMalloc((void **)&d_cnt, 256) != hipSuccess || hipMalloc((void **)&d_n, 4) != hipSuccess) return 2;
	(void)hipMemset(d_out, 0xEE, na_tot * 16); (void)hipMemset(d_cnt, 0, 256); (void)hipMemcpy(d_n, &nl, 4, hipMemcpyHostToDevice);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, 0);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_anchor_heap_lanes<2, 16>), dim3(n_frag), dim3(64), 0, 0, d_pos, d_ff, d_rl, d_mo, d_match, d_nm, d_na, d_ao, d_out, d_tie, d_list, d_n, 63, d_cnt, 21);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_anchor_heap_lanes<1, 16>), dim3(n_frag), dim3(64), 0, 0, d_pos, d_ff, d_rl, d_mo, d_match, d_nm, d_na, d_ao, d_out, d_tie, d_list, d_n, -1, d_cnt, 21);
	(void)hipEventRecord(e1, 0);
	if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
	(void)hipMemcpy(got.data(), d_out, na_tot * 16, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int f = 0; f < n_frag; ++f) {
		const uint64_t o = a_off[f]; uint32_t i = 0;
		for (; i < frag_na[f]; ++i) if (got[o + i].x != exp[o + i].x || got[o + i].y != exp[o + i].y) break;
		if (i < frag_na[f]) { if (bad < 8) fprintf(stderr, "fragment %d (lists %u, anchors %u): first difference at %u: got %016llx %016llx, expected %016llx %016llx\n", f, frag_nm[f], frag_na[f], i,
			(unsigned long long)got[o + i].x, (unsigned long long)got[o + i].y, (unsigned long long)exp[o + i].x, (unsigned long long)exp[o + i].y); ++bad; }
	}
	float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
	printf("%d fragments, %llu anchors: %d differ; both kernels %.3f ms%s", n_frag, (unsigned long long)na_tot, bad, ms, big_lists ? "" : "\n");
	if (big_lists) printf(" = %.1f ns per pop of the longest fragment (%u anchors, %d lists)\n", 1e6 * ms / frag_na[0], frag_na[0], big_lists);
	return bad ? 1 : 0;
}
