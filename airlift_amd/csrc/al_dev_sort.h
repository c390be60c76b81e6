// al_dev_sort.h -- device restatement of the reference's sort (klib ksort.h KRADIX_SORT_INIT), shared by the chain
// kernels and the region bookkeeping.
// AL_SORT_HOST compiles the same code for the CPU (tests/test_dev_sort_cpu.py checks it against the oracle's and the reference's sort).
#pragma once
#include <stdint.h>
#ifdef AL_SORT_HOST
#define AL_D static inline
#else
#include <hip/hip_runtime.h>
#ifndef AL_D
#define AL_D __device__ __forceinline__
#endif
#endif

// ---- sorts.  The reference uses a stable insertion sort up to 64 elements and an unstable in-place MSD radix sort
// above (ksort.h:105-151).  Both are restated step for step, so the order the radix permutation leaves among
// equal keys is the reference's too.  d_rs_sort works through an accessor (key(i), keyof(e), get(i), set(i,e)) so
// the same code sorts anchors, pairing entries and the chain permutation.
#define AL_RS_SCRATCH (2 * (512 + 4 * 8) + 4 * 256)   // bytes of scratch d_rs_sort needs: bucket bounds + one frame per digit (+ the 32-bit histogram of the wavefront form)
template <class A>
AL_D void d_rs_isort(A &acc, int beg, int end)
{   // rs_insertsort, ksort.h:105-115
	for (int i = beg + 1; i < end; ++i)
		if (acc.key(i) < acc.key(i - 1)) {
			typename A::E t = acc.get(i); const uint64_t kt = acc.keyof(t); int j = i;
			for (; j > beg && kt < acc.key(j - 1); --j) acc.set(j, acc.get(j - 1));
			acc.set(j, t);
		}
}
template <class A>
AL_D bool d_rs_sort(A &acc, int n, uint16_t *scr)
{   // radix_sort + rs_sort, ksort.h:116-151 (8-bit digits from bit 56 down).  The reference recurses into every bucket of more
	// than 64 elements with a fresh bucket table on the C stack; here one table is shared: a frame per digit remembers
	// (range, shift, next bucket), and on return from a child the parent's bounds are recounted from its (already
	// partitioned) range.  Buckets are disjoint, so the order they are visited in does not change the result.
	// Returns true if the order could not be reproduced (more than 65535 elements: stable order then, the caller counts it).
	if (n <= 64) { d_rs_isort(acc, 0, n); return false; }
	if (n > 65535) { d_rs_isort(acc, 0, n); return true; }
	uint16_t *bb = scr, *be = scr + 256, *stk = scr + 512;                 // frame: beg, end, shift, next bucket + 1 (0 = not partitioned yet)
	int sp = 1;
	stk[0] = 0; stk[1] = (uint16_t)n; stk[2] = 56; stk[3] = 0;
	while (sp > 0) {
		uint16_t *fr = stk + 4 * (sp - 1);
		const int beg = fr[0], end = fr[1];
		if (fr[3] == 0) {
			// A digit that all keys of the range share gives one bucket holding the whole range, an identity permutation and a
			// recursion into the same range at the next digit (or nothing at the last digit): go straight to the highest digit
			// in which the keys differ.  Equal keys throughout: every remaining level is such an identity.
			const uint64_t k0 = acc.key(beg); uint64_t diff = 0;
			for (int i = beg + 1; i < end; ++i) diff |= acc.key(i) ^ k0;
			if (diff == 0) { --sp; continue; }
			const int top = (63 - __builtin_clzll(diff)) & ~7;
			if (top < (int)fr[2]) fr[2] = (uint16_t)top;
		}
		const int s = fr[2];
		for (int k = 0; k < 256; ++k) bb[k] = be[k] = (uint16_t)beg;
		for (int i = beg; i < end; ++i) ++be[acc.key(i) >> s & 255];
		for (int k = 1; k < 256; ++k) { be[k] = (uint16_t)(be[k] + be[k - 1] - beg); bb[k] = be[k - 1]; }
		int k0 = 0;
		if (fr[3] == 0) {
			for (int k = 0; k < 256;) {
				if (bb[k] != be[k]) {
					int l = (int)(acc.key(bb[k]) >> s & 255);
					if (l != k) {
						typename A::E tmp = acc.get(bb[k]);
						do {
							const typename A::E swap = tmp; tmp = acc.get(bb[l]); acc.set(bb[l]++, swap);
							l = (int)(acc.keyof(tmp) >> s & 255);
						} while (l != k);
						acc.set(bb[k]++, tmp);
					} else ++bb[k];
				} else ++k;
			}
			bb[0] = (uint16_t)beg; for (int k = 1; k < 256; ++k) bb[k] = be[k - 1];
		} else k0 = fr[3] - 1;                                                // back from a child: bounds recounted above, go on behind it
		bool descended = false;
		if (s) {
			const int s2 = s > 8 ? s - 8 : 0;
			for (int k = k0; k < 256; ++k) {
				const int sz = be[k] - bb[k];
				if (sz > 64) {
					fr[3] = (uint16_t)(k + 2);
					uint16_t *ch = stk + 4 * sp; ch[0] = bb[k]; ch[1] = be[k]; ch[2] = (uint16_t)s2; ch[3] = 0; ++sp;   // depth <= 8: one frame per digit
					descended = true; break;
				} else if (sz > 1) d_rs_isort(acc, bb[k], be[k]);
			}
		}
		if (!descended) --sp;
	}
	return false;
}

#ifndef AL_SORT_HOST
// The same sort run by a whole wavefront (all 64 lanes call it, converged).  What depends on the order of operations -- the in-place
// cycle-leader permutation of a range into its buckets -- stays with lane 0, step for step as above.  Everything else does not: the
// digit scan and the bucket counts (recounted every time the walk comes back to a range from one of its children) are a strided pass
// with LDS atomics, and the insertion sorts of the buckets of at most 64 elements -- disjoint ranges -- run a lane per bucket.
// On thousands of chains with a few equal keys the serial form spent milliseconds recounting; this one is bounded by the permutation.
template <class A>
__device__ bool d_rs_sort_wave(A &acc, int n, uint16_t *scr, const int lane)
{
#define AL_WSYNC() do { __threadfence_block(); __builtin_amdgcn_wave_barrier(); } while (0)
	if (n <= 64 || n > 65535) {
		int r = 0;
		if (lane == 0) r = d_rs_sort(acc, n, scr) ? 1 : 0;
		AL_WSYNC();
		return __shfl(r, 0) != 0;
	}
	uint16_t *bb = scr, *be = scr + 256, *stk = scr + 512; uint32_t *hist = (uint32_t *)(scr + 512 + 32);
	if (lane == 0) { stk[0] = 0; stk[1] = (uint16_t)n; stk[2] = 56; stk[3] = 0; }
	int sp = 1;                                                              // wave-uniform: every decision below is made on reduced values
	while (sp > 0) {
		AL_WSYNC();
		uint16_t *fr = stk + 4 * (sp - 1);
		const int beg = fr[0], end = fr[1], f3 = fr[3]; int s = fr[2];
		if (f3 == 0) {
			const uint64_t k0 = acc.key(beg); uint64_t diff = 0;
			for (int i = beg + 1 + lane; i < end; i += 64) diff |= acc.key(i) ^ k0;
			for (int d = 32; d > 0; d >>= 1) diff |= (uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)diff, d) | (uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(diff >> 32), d) << 32;
			if (diff == 0) { --sp; continue; }
			const int top = (63 - __builtin_clzll(diff)) & ~7;
			if (top < s) { s = top; if (lane == 0) fr[2] = (uint16_t)top; }
		}
		for (int k = lane; k < 256; k += 64) hist[k] = 0;
		AL_WSYNC();
		for (int i = beg + lane; i < end; i += 64) atomicAdd(&hist[acc.key(i) >> s & 255], 1u);
		AL_WSYNC();
		if (lane == 0) { uint32_t run = (uint32_t)beg; for (int k = 0; k < 256; ++k) { bb[k] = (uint16_t)run; run += hist[k]; be[k] = (uint16_t)run; } }
		AL_WSYNC();
		int k0 = 0;
		if (f3 == 0) {
			if (lane == 0) {
				for (int k = 0; k < 256;) {
					if (bb[k] != be[k]) {
						int l = (int)(acc.key(bb[k]) >> s & 255);
						if (l != k) {
							typename A::E tmp = acc.get(bb[k]);
							do {
								const typename A::E swap = tmp; tmp = acc.get(bb[l]); acc.set(bb[l]++, swap);
								l = (int)(acc.keyof(tmp) >> s & 255);
							} while (l != k);
							acc.set(bb[k]++, tmp);
						} else ++bb[k];
					} else ++k;
				}
				bb[0] = (uint16_t)beg; for (int k = 1; k < 256; ++k) bb[k] = be[k - 1];
			}
			AL_WSYNC();
		} else k0 = f3 - 1;                                                   // back from a child: bounds recounted above, go on behind it
		bool descended = false;
		if (s) {
			const int s2 = s > 8 ? s - 8 : 0;
			for (int kb = k0; kb < 256 && !descended; kb += 64) {
				const int k = kb + lane;
				const int b0 = k < 256 ? (int)bb[k] : 0, e0 = k < 256 ? (int)be[k] : 0, sz = e0 - b0;
				const unsigned long long big = __ballot(sz > 64);
				const int first_big = big ? __ffsll((long long)big) - 1 : 64;
				if (lane < first_big && sz > 1) d_rs_isort(acc, b0, e0);
				if (big) {
					if (lane == 0) {
						const int kk = kb + first_big;
						fr[3] = (uint16_t)(kk + 2);
						uint16_t *ch = stk + 4 * sp; ch[0] = bb[kk]; ch[1] = be[kk]; ch[2] = (uint16_t)s2; ch[3] = 0;   // depth <= 8: one frame per digit
					}
					++sp; descended = true;
				}
			}
		}
		if (!descended) --sp;
	}
	AL_WSYNC();
	return false;
#undef AL_WSYNC
}
#endif
