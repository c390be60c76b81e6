// al_dev_net.h -- bitonic sorting network on 64-bit keys held in registers (round 2; a header of its own since round 6: the anchor sorts of
// al_kernels_seed.hip and chain_post's key sort of al_kernels_align.hip share it).
// PER keys per thread, element e = thread * PER + r.  Compare-exchange distances below PER are register-to-register, distances inside a
// wavefront are cross-lane moves (DPP quad permutes, ds_swizzle, ds_bpermute: no LDS storage, no barrier), and only the top log2(NT / 64)
// bits of the index go through an LDS exchange (sx: PER * NT words; untouched by a one-wavefront sort).  All comparators ascend: the first step of a
// merge level pairs e with e ^ (kk - 1).  d_bt_levels<PER, NT, 2>(k, sx, thread) sorts PER * NT keys.
#pragma once
#include <stdint.h>
template <int M> __device__ __forceinline__ uint32_t d_lane_xor32(uint32_t v, int lane)
{
	if (M == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);          // quad_perm [1,0,3,2]
	else if (M == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
	else if (M == 3) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x1B, 0xf, 0xf, true);     // quad_perm [3,2,1,0]
	else if (M < 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x1f | ((M & 31) << 10));   // bit mode: and 0x1f, or 0, xor M
	else return (uint32_t)__builtin_amdgcn_ds_bpermute(((lane ^ M) & 63) << 2, (int)v);
}
template <int M> __device__ __forceinline__ uint64_t d_lane_xor64(uint64_t v, int lane)
{
	return (uint64_t)d_lane_xor32<M>((uint32_t)v, lane) | (uint64_t)d_lane_xor32<M>((uint32_t)(v >> 32), lane) << 32;
}
__device__ __forceinline__ void d_cx64(uint64_t &a, uint64_t &b) { const bool sw = a > b; const uint64_t lo = sw ? b : a, hi = sw ? a : b; a = lo; b = hi; }

// one cross-thread step: partner thread T ^ TM, partner register PER-1-r (FLIP: first step of a level) or r
template <int PER, int NT, int TM, bool FLIP>
__device__ __forceinline__ void d_bt_x(uint64_t (&k)[PER], uint64_t *sx, const int T)
{
	constexpr int TOP = 1 << (31 - __builtin_clz((unsigned)TM));
	const bool lower = (T & TOP) == 0;
	uint64_t p[PER];
	if (TM < 64) {
		const int lane = T & 63;
#pragma unroll
		for (int r = 0; r < PER; ++r) p[r] = d_lane_xor64<(TM < 64 ? TM : 1)>(k[FLIP ? PER - 1 - r : r], lane);
	} else {                                                          // across wavefronts: register-major LDS tile, conflict-free both ways
		__syncthreads();
#pragma unroll
		for (int r = 0; r < PER; ++r) sx[r * NT + T] = k[r];
		__syncthreads();
#pragma unroll
		for (int r = 0; r < PER; ++r) p[r] = sx[(FLIP ? PER - 1 - r : r) * NT + (T ^ TM)];
	}
#pragma unroll
	for (int r = 0; r < PER; ++r) { const bool gt = k[r] > p[r]; k[r] = (gt == lower) ? p[r] : k[r]; }
}
template <int PER, int NT, int J>
__device__ __forceinline__ void d_bt_down(uint64_t (&k)[PER], uint64_t *sx, const int T)
{
	if constexpr (J >= 1) {
		if constexpr (J >= PER) d_bt_x<PER, NT, J / PER, false>(k, sx, T);
		else {
#pragma unroll
			for (int r = 0; r < PER; ++r) if ((r & J) == 0) d_cx64(k[r], k[r | J]);
		}
		d_bt_down<PER, NT, J / 2>(k, sx, T);
	}
}
template <int PER, int NT, int KK>
__device__ __forceinline__ void d_bt_levels(uint64_t (&k)[PER], uint64_t *sx, const int T)
{
	if constexpr (KK <= PER * NT) {
		if constexpr (KK <= PER) {
#pragma unroll
			for (int r = 0; r < PER; ++r) { const int r2 = r ^ (KK - 1); if (r < r2) d_cx64(k[r], k[r2]); }
		} else d_bt_x<PER, NT, KK / PER - 1, true>(k, sx, T);
		d_bt_down<PER, NT, KK / 4>(k, sx, T);
		d_bt_levels<PER, NT, KK * 2>(k, sx, T);
	}
}

