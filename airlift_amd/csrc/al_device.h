// al_device.h -- small device-side helpers shared by the kernel files
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AL_HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	fprintf(stderr, "[airlift] HIP error %s at %s:%d: %s\n", hipGetErrorName(e_), __FILE__, __LINE__, hipGetErrorString(e_)); return -1; } } while (0)

// 4-bit packed base fetch (mm_seq4_get, mmpriv.h:29)
__device__ __forceinline__ uint32_t d_seq4(const uint32_t *__restrict__ S, uint64_t i) { return S[i >> 3] >> ((i & 7) << 2) & 0xf; }

// ---- float arithmetic that must match the reference's x86-64 SSE results bit for bit ----------
// * every float op individually rounded (no FMA contraction): use the _rn intrinsics
// * division: double divide of two floats rounded once to float == correctly rounded float divide
// * logf: the reference calls glibc's logf on a small discrete set of arguments (k / match_sc, or an integer k).
//   The host tabulates those with ITS libm at context creation and the device looks them up; an argument outside
//   the table falls back to the device logf and bumps a counter that the host reports (never silent).
#define AL_LOGTAB_N 16384                 // smallest table; the host sizes it to the longest read of the batch (scores <= match_sc * read length)
__device__ __forceinline__ float al_fdiv(float a, float b) { return (float)((double)a / (double)b); }
struct AlLogTab { const float *t; int n; unsigned long long *miss; };   // t[0 .. n): logf(k / match_sc), t[n .. 2n): logf(k)
__device__ __forceinline__ float al_logf_q(const AlLogTab &lt, int k)       // logf((float)k / match_sc)
{
	if (k >= 0 && k < lt.n) return lt.t[k];
	atomicAdd(lt.miss, 1ULL); return 0.0f;
}
__device__ __forceinline__ float al_logf_i(const AlLogTab &lt, int k)       // logf((float)k)
{
	if (k >= 0 && k < lt.n) return lt.t[lt.n + k];
	atomicAdd(lt.miss, 1ULL); return 0.0f;
}
