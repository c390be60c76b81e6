// al_device.h -- small device-side helpers shared by the kernel files
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AL_HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	fprintf(stderr, "[airlift] HIP error %s at %s:%d: %s\n", hipGetErrorName(e_), __FILE__, __LINE__, hipGetErrorString(e_)); return -1; } } while (0)

// 4-bit packed base fetch (mm_seq4_get, mmpriv.h:29)
__device__ __forceinline__ uint32_t d_seq4(const uint32_t *__restrict__ S, uint64_t i) { return S[i >> 3] >> ((i & 7) << 2) & 0xf; }
