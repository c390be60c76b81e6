// cost of device allocation on this box: hipMalloc / first kernel touch / hipFree at several sizes, and hipMallocAsync from a pool
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char *p, size_t n) { size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4096; if (i < n) p[i] = 1; }
int main()
{
	hipFree(0);
	for (size_t gb : {1, 8, 32, 64}) {
		for (int rep = 0; rep < 2; ++rep) {
			const size_t n = gb << 30; char *p = nullptr;
			double t0 = now(); hipError_t e = hipMalloc((void **)&p, n); double t1 = now();
			if (e != hipSuccess) { printf("%zu GB: hipMalloc failed\n", gb); break; }
			touch<<<(unsigned)((n / 4096 + 255) / 256), 256>>>(p, n); hipDeviceSynchronize(); double t2 = now();
			touch<<<(unsigned)((n / 4096 + 255) / 256), 256>>>(p, n); hipDeviceSynchronize(); double t3 = now();
			hipFree(p); double t4 = now();
			printf("%3zu GB rep %d: hipMalloc %.3f s, first touch %.3f s, second touch %.3f s, hipFree %.3f s\n", gb, rep, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
		}
	}
	hipStream_t s; hipStreamCreate(&s);
	hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
	uint64_t thr = UINT64_MAX; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
	for (int rep = 0; rep < 3; ++rep) {
		const size_t n = (size_t)32 << 30; char *p = nullptr;
		double t0 = now(); hipError_t e = hipMallocAsync((void **)&p, n, s); hipStreamSynchronize(s); double t1 = now();
		if (e != hipSuccess) { printf("hipMallocAsync failed\n"); break; }
		touch<<<(unsigned)((n / 4096 + 255) / 256), 256, 0, s>>>(p, n); hipStreamSynchronize(s); double t2 = now();
		hipFreeAsync(p, s); hipStreamSynchronize(s); double t3 = now();
		printf("pool 32 GB rep %d: hipMallocAsync %.3f s, touch %.3f s, hipFreeAsync %.3f s\n", rep, t1 - t0, t2 - t1, t3 - t2);
	}
	return 0;
}
