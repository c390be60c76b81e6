// Do scalar readbacks serialise independent streams driven by different host threads?  Each thread: N x { thin kernel of ~T us on its
// own non-blocking stream; 8-byte device-to-host copy + stream synchronise }.  Destination pageable (stack) or page-locked.
// usage: readback_conc <threads> <iterations> <kernel_us> <pinned 0|1>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <thread>
#include <vector>
__global__ void k_spin(unsigned long long *out, long long cycles)
{
	const long long t0 = wall_clock64(); long long t;
	do { t = wall_clock64(); } while (t - t0 < cycles);
	if (threadIdx.x == 0) out[0] = (unsigned long long)t;
}
int main(int argc, char **argv)
{
	const int nt = argc > 1 ? atoi(argv[1]) : 2, it = argc > 2 ? atoi(argv[2]) : 200, us = argc > 3 ? atoi(argv[3]) : 500, pinned = argc > 4 ? atoi(argv[4]) : 0;
	(void)hipSetDevice(0);
	const long long cyc = (long long)us * 100;          // wall_clock64 runs at 100 MHz
	std::vector<std::thread> th; std::vector<double> el(nt);
	const auto T0 = std::chrono::steady_clock::now();
	for (int t = 0; t < nt; ++t) th.emplace_back([&, t] {
		(void)hipSetDevice(0);
		hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
		unsigned long long *d = nullptr, *hp = nullptr, stack_v = 0; (void)hipMalloc((void **)&d, 64); (void)hipHostMalloc((void **)&hp, 64, hipHostMallocDefault);
		const auto t0 = std::chrono::steady_clock::now();
		for (int i = 0; i < it; ++i) {
			hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, d, cyc);
			(void)hipMemcpyAsync(pinned ? hp : &stack_v, d, 8, hipMemcpyDeviceToHost, s);
			(void)hipStreamSynchronize(s);
		}
		el[t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		(void)hipFree(d); (void)hipHostFree(hp); (void)hipStreamDestroy(s);
	});
	for (auto &x : th) x.join();
	const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count();
	printf("%d thread(s) x %d iterations of a %d us kernel + 8-byte readback (%s): wall %.3f s, per iteration and thread %.1f us (ideal %d)\n", nt, it, us, pinned ? "page-locked" : "pageable", wall, el[0] / it * 1e6, us);
	return 0;
}
