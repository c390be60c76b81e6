// Micro-benchmark: VALU issue rate of one SIMD on gfx950 with 1, 2, 4 wavefronts resident per SIMD, for the
// instruction kinds the extension DP is made of (v_add_u32, v_max_i32, DPP row_shr mov).  Prints cycles per wave-instruction per
// SIMD; MI355X_MICROARCH.md quotes 4 cycles for one wave alone and 2 cycles when more than one wave is resident.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue tools/micro/valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int KIND>
__global__ void __launch_bounds__(1024) k_issue(int iters, int *out, long long *cyc)
{
	int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	const long long t0 = clock64();
	for (int i = 0; i < iters; ++i) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {      // 8 independent chains x 8 = 64 instructions per iteration
			if (KIND == 0) {   // (written out: the compiler folds eight `a += i` into one multiply-add)
#define ADDU(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(i))
				ADDU(a0); ADDU(a1); ADDU(a2); ADDU(a3); ADDU(a4); ADDU(a5); ADDU(a6); ADDU(a7);
#undef ADDU
			}
			else if (KIND == 3) {
#define PKADD(x, y) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(y))
				PKADD(a0, a1); PKADD(a1, a2); PKADD(a2, a3); PKADD(a3, a4); PKADD(a4, a5); PKADD(a5, a6); PKADD(a6, a7); PKADD(a7, a0);
#undef PKADD
			}
			else if (KIND == 4) {   // the guide's 2-cycle figure is for floating-point FMA with more than one wave resident
#define FMA(x, y) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y))
				FMA(a0, a1); FMA(a1, a2); FMA(a2, a3); FMA(a3, a4); FMA(a4, a5); FMA(a5, a6); FMA(a6, a7); FMA(a7, a0);
#undef FMA
			}
			else if (KIND == 1) { a0 = max(a0, i ^ a1); a1 = max(a1, i ^ a2); a2 = max(a2, i ^ a3); a3 = max(a3, i ^ a4); a4 = max(a4, i ^ a5); a5 = max(a5, i ^ a6); a6 = max(a6, i ^ a7); a7 = max(a7, i ^ a0); }
			else {
				a0 = __builtin_amdgcn_update_dpp(a0, a1, 0x111, 0xf, 0xf, false); a1 = __builtin_amdgcn_update_dpp(a1, a2, 0x111, 0xf, 0xf, false);
				a2 = __builtin_amdgcn_update_dpp(a2, a3, 0x111, 0xf, 0xf, false); a3 = __builtin_amdgcn_update_dpp(a3, a4, 0x111, 0xf, 0xf, false);
				a4 = __builtin_amdgcn_update_dpp(a4, a5, 0x111, 0xf, 0xf, false); a5 = __builtin_amdgcn_update_dpp(a5, a6, 0x111, 0xf, 0xf, false);
				a6 = __builtin_amdgcn_update_dpp(a6, a7, 0x111, 0xf, 0xf, false); a7 = __builtin_amdgcn_update_dpp(a7, a0, 0x111, 0xf, 0xf, false);
			}
		}
	}
	const long long t1 = clock64();
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
	const int iters = 20000; int *out; long long *cyc;
	hipMalloc(&out, 256 * 2048 * 4); hipMalloc(&cyc, 2048 * 8);
	const char *names[5] = {"v_add_u32", "v_max_i32 + v_xor", "v_mov_dpp row_shr:1", "v_pk_add_u16", "v_fma_f32"};
	for (int kind = 0; kind < 5; ++kind)
		for (int wps = 1; wps <= 8; wps *= 2) {                 // waves per SIMD: one block of 4*wps waves per CU (256 CUs x 1 block)
			const int threads = 64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps, blocks = 256 * (64 * 4 * wps / threads);
			hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
			for (int rep = 0; rep < 2; ++rep) {
				hipEventRecord(e0);
				if (kind == 0) hipLaunchKernelGGL(k_issue<0>, dim3(blocks), dim3(threads), 0, 0, iters, out, cyc);
				else if (kind == 1) hipLaunchKernelGGL(k_issue<1>, dim3(blocks), dim3(threads), 0, 0, iters, out, cyc);
				else if (kind == 2) hipLaunchKernelGGL(k_issue<2>, dim3(blocks), dim3(threads), 0, 0, iters, out, cyc);
				else if (kind == 3) hipLaunchKernelGGL(k_issue<3>, dim3(blocks), dim3(threads), 0, 0, iters, out, cyc);
				else hipLaunchKernelGGL(k_issue<4>, dim3(blocks), dim3(threads), 0, 0, iters, out, cyc);
				hipEventRecord(e1); hipEventSynchronize(e1);
			}
			float ms = 0; hipEventElapsedTime(&ms, e0, e1);
			std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
			double c = 0; for (long long v : h) c += (double)v; c /= blocks;
			const double inst_per_wave = (double)iters * 64.0 * (kind == 1 ? 2.0 : 1.0);
			// wall-clock figure: every SIMD of the chip (1024) issues inst_per_wave * wps wave-instructions in `ms`; at the 2.4 GHz shader clock that is
			// ms * 2.4e6 cycles.  (The in-kernel column times wave 0 of a block only, which the oldest-first arbiter lets run unimpeded: it under-reads
			// with several waves per SIMD; the wall-clock column is the one the roofline.valu rows of bench.py use.)
			printf("%-22s waves/SIMD %d: wall %.3f ms = %.2f cycles per wave-instruction per SIMD at 2.4 GHz = %.3g wave-instructions/s chip-wide  (in-kernel clock of wave 0: %.2f)\n",
			       names[kind], wps, ms, ms * 2.4e6 / (inst_per_wave * wps), inst_per_wave * wps * 1024.0 / (ms * 1e-3), c / (inst_per_wave * wps));
		}
	return 0;
}
