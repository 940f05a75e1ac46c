// Does the cost of an LDS gather (ds_read_b32, a different address in every lane) depend on WHERE in the workgroup's LDS the
// addresses lie, or on the instruction's offset field?  Two 1024-thread workgroups per CU with 79 360 bytes each, like the score kernel.
//   hipcc --offload-arch=gfx950 -O2 lds_region.hip -o lds_region && ./lds_region
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OFFSET>
__global__ __launch_bounds__(1024) void k_gather(unsigned base, unsigned span_mask, int iters, unsigned *out)
{
	extern __shared__ unsigned dyn[];
	for (unsigned i = threadIdx.x; i < 79360 / 4; i += blockDim.x) dyn[i] = i * 2654435761u;
	__syncthreads();
	unsigned a = (threadIdx.x * 2654435761u) >> 7, acc = 0;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			unsigned v;
			const unsigned addr = base - OFFSET + ((a + u * 977u) & span_mask & ~3u);
			if (OFFSET == 0) asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
			else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFFSET));
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			acc += v;
		}
		a = a * 1664525u + 1013904223u + (acc & 1u);
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OFFSET>
static void run(const char *what, unsigned base, unsigned span_mask, unsigned *out)
{
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int iters = 4000;
	hipLaunchKernelGGL(k_gather<OFFSET>, dim3(512), dim3(1024), 79360, 0, base, span_mask, 10, out);
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(k_gather<OFFSET>, dim3(512), dim3(1024), 79360, 0, base, span_mask, iters, out);
	(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
	float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
	// per CU: 2 workgroups x 16 waves x iters x 8 gathers
	const double per_cu = 2.0 * 16 * iters * 8;
	printf("%-58s %8.3f ms  -> %.2f cycles of the CU per wave-gather (2.4 GHz)\n", what, ms, ms * 1e-3 * 2.4e9 / per_cu);
}

int main()
{
	(void)hipFuncSetAttribute((const void*)k_gather<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	(void)hipFuncSetAttribute((const void*)k_gather<58880>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	unsigned *out; (void)hipMalloc(&out, 512 * 1024 * 4);
	run<0>("table at 0 .. 16 KB, address in the VGPR", 0u, 0x3fffu, out);
	run<0>("table at 32 .. 48 KB, address in the VGPR", 32768u, 0x3fffu, out);
	run<0>("table at 48 .. 64 KB, address in the VGPR", 49152u, 0x3fffu, out);
	run<0>("table at 58 880 .. +16 KB (crosses 64 KB), VGPR", 58880u, 0x3fffu, out);
	run<58880>("table at 58 880 .. +16 KB (crosses 64 KB), offset field", 58880u, 0x3fffu, out);
	run<0>("table at 62 976 .. +16 KB (all but 2.5 KB above 64 KB), VGPR", 62976u, 0x3fffu, out);
	run<0>("table at 58 880 .. +2 KB (below 64 KB), VGPR", 58880u, 0x7ffu, out);
	run<58880>("table at 58 880 .. +2 KB (below 64 KB), offset field", 58880u, 0x7ffu, out);
	run<0>("table at 0 .. 2 KB, VGPR", 0u, 0x7ffu, out);
	return 0;
}
