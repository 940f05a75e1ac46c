// What do LDS reads beyond a workgroup's allocation return, with two 1024-thread workgroups per CU, right after a kernel that
// filled ALL of a CU's LDS with a pattern?  The score kernel's unchecked sweep (chain_kernels.hip, sweep_block_lut2_free)
// relies on 0.  Addresses are given as VGPR + the instruction's offset field, like the kernel's gathers.
//   hipcc --offload-arch=gfx950 -O2 lds_oob.hip -o lds_oob && ./lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(1024) void k_fill_all(unsigned *sink)
{
	extern __shared__ unsigned dyn[];
	const unsigned words = 160 * 1024 / 4;
	for (unsigned i = threadIdx.x; i < words; i += blockDim.x) dyn[i] = 0xdead0000u | (i & 0xffffu);
	__syncthreads();
	if (threadIdx.x == 0) sink[blockIdx.x] = dyn[(blockIdx.x * 7u) % words];
}

constexpr unsigned OFFSET_FIELD = 58880;
// out[wg * n + k] = value read at byte address probes[k]; mode 0: address in the VGPR, 1: address - 58 880 in the VGPR + offset field
__global__ __launch_bounds__(1024) void k_probe(unsigned alloc_bytes, const unsigned *probes, int n, int mode, unsigned *out)
{
	extern __shared__ unsigned dyn[];
	for (unsigned i = threadIdx.x; i < alloc_bytes / 4; i += blockDim.x) dyn[i] = 0x5a000000u | (blockIdx.x << 12 & 0xfff000u) | (i & 0xfffu);
	__syncthreads();
	if (threadIdx.x < (unsigned)n) {
		unsigned v;
		const unsigned a = probes[threadIdx.x];
		if (mode == 0) asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
		else asm volatile("ds_read_b32 %0, %1 offset:58880\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a - OFFSET_FIELD) : "memory");
		out[blockIdx.x * n + threadIdx.x] = v;
	}
	// keep the workgroup resident for a while so that two share a CU
	for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(8);
}

int main()
{
	hipFuncSetAttribute((const void*)k_fill_all, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	const int wgs = 512;
	unsigned *sink; hipMalloc(&sink, wgs * 4);
	const unsigned allocs[] = { 79360u, 78848u, 81920u, 1024u, 20480u };
	for (unsigned alloc : allocs) {
		std::vector<unsigned> probes;
		for (unsigned d : { 4u }) probes.push_back(alloc - d);                    // last word inside
		for (unsigned d = 0; d < 6144; d += 252) probes.push_back(alloc + d);     // just beyond, across every plausible granule
		for (unsigned a : { 81916u, 81920u, 98304u, 131072u, 163836u, 163840u, 200000u, 1u << 20, 1u << 25, 0x40000000u }) if (a >= alloc) probes.push_back(a);
		const int n = (int)probes.size();
		unsigned *d_p, *d_o; hipMalloc(&d_p, n * 4); hipMalloc(&d_o, (size_t)wgs * n * 4);
		hipMemcpy(d_p, probes.data(), n * 4, hipMemcpyHostToDevice);
		for (int mode = 0; mode < 2; ++mode) {
			if (mode == 1 && alloc < OFFSET_FIELD) continue;
			hipLaunchKernelGGL(k_fill_all, dim3(256), dim3(1024), 160 * 1024, 0, sink);
			hipMemset(d_o, 0xff, (size_t)wgs * n * 4);
			hipLaunchKernelGGL(k_probe, dim3(wgs), dim3(1024), alloc, 0, alloc, d_p, n, mode, d_o);
			if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed for alloc %u\n", alloc); continue; }
			std::vector<unsigned> o((size_t)wgs * n);
			hipMemcpy(o.data(), d_o, o.size() * 4, hipMemcpyDeviceToHost);
			printf("alloc %u bytes, %s: ", alloc, mode ? "VGPR + offset:58880" : "VGPR address");
			for (int k = 0; k < n; ++k) {
				int nonzero = 0; unsigned sample = 0;
				for (int w = 0; w < wgs; ++w) if (o[(size_t)w * n + k] != 0) { ++nonzero; sample = o[(size_t)w * n + k]; }
				if (k == 0) printf("[%u inside: %d/%d nonzero] ", probes[k], nonzero, wgs);
				else if (nonzero) printf("[+%u: %d nonzero e.g. 0x%x] ", probes[k] - alloc, nonzero, sample);
			}
			printf("(all other probes beyond the allocation read 0 in all %d workgroups)\n", wgs);
		}
		hipFree(d_p); hipFree(d_o);
	}
	return 0;
}
