// lone_wave.hip -- what does ONE wave cost per instruction when it has a SIMD to itself?  (Round 4: the in-tile steps of the score kernel
// are issued by one wave in order; is its pace set by dependences or by a per-wave issue interval?)  One 64-thread workgroup per CU,
// every kernel a loop of 64 copies of a pattern; cycles from s_memtime around the loop, per pattern instance.
// Build: hipcc --offload-arch=gfx950 -O2 lone_wave.hip -o lone_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define REP64(...) REP8(REP8(__VA_ARGS__))

#define KERNEL(name, ...)                                                              \
	__global__ __launch_bounds__(64) void name(long long *out, int iters, int a, int b)      \
	{                                                                                   \
		__shared__ int lds[4096];                                                       \
		for (int k = threadIdx.x; k < 4096; k += 64) lds[k] = (k * 4 + 64) & 16383;               \
		__syncthreads();                                                                \
		int v0 = threadIdx.x + a, v1 = v0 ^ b, v2 = v1 + 3, v3 = v2 * 5, v4 = a, v5 = b, v6 = 7, v7 = 9; \
		int addr = (threadIdx.x * 4) & 16383;                                                    \
		long long t0 = __builtin_amdgcn_s_memtime();                                    \
		for (int it = 0; it < iters; ++it) { REP64(__VA_ARGS__) }                       \
		long long t1 = __builtin_amdgcn_s_memtime();                                    \
		if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                \
		if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + addr == 0x12345678) out[blockIdx.x + 1024] = lds[v0 & 4095];                   \
	}

KERNEL(k_dep_add,    asm volatile("v_add_u32 %0, %1, %0" : "+v"(v0) : "v"(v1));)
KERNEL(k_ind_add4,   asm volatile("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));)
KERNEL(k_dep_max,    asm volatile("v_max_i32 %0, %1, %0" : "+v"(v0) : "v"(v1));)
KERNEL(k_ind_max4,   asm volatile("v_max_i32 %0, %4, %0\n v_max_i32 %1, %4, %1\n v_max_i32 %2, %4, %2\n v_max_i32 %3, %4, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));)
KERNEL(k_dep_salu,   asm volatile("s_add_i32 s20, s20, 1" ::: "s20", "scc");)
KERNEL(k_ind_salu4,  asm volatile("s_add_i32 s20, s20, 1\n s_add_i32 s21, s21, 1\n s_add_i32 s22, s22, 1\n s_add_i32 s23, s23, 1" ::: "s20", "s21", "s22", "s23", "scc");)
KERNEL(k_valu_salu,  asm volatile("v_add_u32 %0, %1, %0\n s_add_i32 s20, s20, 1" : "+v"(v0) : "v"(v1) : "s20", "scc");)
// the in-tile step: lane 5's value -> scalar -> or -> add -> max (dependent through v0)
KERNEL(k_step,       asm volatile("v_readlane_b32 s20, %0, 5\n s_or_b32 s20, s20, 0x7f\n s_nop 0\n v_add_u32 %1, s20, %2\n v_max_i32 %0, %1, %0" : "+v"(v0), "+v"(v1) : "v"(v2) : "s20", "scc");)
// the same with four independent full-rate VALU instructions in between (are they free?)
KERNEL(k_step_fill4, asm volatile("v_readlane_b32 s20, %0, 5\n v_add_u32 %3, %2, %3\n s_or_b32 s20, s20, 0x7f\n v_add_u32 %4, %2, %4\n v_add_u32 %1, s20, %2\n v_add_u32 %5, %2, %5\n v_max_i32 %0, %1, %0\n v_add_u32 %6, %2, %6" : "+v"(v0), "+v"(v1) , "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6) :: "s20", "scc");)
// readlane -> v_add directly (no scalar op between)
KERNEL(k_step_noor,  asm volatile("v_readlane_b32 s20, %0, 5\n s_nop 0\n v_add_u32 %1, s20, %2\n v_max_i32 %0, %1, %0" : "+v"(v0), "+v"(v1) : "v"(v2) : "s20");)
// a DPP form of the broadcast: row_bcast / v_mov_dpp cannot address an arbitrary lane; ds_bpermute can (LDS crossbar, no memory)
KERNEL(k_step_bperm, asm volatile("ds_bpermute_b32 %1, %3, %0\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, %2\n v_max_i32 %0, %1, %0" : "+v"(v0), "+v"(v1) : "v"(v2), "v"(v4));)
// v_cmp -> s_and -> v_cndmask: the old step's tail
KERNEL(k_cmp_and_cnd, asm volatile("v_cmp_gt_i32 s[20:21], %1, %0\n s_and_b64 s[20:21], s[20:21], s[22:23]\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v0) : "v"(v1) : "s20", "s21", "scc");)
// LDS round trips of a lone wave: dependent ds_read chain (address from the value just read)
KERNEL(k_lds_chain,  asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(addr));)
// two independent reads then wait
KERNEL(k_lds_2ind,   { int t; int u; asm volatile("ds_read_b32 %0, %2\n ds_read_b32 %1, %2 offset:256\n s_waitcnt lgkmcnt(0)" : "=v"(t), "=v"(u) : "v"(addr)); v1 ^= t + u; })
// 8 independent reads then wait
KERNEL(k_lds_8ind,   { int t0; int t1; int t2; int t3; int t4; int t5; int t6; int t7; asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)" : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3), "=v"(t4), "=v"(t5), "=v"(t6), "=v"(t7) : "v"(addr)); v1 ^= t0 + t1 + t2 + t3 + t4 + t5 + t6 + t7; })
// half-rate dependent pair typical of the row arithmetic
KERNEL(k_dep_sad_min, asm volatile("v_sad_u32 %0, %1, %0, 0\n v_min_u32 %0, %2, %0" : "+v"(v0) : "v"(v1), "v"(v2));)
KERNEL(k_ind_sad4,   asm volatile("v_sad_u32 %0, %4, %0, 0\n v_sad_u32 %1, %4, %1, 0\n v_sad_u32 %2, %4, %2, 0\n v_sad_u32 %3, %4, %3, 0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));)
KERNEL(k_readlane4,  asm volatile("v_readlane_b32 s20, %0, 5\n v_readlane_b32 s21, %1, 6\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 8" :: "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "s20", "s21", "s22", "s23");)
KERNEL(k_nop,        asm volatile("s_nop 0");)

struct Case { const char *name; void (*fn)(long long*, int, int, int); int per; };

int main(int argc, char **argv)
{
	int waves = argc > 1 ? atoi(argv[1]) : 1;   // workgroups per CU (each one wave)
	hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
	const int n_cu = prop.multiProcessorCount;
	long long *d; (void)hipMalloc(&d, 4096 * sizeof(long long));
	std::vector<Case> cases = {
		{"dependent v_add_u32", k_dep_add, 1}, {"4 independent v_add_u32", k_ind_add4, 4}, {"dependent v_max_i32", k_dep_max, 1}, {"4 independent v_max_i32", k_ind_max4, 4},
		{"dependent s_add", k_dep_salu, 1}, {"4 independent s_add", k_ind_salu4, 4}, {"v_add + s_add alternating", k_valu_salu, 2},
		{"step: readlane, s_or, nop, v_add, v_max", k_step, 1}, {"step + 4 independent v_add", k_step_fill4, 1}, {"step without s_or", k_step_noor, 1},
		{"step by ds_bpermute", k_step_bperm, 1}, {"v_cmp, s_and, v_cndmask", k_cmp_and_cnd, 1},
		{"dependent ds_read chain", k_lds_chain, 1}, {"2 independent ds_read + wait", k_lds_2ind, 1}, {"8 independent ds_read + wait", k_lds_8ind, 1},
		{"dependent v_sad, v_min", k_dep_sad_min, 2}, {"4 independent v_sad", k_ind_sad4, 4}, {"4 v_readlane", k_readlane4, 4}, {"s_nop 0", k_nop, 1},
	};
	setvbuf(stdout, nullptr, _IONBF, 0);
	printf("%s, %d CUs; %d one-wave workgroup(s) per CU; cycles (s_memtime ticks) per pattern, and per instruction where the pattern has several\n", prop.name, n_cu, waves);
	setvbuf(stdout, nullptr, _IONBF, 0);
	for (auto &c : cases) {
		const int iters = 200;
		printf("%-44s ", c.name);
		hipLaunchKernelGGL(c.fn, dim3(n_cu * waves), dim3(64), 0, 0, d, iters, 1, 2);
		hipLaunchKernelGGL(c.fn, dim3(n_cu * waves), dim3(64), 0, 0, d, iters, 1, 2);
		(void)hipDeviceSynchronize();
		std::vector<long long> h(n_cu * waves);
		(void)hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
		double sum = 0; for (auto v : h) sum += (double)v;
		const double per = sum / h.size() / (iters * 64.0);
		printf("%8.2f ticks per pattern  %8.2f per instruction\n", per, per / c.per);
	}
	return 0;
}
