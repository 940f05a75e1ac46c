// valu_rate.hip -- how many SIMD cycles does one wave64 instruction of each kind cost on gfx950 when 8 waves/SIMD issue it
// back to back?  (Input for DESIGN.md "what bounds the score kernel".)  Build: hipcc --offload-arch=gfx950 -O2 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define KERNEL(name, ...)                                                              \
	__global__ __launch_bounds__(256) void name(int *out, int iters, int a, int b)      \
	{                                                                                   \
		int v0 = threadIdx.x + a, v1 = v0 ^ b, v2 = v1 + 3, v3 = v2 * 5, v4 = a, v5 = b, v6 = 7, v7 = 9; \
		float f0 = v0, f1 = v1, f2 = v2, f3 = v3;                                       \
		int s0 = 0;                                                                     \
		for (int it = 0; it < iters; ++it) { REP64(__VA_ARGS__) }                              \
		out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + (int)(f0 + f1 + f2 + f3) + s0; \
	}

KERNEL(k_add_u32,  asm volatile("v_add_u32 %0, %1, %0\n v_add_u32 %2, %3, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_sub_u32,  asm volatile("v_sub_u32 %0, %1, %0\n v_sub_u32 %2, %3, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_min_u32,  asm volatile("v_min_u32 %0, %1, %0\n v_min_u32 %2, %3, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_min3_i32, asm volatile("v_min3_i32 %0, %1, %0, %2\n v_min3_i32 %2, %3, %2, %0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_sad_u32,  asm volatile("v_sad_u32 %0, %1, %0, 0\n v_sad_u32 %2, %3, %2, 0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_lshl_add, asm volatile("v_lshl_add_u32 %0, %1, 2, %0\n v_lshl_add_u32 %2, %3, 2, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_add3,     asm volatile("v_add3_u32 %0, %1, %0, %2\n v_add3_u32 %2, %3, %2, %0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_cndmask,  asm volatile("v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %2, %3, %2, vcc" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) :: "vcc");)
KERNEL(k_cmp_u32,  asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 vcc, %2, %3" :: "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "vcc");)
KERNEL(k_cmp_e64,  asm volatile("v_cmp_gt_u32 s[20:21], %0, %1\n v_cmp_gt_u32 s[22:23], %2, %3" :: "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "s20", "s21", "s22", "s23");)
KERNEL(k_readlane, asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5" :: "v"(v0), "v"(v1) : "s20", "s21");)
KERNEL(k_mov,      asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_add_f32,  asm volatile("v_add_f32 %0, %1, %0\n v_add_f32 %2, %3, %2" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
KERNEL(k_fma_f32,  asm volatile("v_fma_f32 %0, %1, %0, %2\n v_fma_f32 %2, %3, %2, %0" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
KERNEL(k_min_f32,  asm volatile("v_min_f32 %0, %1, %0\n v_min_f32 %2, %3, %2" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
KERNEL(k_cmp_f32,  asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %2, %3" :: "v"(f0), "v"(f1), "v"(f2), "v"(f3) : "vcc");)
KERNEL(k_cvt_i2f,  asm volatile("v_cvt_f32_i32 %0, %1\n v_cvt_f32_i32 %2, %3" : "+v"(f0), "+v"(v1), "+v"(f2), "+v"(v3));)
KERNEL(k_pk_add_f32, asm volatile("v_pk_add_f32 %0, %1, %0\n v_pk_add_f32 %2, %3, %2" : "+v"(*(double*)&v0), "+v"(*(double*)&v2), "+v"(*(double*)&v4), "+v"(*(double*)&v6));)
KERNEL(k_max_i32,  asm volatile("v_max_i32 %0, %1, %0\n v_max_i32 %2, %3, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_and_or,   asm volatile("v_and_or_b32 %0, %1, %0, %2\n v_and_or_b32 %2, %3, %2, %0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_sub_sgpr, asm volatile("v_sub_u32 %0, s20, %0\n v_sub_u32 %1, s21, %1" : "+v"(v0), "+v"(v2));)
KERNEL(k_subrev_sgpr, asm volatile("v_subrev_u32 %0, s20, %1\n v_subrev_u32 %2, s21, %3" : "=v"(v0), "+v"(v1), "=v"(v2), "+v"(v3));)
KERNEL(k_cmp_sgpr, asm volatile("v_cmp_gt_u32 vcc, s20, %0\n v_cmp_gt_u32 vcc, s21, %1" :: "v"(v0), "v"(v2) : "vcc");)
KERNEL(k_min3_sgpr, asm volatile("v_min3_i32 %0, s20, %1, %0\n v_min3_i32 %2, s21, %3, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_cmp_cnd, asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) :: "vcc");)
KERNEL(k_cnd_only, asm volatile("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %3, %2, %1, vcc" : "=v"(v0), "+v"(v1), "+v"(v2), "=v"(v3));)
KERNEL(k_readlane_use, asm volatile("v_readlane_b32 s20, %0, 3\n v_add_u32 %1, s20, %1" : "+v"(v0), "+v"(v1) :: "s20");)
KERNEL(k_mad_u24,  asm volatile("v_mad_u32_u24 %0, %1, 32, %0\n v_mad_u32_u24 %2, %3, 32, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_mad_i24,  asm volatile("v_mad_i32_i24 %0, %1, 32, %0\n v_mad_i32_i24 %2, %3, 32, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_mad_i24v, asm volatile("v_mad_i32_i24 %0, %1, %2, %0\n v_mad_i32_i24 %2, %3, %1, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_sub_co,   asm volatile("v_sub_co_u32 %0, vcc, %1, %0\n v_sub_co_u32 %2, vcc, %3, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) :: "vcc");)
KERNEL(k_max3_i32, asm volatile("v_max3_i32 %0, %1, %0, %2\n v_max3_i32 %2, %3, %2, %0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
// packed 16-bit integer forms (round 2: would two tiles' coordinate math fit one instruction?)
KERNEL(k_pk_sub_i16, asm volatile("v_pk_sub_i16 %0, %1, %0\n v_pk_sub_i16 %2, %3, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_pk_sub_i16_clamp, asm volatile("v_pk_sub_i16 %0, %1, %0 clamp\n v_pk_sub_i16 %2, %3, %2 clamp" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_pk_min_i16, asm volatile("v_pk_min_i16 %0, %1, %0\n v_pk_min_i16 %2, %3, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_pk_max_i16, asm volatile("v_pk_max_i16 %0, %1, %0\n v_pk_max_i16 %2, %3, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_pk_add_u16, asm volatile("v_pk_add_u16 %0, %1, %0\n v_pk_add_u16 %2, %3, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_pk_lshl_b16, asm volatile("v_pk_lshlrev_b16 %0, 2, %1\n v_pk_lshlrev_b16 %2, 2, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_pk_mad_i16, asm volatile("v_pk_mad_i16 %0, %1, %0, %2\n v_pk_mad_i16 %2, %3, %2, %0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_cvt_pk_i16, asm volatile("v_cvt_pk_i16_i32 %0, %1, %0\n v_cvt_pk_i16_i32 %2, %3, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_perm_b32, asm volatile("v_perm_b32 %0, %1, %0, %2\n v_perm_b32 %2, %3, %2, %0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_sad_u16, asm volatile("v_sad_u16 %0, %1, %0, 0\n v_sad_u16 %2, %3, %2, 0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_lshlrev, asm volatile("v_lshlrev_b32 %0, 5, %1\n v_lshlrev_b32 %2, 5, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_and_b32, asm volatile("v_and_b32 %0, %1, %0\n v_and_b32 %2, %3, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_cmpx, asm volatile("v_cmpx_gt_u32 vcc, %0, %1\n s_mov_b64 exec, -1\n v_cmpx_gt_u32 vcc, %2, %3\n s_mov_b64 exec, -1" :: "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "vcc");)
KERNEL(k_bfe, asm volatile("v_bfe_u32 %0, %1, 7, 8\n v_bfe_u32 %2, %3, 7, 8" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_med3, asm volatile("v_med3_i32 %0, %1, %0, %2\n v_med3_i32 %2, %3, %2, %0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
KERNEL(k_mbcnt, asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0\n v_mbcnt_lo_u32_b32 %2, %3, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));)
// same-address (broadcast) LDS reads of growing width: what one source broadcast of the score sweep costs the LDS pipe
KERNEL(k_ds_b32_bc,  { int t; asm volatile("ds_read_b32 %0, %1\n ds_read_b32 %0, %1 offset:16\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((v4 & 0x3f) << 4)); v1 ^= t; })
KERNEL(k_ds_b64_bc,  { long long t; asm volatile("ds_read_b64 %0, %1\n ds_read_b64 %0, %1 offset:16\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((v4 & 0x3f) << 4)); v1 ^= (int)t; })
KERNEL(k_ds_b96_bc,  { int t0; int t1; int t2; asm volatile("ds_read_b96 v[40:42], %3\n ds_read_b96 v[40:42], %3 offset:16\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v40\n v_mov_b32 %1, v41\n v_mov_b32 %2, v42" : "=v"(t0), "=v"(t1), "=v"(t2) : "v"((v4 & 0x3f) << 4) : "v40", "v41", "v42"); v1 ^= t0 + t1 + t2; })
KERNEL(k_ds_b128_bc, { int t0; int t1; asm volatile("ds_read_b128 v[40:43], %2\n ds_read_b128 v[40:43], %2 offset:16\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v40\n v_mov_b32 %1, v43" : "=v"(t0), "=v"(t1) : "v"((v4 & 0x3f) << 4) : "v40", "v41", "v42", "v43"); v1 ^= t0 + t1; })
KERNEL(k_ds_read,  { int t; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((v0 & 0xff) << 2)); v1 ^= t; })
KERNEL(k_ds_read_nw, { int t; int u; asm volatile("ds_read_b32 %0, %2\n ds_read_b32 %1, %2 offset:4" : "=v"(t), "=v"(u) : "v"((v0 & 0xff) << 2)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); v1 ^= t + u; })

// What does a ds_read beyond the workgroup's LDS allocation return?  (The score kernel may index its penalty table with a
// distance that a later test rejects.)  1 KB allocated and filled with 0x5a5a5a5a; byte offsets given by the host.
__global__ void k_lds_oob(const unsigned *offsets, int n, unsigned *out)
{
	extern __shared__ unsigned dyn[];
	for (int i = threadIdx.x; i < 256; i += blockDim.x) dyn[i] = 0x5a5a5a5au;
	__syncthreads();
	if (threadIdx.x < (unsigned)n) {
		unsigned v;
		asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(offsets[threadIdx.x]) : "memory");
		out[threadIdx.x] = v;
	}
}

typedef void (*kfn)(int*, int, int, int);

int main()
{
	struct { const char *name; kfn fn; } tests[] = {
		{"v_add_u32", k_add_u32}, {"v_sub_u32", k_sub_u32}, {"v_min_u32", k_min_u32}, {"v_max_i32", k_max_i32}, {"v_min3_i32", k_min3_i32},
		{"v_sad_u32", k_sad_u32}, {"v_lshl_add_u32", k_lshl_add}, {"v_add3_u32", k_add3}, {"v_and_or_b32", k_and_or}, {"v_cndmask_b32", k_cndmask},
		{"v_cmp_gt_u32 vcc", k_cmp_u32}, {"v_cmp_gt_u32 sgpr", k_cmp_e64}, {"v_readlane_b32", k_readlane}, {"v_mov_b32", k_mov},
		{"v_add_f32", k_add_f32}, {"v_fma_f32", k_fma_f32}, {"v_min_f32", k_min_f32}, {"v_cmp_gt_f32", k_cmp_f32}, {"v_cvt_f32_i32", k_cvt_i2f},
		{"v_pk_add_f32", k_pk_add_f32},
		{"v_pk_sub_i16", k_pk_sub_i16}, {"v_pk_sub_i16 clamp", k_pk_sub_i16_clamp}, {"v_pk_min_i16", k_pk_min_i16}, {"v_pk_max_i16", k_pk_max_i16},
		{"v_pk_add_u16", k_pk_add_u16}, {"v_pk_lshlrev_b16", k_pk_lshl_b16}, {"v_pk_mad_i16", k_pk_mad_i16}, {"v_cvt_pk_i16_i32", k_cvt_pk_i16},
		{"v_perm_b32", k_perm_b32}, {"v_sad_u16", k_sad_u16}, {"v_lshlrev_b32", k_lshlrev}, {"v_and_b32", k_and_b32}, {"v_cmpx + s_mov exec", k_cmpx},
		{"v_bfe_u32", k_bfe}, {"v_med3_i32", k_med3}, {"v_mbcnt_lo", k_mbcnt},
		{"v_sub_u32 sgpr src", k_sub_sgpr}, {"v_subrev_u32 sgpr src", k_subrev_sgpr}, {"v_cmp vcc, sgpr, v", k_cmp_sgpr}, {"v_min3_i32 sgpr src", k_min3_sgpr}, {"v_cmp + v_cndmask (2 instr)", k_cmp_cnd}, {"v_cndmask indep", k_cnd_only}, {"v_readlane + v_add using it", k_readlane_use}, {"v_mad_u32_u24 (x32 literal)", k_mad_u24}, {"v_mad_i32_i24 (x32 literal)", k_mad_i24}, {"v_mad_i32_i24 (vgpr)", k_mad_i24v}, {"v_sub_co_u32", k_sub_co}, {"v_max3_i32", k_max3_i32}, {"ds_read_b32 broadcast x2", k_ds_b32_bc}, {"ds_read_b64 broadcast x2", k_ds_b64_bc}, {"ds_read_b96 broadcast x2", k_ds_b96_bc}, {"ds_read_b128 broadcast x2", k_ds_b128_bc}, {"ds_read_b32+wait", k_ds_read}, {"ds_read_b32 x2 then wait", k_ds_read_nw},
	};
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	{
		const unsigned offs[] = { 0u, 1020u, 1024u, 4096u, 65532u, 65536u, 163836u, 163840u, 1u << 20, 1u << 24, 0x7ffffffcu, 0xfffffffcu };
		const int n = (int)(sizeof(offs) / 4);
		unsigned *d_off, *d_out, h_out[16];
		hipMalloc(&d_off, sizeof(offs)); hipMalloc(&d_out, sizeof(offs));
		hipMemcpy(d_off, offs, sizeof(offs), hipMemcpyHostToDevice);
		hipLaunchKernelGGL(k_lds_oob, dim3(1), dim3(64), 1024, 0, d_off, n, d_out);
		hipMemcpy(h_out, d_out, sizeof(offs), hipMemcpyDeviceToHost);
		printf("ds_read_b32 with 1 KB of LDS allocated (filled 0x5a5a5a5a): ");
		for (int i = 0; i < n; ++i) printf("[%u]=0x%x ", offs[i], h_out[i]);
		printf("\n"); fflush(stdout);
	}
	const int cus = prop.multiProcessorCount;
	int *out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const double clk = prop.clockRate * 1e3;   // Hz
	printf("device %s, %d CUs, clock %.0f MHz; 8 waves/SIMD, 128 instr per loop body\n", prop.name, cus, clk / 1e6);
	for (auto &t : tests) {
		const int iters = 20000;
		for (int rep = 0; rep < 2; ++rep) {
			hipEventRecord(e0);
			hipLaunchKernelGGL(t.fn, dim3(cus * 8), dim3(256), 0, 0, out, iters, 1, 2);
			hipEventRecord(e1); hipEventSynchronize(e1);
		}
		float ms; hipEventElapsedTime(&ms, e0, e1);
		// per SIMD: 8 waves x iters x 128 instructions
		const double instr = 8.0 * iters * 128;  /* kernels with other bodies: scale by hand */
		fflush(stdout); printf("%-28s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD (at nominal clock)\n", t.name, ms, ms * 1e-3 * clk / instr);
	}
	return 0;
}
