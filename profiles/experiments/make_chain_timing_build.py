#!/usr/bin/env python3
"""Instrumented build of the score kernel: where the serial chain through the tiles of a team's chunk spends its time.  A COPY of
csrc/chain_kernels.hip with 100 MHz time stamps around the waits, the in-tile phases and the rescue rescans of the team code,
compiled into mm2-gb_amd/ab/libchain.so (git-ignored, travels to the GPU box).   python profiles/experiments/make_chain_timing_build.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "mm2-gb_amd")
src = open(os.path.join(PKG, "csrc", "chain_kernels.hip")).read()


def sub(old, new, count=1):
    global src
    assert src.count(old) >= 1, old
    src = src.replace(old, new, count)


sub("constexpr int SCORE_THREADS = 1024;\n",
    "constexpr int SCORE_THREADS = 1024;\n"
    "// 0 ticks waiting for the earlier tiles before an in-tile phase, 1 ticks in in-tile phases, 2 in-tile phases, 3 rescans, 4 ticks in rescans, 5 full-state-machine steps,\n"
    "// 6 entry-mode steps, 7 ticks from 'earlier tiles final' to 'this tile published' (the chain), 8 tiles of rescue chunks in team code, 9 blocks read by rescans\n"
    "__device__ unsigned long long g_chain[16];\n"
    "__device__ __forceinline__ long long tick() { return (long long)__builtin_amdgcn_s_memrealtime(); }\n"
    "__device__ __forceinline__ void chain_add(int k, long long v) { if ((threadIdx.x & 63) == 0) atomicAdd(&g_chain[k], (unsigned long long)v); }\n")
# whole-workgroup / big-team path with one tile per wave (coop_chunk)
sub("		const int my_slot = (int)((unsigned)t % (unsigned)n_slots);\n		wait_done(t);                                                    // every earlier tile is final\n",
    "		const int my_slot = (int)((unsigned)t % (unsigned)n_slots);\n		const long long tk0 = tick();\n		wait_done(t);                                                    // every earlier tile is final\n		const long long tk1 = tick();\n")
sub("		if (lane == 0) __hip_atomic_store(&sh->done, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);\n	}\n}",
    "		if (lane == 0) __hip_atomic_store(&sh->done, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);\n		if (TRACK) { chain_add(0, tk1 - tk0); chain_add(7, tick() - tk1); chain_add(8, 1); }\n	}\n}")
# the rescue rescan of the table build (first occurrence = in_tile_lut)
sub("				int bf = INT_MIN, bi = -1;\n				for (int jj = stt + lane; jj < i0; jj += WAVE) {              // earlier tiles (ascending per lane)\n",
    "				const long long rk0 = tick();\n				chain_add(3, 1); chain_add(9, (i0 - stt + 63) / 64);\n				int bf = INT_MIN, bi = -1;\n				for (int jj = stt + lane; jj < i0; jj += WAVE) {              // earlier tiles (ascending per lane)\n")
sub("					{ const size_t at = (size_t)keep.idx * 4; keep.x = sr[at]; keep.y = sr[at + 2]; keep.tag = tag_of((unsigned)sr[at + 3]); keep.hi = ht; }\n				}\n",
    "					{ const size_t at = (size_t)keep.idx * 4; keep.x = sr[at]; keep.y = sr[at + 2]; keep.tag = tag_of((unsigned)sr[at + 3]); keep.hi = ht; }\n				}\n				chain_add(4, tick() - rk0);\n")
sub("			mode = FULL;                                                         // (only ever from ENTRY: IN_TILE stays in the branch above)\n",
    "			mode = FULL;                                                         // (only ever from ENTRY: IN_TILE stays in the branch above)\n			chain_add(5, 1);\n")
sub("			if (mode == ENTRY && !(slow >> t & 1)) {\n", "			if (mode == ENTRY && !(slow >> t & 1)) {\n				chain_add(6, 1);\n")
# in-tile phases of the rescue build: ticks
sub("		enum { ENTRY = 0, IN_TILE = 1, FULL = 2 };\n", "		enum { ENTRY = 0, IN_TILE = 1, FULL = 2 };\n		const long long ik0 = tick();\n		chain_add(2, 1);\n")
sub("		if (mode == IN_TILE) {                                                   // the anchor remembered now is one of this tile\n",
    "		chain_add(1, tick() - ik0);\n		if (mode == IN_TILE) {                                                   // the anchor remembered now is one of this tile\n")
# split of the rescue build's in-tile phase: 10 ticks before the step loop (set-up, entry precomputation), 11 ticks in the step loop (entry / full steps), 12 ticks in the plain steps + keep update
sub("		StepPre cur = tile_pre(tl, 0);\n		int t = done_fast ? n_here : 0;\n		for (; t < n_here; ++t) {\n			if (mode == IN_TILE) break;",
    "		const long long ik1 = tick();\n		chain_add(10, ik1 - ik0);\n		StepPre cur = tile_pre(tl, 0);\n		int t = done_fast ? n_here : 0;\n		for (; t < n_here; ++t) {\n			if (mode == IN_TILE) break;")
sub("		if (mode == IN_TILE && t < n_here) {\n", "		const long long ik2 = tick();\n		chain_add(11, ik2 - ik1);\n		if (mode == IN_TILE && t < n_here) {\n")
sub("		chain_add(1, tick() - ik0);\n", "		chain_add(1, tick() - ik0); chain_add(12, tick() - ik2);\n")

# ---- round 4: a per-tile trace of the chain in the one-tile-per-wave team code (coop_chunk): s_memtime stamps per tile of the FIRST chunk a
# launch scores that way: 0 before the wait for the last source block, 1 after it, 2 after that block's sweep, 3 after "every earlier tile is
# final", 4 after the in-tile phase, 5 after the publication, 6 the wave's first look at this tile (tile loop top)
sub("__device__ unsigned long long g_chain[16];\n",
    "__device__ unsigned long long g_chain[16];\n__device__ long long g_trace[8192 * 8];\n"
    "__device__ __forceinline__ long long ctick() { return (long long)__builtin_amdgcn_s_memtime(); }\n"
    "__device__ __forceinline__ void trace(int t, int k, long long v) { if ((threadIdx.x & 63) == 0 && t < 8192) g_trace[t * 8 + k] = v; }\n")
sub("		const Target T = load_target(b, i0, ce, TRACK);\n		const int n_here = min(WAVE, ce - i0);\n		int best = T.q + 1, arg = -1;\n		const int tile_lo = first_lane(T.st);\n		const int st_hi = bcast(T.st, n_here - 1);\n		int jb = cs + ((tile_lo - cs) & ~(WAVE - 1));\n		const int eq_lo = MODE == MODE_LUT && jb < i0",
    "		trace(t, 6, ctick());\n		const Target T = load_target(b, i0, ce, TRACK);\n		const int n_here = min(WAVE, ce - i0);\n		int best = T.q + 1, arg = -1;\n		const int tile_lo = first_lane(T.st);\n		const int st_hi = bcast(T.st, n_here - 1);\n		int jb = cs + ((tile_lo - cs) & ~(WAVE - 1));\n		const int eq_lo = MODE == MODE_LUT && jb < i0")
sub("			wait_done(k + 1);                                          // that tile's scores are in the ring\n			const int sf = ring[slot * WAVE + lane];\n			slot = slot + 1 == n_slots ? 0 : slot + 1;\n			sweep_any<MODE>(b, T, jb, k_from, sf, sq, jb >= st_hi && jb + WAVE <= eq_lo, stage, P, lut, best, arg);\n",
    "			const bool last_blk = jb + WAVE >= i0;\n			if (last_blk) trace(t, 0, ctick());\n			wait_done(k + 1);                                          // that tile's scores are in the ring\n			if (last_blk) trace(t, 1, ctick());\n			const int sf = ring[slot * WAVE + lane];\n			slot = slot + 1 == n_slots ? 0 : slot + 1;\n			sweep_any<MODE>(b, T, jb, k_from, sf, sq, jb >= st_hi && jb + WAVE <= eq_lo, stage, P, lut, best, arg);\n			if (last_blk) trace(t, 2, ctick());\n")
sub("		const long long tk1 = tick();\n", "		const long long tk1 = tick();\n		trace(t, 3, ctick());\n")
sub("		const int i = i0 + lane;\n		const int fi = arg < 0 ? T.q : best;\n		if (T.live) {\n			ring[my_slot * WAVE + lane] = fi;",
    "		trace(t, 4, ctick());\n		const int i = i0 + lane;\n		const int fi = arg < 0 ? T.q : best;\n		if (T.live) {\n			ring[my_slot * WAVE + lane] = fi;")
sub("		if (TRACK) { chain_add(0, tk1 - tk0); chain_add(7, tick() - tk1); chain_add(8, 1); }\n", "		trace(t, 5, ctick());\n		if (TRACK) { chain_add(0, tk1 - tk0); chain_add(7, tick() - tk1); chain_add(8, 1); }\n")

# ---- gangs: per PAIR of the first gang chunk: 0 before the wait for the last source block,
# 1 before 'every earlier tile is final', 2 after it, 3 after tile A's fields are loaded, 4 after in-tile A, 5 after A is stored + published in the workgroup,
# 6 after A is swept into B, 7 after in-tile B   (the global publication follows: stamp 0 of the next pair shows it)
GANG = "gang_chunk_pairs(const DevBatch &b"
def gsub(old, new):
    global src
    at = src.index(GANG)
    assert src.count(old, at) >= 1, old
    src = src[:at] + src[at:].replace(old, new, 1)
gsub("			sweep_pair_block(b, t, jb, eq_lo, sf, sq, stage, P);\n		}\n		const int slot_a", "			if (jb + 2 * WAVE >= i0 && jb + WAVE < i0) trace(pr, 0, ctick());\n			sweep_pair_block(b, t, jb, eq_lo, sf, sq, stage, P);\n		}\n		trace(pr, 1, ctick());\n		const int slot_a")
gsub("		wait_done(ta);                                               // every earlier tile is final\n		Keep keep;", "		wait_done(ta);                                               // every earlier tile is final\n		trace(pr, 2, ctick());\n		Keep keep;")
gsub("		in_tile<MODE_LUT, TRACK>(b, TA, i0, t.n_a, P, lut, stage, t.best_a, t.arg_a, keep, f_old, prog);\n",
     "		asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); trace(pr, 3, ctick());\n		in_tile<MODE_LUT, TRACK>(b, TA, i0, t.n_a, P, lut, stage, t.best_a, t.arg_a, keep, f_old, prog);\n		trace(pr, 4, ctick());\n")
gsub("			sweep_a_into_b(b, t, cs, i0, f_a, TA.q, stage, P);\n", "			trace(pr, 5, ctick());\n			sweep_a_into_b(b, t, cs, i0, f_a, TA.q, stage, P);\n			trace(pr, 6, ctick());\n")
gsub("			in_tile<MODE_LUT, TRACK>(b, TB, i0 + WAVE, t.n_b, P, lut, stage, t.best_b, t.arg_b, keep, f_old);\n", "			in_tile<MODE_LUT, TRACK>(b, TB, i0 + WAVE, t.n_b, P, lut, stage, t.best_b, t.arg_b, keep, f_old);\n			trace(pr, 7, ctick());\n")

# ---- the plain steps alone: one wave per workgroup runs plain_steps on a made-up dense tile `iters` times; s_memtime ticks per call
src = src.replace("} // namespace mm2gb\n", r"""
template <int KIND>
__global__ __launch_bounds__(64) void k_bench_steps(long long *out, int iters, DevParams P, const int *lut_g, unsigned long long need)
{
	extern __shared__ __attribute__((aligned(16))) int smem[];
	int *lut = smem + P.lut_base / 4;
	for (int k = threadIdx.x; k <= P.lut_last; k += 64) lut[k] = lut_g[k];
	int4 *stage = (int4*)smem;
	const int lane = threadIdx.x;
	const int x = 1000 + 3 * lane, y = 500 + 3 * lane + (lane & 1), q = 15;
	stage[lane] = make_int4(128 - LUT_BIAS, (q - 1) * 4, x << 2, y << 2);
	__syncthreads();
	TileLut tl;
	tl.tx4 = (x - 1) << 2; tl.ty4 = (y - 1) << 2; tl.lo = 0;
	tl.lim4 = (unsigned)P.dq_lim << 2; tl.base = (unsigned)P.lut_base; tl.last_at = tl.base + ((unsigned)P.lut_last << 2);
	tl.stage = stage; tl.edges = false; tl.kind = KIND < 3 ? KIND : 0;
	int bestv = (q + 1) << 7;
	long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it) { tl.tx4 += 4; tl.ty4 += 4; asm volatile("" ::: "memory"); plain_steps_impl<KIND < 3 ? KIND : 0>(tl, need, bestv); bestv = (bestv & 0xfffff) | (1 << 12); asm volatile("" : "+v"(bestv)); }
	long long t1 = __builtin_amdgcn_s_memtime();
	if (lane == 0) out[blockIdx.x] = t1 - t0;
	out[1024 + blockIdx.x * 64 + lane] = bestv;
}
} // namespace mm2gb
extern "C" double mm2gb_debug_bench_steps(int kind, int n_wg, int iters, unsigned long long need)
{
	using namespace mm2gb;
	DevParams P; memset(&P, 0, sizeof P);
	P.max_dist_x = P.max_dist_y = P.dq_lim = 5000; P.bw = 500; P.max_iter = 5000; P.n_seg = 1; P.gap = 0.01f * 15; P.skip = 0;
	P.lut_last = P.bw + 1; P.lut_base = LUT_LDS_TOTAL - 4 * (P.lut_last + 1); P.lut_clamp = kind == ROWS_CLAMPED; P.free_sweep = 1;
	int *lut; long long *out;
	(void)hipMalloc(&lut, LUT_ENTRIES * 4); (void)hipMalloc(&out, (1024 + 2048 * 64) * 8);
	launch_build_lut(lut, P, 0);
	auto fn = kind == 3 ? k_bench_steps<3> : kind == ROWS_FREE ? k_bench_steps<ROWS_FREE> : kind == ROWS_CHECKED ? k_bench_steps<ROWS_CHECKED> : k_bench_steps<ROWS_CLAMPED>;
	(void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LUT_LDS_TOTAL);
	for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(fn, dim3(n_wg), dim3(64), LUT_LDS_TOTAL, 0, out, iters, P, lut, need);
	(void)hipDeviceSynchronize();
	std::vector<long long> h(n_wg);
	(void)hipMemcpy(h.data(), out, n_wg * 8, hipMemcpyDeviceToHost);
	double s = 0; for (auto v : h) s += (double)v;
	(void)hipFree(lut); (void)hipFree(out);
	return s / n_wg / iters;
}
""")
src = src.replace("#include <algorithm>\n", "#include <algorithm>\n#include <vector>\n#include <string.h>\n", 1)

# ---- round 4: the rescue build's in-tile phase in three parts (ticks of s_memtime summed over phases): 13 from its start to the plain steps of the
# fast path, 14 those steps (incl. the quarters handed out), 15 from their end to the end of the phase; in g_chain2[0..2], phases counted in [3]
sub("__device__ unsigned long long g_chain[16];\n", "__device__ unsigned long long g_chain[16];\n__device__ unsigned long long g_chain2[8];\n"
    "__device__ __forceinline__ void chain2_add(int k, long long v) { if ((threadIdx.x & 63) == 0) atomicAdd(&g_chain2[k], (unsigned long long)v); }\n")
sub("		enum { ENTRY = 0, IN_TILE = 1, FULL = 2 };\n", "		enum { ENTRY = 0, IN_TILE = 1, FULL = 2 };\n		const long long ck0 = (long long)__builtin_amdgcn_s_memtime();\n		long long ck1 = ck0, ck2 = ck0;\n")
sub("			if (prog.ring_slot) plain_steps_by_quarters(tl, __ballot(T.live && T.st < i) >> 1, bestv, hand_out);\n			else plain_steps(tl, __ballot(T.live && T.st < i) >> 1, bestv);\n",
    "			ck1 = (long long)__builtin_amdgcn_s_memtime();\n			if (prog.ring_slot) plain_steps_by_quarters(tl, __ballot(T.live && T.st < i) >> 1, bestv, hand_out);\n			else plain_steps(tl, __ballot(T.live && T.st < i) >> 1, bestv);\n			ck2 = (long long)__builtin_amdgcn_s_memtime();\n")
sub("		if (mode == IN_TILE) {                                                   // the anchor remembered now is one of this tile\n",
    "		if (ck2 != ck0) { const long long ck3 = (long long)__builtin_amdgcn_s_memtime(); chain2_add(0, ck1 - ck0); chain2_add(1, ck2 - ck1); chain2_add(2, ck3 - ck2); chain2_add(3, 1); }\n		if (mode == IN_TILE) {                                                   // the anchor remembered now is one of this tile\n")
src += '''
extern "C" void mm2gb_debug_chain_ticks(unsigned long long *out, int reset)
{
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mm2gb::g_chain), 128);
	if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(mm2gb::g_chain), z, 128); }
}
extern "C" void mm2gb_debug_chain2(unsigned long long *out, int reset)
{
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mm2gb::g_chain2), 64);
	if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(mm2gb::g_chain2), z, 64); }
}
extern "C" void mm2gb_debug_chain_trace(long long *out, int n_tiles)
{
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mm2gb::g_trace), (size_t)n_tiles * 8 * sizeof(long long));
}
'''
out_dir = os.path.join(PKG, "ab")
os.makedirs(out_dir, exist_ok=True)
tmp = os.path.join(PKG, "csrc", "chain_kernels_timing_tmp.hip")
open(tmp, "w").write(src)
try:
    subprocess.check_call(["make", "-s", "-C", PKG])
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include")]
    obj = os.path.join(out_dir, "chain_kernels_timing.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", tmp, "-o", obj])
    others = [os.path.join(PKG, "build", f) for f in os.listdir(os.path.join(PKG, "build")) if f.endswith(".o") and f != "chain_kernels.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + others + ["-o", os.path.join(out_dir, "libchain.so"), "-lpthread"])
finally:
    os.rename(tmp, "/tmp/timing_tmp.hip")
print(os.path.join(out_dir, "libchain.so"))
