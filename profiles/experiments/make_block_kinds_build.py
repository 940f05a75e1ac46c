#!/usr/bin/env python3
"""Instrumented build of the score kernel for profiles/block_kinds.py: a COPY of csrc/chain_kernels.hip with a counter per sweep build
(tile-blocks = 64 sources x 64 targets), compiled with the other objects into mm2-gb_amd/ab/libcount.so (git-ignored, travels to the GPU
box).  The tracked sources are not touched.   python profiles/experiments/make_block_kinds_build.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "mm2-gb_amd")
src = open(os.path.join(PKG, "csrc", "chain_kernels.hip")).read()


def put(after, text, count=1):
    global src
    assert src.count(after) >= 1, after
    src = src.replace(after, after + text, count)


put("constexpr int SCORE_THREADS = 1024;\n",
    "// counters of tile-blocks by sweep build: 0 two tiles FAR, 1 two tiles unchecked (not FAR), 2 two tiles with range test, 3 one tile FAR, 4 one tile unchecked,\n"
    "// 5 one tile with range test only, 6 one tile with window / equal-position tests, 7 in-tile phases\n"
    "__device__ unsigned long long g_block_counts[16];\n"
    "__device__ __forceinline__ void count_block(int kind, int n) { if ((threadIdx.x & 63) == 0) atomicAdd(&g_block_counts[kind], (unsigned long long)n); }\n")
put("\tconst int tx4 = (int)(((unsigned)T.x - 1u) << 2), ty4 = (int)(((unsigned)T.y - 1u) << 2);\n\tint bestv = best << 7;\n",
    "\tcount_block(far_block ? 3 : free_block ? 4 : no_check ? 5 : 6, 1);\n")
put("\tint bva = best_a << 7, bvb = best_b << 7;\n", "\tcount_block(far_block ? 0 : free_block ? 1 : 2, 2);\n")
put("\tconst int lane = lane_id(), i = i0 + lane;\n\t// The in-tile phase is a chain of dependent instructions", "")
src = src.replace("\t__builtin_amdgcn_s_setprio(MM2GB_INTILE_PRIO);\n\tstage[lane] = make_int4(0, (T.q - 1) * 4,", "\tcount_block(7, 1);\n\t__builtin_amdgcn_s_setprio(MM2GB_INTILE_PRIO);\n\tstage[lane] = make_int4(0, (T.q - 1) * 4,", 1)
# steps of the rescue build's in-tile phase by mode: 8 entry, 9 after an update inside the tile, 10 the full state machine; 11 tiles of that build
put("\t\t\tif (mode == IN_TILE || (mode == ENTRY && !(slow >> t & 1))) {\n", "\t\t\t\tcount_block(mode == IN_TILE ? 9 : 8, 1);\n")
put("\t\t\tmode = FULL;                                                         // (only ever from ENTRY: IN_TILE stays in the branch above)\n", "\t\t\tcount_block(10, 1);\n")
put("\t\tenum { ENTRY = 0, IN_TILE = 1, FULL = 2 };\n", "\t\tcount_block(11, 1);\n")
assert src.count("count_block(") == 7
src += '''
extern "C" void mm2gb_debug_block_counts(unsigned long long *out, int reset)
{
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mm2gb::g_block_counts), 128);
	if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(mm2gb::g_block_counts), z, 128); }
}
'''
out_dir = os.path.join(PKG, "ab")
os.makedirs(out_dir, exist_ok=True)
tmp = os.path.join(PKG, "csrc", "chain_kernels_count_tmp.hip")
open(tmp, "w").write(src)
try:
    subprocess.check_call(["make", "-s", "-C", PKG])
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include")]
    obj = os.path.join(out_dir, "chain_kernels_count.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", tmp, "-o", obj])
    others = [os.path.join(PKG, "build", f) for f in os.listdir(os.path.join(PKG, "build")) if f.endswith(".o") and f != "chain_kernels.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + others + ["-o", os.path.join(out_dir, "libcount.so"), "-lpthread"])
finally:
    os.remove(tmp)
print(os.path.join(out_dir, "libcount.so"))
