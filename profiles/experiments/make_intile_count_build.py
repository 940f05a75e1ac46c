#!/usr/bin/env python3
"""Instrumented build: what the in-tile phases of the score kernel do (round 4) -- phases, calls of plain_steps by kind of tile, sources needed
and rows computed, tiles of the rescue build done again the long way.  -> mm2-gb_amd/ab/libintile.so (git-ignored).
python profiles/experiments/make_intile_count_build.py;  python profiles/experiments/intile_counts.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "mm2-gb_amd")
src = open(os.path.join(PKG, "csrc", "chain_kernels.hip")).read()


def sub(old, new):
    global src
    assert src.count(old) >= 1, old
    src = src.replace(old, new, 1)


sub("constexpr int SCORE_THREADS = 1024;\n",
    "constexpr int SCORE_THREADS = 1024;\n"
    "// 0 in-tile phases, 1 calls of plain_steps with sources to do, 2 sources needed in them, 3 rows computed, 4-6 calls by kind (free, checked, clamped),\n"
    "// 7 rescue-build tiles whose fast path was undone, 8 steps of the entry / full loop, 9 in-tile phases of the rescue build, 10 phases with nothing to do\n"
    "__device__ unsigned long long g_it[16];\n"
    "__device__ __forceinline__ void it_add(int k, long long v) { if ((threadIdx.x & 63) == 0) atomicAdd(&g_it[k], (unsigned long long)v); }\n")
sub("	if (!need) return;\n	if (tl.kind == ROWS_FREE)",
    "	if (!need) { it_add(10, 1); return; }\n	it_add(1, 1); it_add(2, __builtin_popcountll(need)); it_add(3, 2 * __builtin_popcountll((need | need >> 1) & 0x5555555555555555ull)); it_add(4 + tl.kind, 1);\n	if (tl.kind == ROWS_FREE)")
sub("	__builtin_amdgcn_s_setprio(MM2GB_INTILE_PRIO);\n	stage[lane] = make_int4(128 - LUT_BIAS,", "	it_add(0, 1); if (TRACK) it_add(9, 1);\n	__builtin_amdgcn_s_setprio(MM2GB_INTILE_PRIO);\n	stage[lane] = make_int4(128 - LUT_BIAS,")
sub("			} else { bestv = bestv0; arg = arg0; }\n", "			} else { bestv = bestv0; arg = arg0; it_add(7, 1); }\n")
sub("			const int j = i0 + t;\n			const StepPre nxt = tile_pre(tl, t + 1 < n_here ? t + 1 : t);\n", "			it_add(8, 1);\n			const int j = i0 + t;\n			const StepPre nxt = tile_pre(tl, t + 1 < n_here ? t + 1 : t);\n")
src += '''
extern "C" void mm2gb_debug_intile_counts(unsigned long long *out, int reset)
{
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mm2gb::g_it), 128);
	if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(mm2gb::g_it), z, 128); }
}
'''
out_dir = os.path.join(PKG, "ab")
os.makedirs(out_dir, exist_ok=True)
tmp = os.path.join(PKG, "csrc", "chain_kernels_intile_tmp.hip")
open(tmp, "w").write(src)
try:
    subprocess.check_call(["make", "-s", "-C", PKG])
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include")]
    obj = os.path.join(out_dir, "chain_kernels_intile.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", tmp, "-o", obj])
    others = [os.path.join(PKG, "build", f) for f in os.listdir(os.path.join(PKG, "build")) if f.endswith(".o") and f != "chain_kernels.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + others + ["-o", os.path.join(out_dir, "libintile.so"), "-lpthread"])
finally:
    os.remove(tmp)
print(os.path.join(out_dir, "libintile.so"))
