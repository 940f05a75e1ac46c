#!/usr/bin/env python3
"""plain_steps alone on a made-up dense tile (instrumented library): s_memtime ticks per 64-source tile, one wave per CU / several."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM2GB_LIB_PATH", os.path.join(ROOT, "mm2-gb_amd", "ab", "libchain.so"))
import mm2gb_amd as mm
L = mm.lib()
L.mm2gb_debug_bench_steps.restype = C.c_double
L.mm2gb_debug_bench_steps.argtypes = [C.c_int, C.c_int, C.c_int, C.c_ulonglong]
for kind, name in ((0, "free"), (1, "checked"), (2, "clamped")):
    for n_wg in (256,):
        for need, nm in ((0x7fffffffffffffff, "63 sources"), (0x00000000ffffffff, "32 sources"), (0x3, "2 sources")):
            t = L.mm2gb_debug_bench_steps(kind, n_wg, 200, need)
            print(f"{name:8s} {n_wg:5d} one-wave workgroups  {nm:11s} {t:9.0f} ticks per call")
