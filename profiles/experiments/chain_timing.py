#!/usr/bin/env python3
"""Run the slowest read of a small bench batch alone through the instrumented library (make_chain_timing_build.py) and print where the
serial chain through its tiles goes.   MM2GB_LIB_PATH=mm2-gb_amd/ab/libchain.so python profiles/experiments/chain_timing.py [read]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MM2GB_LIB_PATH", os.path.join(ROOT, "mm2-gb_amd", "ab", "libchain.so"))
os.environ.setdefault("MM2GB_GANG_MAX", "0")
import numpy as np
import bench, mm2gb_amd as mm

r = int(sys.argv[1]) if len(sys.argv) > 1 else 16
_, n_reads, anchors, off = bench.shard_for_rank(mm, 0, 1, 1, 20_000_000, 100_000, 300_000, 16)
a = anchors[off[r]:off[r + 1]]
o = np.array([0, len(a)], np.int64)
L = mm.lib()
buf = (C.c_ulonglong * 16)()
names = ["ticks waiting before in-tile", "ticks in in-tile phases (rescue build)", "in-tile phases (rescue build)", "rescans", "ticks in rescans", "full-state-machine steps",
         "entry-mode steps", "ticks earlier-tiles-final -> published (one tile per wave)", "tiles (one tile per wave, rescue)", "blocks read by rescans",
         "ticks in-tile before the step loop (incl. a whole tile in entry mode on the plain steps)", "ticks in the entry / full step loop", "ticks in plain steps + keep update"]
for name, env in (("whole workgroup", {}), ("8-wave teams", {"MM2GB_WHOLE_WG_PCT": "0"})):
    os.environ.update(env)
    with mm.Engine() as e:
        e.score(a, o)
        L.mm2gb_debug_chain_ticks(buf, 1)
        st = e.score(a, o)[2]
        L.mm2gb_debug_chain_ticks(buf, 1)
    for k in env:
        del os.environ[k]
    print(f"{name}: read {r}, {len(a)} anchors, {st['n_pairs']} pairs, ms_score {st['ms_score']:.3f}")
    for k, nm in enumerate(names):
        v = buf[k]
        print(f"   {nm:62s} {v:12d}" + (f"  = {v / 100:.1f} us" if "ticks" in nm else ""))
