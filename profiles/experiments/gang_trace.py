#!/usr/bin/env python3
"""Per-pair trace of the chain through ONE heavy chunk scored by a gang of workgroups (instrumented library of make_chain_timing_build.py).
python profiles/experiments/gang_trace.py [gang_max] [anchors] [xwin]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MM2GB_LIB_PATH", os.path.join(ROOT, "mm2-gb_amd", "ab", "libchain.so"))
os.environ["MM2GB_GANG_MAX"] = sys.argv[1] if len(sys.argv) > 1 else "8"
import numpy as np
import mm2gb_amd as mm, synth_cases as sc

n = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
xwin = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
a = sc.sort_by_x(sc.repeat_block(n, 7, xwin=xwin, ywin=6000))
off = np.array([0, len(a)], np.int64)
L = mm.lib()
pairs = os.environ.get("MM2GB_GANG_PAIRS", "0") not in ("", "0")
nt = min(8192, (len(a) + 127) // 128 if pairs else (len(a) + 63) // 64)
buf = np.zeros(nt * 8, np.int64)
c2 = (C.c_ulonglong * 8)()
with mm.Engine() as e:
    e.score(a, off)
    L.mm2gb_debug_chain2(c2, 1)
    st = e.score(a, off)[2]
    L.mm2gb_debug_chain2(c2, 1)
    print("gang counts", e.gang_counts())
    L.mm2gb_debug_chain_trace(C.c_void_p(buf.ctypes.data), nt)
tr = buf.reshape(nt, 8).astype(np.float64)
if c2[3]:
    print(f"rescue build in-tile phases on the fast path: {c2[3]}; per phase: before the steps {c2[0] / c2[3]:.0f}, the steps {c2[1] / c2[3]:.0f}, after them {c2[2] / c2[3]:.0f} ticks")
print(f"{len(a)} anchors, {nt} pairs, {st['n_pairs']} pairs of anchors, ms_score {st['ms_score']:.3f}; stamps are s_memtime ticks (~0.43 ns)")
lo, hi = 50, nt - 2
t = np.arange(lo + 1, hi)
endp = tr[:, 7] if pairs else tr[:, 4]
step = endp[lo + 1:hi] - endp[lo:hi - 1]
print(f"{'pairs' if pairs else 'single tiles'}: end of the last in-tile phase to the next turn's: mean {step.mean():.0f}  median {np.median(step):.0f}  min {step.min():.0f}  max {step.max():.0f}")
parts = [("previous pair's in-tile B ends -> 'every earlier tile is final' seen", tr[t, 2] - endp[t - 1]),
         ("   sweeps done -> seen (negative = the sweeps ended after the previous pair)", tr[t, 2] - tr[t, 1]),
         ("tile A's fields loaded", tr[t, 3] - tr[t, 2]),
         ("in-tile A", tr[t, 4] - tr[t, 3]),
         ] + ([("store + publish A", tr[t, 5] - tr[t, 4]), ("A swept into B", tr[t, 6] - tr[t, 5]), ("tile B's fields + in-tile B", tr[t, 7] - tr[t, 6])] if pairs else [])
for name, v in parts:
    print(f"  {name:86s} mean {v.mean():9.0f}  median {np.median(v):9.0f}  p90 {np.percentile(v, 90):9.0f}")
# by position of the pair in its strip (wave): the strip's first pair waits for another workgroup
w = t % 16
for k in (0, 1, 8, 15):
    sel = w == k
    print(f"  wave {k:2d}: previous end -> seen  median {np.median((tr[t, 2] - endp[t - 1])[sel]):9.0f}   whole step median {np.median((endp[t] - endp[t - 1])[sel]):9.0f}")
