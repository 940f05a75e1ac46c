#!/usr/bin/env python3
"""Per-tile trace of the chain through ONE heavy chunk scored by a whole workgroup (instrumented library of make_chain_timing_build.py):
what the interval between two consecutive publications is made of.   python profiles/experiments/chain_trace.py [anchors] [xwin]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MM2GB_LIB_PATH", os.path.join(ROOT, "mm2-gb_amd", "ab", "libchain.so"))
import numpy as np
import mm2gb_amd as mm, synth_cases as sc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
xwin = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
a = sc.sort_by_x(sc.repeat_block(n, 7, xwin=xwin, ywin=6000))
off = np.array([0, len(a)], np.int64)
os.environ["MM2GB_BIG_TEAM"] = "16"
L = mm.lib()
nt = min(8192, (len(a) + 63) // 64)
buf = np.zeros(nt * 8, np.int64)
with mm.Engine() as e:
    e.score(a, off)
    st = e.score(a, off)[2]
    L.mm2gb_debug_chain_trace(C.c_void_p(buf.ctypes.data), nt)
tr = buf.reshape(nt, 8).astype(np.float64)
print(f"{len(a)} anchors, {nt} tiles, {st['n_pairs']} pairs, ms_score {st['ms_score']:.3f}; stamps are s_memtime ticks")
pub = tr[:, 5]
lo, hi = 100, nt - 2                       # steady state: windows full
step = pub[lo + 1:hi] - pub[lo:hi - 1]
print(f"publication to publication: mean {step.mean():.0f} ticks (min {step.min():.0f}, median {np.median(step):.0f}, max {step.max():.0f}); whole kernel {pub[hi - 1] - tr[0, 6]:.0f} ticks"
      f" -> {st['ms_score'] * 1e6 / (pub[hi - 1] - tr[0, 6]):.2f} ns per tick")
t = np.arange(lo + 1, hi)
parts = [("previous tile published -> this tile's wave sees it (wait for the last block ends)", tr[t, 1] - pub[t - 1]),
         ("   (that wave had been waiting since: wait begin -> previous publication; negative = arrived late)", pub[t - 1] - tr[t, 0]),
         ("sweep of the last source block", tr[t, 2] - tr[t, 1]),
         ("-> 'every earlier tile is final'", tr[t, 3] - tr[t, 2]),
         ("in-tile phase", tr[t, 4] - tr[t, 3]),
         ("stores + publication", tr[t, 5] - tr[t, 4]),
         ("tile loop top -> wait for the last block (all the other source blocks)", tr[t, 0] - tr[t, 6])]
for name, v in parts:
    print(f"  {name:100s} mean {v.mean():9.0f}  median {np.median(v):9.0f}  p90 {np.percentile(v, 90):9.0f}")
