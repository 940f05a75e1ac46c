#!/usr/bin/env python3
"""Debugging aid: the shortest prefix of a read on which the device's skip-limited RMQ fill and the oracle disagree, and the oracle's view of its last anchor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mm2gb_amd as mm, orc
from test_gpu_rmq import first_pass, to_lib
skip = int(sys.argv[1]) if len(sys.argv) > 1 else 25
which = int(sys.argv[2]) if len(sys.argv) > 2 else 9
a, off = mm.synth_reads(41, 0, 40, 10_000, 120_000)
x = first_pass(a[off[which]:off[which + 1]])
prm = orc.default_rmq_param(max_chn_skip=skip)
with mm.Engine() as e:
    def same(n):
        if n == 0: return True
        res, tied, _ = e.rmq_chain(x[:n], np.array([0, n], np.int64), to_lib(prm))
        o = orc.lchain_rmq(x[:n], prm)
        return np.array_equal(res[0][0], o["u"]) and np.array_equal(res[0][1], o["a_out"])
    import ctypes as C
    res, tied, _ = e.rmq_chain(x, np.array([0, len(x)], np.int64), to_lib(prm))
    f = np.zeros(len(x), np.int32); p = np.zeros(len(x), np.int32)
    assert mm.lib().mm2gb_debug_last_fill(C.c_void_p(e._h), C.c_int64(len(x)), C.c_void_p(f.ctypes.data), C.c_void_p(p.ctypes.data)) == 0
    o = orc.lchain_rmq(x, prm)
    op = np.where(o["p"] >= 0, np.arange(len(x)) - o["p"], 0)
    bad = np.nonzero((f != o["f"]) | (p != op))[0]
    print("anchors that differ:", len(bad), bad[:10])
    if len(bad):
        i = int(bad[0])
        print("first: anchor", i, "device f, p", int(f[i]), i - int(p[i]) if p[i] else -1, "oracle", int(o["f"][i]), int(o["p"][i]))
        yi = int(x[i, 1] & 0xffffffff); xi = int(x[i, 0] & 0xffffffff)
        ys = (x[:i, 1] & np.uint64(0xffffffff)).astype(np.int64); xs = (x[:i, 0] & np.uint64(0xffffffff)).astype(np.int64)
        cand = [j for j in range(i) if yi - prm.max_dist_inner <= ys[j] <= yi - 1 and xi - xs[j] <= prm.max_dist_inner and xs[j] != xi]
        cand.sort(key=lambda j: (-ys[j], -j))
        print("inner candidates (first 60): j y x f p", [(j, int(ys[j]), int(xs[j]), int(o["f"][j]), int(o["p"][j])) for j in cand[:60]])
        print("anchor x y", xi, yi)
    sys.exit(0)
    print("whole read", len(x), same(len(x)))
    lo, hi = 0, len(x)          # same(lo) true, same(hi) false
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if same(mid): lo = mid
        else: hi = mid
    n = hi
    print("first prefix that differs:", n)
    o = orc.lchain_rmq(x[:n], prm)
    i = n - 1
    print("anchor", i, "x", int(x[i, 0] & 0xffffffff), "y", int(x[i, 1] & 0xffffffff), "oracle f", int(o["f"][i]), "p", int(o["p"][i]), "n_tied", o["n_tied"])
    res, tied, _ = e.rmq_chain(x[:n], np.array([0, n], np.int64), to_lib(prm))
    print("device chains", res[0][0][:6], "oracle", o["u"][:6], "device tied", tied)
    # candidates of the inner walk in order
    yi = int(x[i, 1] & 0xffffffff); xi = int(x[i, 0] & 0xffffffff)
    ys = (x[:i, 1] & np.uint64(0xffffffff)).astype(np.int64); xs = (x[:i, 0] & np.uint64(0xffffffff)).astype(np.int64)
    cand = [j for j in range(i) if yi - prm.max_dist_inner <= ys[j] <= yi - 1 and xi - xs[j] <= prm.max_dist_inner]
    cand.sort(key=lambda j: (-ys[j], -j))
    print("inner candidates (first 40): j y f p", [(j, int(ys[j]), int(o["f"][j]), int(o["p"][j])) for j in cand[:40]])
