#!/usr/bin/env python3
"""One golden vector through engine.score under a few env settings; prints where f/p differ from the vector."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mm2gb_amd as mm, golden_io, orc
from test_gpu_parity import misc_from, rel
name = sys.argv[1] if len(sys.argv) > 1 else "synth_read_like"
g = golden_io.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
a = g["a"]; off = np.array([0, len(a)], np.int64)
for env in ({"MM2GB_FREE_SWEEP": "1"}, {"MM2GB_FREE_SWEEP": "0"}, {"MM2GB_LUT_CLAMP": "1"}):
    for k in ("MM2GB_FREE_SWEEP", "MM2GB_LUT_CLAMP"): os.environ.pop(k, None)
    os.environ.update(env)
    with mm.Engine() as e:
        e.set_misc(misc_from(g["prm"]))
        f, p, st = e.score(a, off)
    bad = np.flatnonzero((f != g["f"]) | (p != rel(g["p"])))
    print(env, "mismatches:", bad.size, "first:", bad[:8], "gpu f", f[bad[:8]], "want", g["f"][bad[:8]], "gpu p", p[bad[:8]], "want", rel(g["p"])[bad[:8]], flush=True)
    if bad.size:
        print("   stats:", {k: st[k] for k in list(st)[:12]})
