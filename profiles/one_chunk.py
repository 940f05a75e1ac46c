#!/usr/bin/env python3
"""How long does ONE heavy chunk take on an otherwise idle GPU, per way of running it?  (The serial chain through the tiles
of a chunk bounds small batches and the tail of large ones.)   python profiles/one_chunk.py [anchors] [xwin]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mm2gb_amd as mm, synth_cases as sc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
xwin = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
a = sc.sort_by_x(sc.repeat_block(n, 7, xwin=xwin, ywin=6000))
off = np.array([0, len(a)], np.int64)
configs = [("whole workgroup (16 waves)", {"MM2GB_BIG_TEAM": "16"}), ("8-wave team", {"MM2GB_BIG_TEAM": "8", "MM2GB_WHOLE_WG_PCT": "0"}),
           ("one wave", {"MM2GB_NO_COOP": "1"})]
for name, env in configs:
    os.environ.update(env)
    with mm.Engine() as e:
        e.score(a, off)
        best = min(e.score(a, off)[2]["ms_score"] for _ in range(3))
        st = e.score(a, off)[2]
    for k in env:
        del os.environ[k]
    tiles = (len(a) + 63) // 64
    print(f"{name:28s} {best:8.3f} ms   {st['n_pairs'] / best / 1e6:8.1f} G pairs/s   {best * 1e3 / tiles:6.2f} us per tile   "
          f"(pairs {st['n_pairs']}, tracked {st['n_tracked_chunks']}, big-team chunks {st['n_long_chunks']}, small-team {st['n_mid_chunks']})", flush=True)
