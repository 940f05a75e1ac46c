#!/usr/bin/env python3
"""Which read of a bench batch sets the pace of a small batch, and how long does it take alone per way of running it?
usage: python profiles/experiments/slowest_read.py [anchors]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import bench, mm2gb_amd as mm

target = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
_, n_reads, anchors, off = bench.shard_for_rank(mm, 0, 1, 1, target, 100_000, 300_000, 16)
with mm.Engine() as e:
    e.score(anchors, off)
    st = e.score(anchors, off)[2]
    print("batch:", n_reads, "reads", int(off[-1]), "anchors; ms_score", round(st["ms_score"], 3), "gangs", e.gang_counts(), flush=True)
    rows = []
    for r in range(n_reads):
        a = anchors[off[r]:off[r + 1]]
        o = np.array([0, len(a)], np.int64)
        e.score(a, o)
        s = e.score(a, o)[2]
        rows.append((s["ms_score"], r, len(a), s["n_pairs"], s["n_tracked_chunks"], s["n_long_chunks"], s["n_mid_chunks"], s["n_chunks"]))
rows.sort(reverse=True)
print("slowest reads alone (ms_score, read, anchors, pairs, tracked chunks, big-team chunks, small-team chunks, chunks):")
for row in rows[:6]:
    print("  ", row, flush=True)
r = rows[0][1]
a = anchors[off[r]:off[r + 1]]
o = np.array([0, len(a)], np.int64)
for name, env in (("default", {}), ("no gangs", {"MM2GB_GANG_MAX": "0"}), ("gang of 2", {"MM2GB_GANG_MAX": "2"}), ("gang of 4", {"MM2GB_GANG_MAX": "4"}),
                  ("no gangs, no whole workgroup", {"MM2GB_GANG_MAX": "0", "MM2GB_WHOLE_WG_PCT": "0"}), ("no gangs, one wave per chunk", {"MM2GB_GANG_MAX": "0", "MM2GB_NO_COOP": "1"})):
    os.environ.update(env)
    with mm.Engine() as e:
        e.score(a, o)
        best = min(e.score(a, o)[2]["ms_score"] for _ in range(3))
        g = e.gang_counts()
    for k in env:
        del os.environ[k]
    print(f"read {r} alone, {name:32s} {best:8.3f} ms  gangs {g}", flush=True)
