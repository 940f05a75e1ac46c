#!/usr/bin/env python3
"""The device post-pass of single reads of the bench batch, alone on the GPU, with the kernels' own accounting (MM2GB_DEBUG_PHASES=1):
what the reads that end k_post_chains spend where.  usage: MM2GB_DEBUG_PHASES=1 python profiles/experiments/post_one_read.py [read ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["MM2GB_DEBUG_PHASES"] = "1"
import numpy as np
import bench, mm2gb_amd as mm

reads = [int(x) for x in sys.argv[1:]] or [5468, 6614, 6920, 4454]
_, n_reads, anchors, off = bench.shard_for_rank(mm, 0, 1, 2024, 500_000_000, 100_000, 300_000, 16)
print("batch:", n_reads, "reads", int(off[-1]), "anchors", flush=True)
with mm.Engine() as e:
    for r in reads:
        a = np.ascontiguousarray(anchors[off[r]:off[r + 1]])
        o = np.array([0, len(a)], np.int64)
        for team in ("1", "0"):
            os.environ["MM2GB_POST_TEAM_READS"] = team
            print(f"=== read {r}: {len(a)} anchors, alone, team_reads {team}", file=sys.stderr, flush=True)
            e.chain_gpu(a, o)
            e.chain_gpu(a, o)
