#!/usr/bin/env python3
"""The device RMQ fill on the reads of profiles/rmq_rate.py per kernel form and skip limit: the C call's seconds (mm2gb_rmq_chain_gpu alone)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, mm2gb_amd as mm, orc
threads = bench.cpu_quota() or os.cpu_count() or 8
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
a, off = mm.synth_reads(2024, 0, n_reads, 100_000, 300_000, threads=threads)
with mm.Engine() as e:
    first, _ = e.chain(a, off, threads=threads)
    reads = [orc.radix_sort_x(x[1]) if len(x[1]) else x[1] for x in first]
    o2 = np.zeros(len(reads) + 1, np.int64); o2[1:] = np.cumsum([len(x) for x in reads])
    allr = np.concatenate(reads)
    for kernel, skip, kw in (("tiles", orc.INT32_MAX, {}), ("steps", orc.INT32_MAX, {}), ("steps", 25, {}), ("steps", 0, {}), ("steps", 1000, {}), ("steps", 25, dict(max_dist_inner=0)), ("tiles", orc.INT32_MAX, dict(max_dist_inner=0))):
        os.environ["MM2GB_RMQ_KERNEL"] = kernel
        prm = mm.default_rmq_param(max_chn_skip=skip, **kw)
        if kw: print(kw, end=" ")
        best = None
        for _ in range(2):
            res, tied, st = e.rmq_chain(allr, o2, prm)
            best = st["ms_total"] if best is None else min(best, st["ms_total"])
        print(kernel, "skip", skip, ": %.1f ms for %d reads, %d anchors (%.2f us per anchor of the longest read's %d), post-pass %.1f ms, reads with ties %d" % (best, len(reads), o2[-1], best * 1e3 / max(len(x) for x in reads), max(len(x) for x in reads), st["ms_post"], int((tied > 0).sum())), flush=True)
