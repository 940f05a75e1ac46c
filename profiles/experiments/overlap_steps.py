#!/usr/bin/env python3
"""Experiment: what consecutive steps gain when they alternate between two sets of work arenas and compute streams
(here: two engines), so that step k+1's split / range / plan kernels (HBM-bound) run in the tail of step k's score kernel
(vector-ALU-bound, persistent workgroups that leave the chip one by one).  Prints ms per step for one engine, synchronised per
step (what bench.py times), one engine without the per-step sync, and two engines alternating.
usage: python profiles/experiments/overlap_steps.py [anchors] [steps]"""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
import mm2gb_amd as mm


def main():
    target = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    threads = bench.cpu_quota() or os.cpu_count() or 8
    _, n_reads, anchors, off = bench.shard_for_rank(mm, 0, 1, 1, target, 100_000, 300_000, threads)
    n = int(off[-1])
    dev = torch.device("cuda", 0)
    d_anchors = torch.from_numpy(anchors.view(np.int64)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    outs = [(torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)) for _ in range(2)]
    engs = [mm.Engine(device=0), mm.Engine(device=0)]

    def run(n_eng, sync_each):
        for e in engs:
            e.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            k = i % n_eng
            engs[k].score_device(n_reads, d_off.data_ptr(), d_anchors.data_ptr(), n, outs[k][0].data_ptr(), outs[k][1].data_ptr())
            if sync_each:
                engs[k].sync()
        for e in engs:
            e.sync()
        return (time.perf_counter() - t0) * 1e3 / steps

    for k in range(2):                       # warm-up: arenas of both engines
        engs[k].score_device(n_reads, d_off.data_ptr(), d_anchors.data_ptr(), n, outs[k][0].data_ptr(), outs[k][1].data_ptr())
        engs[k].sync()
    res = {"anchors": n, "steps": steps}
    for name, n_eng, sync_each in (("one_engine_sync_each_step", 1, True), ("one_engine_no_sync", 1, False), ("two_engines_alternating", 2, False),
                                   ("one_engine_sync_each_step_again", 1, True)):
        res[name + "_ms_per_step"] = round(min(run(n_eng, sync_each) for _ in range(2)), 3)
    same = bool(torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]))
    res["same_results_from_both_engines"] = same
    print(json.dumps(res))


if __name__ == "__main__":
    main()
