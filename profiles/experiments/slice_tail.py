#!/usr/bin/env python3
"""mm2gb_score_host from page-locked buffers against the size of the LAST slice of its H2D / kernel / D2H pipeline (what follows the last
copy is exposed: that slice's kernels and its D2H).   python profiles/experiments/slice_tail.py [anchors]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import bench, mm2gb_amd as mm

target = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
_, n_reads, anchors, off = bench.shard_for_rank(mm, 0, 1, 1, target, 100_000, 300_000, threads=16)
n = len(anchors)
h_a = torch.empty((n, 2), dtype=torch.int64).pin_memory(); h_a.numpy()[:] = anchors.view(np.int64)
h_f = torch.empty(n, dtype=torch.int32).pin_memory(); h_p = torch.empty(n, dtype=torch.int32).pin_memory()
for tail, first, slice_m, dual in ((24, 96, 96, 16), (24, 32, 96, 16), (24, 16, 96, 16), (24, 32, 64, 16), (24, 32, 96, 1000), (24, 32, 64, 1000), (24, 16, 48, 1000)):
    os.environ["MM2GB_SLICE_TAIL_ANCHORS"] = str(tail * 1000 * 1000)
    os.environ["MM2GB_SLICE_FIRST_ANCHORS"] = str(first * 1000 * 1000)
    os.environ["MM2GB_SLICE_ANCHORS"] = str(slice_m * 1000 * 1000)
    os.environ["MM2GB_DUAL_STREAM_MAX"] = str(dual * 1000 * 1000)
    with mm.Engine() as eng:
        st = mm.Stats()
        times = []
        for _ in range(4):
            t0 = time.perf_counter()
            rc = mm.lib().mm2gb_score_host(eng._h, n_reads, off.ctypes.data, h_a.data_ptr(), h_f.data_ptr(), h_p.data_ptr(), ctypes.byref(st))
            times.append(time.perf_counter() - t0)
            assert rc == 0
    best = min(times)
    print(f"{n} anchors, first {first} M, slices {slice_m} M, last {tail} M, two compute streams up to {dual} M: {best * 1e3:7.1f} ms  input {n * 16 / best / 1e9:5.1f} GB/s  (all: {[round(t * 1e3, 1) for t in times]})", flush=True)
