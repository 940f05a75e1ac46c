#!/usr/bin/env python3
"""PCIe-inclusive rate of mm2gb_score_host (page-locked buffers) against the slice size of its H2D / kernel / D2H pipeline."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, mm2gb_amd as mm

target = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
_, n_reads, anchors, off = bench.shard_for_rank(mm, 0, 1, 2024, target, 100_000, 300_000, threads=16)
n = len(anchors)
h_a = torch.empty((n, 2), dtype=torch.int64).pin_memory(); h_a.numpy()[:] = anchors.view(np.int64)
h_f = torch.empty(n, dtype=torch.int32).pin_memory(); h_p = torch.empty(n, dtype=torch.int32).pin_memory()
for sl in (64, 96, 128):
    os.environ["MM2GB_SLICE_ANCHORS"] = str(sl * 1000 * 1000)
    with mm.Engine() as eng:
        st = mm.Stats()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            rc = mm.lib().mm2gb_score_host(eng._h, n_reads, off.ctypes.data, h_a.data_ptr(), h_f.data_ptr(), h_p.data_ptr(), ctypes.byref(st))
            dt = time.perf_counter() - t0
            assert rc == 0
            best = dt if best is None else min(best, dt)
    print(f"slice {sl:6d} M anchors: {best * 1e3:7.1f} ms  {n / best / 1e9:5.2f} G anchors/s  {st.n_pairs / best / 1e12:5.2f} T pairs/s  kernels {st.ms_prep + st.ms_score:6.1f} ms", flush=True)
