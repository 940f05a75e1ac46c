#!/usr/bin/env python3
"""Whole-batch chaining on host buffers (mm2gb_chain_host: scores on the GPU, backtrack + compaction on host threads that
start on each slice as soon as it is back) next to the score call alone, per thread count; then two engines on one GPU.
    python profiles/postpass.py --anchors 100000000 --out profiles/r01_postpass.json"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, mm2gb_amd as mm

ap = argparse.ArgumentParser()
ap.add_argument("--anchors", type=int, default=100_000_000)
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "postpass.json"))
args = ap.parse_args()
_, n_reads, a, off = bench.shard_for_rank(mm, 0, 1, 2024, args.anchors, 100_000, 300_000, threads=64)
eng = mm.Engine()
f, p, st = eng.score(a, off)                      # warm-up + scores for the post-pass
rows = []
misc = eng.misc
L = mm.lib()
for th in (1, 8, 32, 64, 128):
    out, stats = mm.Chains(), mm.Stats()
    t0 = time.perf_counter()
    rc = L.mm2gb_chain_host(eng._h, n_reads, off.ctypes.data, a.ctypes.data, th, C.byref(out), C.byref(stats))
    dt = time.perf_counter() - t0
    assert rc == 0
    n_chains = int(np.ctypeslib.as_array(out.u_off, shape=(n_reads + 1,))[-1])
    L.mm2gb_chains_free(C.byref(out))
    rows.append({"threads": th, "chain_host_s": round(dt, 3), "gpu_score_call_ms": round(stats.ms_total, 1),
                 "beyond_the_score_call_s": round(dt - stats.ms_total / 1e3, 3), "anchors_per_s": len(a) / dt, "chains": n_chains})
    print(rows[-1], flush=True)
eng.close()
pool_rows = []
with mm.Pool(devices=[0, 0]) as pool:
    for th in (8, 32):
        out, stats = mm.Chains(), mm.Stats()
        t0 = time.perf_counter()
        rc = L.mm2gb_pool_chain_host(pool._h, n_reads, off.ctypes.data, a.ctypes.data, th, C.byref(out), C.byref(stats))
        dt = time.perf_counter() - t0
        assert rc == 0
        L.mm2gb_chains_free(C.byref(out))
        pool_rows.append({"engines": 2, "threads": th, "chain_host_s": round(dt, 3), "slowest_engine_score_call_ms": round(stats.ms_total, 1)})
        print(pool_rows[-1], flush=True)
json.dump({"anchors": len(a), "reads": n_reads, "rows": rows, "two_engines_on_one_gpu": pool_rows,
           "note": "pageable numpy buffers in and out; the score call is sliced (64 M anchors) and overlapped on three streams; post-pass threads consume slices as they land"},
          open(args.out, "w"), indent=1)
