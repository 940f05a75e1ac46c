#!/usr/bin/env python3
"""The device post-pass alone (backtrack + compaction kernels) on one HBM-resident micro-batch of the bench's workload: score once, then
mm2gb_post_device a few times; per-run milliseconds, totals and a digest of the chains left on the device (to compare builds / settings at
full size).  MM2GB_DEBUG_PHASES=1 in the environment adds the library's per-phase sums and the schedule of the reads that end last (stderr).
    python3 profiles/post_only.py [--anchors 500000000] [--runs 3] [--lo 100000 --hi 300000] [--out file.json]"""
import argparse, ctypes as C, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, mm2gb_amd as mm
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--anchors", type=int, default=500_000_000)
ap.add_argument("--runs", type=int, default=3)
ap.add_argument("--lo", type=int, default=100_000)
ap.add_argument("--hi", type=int, default=300_000)
ap.add_argument("--seed", type=int, default=2024)
ap.add_argument("--out", default="")
args = ap.parse_args()
_, n_reads, a, off = bench.shard_for_rank(mm, 0, 1, args.seed, args.anchors, args.lo, args.hi, threads=min(64, os.cpu_count() or 8))
n = int(off[-1])
dev = torch.device("cuda", 0)
d_a = torch.from_numpy(a.view(np.int64)).to(dev)
d_off = torch.from_numpy(off).to(dev)
d_f = torch.empty(n, dtype=torch.int32, device=dev)
d_p = torch.empty(n, dtype=torch.int32, device=dev)
eng = mm.Engine(device=0)
eng.score_device(n_reads, d_off.data_ptr(), d_a.data_ptr(), n, d_f.data_ptr(), d_p.data_ptr())
eng.sync()
st = eng.stats()
L = mm.lib()
rows = []
for k in range(args.runs):
    n_ch, n_kept, ms = C.c_int64(0), C.c_int64(0), C.c_float(0)
    t0 = time.perf_counter()
    rc = L.mm2gb_post_device(eng._h, n_reads, d_off.data_ptr(), d_a.data_ptr(), n, d_f.data_ptr(), d_p.data_ptr(), C.byref(n_ch), C.byref(n_kept), C.byref(ms))
    assert rc == 0, L.mm2gb_last_error().decode()
    rows.append({"ms": round(ms.value, 3), "call_s": round(time.perf_counter() - t0, 4), "chains": n_ch.value, "kept": n_kept.value})
    print(rows[-1], flush=True)
dg = (C.c_uint64 * 4)()
assert L.mm2gb_post_device_digest(eng._h, n_reads, dg) == 0, L.mm2gb_last_error().decode()
res = {"anchors": n, "reads": n_reads, "pairs": st["n_pairs"], "score_ms": st["ms_score"], "runs": rows, "best_ms": min(r["ms"] for r in rows),
       "digest": ["%016x" % v for v in dg]}
print(json.dumps(res))
if args.out:
    json.dump(res, open(args.out, "w"), indent=1)
