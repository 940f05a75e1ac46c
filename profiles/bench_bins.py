#!/usr/bin/env python3
"""Per-length-bin chaining throughput (SURVEY 8d bins) with the CPU baseline beside every GPU number.

For each bin: synthetic reads of that length range (same generator and seed policy as bench.py), ~--anchors anchors resident
in HBM, K timed steps of the hot path; then the oracle's score fill on a bounded sample with 1 thread and with one socket's
worth of threads, built -O3 (reference flags) and -O3 -march=native.  Writes one JSON document.

    python profiles/bench_bins.py --anchors 100000000 --out profiles/r01_bins.json
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench            # noqa: E402
import mm2gb_amd as mm  # noqa: E402
import orc              # noqa: E402

BINS = [(10_000, 30_000), (30_000, 100_000), (100_000, 200_000), (200_000, 300_000)]


def native_oracle():
    """Same source, -march=native (FMA contraction stays off so results are unchanged)."""
    path = os.path.join(ROOT, "oracle", "libchain_oracle_native.so")
    src = os.path.join(ROOT, "oracle", "chain_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fPIC", "-shared", src, "-o", path, "-lm", "-lpthread"])
    L = C.CDLL(path)
    L.orc_chain_fill_reads_mt.restype = C.c_int64
    L.orc_chain_fill_reads_mt.argtypes = [C.POINTER(orc.Param), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return L


def cpu_rate(L, anchors, off, threads, budget_s):
    """pairs/s of orc_chain_fill over reads dealt to `threads` pthreads; sample sized to ~budget_s."""
    prm = orc.default_param()
    n_reads = len(off) - 1
    f = np.empty(len(anchors), np.int32)
    p = np.empty(len(anchors), np.int64)
    off = np.ascontiguousarray(off, dtype=np.int64)

    def run(k):
        t0 = time.perf_counter()
        pairs = L.orc_chain_fill_reads_mt(C.byref(prm), k, off.ctypes.data, anchors.ctypes.data, f.ctypes.data, p.ctypes.data, threads)
        return pairs, time.perf_counter() - t0

    cal = min(n_reads, max(threads, 4))
    pairs, dt = run(cal)
    want = int(min(n_reads, max(cal, cal * budget_s / max(dt, 1e-3))))
    pairs, dt = run(want)
    return pairs / dt, want, pairs, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--anchors", type=int, default=100_000_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--cpu-seconds", type=float, default=4.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "bins.json"))
    args = ap.parse_args()
    import torch
    dev = torch.device("cuda", 0)
    cores, model = bench.cores_per_socket()
    quota = bench.cpu_quota()
    threads = max(1, min(cores, quota or cores))         # more threads than usable CPUs only get throttled
    L_ref = orc.lib()
    L_nat = native_oracle()
    doc = {"gpu": torch.cuda.get_device_name(0), "cpu": model, "cores_per_socket": cores, "usable_cpus": quota, "cpu_threads": threads, "anchors_per_bin": args.anchors, "bins": []}
    eng = mm.Engine(device=0)
    for lo, hi in BINS:
        first, n_reads, a, off = bench.shard_for_rank(mm, 0, 1, 2024, args.anchors, lo, hi, threads=64)
        n = int(off[-1])
        d_a = torch.from_numpy(a.view(np.int64)).to(dev)
        d_off = torch.from_numpy(off).to(dev)
        d_f = torch.empty(n, dtype=torch.int32, device=dev)
        d_p = torch.empty(n, dtype=torch.int32, device=dev)
        eng.score_device(n_reads, d_off.data_ptr(), d_a.data_ptr(), n, d_f.data_ptr(), d_p.data_ptr()); eng.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.score_device(n_reads, d_off.data_ptr(), d_a.data_ptr(), n, d_f.data_ptr(), d_p.data_ptr()); eng.sync()
        wall = (time.perf_counter() - t0) / args.steps
        st = eng.stats()
        row = {"read_len": [lo, hi], "reads": n_reads, "anchors": n, "pairs": st["n_pairs"], "pairs_per_anchor": round(st["n_pairs"] / n, 1),
               "gpu_pairs_per_s": st["n_pairs"] / wall, "gpu_ms_per_step": wall * 1e3, "gpu_score_kernel_ms": st["ms_score"], "gpu_prep_ms": st["ms_prep"],
               "long_chunks": st["n_long_chunks"], "mid_chunks": st["n_mid_chunks"], "tracked_chunks": st["n_tracked_chunks"], "cpu": {}}
        for tag, L in (("O3", L_ref), ("O3_native", L_nat)):
            for th in (1, threads):
                rate, reads_used, pairs, dt = cpu_rate(L, a, off, th, args.cpu_seconds)
                row["cpu"][f"{tag}_{th}t"] = {"pairs_per_s": rate, "reads": reads_used, "seconds": round(dt, 2)}
        row["speedup_vs_cpu_O3"] = row["gpu_pairs_per_s"] / row["cpu"][f"O3_{threads}t"]["pairs_per_s"]
        doc["bins"].append(row)
        print(json.dumps(row), flush=True)
        del d_a, d_off, d_f, d_p
    json.dump(doc, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
