#!/usr/bin/env python3
"""Ties of the RMQ re-chaining's range minimum on the reads of profiles/rmq_rate.py: how many reads meet one, and how many of those ties can
change the anchor's score or predecessor (the others need not be broken the reference's way: csrc/rmq_host.cpp).  Host form on all threads,
MM2GB_RMQ_TIES=strict (every tie -> the reference's tree) against the default.   python profiles/tie_stats.py [--reads 2000]"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=2000)
ap.add_argument("--child", default="")
args = ap.parse_args()
import numpy as np
import bench, mm2gb_amd as mm, orc
threads = bench.cpu_quota() or os.cpu_count() or 8
cache = os.path.join(ROOT, "gpurun_out", "tie_stats_input.npz")
if not args.child:
    a, off = mm.synth_reads(2024, 0, args.reads, 100_000, 300_000, threads=threads)
    with mm.Engine() as e:
        first, _ = e.chain(a, off, threads=threads)
    reads = [orc.radix_sort_x(x[1]) if len(x[1]) else x[1] for x in first]
    o2 = np.zeros(len(reads) + 1, np.int64); o2[1:] = np.cumsum([len(x) for x in reads])
    os.makedirs(os.path.dirname(cache), exist_ok=True)
    np.savez(cache, a=np.concatenate(reads), off=o2)
    for mode in ("strict", "weigh"):
        env = dict(os.environ, MM2GB_DEBUG_PHASES="1")
        if mode == "strict": env["MM2GB_RMQ_TIES"] = "strict"
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode], env=env, check=True)
    os.remove(cache)
else:
    z = np.load(cache)
    prm = mm.default_rmq_param()
    best = None
    for _ in range(3):
        t0 = time.perf_counter(); res, tied = mm.rmq_chain_host(z["a"], z["off"], prm, threads=threads); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    import hashlib
    h = hashlib.sha256()
    for u, ao in res: h.update(np.ascontiguousarray(u).tobytes()); h.update(np.ascontiguousarray(ao).tobytes())
    print(json.dumps({"ties": args.child, "reads": len(res), "anchors": int(z["off"][-1]), "reads_done_again": int((tied > 0).sum()), "host_form_s": round(best, 4), "threads": threads, "chains_sha256_16": h.hexdigest()[:16]}), flush=True)
