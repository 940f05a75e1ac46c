#!/usr/bin/env python3
"""Rate of the batch accumulator + dispatcher alone (SURVEY 8f N1, mm2gb_batcher_*): P producer threads feed reads one at a time
(mm2gb_batcher_add copies a read into the open batch OUTSIDE the batcher's mutex), batches of up to max_total_n anchors close by the rule of
map.c:887-920 and go to the engines' workers, every read's chains come back through the callback (counted, not copied).
  python profiles/batcher_rate.py [--reads 4000] [--producers 1 4] [--engines 1 2] [--out gpurun_out/batcher_rate.json]"""
import argparse, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import bench, mm2gb_amd as mm

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=4000)
ap.add_argument("--producers", type=int, nargs="+", default=[1, 4, 8])
ap.add_argument("--engines", type=int, nargs="+", default=[1, 2])
ap.add_argument("--max-total-n", type=int, default=40_000_000)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "batcher_rate.json"))
args = ap.parse_args()
threads = bench.cpu_quota() or os.cpu_count() or 8
a, off = mm.synth_reads(2024, 0, args.reads, 10_000, 100_000, threads=threads)
reads = [a[off[r]:off[r + 1]] for r in range(args.reads)]
n = int(off[-1])
cfg = mm.default_config()
cfg.max_total_n, cfg.max_read, cfg.min_n = args.max_total_n, 100_000, 0
cfg.has_max_total_n = cfg.has_max_read = 1
cfg.score_kernel.micro_batch = 1
rows = []
for post_threads, label in ((max(1, threads - 4), "host post-pass"), (0, "device post-pass")):
    for n_eng in args.engines:
        for prod in args.producers:
            with mm.Batcher(devices=[0] * n_eng, config=cfg, post_threads=post_threads, keep_results=False) as b:
                for rep in range(4):                              # the first rounds grow the page-locked batch buffers (every batch object once)
                    t0 = time.perf_counter()
                    b.feed(0, a, off, producers=prod)             # native producer threads (a Python loop would measure the interpreter)
                    t_fed = time.perf_counter() - t0
                    b.flush()
                    dt = time.perf_counter() - t0
                st = b.stats()
            rows.append({"post": label, "post_threads": post_threads, "engines_on_gpu0": n_eng, "producers": prod, "reads": args.reads, "anchors": n,
                         "seconds": round(dt, 4), "seconds_until_all_reads_were_added": round(t_fed, 4), "anchors_per_s": n / dt, "batches_per_round": st["batches"][0] // 4})
            print(json.dumps(rows[-1]), flush=True)
json.dump({"max_total_n": args.max_total_n, "usable_cpus": threads, "rows": rows}, open(args.out, "w"), indent=1)
