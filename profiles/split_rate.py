#!/usr/bin/env python3
"""Kernel time of small and medium micro-batches with and without the SPLIT build of k_score (heavy chunks scored strip by strip with
the help of idle workgroups): ms of split+window+plan+score per batch (HIP events, best of 5), chunks split, items helpers took.
python profiles/split_rate.py > gpurun_out/split_rate.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mm2gb_amd as mm
cases = [("64 reads of 10-100 kb", 64, 10_000, 100_000), ("64 reads of 100-300 kb", 64, 100_000, 300_000), ("512 reads of 10-100 kb", 512, 10_000, 100_000),
         ("2000 reads of 30-100 kb", 2000, 30_000, 100_000), ("1426 reads of 200-300 kb (100 M anchors)", 1426, 200_000, 300_000)]
out = []
for name, n_reads, lo, hi in cases:
    a, off = mm.synth_reads(2024, 0, n_reads, lo, hi, threads=16)
    row = {"batch": name, "anchors": int(len(a))}
    ref = None
    for label, env in (("plain", "0"), ("split", "200000000")):
        os.environ["MM2GB_SPLIT_MAX_ANCHORS"] = env
        with mm.Engine() as e:
            e.set_misc(mm.default_misc())
            best = 1e9
            for _ in range(6):
                f, p, st = e.score(a, off)
                best = min(best, st["ms_prep"] + st["ms_score"])
            row[label] = {"ms": round(best, 3), "pairs_per_s": st["n_pairs"] / best * 1e3, "split_chunks_helped_items": e.split_counts()}
            if ref is None: ref = (f.copy(), p.copy())
            else: row["same_results"] = bool(np.array_equal(ref[0], f) and np.array_equal(ref[1], p))
    row["speedup"] = round(row["plain"]["ms"] / row["split"]["ms"], 3)
    out.append(row); print(json.dumps(row), file=sys.stderr, flush=True)
print(json.dumps(out))
