#!/usr/bin/env python3
"""Soak of the split build (one chunk on several workgroups): the same batches scored again and again, every result compared with the
oracle's.  python profiles/split_soak.py [rounds]   (MM2GB_LIB_PATH picks the build)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mm2gb_amd as mm, orc, synth_cases as sc
from test_gpu_parity import misc_from, rel, band_cloud
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
parts = [sc.sort_by_x(np.concatenate([sc.repeat_block(23000, 401, xwin=4500, ywin=7000), sc.colinear(900, 402)])),
         sc.sort_by_x(sc.repeat_block(9000, 403, xwin=9000, ywin=9000, r0=4_000_000)),
         band_cloud(12345, 404, xwin=11000, jitter=800), sc.read_like(9000, 405), sc.rescue_case(n_noise=9000, n_chain=60, seed=23),
         sc.sort_by_x(sc.repeat_block(17000 + 64 * 3 + 7, 406, xwin=3000, ywin=4000, r0=6_000_000))]
off = np.zeros(len(parts) + 1, dtype=np.int64); off[1:] = np.cumsum([len(x) for x in parts])
a = np.concatenate(parts)
prm = orc.default_param()
fo, po, _ = orc.chain_fill_many(a, off, prm, threads=8)
po_rel = np.concatenate([rel(po[off[r]:off[r + 1]]) for r in range(len(off) - 1)])
bad_runs = 0; worst = 0; where = {}
for env in ({}, {"MM2GB_WHOLE_WG_PCT": "1"}):
    os.environ.pop("MM2GB_WHOLE_WG_PCT", None); os.environ.update(env)
    with mm.Engine() as e:
        e.set_misc(misc_from(prm))
        t0 = time.perf_counter()
        for r in range(rounds):
            f, p, st = e.score(a, off)
            bad = np.flatnonzero((f != fo) | (p != po_rel))
            if bad.size:
                bad_runs += 1; worst = max(worst, bad.size)
                rd = int(np.searchsorted(off, bad[0], side="right") - 1)
                where.setdefault((tuple(env.items()), rd, int((bad[0] - off[rd]) // 1024)), 0)
                where[(tuple(env.items()), rd, int((bad[0] - off[rd]) // 1024))] += 1
        print(env, "split chunks / helped items of the last call:", e.split_counts(), "ms per call", round((time.perf_counter() - t0) / rounds * 1e3, 2), flush=True)
print("lib", os.environ.get("MM2GB_LIB_PATH", "main"), "rounds", 2 * rounds, "runs with mismatches", bad_runs, "worst", worst, "first mismatch (env, read, strip):", where)
