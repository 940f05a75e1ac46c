"""Soak tool (not collected by pytest): python tests/fuzz_soak.py SEED [SEED ...] [--iters N]
Runs tests/synth_cases.fuzz_case batches through the HIP path and the oracle; the first batch that differs is written
to gpurun_out/fuzz_fail_<seed>_<iteration>.npz (anchors, offsets, GPU f/p, parameters) and the exit code is 1."""
import argparse, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import mm2gb_amd as mm, orc, synth_cases as sc
from test_gpu_parity import misc_from, rel

ap = argparse.ArgumentParser()
ap.add_argument("seeds", type=int, nargs="+")
ap.add_argument("--iters", type=int, default=200)
args = ap.parse_args()
out_dir = os.path.join(os.path.dirname(HERE), "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
failed = False
with mm.Engine() as eng:
    for seed in args.seeds:
        rng = np.random.default_rng(seed)
        for it in range(args.iters):
            a, off, kw = sc.fuzz_case(rng)
            prm = orc.default_param(**kw)
            eng.set_misc(misc_from(prm))
            f, p, st = eng.score(a, off)
            fo, po, pairs = orc.chain_fill_many(a, off, prm, threads=4)
            po_rel = np.concatenate([rel(po[off[r]:off[r + 1]]) for r in range(len(off) - 1)]) if len(a) else np.zeros(0, np.int32)
            bad = np.flatnonzero((f != fo) | (p != po_rel))
            if bad.size or st["n_pairs"] != pairs:
                plain = {k: float(v) for k, v in kw.items()}
                print("seed", seed, "iteration", it, "differs at", bad[:10], "pairs", st["n_pairs"], pairs, "parameters", plain, "stats", st, flush=True)
                np.savez(os.path.join(out_dir, f"fuzz_fail_{seed}_{it}.npz"), a=a, off=off, f=f, p=p, kw=json.dumps(plain))
                failed = True
                break
        else:
            print("seed", seed, "clean over", args.iters, "batches", flush=True)
sys.exit(1 if failed else 0)
