"""Soak tool (not collected by pytest): python tests/fuzz_soak.py SEED [SEED ...] [--iters N] [--teams] [--post] [--big N] [--huge N]
Runs tests/synth_cases.fuzz_case batches through the HIP path and the oracle; the first batch that differs is written
to gpurun_out/fuzz_fail_<seed>_<iteration>.npz (anchors, offsets, GPU f/p, parameters) and the exit code is 1.
--teams adds three engines whose planner thresholds send every chunk that fits the LDS ring to the big (8/16-wave) teams and to
the 4-wave teams (normally reserved for long chunks), and one that sends every chunk of two strips or more to a gang of workgroups, so the cooperative
paths see the same odd shapes."""
import argparse, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import mm2gb_amd as mm, orc, synth_cases as sc
from test_gpu_parity import misc_from, rel

ap = argparse.ArgumentParser()
ap.add_argument("seeds", type=int, nargs="+")
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--teams", action="store_true")
ap.add_argument("--post", action="store_true", help="also the device post-pass (mm2gb_chain_gpu): chains and compacted anchors against the host post-pass of the same scores")
ap.add_argument("--big", type=int, default=0, help="per seed, also this many batches of bench-like reads (10-100 kb, ~1-2 M anchors) with random parameters")
ap.add_argument("--huge", type=int, default=0, help="per seed, also this many batches of >= 20 M anchors (30-300 kb reads): the size at which team modes, gangs and the planner's "
                                                    "lists carry real load; every anchor against the oracle under every engine configuration")
args = ap.parse_args()
out_dir = os.path.join(os.path.dirname(HERE), "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
failed = False


def make_engine(env):
    for k, v in env.items():
        os.environ[k] = v
    e = mm.Engine()                        # the planner thresholds are read when the engine is created
    for k in env:
        del os.environ[k]
    return e


def post_differs(eng, a, off, prm, what):
    """mm2gb_chain_gpu (scores + device post-pass) against mm2gb_chain_host (same scores, host post-pass); True and a dump if they differ."""
    global failed
    if prm.n_seg > 1 and False:
        return False
    dev, _ = eng.chain_gpu(a, off)
    host, _ = eng.chain(a, off, threads=8)
    for r in range(len(off) - 1):
        if not (np.array_equal(dev[r][0], host[r][0]) and np.array_equal(dev[r][1], host[r][1])):
            print("POST-PASS differs:", what, "read", r, "chains", len(dev[r][0]), len(host[r][0]), flush=True)
            np.savez(os.path.join(out_dir, "fuzz_fail_post.npz"), a=a, off=off, prm=json.dumps(orc.param_to_dict(prm), default=float))
            failed = True
            return True
    return False


engines = [("default", make_engine({}))]
if args.teams:
    engines.append(("team8", make_engine({"MM2GB_LONG_MIN_COST": "1", "MM2GB_LONG_MIN_WINDOW": "1", "MM2GB_WIDE_WINDOW": "1", "MM2GB_WHOLE_WG_PCT": "0"})))
    engines.append(("team16", make_engine({"MM2GB_LONG_MIN_COST": "1", "MM2GB_LONG_MIN_WINDOW": "1", "MM2GB_WIDE_WINDOW": "1", "MM2GB_WHOLE_WG_PCT": "1"})))
    engines.append(("team4", make_engine({"MM2GB_LONG_MIN_COST": "1", "MM2GB_LONG_MIN_WINDOW": "1", "MM2GB_WIDE_WINDOW": "100000000"})))
    # every heavy chunk on a 4-wave team, also with windows wider than the team's share of the ring (older scores from global memory)
    engines.append(("team4all", make_engine({"MM2GB_LONG_MIN_COST": "1", "MM2GB_LONG_MIN_WINDOW": "1", "MM2GB_TEAM4_ALL": "1", "MM2GB_WHOLE_WG_PCT": "0"})))
    # every chunk of two strips or more on a gang of workgroups (round 4): scores and the rescue state across workgroups, quarters handed out early
    engines.append(("gangs", make_engine({"MM2GB_LONG_MIN_COST": "1", "MM2GB_LONG_MIN_WINDOW": "1", "MM2GB_WIDE_WINDOW": "1", "MM2GB_GANG_PCT": "100000", "MM2GB_GANG_MAX": "64",
                                          "MM2GB_GANG_MAX_ANCHORS": "2000000000"})))
seen = {name: [0, 0] for name, _ in engines}
gang_chunks = 0
for seed in args.seeds:
    rng = np.random.default_rng(seed)
    for it in range(args.iters):
        a, off, kw = sc.fuzz_case(rng)
        prm = orc.default_param(**kw)
        fo, po, pairs = orc.chain_fill_many(a, off, prm, threads=4)
        po_rel = np.concatenate([rel(po[off[r]:off[r + 1]]) for r in range(len(off) - 1)]) if len(a) else np.zeros(0, np.int32)
        for name, eng in engines:
            eng.set_misc(misc_from(prm))
            f, p, st = eng.score(a, off)
            seen[name][0] += st["n_long_chunks"]; seen[name][1] += st["n_mid_chunks"]
            if name == "gangs":
                gang_chunks += eng.gang_counts()[0]
            if args.post and name == "default":
                post_differs(eng, a, off, prm, f"seed {seed} iteration {it}")
            bad = np.flatnonzero((f != fo) | (p != po_rel))
            if bad.size or st["n_pairs"] != pairs:
                plain = {k: float(v) for k, v in kw.items()}
                print("engine", name, "seed", seed, "iteration", it, "differs at", bad[:10], "pairs", st["n_pairs"], pairs, "parameters", plain, "stats", st, flush=True)
                np.savez(os.path.join(out_dir, f"fuzz_fail_{name}_{seed}_{it}.npz"), a=a, off=off, f=f, p=p, kw=json.dumps(plain))
                failed = True
                break
        if failed:
            break
    else:
        print("seed", seed, "clean over", args.iters, "batches", flush=True)
    if failed:
        break
    # many planning blocks per read, look-back across blocks, every list of the planner in one batch
    for it in range(args.big + args.huge):
        huge = it >= args.big
        if huge:
            # >= 20 M anchors: reads are added until the batch is there (the generator is deterministic per read id)
            n_reads = 64
            while mm.synth_count(seed * 1000 + it, 0, n_reads, 30_000, 300_000) < 20_000_000:
                n_reads += 32
            a, off = mm.synth_reads(seed * 1000 + it, 0, n_reads, 30_000, 300_000, threads=16)
            kw = dict(max_iter=int(rng.choice([5000, 5000, 2000])), bw=int(rng.choice([500, 500, 100])), max_dist_x=5000, max_dist_y=5000,
                      pen_gap=np.float32(rng.choice([0.12, 0.19])), pen_skip=np.float32(0.0))
        else:
            a, off = mm.synth_reads(seed * 1000 + it, 0, int(rng.integers(8, 40)), 10_000, 100_000, threads=8)
            kw = dict(max_iter=int(rng.choice([100, 1000, 5000, 20000])), bw=int(rng.choice([100, 500, 2000])),
                      max_dist_x=int(rng.choice([1000, 5000, 10000])), max_dist_y=int(rng.choice([1000, 5000, 10000])),
                      pen_gap=np.float32(rng.choice([0.12, 0.19])), pen_skip=np.float32(0.0))
        prm = orc.default_param(**kw)
        fo, po, pairs = orc.chain_fill_many(a, off, prm, threads=16)
        po_rel = np.concatenate([rel(po[off[r]:off[r + 1]]) for r in range(len(off) - 1)])
        for name, eng in engines:
            eng.set_misc(misc_from(prm))
            f, p, st = eng.score(a, off)
            if args.post and name == "default" and not post_differs(eng, a, off, prm, f"BIG seed {seed} iteration {it}"):
                pass
            bad = np.flatnonzero((f != fo) | (p != po_rel))
            if bad.size or st["n_pairs"] != pairs:
                plain = {k: float(v) for k, v in kw.items()}
                print("BIG engine", name, "seed", seed, "iteration", it, "differs at", bad[:10], "pairs", st["n_pairs"], pairs, "parameters", plain, "stats", st, flush=True)
                np.savez(os.path.join(out_dir, f"fuzz_fail_big_{name}_{seed}_{it}.npz"), a=a, off=off, f=f, p=p, kw=json.dumps(plain))
                failed = True
                break
        if failed:
            break
    else:
        if args.big + args.huge:
            print("seed", seed, "clean over", args.big, "large batches and", args.huge, "of >= 20 M anchors", flush=True)
    if failed:
        break
print("chunks sent to big / 4-wave teams per engine:", seen, "| chunks scored by gangs (engine 'gangs'):", gang_chunks)
for _, e in engines:
    e.close()
sys.exit(1 if failed else 0)
