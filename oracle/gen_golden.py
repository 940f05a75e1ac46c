#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE built by oracle/Makefile (dev container only).

What is recorded (data only -- inputs and expected outputs, no reference source text):
  * real_<pair>_<skip>.npz : for each reference test pair (test/*.fa) the anchors the reference's own
    seeding handed to mg_lchain_dp (map.c:523), the per-anchor f[]/p[] it computed (lchain.c:169-207, observed through
    oracle/capture_hooks.c), and the chains u[] / compacted anchors it returned; plus the PAF it printed.
  * synth_<case>.npz       : seeded synthetic anchor sets (tests/synth_cases.py) pushed through the reference's
    mg_lchain_dp in-process, same arrays.
  * data/*.fa              : the reference's test FASTA files (its only test data).

Run:  make -C oracle all && python oracle/gen_golden.py
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc            # noqa: E402
import synth_cases as sc  # noqa: E402

REF = os.environ.get("MM2GB_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")
PAIRS = {"mt": ("MT-human.fa", "MT-orang.fa"), "inv": ("t-inv.fa", "q-inv.fa"), "q2": ("t2.fa", "q2.fa")}
SKIPS = {"inf": orc.INT32_MAX, "s25": 25}


def save_case(path, a, prm, f, p, u, a_out, extra=None):
    assert np.all(p < 2**31) and np.all(p >= -1)
    meta = dict(param=orc.param_to_dict(prm))
    meta["param"]["pen_gap"] = float(np.float32(prm.pen_gap))
    meta["param"]["pen_skip"] = float(np.float32(prm.pen_skip))
    if extra:
        meta.update(extra)
    np.savez_compressed(path, a=a, f=f.astype(np.int32), p=p.astype(np.int32), u=u, a_out=a_out,
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))


def real_pairs():
    os.makedirs(os.path.join(GOLD, "data"), exist_ok=True)
    exe = os.path.join(orc.REF_DIR, "minimap2_cpu")
    hook = os.path.join(orc.REF_DIR, "libcapture.so")
    for name, (tgt, qry) in PAIRS.items():
        for fn in (tgt, qry):
            shutil.copyfile(os.path.join(REF, "test", fn), os.path.join(GOLD, "data", fn))
        for tag, skip in SKIPS.items():
            with tempfile.TemporaryDirectory() as td:
                cap = os.path.join(td, "cap.bin")
                env = dict(os.environ, LD_PRELOAD=hook, MM2GB_CAPTURE=cap)
                paf = subprocess.run([exe, "-t", "1", f"--max-chain-skip={skip}", os.path.join(REF, "test", tgt),
                                      os.path.join(REF, "test", qry)], env=env, check=True, capture_output=True).stdout.decode()
                recs = orc.read_capture(cap) if os.path.exists(cap) else []
            with open(os.path.join(GOLD, f"real_{name}_{tag}.paf"), "w") as fh:
                fh.write(paf)
            for k, r in enumerate(recs):
                assert r["f"] is not None
                save_case(os.path.join(GOLD, f"real_{name}_{tag}_{k}.npz"), r["a"], r["prm"], r["f"], r["p"], r["u"], r["a_out"],
                          extra=dict(source=f"{tgt} x {qry}", max_chain_skip=skip, record=k))
            print(f"real {name} {tag}: {len(recs)} chaining calls, {sum(len(r['a']) for r in recs)} anchors, {paf.count(chr(10))} PAF lines")


def synth_cases():
    P = orc.default_param
    cases = {
        "noise": (sc.noise(4000, 1, n_rid=2, span=200000), P()),
        "read_like": (sc.read_like(20000, 11), P()),
        "repeat_sat": (sc.sort_by_x(np.concatenate([sc.repeat_block(6200, 3), sc.colinear(400, 4)])), P()),
        "rescue9000": (sc.rescue_case(), P()),
        "rescue_small": (sc.rescue_case(n_noise=300, n_chain=30), P(max_iter=100)),
        "ties": (sc.grid_ties(), P()),
        "two_seg": (sc.two_segments(400, 5), P(n_seg=2)),
        "cdna_two_seg": (sc.two_segments(400, 6), P(is_cdna=1, n_seg=2)),
        "cdna": (sc.read_like(5000, 8), P(is_cdna=1)),
        "varspan": (sc.variable_span(800, 9), P()),
        "skip25": (sc.read_like(20000, 12), P(max_skip=25)),
        "skip0": (sc.read_like(20000, 13), P(max_skip=0)),
        "penskip": (sc.read_like(8000, 14), P(pen_skip=np.float32(0.05))),
        "smallbw": (sc.read_like(8000, 15), P(bw=100, max_dist_x=50, max_dist_y=60)),
        "iter64": (sc.read_like(8000, 16), P(max_iter=64)),
        "single": (sc.noise(1, 17), P()),
        "pair": (sc.colinear(2, 18), P(min_cnt=1, min_sc=1)),
    }
    for name, (a, prm) in cases.items():
        r = orc.ref_lchain_dp(a, prm)
        save_case(os.path.join(GOLD, f"synth_{name}.npz"), a, prm, r["f"], r["p"], r["u"], r["a_out"], extra=dict(source="tests/synth_cases.py"))
        print(f"synth {name}: {len(a)} anchors, {len(r['u'])} chains")


if __name__ == "__main__":
    if not orc.ref_available():
        sys.exit("reference build missing: run `make -C oracle all` in a container that has /root/reference")
    os.makedirs(GOLD, exist_ok=True)
    real_pairs()
    synth_cases()
