#!/usr/bin/env python3
"""Generate tests/golden/rmq/*.npz from the REFERENCE's mg_lchain_rmq (lchain.c:250-369), built by oracle/Makefile and called
in-process (dev container only).  Data only: the anchors given to the call, its parameters, the per-anchor f[] / p[] it computed
(observed through oracle/capture_hooks.c at its call of mg_chain_backtrack, lchain.c:355) and what it returned.

Inputs are what post_chaining_helper hands it (map.c:444-451): the anchors the first chaining kept, re-sorted by x with the
reference's radix_sort_128x.  Cases where the reference had to break a tie between equal range-minimum priorities by the shape
of its tree are kept too and marked (`tied` = what the oracle counted): implementations cannot be expected to reproduce those,
only to report them.

Run:  make -C oracle all && python oracle/gen_golden_rmq.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc                 # noqa: E402
import synth_cases as sc   # noqa: E402
import mm2gb_amd as mm     # noqa: E402  (only its deterministic read generator, mm2gb_synth_*: no GPU involved)

OUT = os.path.join(ROOT, "tests", "golden", "rmq")


def first_pass(a):
    o = orc.lchain_dp(a, orc.default_param(), want_fp=False)
    return orc.ref_radix_sort(o["a_out"]) if len(o["a_out"]) else o["a_out"]


def save(name, a, prm):
    rf = orc.ref_lchain_rmq(a, prm)
    o = orc.lchain_rmq(a, prm)
    d = {k: (float(np.float32(getattr(prm, k))) if k.startswith("pen_") else int(getattr(prm, k))) for k, _ in orc.RmqParam._fields_}
    if o["n_tied"] == 0:
        assert np.array_equal(o["f"], rf["f"]) and np.array_equal(o["p"], rf["p"]) and np.array_equal(o["u"], rf["u"]), name
    meta = dict(param=d, tied=int(o["n_tied"]))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), a=a, f=rf["f"].astype(np.int32), p=rf["p"].astype(np.int32), u=rf["u"], a_out=rf["a_out"],
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    print(f"{name}: n={len(a)} chains={len(rf['u'])} tied={o['n_tied']}")


def main():
    os.makedirs(OUT, exist_ok=True)
    a, off = mm.synth_reads(3, 0, 6, 10_000, 60_000)
    for r in (0, 2, 5):
        save(f"ont_{r}", first_pass(a[off[r]:off[r + 1]]), orc.default_rmq_param())
    x = first_pass(a[off[1]:off[2]])
    save("ont_skip25", x, orc.default_rmq_param(max_chn_skip=25))
    save("ont_cap40", x, orc.default_rmq_param(cap_rmq_size=40))
    save("ont_no_inner", x, orc.default_rmq_param(max_dist_inner=0))
    save("ont_narrow", x, orc.default_rmq_param(bw=500, max_dist=2000, max_dist_inner=300))
    save("ont_gap_skip", x, orc.default_rmq_param(pen_gap=np.float32(0.3), pen_skip=np.float32(0.05)))
    save("two_chains", orc.ref_radix_sort(sc.sort_by_x(np.concatenate([sc.colinear(900, 7, r0=1_000_000, q0=100), sc.colinear(700, 8, r0=1_030_000, q0=22_000)]))),
         orc.default_rmq_param())
    save("grid_tied", orc.ref_radix_sort(sc.grid_ties(nx=24, ny=9, step=11)), orc.default_rmq_param())
    save("tiny", orc.ref_radix_sort(sc.colinear(5, 9)), orc.default_rmq_param(min_cnt=2, min_sc=10))
    rng = np.random.default_rng(17)
    dense = sc.sort_by_x(sc.pack(np.full(700, 1), np.zeros(700, np.int64), 1000 + rng.integers(0, 120, 700), 100 + rng.integers(0, 120, 700)))
    save("dense_tied", orc.ref_radix_sort(dense), orc.default_rmq_param())
    several = np.concatenate([sc.colinear(300, 21, rid=1, r0=50_000, q0=100), sc.colinear(400, 22, rid=1, r0=900_000, q0=9_000),
                              sc.colinear(250, 23, rid=2, r0=10_000, q0=30_000), sc.colinear(200, 24, rid=2, rev=1, r0=10_000, q0=40_000)])
    save("several_chains", orc.ref_radix_sort(sc.sort_by_x(several)), orc.default_rmq_param())


if __name__ == "__main__":
    main()
