#!/usr/bin/env python3
"""tests/golden/seeds/*.npz from the REFERENCE (dev container only): what mm_collect_matches (seed.c:98) handed to
collect_seed_hits (map.c:295-331) and the sorted anchors that came out, observed through oracle/capture_hooks.c on the
reference's own test data and on simulated long reads, with the option bits that reach skip_seed (map.c:205-227) from
the command line.  Data only.     make -C oracle all && python oracle/gen_golden_seeds.py"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc            # noqa: E402
import sim_reads      # noqa: E402

REF = os.environ.get("MM2GB_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "seeds")
RUNS = {   # name: (target, query, extra arguments, option bits)
    "mt": ("MT-human.fa", "MT-orang.fa", [], 0),
    "mt_for": ("MT-human.fa", "MT-orang.fa", ["--for-only"], orc.MM_F_FOR_ONLY),
    "mt_rev": ("MT-human.fa", "MT-orang.fa", ["--rev-only"], orc.MM_F_REV_ONLY),
    "inv": ("t-inv.fa", "q-inv.fa", [], 0),
    "inv_rev": ("t-inv.fa", "q-inv.fa", ["--rev-only"], orc.MM_F_REV_ONLY),
}
# -X = MM_F_NO_DIAG | MM_F_NO_DUAL (main.c): the name tests of skip_seed.  (target, query, q_rank, ref_rank, ref_len of the one target)
XRUNS = {
    "mt_x_self": ("MT-human.fa", "MT-human.fa", 0, [0], [16569]),       # same name, same length: diagonal dropped, self flag
    "mt_x_larger": ("MT-human.fa", "MT-orang.fa", 1, [0], [16569]),     # "MT_orang" > "MT_human": every hit dropped
    "mt_x_smaller": ("MT-orang.fa", "MT-human.fa", 0, [1], [16499]),    # "MT_human" < "MT_orang": kept
}


def run(tgt, qry, extra, paf_to=None):
    exe = os.path.join(orc.REF_DIR, "minimap2_cpu")
    hook = os.path.join(orc.REF_DIR, "libcapture.so")
    with tempfile.TemporaryDirectory() as td:
        cap = os.path.join(td, "seeds.bin")
        env = dict(os.environ, LD_PRELOAD=hook, MM2GB_CAPTURE_SEEDS=cap)
        subprocess.run([exe, "-t", "1"] + extra + [tgt, qry], env=env, check=True, capture_output=True)
        recs = orc.read_seed_capture(cap) if os.path.exists(cap) else []
    if paf_to:   # what the reference prints for this run at max-chain-skip = infinity (the GPU path's contract): for tests/test_gpu_mapper.py
        r = subprocess.run([exe, "-t", "1", "--max-chain-skip=2147483647"] + extra + [tgt, qry], check=True, capture_output=True)
        open(paf_to, "wb").write(r.stdout)
    return recs


def save(name, k, r, flag, source, **names):
    meta = dict(flag=int(flag), qlen=int(r["qlen"]), rep_len=int(r["rep_len"]), source=source, record=k, **names)
    np.savez_compressed(os.path.join(OUT, f"{name}_{k}.npz"), seeds=r["seeds"], hits=r["hits"], hit_off=r["hit_off"], a=r["a"], mini_pos=r["mini_pos"],
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))


if __name__ == "__main__":
    if not orc.ref_available():
        sys.exit("reference build missing: run `make -C oracle all` in a container that has /root/reference")
    os.makedirs(OUT, exist_ok=True)
    for name, (tgt, qry, extra, flag) in RUNS.items():
        recs = run(os.path.join(REF, "test", tgt), os.path.join(REF, "test", qry), extra, paf_to=os.path.join(OUT, name + ".paf") if extra else None)
        for k, r in enumerate(recs):
            save(name, k, r, flag, f"{tgt} x {qry} {' '.join(extra)}".strip())
        print(f"{name}: {len(recs)} records, {sum(len(r['hits']) for r in recs)} hits, {sum(len(r['a']) for r in recs)} anchors")
    for name, (tgt, qry, q_rank, ref_rank, ref_len) in XRUNS.items():
        recs = run(os.path.join(REF, "test", tgt), os.path.join(REF, "test", qry), ["-X"])
        for k, r in enumerate(recs):
            save(name, k, r, orc.MM_F_NO_DIAG | orc.MM_F_NO_DUAL, f"{tgt} x {qry} -X", q_rank=q_rank, ref_rank=ref_rank, ref_len=ref_len)
        print(f"{name}: {len(recs)} records, {sum(len(r['hits']) for r in recs)} hits, {sum(len(r['a']) for r in recs)} anchors")
    # simulated long reads on a random genome with repeats and tandem arrays (tests/sim_reads.py): of 150 reads the first four and the
    # first six that cross an array (minimizers above mid_occ: dropped or thinned out, rep_len > 0); files are named by read number
    with tempfile.TemporaryDirectory() as td:
        ref_fa, reads_fa = os.path.join(td, "ref.fa"), os.path.join(td, "reads.fa")
        sim_reads.simulate(ref_fa, reads_fa, seed=5, n_reads=150, len_lo=3_000, len_hi=20_000)
        for name, extra, flag in (("sim", [], 0), ("sim_for", ["--for-only"], orc.MM_F_FOR_ONLY)):
            recs = run(ref_fa, reads_fa, extra)
            assert len(recs) == 150
            rep = [k for k, r in enumerate(recs) if r["rep_len"] > 0][:6]
            pick = sorted(set([0, 1, 2, 3] + rep)) if name == "sim" else rep[:2] + [0]
            for k in pick:
                save(name, k, recs[k], flag, "tests/sim_reads.py seed=5 n_reads=150 " + " ".join(extra))
            print(f"{name}: reads {pick}, {sum(len(recs[k]['hits']) for k in pick)} hits, rep_len {[recs[k]['rep_len'] for k in pick]}")
    # a small genome with repeat copies against itself, -X: off-diagonal same-strand hits of a sequence on itself carry MM_SEED_SELF,
    # and of two different sequences only the pair (smaller name -> larger name) is kept
    with tempfile.TemporaryDirectory() as td:
        rng = np.random.default_rng(9)
        chrs = sim_reads.make_genome(rng, n_chr=3, chr_len=20_000, n_rep_families=2, rep_len=300, copies=9, tandem=0)
        names = ["seqB", "seqA", "seqC"]                      # not in lexicographic order on purpose
        fa = os.path.join(td, "g.fa")
        sim_reads.write_fasta(fa, list(zip(names, chrs)))
        order = sorted(names)
        ref_rank = [order.index(nm) for nm in names]
        ref_len = [len(c) for c in chrs]
        recs = run(fa, fa, ["-X"])
        assert len(recs) == len(names)
        for k, r in enumerate(recs):                          # reads come in file order
            save("self_x", k, r, orc.MM_F_NO_DIAG | orc.MM_F_NO_DUAL, "3 x 20 kb with repeat copies, against itself, -X", q_rank=ref_rank[k], ref_rank=ref_rank, ref_len=ref_len)
        print(f"self_x: {len(recs)} records, {sum(len(r['hits']) for r in recs)} hits, {sum(len(r['a']) for r in recs)} anchors, "
              f"{sum(int(((r['a'][:, 1] >> 43) & 1).sum()) for r in recs)} self")
