#!/usr/bin/env python3
"""End-to-end rate of the from-scratch mapper (csrc/seeding.cpp + csrc/mapper.cpp + device path): simulated long reads on a 3 Mbp genome,
index build and mapping timed separately, PAF compared with the reference host's when it is there.   python profiles/mapper_rate.py [n_reads [seed [len_lo len_hi]]]"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mm2gb_amd as mm, sim_reads
from test_seeding_cpu import read_fasta
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
len_lo = int(sys.argv[3]) if len(sys.argv) > 3 else 5_000
len_hi = int(sys.argv[4]) if len(sys.argv) > 4 else 80_000
out = {}
with tempfile.TemporaryDirectory() as td:
    ref, reads = os.path.join(td, "ref.fa"), os.path.join(td, "reads.fa")
    bases = sim_reads.simulate(ref, reads, seed=seed, n_reads=n_reads, len_lo=len_lo, len_hi=len_hi)
    refs, rd = read_fasta(ref), read_fasta(reads)
    bases = sum(len(s) for _, s in rd)
    with mm.Engine() as e:
        t0 = time.perf_counter()
        ix = mm.SeedIndex([s for _, s in refs], threads=16)
        t_index = time.perf_counter() - t0
        opt = mm.map_opt(host_threads=16, rechain_on_device=int(os.environ.get("MAPPER_RATE_RECHAIN_DEVICE", "0")))
        mm.map_reads(e, ix, [n for n, _ in refs], rd[:8], opt=opt)              # warm-up: arenas, first kernel launches
        best = 1e9
        for rep in range(int(os.environ.get("MAPPER_RATE_REPS", "3"))):
            t0 = time.perf_counter()
            paf, st = mm.map_reads(e, ix, [n for n, _ in refs], rd, opt=opt)
            best = min(best, time.perf_counter() - t0)
        ix.close()
    out = {"reads": n_reads, "seed": seed, "read_len": [len_lo, len_hi], "bases": bases, "index_seconds": round(t_index, 3), "map_seconds": round(best, 4), "gbp_per_s_mapping": bases / best / 1e9,
           "gbp_per_s_with_index": bases / (best + t_index) / 1e9, "paf_lines": paf.count("\n"), "stats": st, "host_threads": 16, "rechain_on_device": int(os.environ.get("MAPPER_RATE_RECHAIN_DEVICE", "0"))}
    host = os.path.join(ROOT, "oracle", "_ref", "minimap2_cpu")
    if os.path.exists(host) and not os.environ.get("MAPPER_RATE_NO_REF"):
        t0 = time.perf_counter()
        r = subprocess.run([host, "-t", "16", "--max-chain-skip=2147483647", ref, reads], capture_output=True)
        out["reference_cpu_seconds_t16"] = round(time.perf_counter() - t0, 3)
        want = r.stdout.decode()
        g = {}; w = {}
        for text, d in ((paf, g), (want, w)):
            for ln in text.splitlines():
                d.setdefault(ln.split("\t", 1)[0], []).append(ln)
        out["reads_with_different_paf"] = sum(1 for k in set(g) | set(w) if g.get(k) != w.get(k))
print(json.dumps(out))
