#!/usr/bin/env python3
"""mm2gb_map_reads_stream on ~0.26 Gbp of simulated reads (bench.py's e2e mix) per number of engines on the GPU and chunk size.
usage: python profiles/experiments/e2e_knobs.py [copies [ENGINESxCHUNK_MBP,...]]"""
import os, sys, time, json, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "profiles"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import bench, mm2gb_amd as mm, sim_reads
from test_seeding_cpu import read_fasta

threads = max(1, min(32, bench.cpu_quota() or 16))
with tempfile.TemporaryDirectory() as td:
    ref, ra, rb = os.path.join(td, "ref.fa"), os.path.join(td, "a.fa"), os.path.join(td, "b.fa")
    target = 260e6
    sim_reads.simulate(ref, ra, seed=21, n_reads=max(8, int(target / 2 / 55_000)), len_lo=10_000, len_hi=100_000)
    sim_reads.simulate(ref, rb, seed=21, n_reads=max(4, int(target / 2 / 200_000)), len_lo=100_000, len_hi=300_000)
    refs = read_fasta(ref)
    reads = [("s" + n, s) for n, s in read_fasta(ra)] + [("l" + n, s) for n, s in read_fasta(rb)]
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reads = [(f"{n}_c{c}", s) for c in range(copies) for n, s in reads]
bases = sum(len(s) for _, s in reads)
ix = mm.SeedIndex([s for _, s in refs], threads=threads)
names = [n for n, _ in refs]
engs = [mm.Engine(device=0) for _ in range(6)]
opt = mm.map_opt(host_threads=threads)
mm.map_reads_stream(engs, ix, names, reads[:48], opt=opt, chunk_bases=500_000)
first = None
grid = ((3, 48_000_000), (2, 48_000_000), (4, 48_000_000), (6, 48_000_000), (3, 24_000_000), (4, 24_000_000), (6, 16_000_000), (3, 96_000_000)) if copies == 1 else \
       ((3, 48_000_000), (4, 24_000_000), (4, 32_000_000), (5, 24_000_000), (3, 48_000_000), (4, 24_000_000))
if len(sys.argv) > 2:                                        # e.g. "4x32,4x32": engines x chunk Mbp, in this order
    grid = tuple((int(a), int(b) * 1_000_000) for a, b in (g.split("x") for g in sys.argv[2].split(",")))
for n_eng, chunk in grid:
    t0 = time.perf_counter()
    paf, st = mm.map_reads_stream(engs[:n_eng], ix, names, reads, opt=opt, chunk_bases=chunk)
    dt = time.perf_counter() - t0
    if first is None:
        first = paf
    print(json.dumps({"engines": n_eng, "chunk_bases": chunk, "seconds": round(dt, 2), "gbp_per_s": round(bases / dt / 1e9, 4), "same_paf": paf == first,
                      "stages": {k: round(v, 2) for k, v in st.items() if k.startswith("s_")}}), flush=True)
