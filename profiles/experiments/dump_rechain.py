#!/usr/bin/env python3
"""The re-chaining call's input (offsets + anchors sorted by x) of a small sample of bench.py's e2e read mix, for working on csrc/rmq_host.cpp off
the GPU box: MM2GB_DUMP_RECHAIN (csrc/mapper.cpp) writes int64 n_reads, int64 offsets[n_reads + 1], anchors.   python profiles/experiments/dump_rechain.py [out]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "profiles"))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "rechain_dump.bin")
os.environ["MM2GB_DUMP_RECHAIN"] = out
import bench, mm2gb_amd as mm, sim_reads
from test_seeding_cpu import read_fasta
threads = max(1, min(32, bench.cpu_quota() or 16))
with tempfile.TemporaryDirectory() as td:
    ref, ra, rb = os.path.join(td, "ref.fa"), os.path.join(td, "a.fa"), os.path.join(td, "b.fa")
    sim_reads.simulate(ref, ra, seed=21, n_reads=40, len_lo=10_000, len_hi=100_000)
    sim_reads.simulate(ref, rb, seed=21, n_reads=12, len_lo=100_000, len_hi=300_000)
    refs = read_fasta(ref)
    reads = [("s" + n, s) for n, s in read_fasta(ra)] + [("l" + n, s) for n, s in read_fasta(rb)]
ix = mm.SeedIndex([s for _, s in refs], threads=threads)
with mm.Engine(device=0) as e:
    paf, st = mm.map_reads_stream([e], ix, [n for n, _ in refs], reads, opt=mm.map_opt(host_threads=threads), chunk_bases=10_000_000_000)
print(st, os.path.getsize(out))
