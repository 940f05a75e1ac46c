#!/usr/bin/env python3
"""End-to-end wall time of the reference minimap2 host on simulated long reads: CPU chaining (lchain.c, max-chain-skip
= INT32_MAX and the default 25) vs --gpu-chain on top of libmm2gb_chain.so; PAF compared.  Single host thread (-t 1), as the
reference documents.  Writes one JSON document.

    python profiles/e2e_sim.py --reads 400 --out profiles/r01_e2e.json
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sim_reads  # noqa: E402


def run(cmd):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        raise SystemExit(r.stderr.decode()[-2000:])
    return r.stdout.decode(), dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=400)
    ap.add_argument("--len-lo", type=int, default=5_000)
    ap.add_argument("--len-hi", type=int, default=80_000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "e2e.json"))
    ap.add_argument("--threads", type=int, default=8, help="second comparison with this many host threads (the reference's GPU path is -t 1 only)")
    args = ap.parse_args()
    cpu = os.path.join(ROOT, "oracle", "_ref", "minimap2_cpu")
    gpu = os.path.join(ROOT, "oracle", "_ref", "minimap2_gpuhost")
    cfg = os.path.join(ROOT, "mm2-gb_amd", "mi355x_config.json")
    with tempfile.TemporaryDirectory() as td:
        ref, reads = os.path.join(td, "ref.fa"), os.path.join(td, "reads.fa")
        bases = sim_reads.simulate(ref, reads, seed=11, n_reads=args.reads, len_lo=args.len_lo, len_hi=args.len_hi)
        paf_inf, t_inf = run([cpu, "-t", "1", "--max-chain-skip=2147483647", ref, reads])
        paf_25, t_25 = run([cpu, "-t", "1", ref, reads])
        paf_gpu, t_gpu = run([gpu, "-t", "1", "--gpu-chain", "--gpu-cfg", cfg, ref, reads])
        _, t_gpu2 = run([gpu, "-t", "1", "--gpu-chain", "--gpu-cfg", cfg, ref, reads])
        # several host threads: one stream id (engine) per thread
        mt = {}
        if args.threads > 1:
            cfg_n = os.path.join(td, "cfg_n.json")
            doc_cfg = json.load(open(cfg)); doc_cfg["num_streams"] = args.threads
            json.dump(doc_cfg, open(cfg_n, "w"))
            T = str(args.threads)
            paf_inf_n, t_inf_n = run([cpu, "-t", T, "--max-chain-skip=2147483647", ref, reads])
            paf_gpu_n, t_gpu_n = run([gpu, "-t", T, "--gpu-chain", "--gpu-cfg", cfg_n, ref, reads])
            _, t_gpu_n2 = run([gpu, "-t", T, "--gpu-chain", "--gpu-cfg", cfg_n, ref, reads])
            mt = {"threads": args.threads, "cpu_skip_inf_s": round(t_inf_n, 2), "gpu_chain_s": round(min(t_gpu_n, t_gpu_n2), 2),
                  "paf_gpu_equals_cpu_skip_inf": sorted(paf_gpu_n.splitlines()) == sorted(paf_inf.splitlines()),
                  "paf_cpu_threads_equals_one_thread": sorted(paf_inf_n.splitlines()) == sorted(paf_inf.splitlines())}
    doc = {"reads": args.reads, "bases": bases, "read_len": [args.len_lo, args.len_hi],
           "cpu_skip_inf_s": round(t_inf, 2), "cpu_skip_25_s": round(t_25, 2), "gpu_chain_s": round(min(t_gpu, t_gpu2), 2),
           "gbp_per_s": {"cpu_skip_inf": bases / t_inf / 1e9, "cpu_skip_25": bases / t_25 / 1e9, "gpu_chain": bases / min(t_gpu, t_gpu2) / 1e9},
           "paf_lines": paf_inf.count("\n"), "paf_gpu_equals_cpu_skip_inf": paf_gpu == paf_inf,
           "paf_lines_differing_skip25_vs_inf": sum(a != b for a, b in zip(paf_25.splitlines(), paf_inf.splitlines())),
           "several_threads": mt,
           "note": "whole program wall time incl. index build and process start-up (GPU run: incl. HIP init); -t 1 unless stated"}
    json.dump(doc, open(args.out, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
