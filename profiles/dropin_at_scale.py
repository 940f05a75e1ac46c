"""The drop-in at scale, alone: the reference's host program (oracle/_ref/minimap2_gpuhost[_rmq], untouched sources) on top of the library over
bench.py's at-scale read set, with the library's own account of the run (MM2GB_REPORT) and, on request, the WHOLE PAF compared with what
`minimap2_cpu --max-chain-skip=2147483647` prints for the same file.

    python3 profiles/dropin_at_scale.py [--bases 1e9] [--legs gpuhost_rmq] [--full-cpu-paf] [--env K=V ...] [--debug] [--out FILE]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bases", type=float, default=1.0e9)
    ap.add_argument("--legs", default="gpuhost,gpuhost_rmq")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--full-cpu-paf", action="store_true", help="also run minimap2_cpu over the whole set (minutes) and compare the whole PAF")
    ap.add_argument("--env", action="append", default=[], help="K=V for the host's environment")
    ap.add_argument("--debug", action="store_true", help="MM2GB_DEBUG_PHASES=1 and the library's lines on stderr")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--own-host", action="store_true", help="also the repository's own host (mm2gb_map_reads_stream, four engines) on the same reads; with --full-cpu-paf its whole PAF is compared too")
    ap.add_argument("--cfg", action="append", default=[], help="K=V: top-level key of the gpu config to override (e.g. max_total_n=1000000)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    threads = args.threads or max(1, min(32, bench.cpu_quota() or 16))
    extra = dict(kv.split("=", 1) for kv in args.env)
    cfg_over = {k: json.loads(v) for k, v in (kv.split("=", 1) for kv in args.cfg)}
    if args.debug:
        extra["MM2GB_DEBUG_PHASES"] = "1"
    out = {}
    with tempfile.TemporaryDirectory() as td:
        t0 = time.perf_counter()
        ref, refs, uniq, reads, unique_bases, copies, bases = bench.scale_read_set(td, args.bases)
        out["simulate_seconds"] = round(time.perf_counter() - t0, 1)
        want = None
        if args.full_cpu_paf:
            allfa = os.path.join(td, "all_cpu.fa")
            with open(allfa, "wb") as fh:
                for n, sq in reads:
                    fh.write(b">" + n.encode() + b"\n" + bytes(sq) + b"\n")
            cpu = os.path.join(ROOT, "oracle", "_ref", "minimap2_cpu")
            t0 = time.perf_counter()
            r = subprocess.run([cpu, "-t", str(threads), "--max-chain-skip=2147483647", ref, allfa], capture_output=True, timeout=3000)
            assert r.returncode == 0, r.stderr.decode()[-500:]
            want = r.stdout
            out["reference_cpu_whole_set"] = {"seconds": round(time.perf_counter() - t0, 1), "threads": threads, "paf_lines": want.count(b"\n"), "gbp_per_s": bases / (time.perf_counter() - t0) / 1e9}
            os.unlink(allfa)
        if args.own_host:
            import mm2gb_amd as mm
            t0 = time.perf_counter()
            ix = mm.SeedIndex([sq for _, sq in refs], threads=threads)
            t_index = time.perf_counter() - t0
            engines = [mm.Engine(device=0) for _ in range(4)]
            opt = mm.map_opt(host_threads=threads)
            names = [n for n, _ in refs]
            mm.map_reads_stream(engines, ix, names, reads[:24], opt=opt, chunk_bases=500_000)
            t0 = time.perf_counter()
            paf, st = mm.map_reads_stream(engines, ix, names, reads, opt=opt, chunk_bases=bench.OWN_HOST_CHUNK_BASES)
            dt = time.perf_counter() - t0
            own = {"map_seconds": round(dt, 2), "index_seconds": round(t_index, 3), "gbp_per_s": bases / (dt + t_index) / 1e9, "paf_lines": paf.count("\n"), "engines": 4}
            if want is not None:
                own["whole_paf_identical_to_reference_cpu"] = paf.encode() == want
                if paf.encode() != want:
                    own["whole_paf_identical_as_sorted_lines"] = sorted(paf.encode().splitlines()) == sorted(want.splitlines())
            out["own_host"] = own
            for e in engines:
                e.close()
            ix.close()
        for k in range(args.repeat):
            res = bench.reference_host_at_scale(td, ref, reads, bases, uniq, None, threads, legs=tuple(args.legs.split(",")), extra_env=extra, keep_stderr=True, cfg_override=cfg_over)
            for key in args.legs.split(","):
                leg = res.get(key, {})
                err = leg.pop("_stderr", "")
                if args.debug or any(e.startswith("MM2GB_RMQ_CALLS=") for e in args.env):
                    sys.stderr.write(f"---- {key} ----\n" + "\n".join(l for l in err.splitlines() if "mm2gb" in l or "M::" in l) + "\n")
                if want is not None and "seconds_whole_program" in leg:
                    got = open(os.path.join(td, key + ".paf"), "rb").read()
                    # the host prints a mini-batch's reads in input order whatever the thread count: the files must be equal as they are
                    leg["whole_paf_identical_to_reference_cpu"] = got == want
                    if got != want:
                        leg["whole_paf_identical_as_sorted_lines"] = sorted(got.splitlines()) == sorted(want.splitlines())
            out[f"run{k}"] = res
    text = json.dumps(out, indent=1)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
