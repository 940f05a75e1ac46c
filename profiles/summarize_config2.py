#!/usr/bin/env python3
"""gpurun_out/config2_<tag>/{trace,sq1,sq2} (profiles/profile_config2.sh) -> profiles/<tag>_config2_counters.json + config2_counters_latest.json,
which bench.py's config2_500M extra reads (matched by kernel-source hash and batch size)."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as bench_mod  # noqa: E402
tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out", f"config2_{tag}")
KERNEL = "k_score<0, false, false>"


def one(pattern):
    hits = glob.glob(os.path.join(G, pattern), recursive=True)
    if not hits:
        sys.exit(f"missing {pattern}")
    return max(hits, key=lambda f: sum(1 for _ in open(f)))


def counters(path, first=4):
    acc, n, seen = collections.defaultdict(float), collections.defaultdict(int), collections.defaultdict(int)
    for r in sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"])):
        if KERNEL not in r["Kernel_Name"]:
            continue
        seen[r["Counter_Name"]] += 1
        if seen[r["Counter_Name"]] > first:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    if not acc:
        sys.exit(f"no {KERNEL} row in {path}")
    return {k: acc[k] / n[k] for k in acc}


dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sorted(csv.DictReader(open(one("trace/**/*kernel_trace.csv"))), key=lambda r: int(r["Dispatch_Id"]))
       if KERNEL in r["Kernel_Name"]][:4]
kernel_ms = sum(dur) / len(dur) / 1e6
sq = {}
sq.update(counters(one("sq1/**/*counter_collection.csv")))
sq.update(counters(one("sq2/**/*counter_collection.csv")))
line = json.loads(next(l for l in reversed(open(os.path.join(G, "trace.log")).read().splitlines()) if l.startswith("{")))
anchors, pairs = line["config"]["anchors_per_gpu"], line["config"]["pairs_per_gpu"]
cycles = kernel_ms * 1e-3 * 2.4e9
out = {"workload": "configs[2]: synthetic 10-100 kb reads", "anchors": anchors, "pairs": pairs, "profiled_kernel_ms": round(kernel_ms, 3),
       "valu_insts_per_launch": sq["SQ_INSTS_VALU"], "salu_insts_per_launch": sq["SQ_INSTS_SALU"], "lds_insts_per_launch": sq["SQ_INSTS_LDS"],
       "valu_insts_per_64_pairs": sq["SQ_INSTS_VALU"] * 64 / pairs, "salu_insts_per_64_pairs": sq["SQ_INSTS_SALU"] * 64 / pairs,
       "valu_busy_fraction": round(sq["SQ_ACTIVE_INST_VALU"] / (cycles * 256), 3), "salu_busy_fraction": round(sq["SQ_ACTIVE_INST_SCA"] / (cycles * 256), 3),
       "lds_busy_fraction": round(sq["SQ_ACTIVE_INST_LDS"] / (cycles * 256), 3),
       "pairs_per_s_kernel": pairs / (kernel_ms * 1e-3), "kernel_sha16": bench_mod.kernel_sha16(),
       "source": f"profiles/{tag}_config2_counters.json: rocprofv3 --kernel-trace and two --pmc passes of `bench.py --len-lo 10000 --len-hi 100000` (profiles/profile_config2.sh), "
                 f"{KERNEL}, mean of the first 4 launches"}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_config2_counters.json"), "w"), indent=1)
json.dump(out, open(os.path.join(ROOT, "profiles", "config2_counters_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
