#!/usr/bin/env python3
"""Counters of the device post-pass (gpurun_out/pmc_post_<tag> from collect_pmc_post.sh) -> profiles/<tag>_post_counters.json and
profiles/post_traffic_latest.json (read by bench.py for roofline_post.traffic).   python profiles/summarize_post.py r04"""
import collections, csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out", f"pmc_post_{tag}")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(G + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_post" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(G + "/*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_post" in k:
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, d in acc.items():
    out[k] = {c: max(v) for c, v in d.items()}          # the largest launch = the bench batch (the extras are smaller)
    if dur[k]:
        out[k]["ms_max_launch_under_counters"] = max(dur[k]) / 1e6
bench_log = None
for f in glob.glob(G + "/*.log"):
    for ln in open(f, errors="replace"):
        if ln.startswith("{"):
            bench_log = json.loads(ln)
chains = next((v for k, v in out.items() if "k_post_chains" in k), None)
if not chains or "FETCH_SIZE" not in chains:
    sys.exit("no k_post_chains counters under " + G)
anchors = bench_log["config"]["anchors_per_gpu"] if bench_log else None
k_ms = None
try:
    for r in csv.DictReader(open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))):
        if "k_post_chains" in r["Name"]:
            k_ms = float(r["AverageNs"]) / 1e6
except Exception:
    pass
post_sha = hashlib.sha256(open(os.path.join(ROOT, "mm2-gb_amd", "csrc", "post_kernels.hip"), "rb").read()).hexdigest()[:16]
traffic = {"anchors": anchors, "hbm_bytes_per_launch": (2 * chains["FETCH_SIZE"] + chains["WRITE_SIZE"]) * 1024, "fetch_size_kb": chains["FETCH_SIZE"], "write_size_kb": chains["WRITE_SIZE"],
           "l2_hit_rate": chains.get("TCC_HIT_sum", 0) / max(1.0, chains.get("TCC_HIT_sum", 0) + chains.get("TCC_MISS_sum", 0)),
           "post_sha16": post_sha, "k_post_chains_ms": k_ms,
           "source": f"profiles/{tag}_post_counters.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, profiles/collect_pmc_post.sh), k_post_chains, "
                     "(2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md"}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_post_counters.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(ROOT, "profiles", "post_traffic_latest.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
