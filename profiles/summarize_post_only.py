#!/usr/bin/env python3
"""Counters of the device post-pass collected by profiles/pmc_post_only.sh (one post-pass per rocprofv3 pass: post_only.py --runs 1) ->
profiles/<tag>_post_counters.json (per kernel, summed over its launches of the one post-pass) and profiles/post_traffic_latest.json (what
bench.py reads for roofline_post.traffic).   python3 profiles/summarize_post_only.py gpurun_out/pmc_r6 r06"""
import collections, csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_post" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k][r["Counter_Name"]] += 1
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "sq1", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_post" in k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
anchors = None
for f in glob.glob(os.path.join(src, "*.log")):
    for ln in open(f, errors="replace"):
        if ln.startswith("{\"anchors\""):
            anchors = json.loads(ln)["anchors"]
out = {}
tot = collections.defaultdict(float)
for k, d in acc.items():
    o = dict(d)
    o["launches"] = max(launches[k].values())
    o["ms_under_counters_summed"] = round(sum(dur[k]), 3)
    if "FETCH_SIZE" in d:
        o["hbm_bytes"] = (2 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0)) * 1024
    if "TCC_HIT_sum" in d:
        o["l2_hit_rate"] = round(d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 3)
    if "SQ_WAVE_CYCLES" in d:
        o["wait_any_share"] = round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 3)
    out[k] = o
    for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"):
        tot[c] += d.get(c, 0.0)
post_sha = hashlib.sha256(open(os.path.join(ROOT, "mm2-gb_amd", "csrc", "post_kernels.hip"), "rb").read()).hexdigest()[:16]
traffic = {"anchors": anchors, "hbm_bytes_per_launch": (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024, "fetch_size_kb": tot["FETCH_SIZE"], "write_size_kb": tot["WRITE_SIZE"],
           "l2_hit_rate": round(tot["TCC_HIT_sum"] / max(1.0, tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"]), 3),
           "wait_any_share_all_post_kernels": round(tot["SQ_WAIT_ANY"] / max(1.0, tot["SQ_WAVE_CYCLES"]), 3),
           "wait_any_share_by_kernel": {k.split("::")[-1]: v["wait_any_share"] for k, v in out.items() if "wait_any_share" in v and v.get("ms_under_counters_summed", 0) > 0.5},
           "hbm_gb_by_kernel": {k.split("::")[-1]: round(v["hbm_bytes"] / 1e9, 2) for k, v in out.items() if v.get("hbm_bytes", 0) > 1e8},
           "post_sha16": post_sha,
           "source": f"profiles/{tag}_post_counters.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum / SQ_* (separate passes, profiles/pmc_post_only.sh: one post-pass of "
                     "profiles/post_only.py at the bench's batch), every k_post_* kernel summed over its launches, (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md"}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_post_counters.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(ROOT, "profiles", "post_traffic_latest.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
