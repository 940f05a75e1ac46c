#!/usr/bin/env python3
"""Turn one round's rocprofv3 output (gpurun_out/<tag>_stats, _fetch, _write from profile_round.sh and gpurun_out/pmc_<tag> from
collect_pmc.sh) into the small files kept under profiles/:  <tag>_kernel_stats.csv, <tag>_fetch_pmc.csv, <tag>_write_pmc.csv,
<tag>_sq_counters.json and traffic_latest.json (read by bench.py for roofline.traffic).
    python profiles/summarize_round.py r01b"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as bench_mod  # noqa: E402  (kernel_sha16: which kernel build these counters belong to)
tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
KERNEL = "k_score<0, false, false>"       # MODE_LUT, plain build (not SPLIT, no gang phase)


def one(pattern):
    """The profiled run may have child processes of its own (bench.py's end-to-end extra starts the reference host): every process
    leaves a file.  The bench process is the one with the most rows."""
    hits = glob.glob(os.path.join(G, pattern), recursive=True)
    if not hits:
        sys.exit(f"missing {pattern}")
    return max(hits, key=lambda f: sum(1 for _ in open(f)))


def counters(path, kernel=KERNEL, first=2):
    """mean per launch of every counter of one kernel over its first `first` dispatches = the warm-up and the timed steps on the
    bench workload (later dispatches of the same kernel belong to bench.py's extras: other batch sizes)"""
    acc, n = collections.defaultdict(float), collections.defaultdict(int)
    rows = []
    seen = collections.defaultdict(int)
    for r in sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"])):
        is_k, is_w = kernel in r["Kernel_Name"], "k_window" in r["Kernel_Name"]
        if not (is_k or is_w):
            continue
        tag = (r["Kernel_Name"], r["Counter_Name"])
        seen[tag] += 1
        if seen[tag] > first:
            continue
        rows.append(r)
        if is_k:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    return {k: acc[k] / n[k] for k in acc}, rows


# per-kernel time: the first 4 dispatches (1 warm-up + 3 steps) of each kernel in the trace of the bench process
trace = one(f"{tag}_stats/**/*kernel_trace.csv")
dur = collections.defaultdict(list)
for r in sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Dispatch_Id"])):
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
with open(os.path.join(P, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls_counted", "AverageNs", "MinNs", "MaxNs", "Calls_in_trace", "note"])
    for name, v in sorted(dur.items(), key=lambda kv: -sum(kv[1][:4])):
        head = v[:4]
        w.writerow([name, len(head), sum(head) / len(head), min(head), max(head), len(v), "first 4 dispatches = bench workload (rocprofv3 --kernel-trace of `python bench.py --steps 3 --warmup 1`)"])
score = next(v for k, v in dur.items() if KERNEL in k)
kernel_ms = sum(score[:4]) / len(score[:4]) / 1e6
out = {}
for kind in ("fetch", "write"):
    vals, rows = counters(one(f"{tag}_{kind}/**/*counter_collection.csv"))
    out[kind] = vals
    with open(os.path.join(P, f"{tag}_{kind}_pmc.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    # calibration: k_window reads 16 B per anchor (+ look-back samples and probes) and writes 16 B per anchor
    split = [r for r in rows if "k_window" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE")]
    out[kind + "_split_kb"] = sum(float(r["Counter_Value"]) for r in split) / max(1, len(split))
sq = {}
for grp in ("sq1", "sq2", "lds", "l2"):
    vals, _ = counters(one(f"pmc_{tag}/{grp}/**/*counter_collection.csv"))
    sq.update(vals)
bench = json.loads(next(l for l in reversed(open(os.path.join(G, f"{tag}_stats.log")).read().splitlines()) if l.startswith("{")))
anchors, pairs = bench["config"]["anchors_per_gpu"], bench["config"]["pairs_per_gpu"]
cycles = kernel_ms * 1e-3 * 2.4e9
sq["derived"] = {
    "kernel_ms": kernel_ms, "kernel_cycles_at_2.4GHz": cycles,
    "valu_busy": sq["SQ_ACTIVE_INST_VALU"] / (cycles * 256),
    "salu_busy": sq["SQ_ACTIVE_INST_SCA"] / (cycles * 256),
    "valu_insts_per_source_step": sq["SQ_INSTS_VALU"] * 64 / pairs,
    "salu_insts_per_source_step": sq["SQ_INSTS_SALU"] * 64 / pairs,
    "lds_insts_per_source_step": sq["SQ_INSTS_LDS"] * 64 / pairs,
    "lds_idx_active_fraction": sq["SQ_LDS_IDX_ACTIVE"] / (cycles * 256),
    "lds_bank_conflict_fraction_of_lds_cycles": sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"],
    "l2_hit_rate": sq["TCC_HIT_sum"] / (sq["TCC_HIT_sum"] + sq["TCC_MISS_sum"]),
}
sq["note"] = (f"per launch of {KERNEL} (MODE_LUT) on the default bench workload ({anchors} anchors, {pairs} pairs); rocprofv3 --pmc, one group per pass "
              "(profiles/collect_pmc.sh); SQ_ACTIVE_* are summed over the 4 SIMDs of a CU in units of 4 cycles, hence busy = value / (cycles * 256 CUs)")
json.dump(sq, open(os.path.join(P, f"{tag}_sq_counters.json"), "w"), indent=1)
fetch_kb, write_kb = out["fetch"]["FETCH_SIZE"], out["write"]["WRITE_SIZE"]
traffic = {
    "anchors": anchors, "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024, "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
    "k_window_fetch_kb": out["fetch_split_kb"], "k_window_write_kb": out["write_split_kb"],
    "source": f"profiles/{tag}_fetch_pmc.csv + {tag}_write_pmc.csv: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), {KERNEL} = MODE_LUT, "
              "(2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (the x2 was calibrated on k_split_soa, the pure 16 B-in / 16 B-out kernel of earlier builds, whose read showed as 8 B/anchor: profiles/earlier/r01c_fetch_pmc.csv)",
    "valu_busy_fraction": round(sq["derived"]["valu_busy"], 3),
    "valu_insts_per_launch": sq["SQ_INSTS_VALU"],
    "lds_idx_active_fraction": round(sq["derived"]["lds_idx_active_fraction"], 3),
    "lds_bank_conflict_fraction": round(sq["derived"]["lds_bank_conflict_fraction_of_lds_cycles"], 3),
    "profiled_kernel_ms": round(kernel_ms, 3),
    "kernel_sha16": bench_mod.kernel_sha16(),
    "valu_source": f"profiles/{tag}_sq_counters.json",
}
json.dump(traffic, open(os.path.join(P, "traffic_latest.json"), "w"), indent=1)
print(json.dumps({"kernel_ms": kernel_ms, **sq["derived"], "hbm_GB": traffic["hbm_bytes_per_launch"] / 1e9,
                  "k_window_fetch_B_per_anchor": out["fetch_split_kb"] * 1024 / anchors, "k_window_write_B_per_anchor": out["write_split_kb"] * 1024 / anchors}, indent=1))
