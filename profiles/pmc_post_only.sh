#!/bin/bash
# rocprofv3 counters of the device post-pass kernels on profiles/post_only.py (one --pmc group per pass, never mixed with tracing domains):
# what the walk / sort kernels wait for -- vector-memory instructions, the address unit and the L1 (TA / TCP), L2 hits.
# usage (on the GPU box, from the repo root): bash profiles/pmc_post_only.sh <outdir> [groups...]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$REPO/gpurun_out/pmc_post_only}; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/profiles/post_only.py" --runs 1 > "$OUT/$name.log" 2>&1 || echo "pass $name: failed or timed out (rc $?)"
}
want=${*:-sq1 sq2 fetch write l2}   # (the TA_* counters took rocprofv3 down with signal 6 on this pool: not in the default set)
for g in $want; do case $g in
  sq1) run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM ;;
  sq2) run sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY ;;
  ta)  run ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE ;;
  tcp) run tcp TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE ;;
  lat) run lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum ;;
  l2)  run l2 TCC_HIT_sum TCC_MISS_sum ;;
  fetch) run fetch FETCH_SIZE GRBM_GUI_ACTIVE ;;
  write) run write WRITE_SIZE GRBM_GUI_ACTIVE ;;
esac; done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "k_post" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: max(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open(out + "/counters.json", "w"), indent=1)
for k in sorted(res):
    if any(x in k for x in ("k_post_walk", "k_post_sort", "k_post_emit", "k_post_chains", "k_post_lift", "k_post_partition", "k_post_classes")):
        print(k)
        for c, v in sorted(res[k].items()):
            print(f"   {c:36s} {v:.5g}")
PY
