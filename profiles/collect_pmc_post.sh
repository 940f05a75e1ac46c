#!/bin/bash
# rocprofv3 counters of the device post-pass kernels (k_post_chains first of all); one --pmc group per pass, never mixed with tracing domains.
# usage (on the GPU box, from the repo root): bash profiles/collect_pmc_post.sh <outdir>
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$REPO/gpurun_out/pmc_post}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --cpu-seconds 0 --no-pcie --no-e2e --no-bins --no-config2 > "$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
run sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE GRBM_GUI_ACTIVE
run l2 TCC_HIT_sum TCC_MISS_sum
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if not k.startswith("mm2gb::k_post") and "k_post" not in k:
            continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} {v:.4g}")
PY
