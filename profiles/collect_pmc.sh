#!/bin/bash
# Collect rocprofv3 counters for the score kernel (one --pmc group per pass; never mixed with tracing domains).
# usage (on the GPU box, from the repo root): bash profiles/collect_pmc.sh <anchors> <outdir>
set -u
ANCH=${1:-50000000}
OUT=${2:-$GRAFT_REPO_ROOT/gpurun_out/pmc}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/bench.py" --anchors "$ANCH" --steps 1 --warmup 1 --cpu-seconds 0 --no-pcie --no-e2e --no-bins --no-post --no-config2 > "$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
run sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE GRBM_GUI_ACTIVE
run l2 TCC_HIT_sum TCC_MISS_sum
find "$OUT" -name "*counter_collection.csv" | head
