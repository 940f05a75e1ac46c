#!/bin/bash
# LDS and L2 counters of the score kernel on the default bench workload (separate --pmc passes, kernel-trace only).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_lds_l2}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --cpu-seconds 0 --no-pcie > "$OUT/$name.log" 2>&1; }
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAIT_INST_LDS
run l2 TCC_HIT_sum TCC_MISS_sum
