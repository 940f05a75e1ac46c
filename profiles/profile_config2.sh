#!/bin/bash
# Counters of k_score on BASELINE configs[2] at its stated size (10-100 kb reads, 500 M anchors), on the GPU box from the repo root:
#   bash profiles/profile_config2.sh r05
# kernel trace, then two --pmc passes (instruction counts; busy cycles), each its own run (never mixed with tracing domains);
# profiles/summarize_config2.py turns them into profiles/<tag>_config2_counters.json and profiles/config2_counters_latest.json.
set -u
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/config2_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--len-lo 10000 --len-hi 100000 --steps 3 --warmup 1 --cpu-seconds 0 --no-pcie --no-e2e --no-bins --no-post --no-config2"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" $ARGS > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/sq1" -- python3 "$REPO/bench.py" $ARGS > "$OUT/sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/sq2" -- python3 "$REPO/bench.py" $ARGS > "$OUT/sq2.log" 2>&1
cd "$REPO" && python3 profiles/summarize_config2.py "$TAG"
