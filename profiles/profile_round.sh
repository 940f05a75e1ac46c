#!/bin/bash
# One round's judged profile of the default bench (run on the GPU box from the repo root):
#   bash profiles/profile_round.sh r01
# 1. rocprofv3 --kernel-trace --stats of `python bench.py` (3 steps + 1 warmup)  -> gpurun_out/<tag>_stats/
# 2. FETCH_SIZE and WRITE_SIZE in separate --pmc passes (1 step + 1 warmup)       -> gpurun_out/<tag>_fetch/, _write/
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $REPO/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-pcie --no-e2e --no-bins --no-config2 > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -- python3 $REPO/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-pcie --no-e2e --no-bins --no-post --no-config2 > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -- python3 $REPO/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-pcie --no-e2e --no-bins --no-post --no-config2 > $OUT/${TAG}_write.log 2>&1
tail -1 $OUT/${TAG}_stats.log | cut -c1-400
