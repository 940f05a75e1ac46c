#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout 300 python -u -m pytest tests/test_gpu_post.py tests/test_gpu_formats.py tests/test_gpu_rmq.py -x -q 2>&1 | tail -2
MM2GB_DEBUG_PHASES=1 timeout 300 python bench.py --cpu-seconds 0 --no-pcie --no-bins --no-e2e --steps 1 > $O/r02q_bench.json 2> $O/r02q_bench.err
grep "post-pass" $O/r02q_bench.err | tail -5 | cut -c1-300
python - <<PY
import json
d = json.load(open("gpurun_out/r02q_bench.json")); print(d["post_pass_device"]["ms"])
PY
