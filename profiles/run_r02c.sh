#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_batcher.py tests/test_gpu_post.py -x -q > $O/r02c_test.log 2>&1; echo "tests rc=$?"; tail -5 $O/r02c_test.log
MM2GB_DEBUG_PHASES=1 timeout 600 python bench.py --cpu-seconds 0 --no-pcie --no-bins --no-e2e --steps 1 > $O/r02c_bench.json 2> $O/r02c_bench.err; echo "bench rc=$?"
grep "post-pass" $O/r02c_bench.err | tail -2
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02c_bench.json"))
print(json.dumps(d.get("post_pass_device"))[:200])
PY
