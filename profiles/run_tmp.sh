#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout 300 python -u -m pytest tests/test_gpu_post.py tests/test_gpu_rmq.py tests/test_gpu_batcher.py -x -q > $O/r02n_test.log 2>&1; echo "tests rc=$?"; tail -4 $O/r02n_test.log
MM2GB_DEBUG_PHASES=1 timeout 300 python bench.py --cpu-seconds 0 --no-pcie --no-bins --no-e2e --steps 1 > $O/r02n_bench.json 2> $O/r02n_bench.err; echo "bench rc=$?"
grep "post-pass" $O/r02n_bench.err | tail -1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02n_bench.json")); print(json.dumps(d.get("post_pass_device"))[:160])
PY
