#!/bin/bash
# round 2, second GPU pass: the device post-pass (N2) -- tests, then its time at bench size
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_post.py -x -q --durations=10 > $O/r02b_post_test.log 2>&1; echo "post test rc=$?"
tail -15 $O/r02b_post_test.log
timeout 600 python bench.py --cpu-seconds 0 --no-pcie --no-bins --no-e2e > $O/r02b_bench.json 2> $O/r02b_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02b_bench.json"))
print(json.dumps(d.get("post_pass_device")), d["stage_ms"])
PY
tail -3 $O/r02b_bench.err
