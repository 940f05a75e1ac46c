#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout 300 python -u -m pytest tests/test_gpu_post.py tests/test_gpu_rmq.py tests/test_gpu_formats.py tests/test_gpu_batcher.py -x -q > $O/r02q_test.log 2>&1; echo "tests rc=$?"; tail -3 $O/r02q_test.log
for w in 8192 3072 2048; do
MM2GB_POST_WAVES=$w MM2GB_DEBUG_PHASES=1 timeout 300 python bench.py --cpu-seconds 0 --no-pcie --no-bins --no-e2e --steps 1 > $O/r02q_bench_$w.json 2> $O/r02q_bench_$w.err
echo "waves=$w"; grep "post-pass" $O/r02q_bench_$w.err | tail -1 | cut -c1-260
python - <<PY
import json
d = json.load(open("gpurun_out/r02q_bench_$w.json")); print(d["post_pass_device"]["ms"])
PY
done
