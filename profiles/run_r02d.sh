#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
for w in 1024 2048 4096 6144; do
  MM2GB_POST_WAVES=$w MM2GB_DEBUG_PHASES=1 timeout 600 python bench.py --cpu-seconds 0 --no-pcie --no-bins --no-e2e --steps 1 > $O/r02d_bench_$w.json 2> $O/r02d_bench_$w.err
  echo "waves=$w"; grep "post-pass" $O/r02d_bench_$w.err | tail -1
  python - <<PY
import json
d = json.load(open("gpurun_out/r02d_bench_$w.json"))
print(d["post_pass_device"]["ms"])
PY
done
