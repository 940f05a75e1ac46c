#!/bin/bash
# dual compute streams + XCD-aware k_window: parity, then the host-buffer rates with one and two compute streams
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > $O/r02e_test.log 2>&1; echo "tests rc=$?"; tail -3 $O/r02e_test.log
for one in 1 0; do
  echo "== MM2GB_ONE_COMPUTE_STREAM=$one"
  MM2GB_ONE_COMPUTE_STREAM=$one timeout 600 python profiles/pcie_slices.py 400000000 2>&1 | grep slice
  MM2GB_ONE_COMPUTE_STREAM=$one timeout 600 python profiles/stream_api_rate.py --out $O/r02e_stream_rate_$one.json 2>&1 | tail -4
done
timeout 600 python bench.py --cpu-seconds 0 --no-bins --no-e2e --no-post > $O/r02e_bench.json 2> $O/r02e_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02e_bench.json"))
print(d["stage_ms"], d["value"], json.dumps(d["host_path"]["scores_only"])[:200])
PY
