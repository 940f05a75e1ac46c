#!/bin/bash
# free sweep (blocks swept without range test, table at the end of LDS): parity first, then A/B on the 500 M-anchor bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream_api.py tests/test_gpu_pool.py -m gpu -x -q --durations=5 > $O/r02w_parity.log 2>&1; echo "parity rc=$?"; tail -6 $O/r02w_parity.log
run() { # name env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-pcie --no-e2e --no-post --no-bins --cpu-seconds 0 2>/dev/null | tail -1 > $O/r02w_ab_$name.json
  python -c "
import json; d=json.load(open('$O/r02w_ab_$name.json')); print('$name', round(d['value']/1e12,3), d['ms_per_step'], d['roofline']['kernel_ms'], d.get('plan'))"
}
for rep in 1 2; do
  run old MM2GB_LIB_PATH=$PWD/mm2-gb_amd/variants/libold.so


  run new_free0 MM2GB_FREE_SWEEP=0
  run new_free1 MM2GB_FREE_SWEEP=1
done
