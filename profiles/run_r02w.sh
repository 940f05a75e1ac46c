#!/bin/bash
# free sweep (blocks swept without range test, table at the end of LDS): parity first, then A/B on the 500 M-anchor bench and the bins
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream_api.py tests/test_gpu_pool.py -m gpu -x -q --durations=5 > $O/r02w_parity.log 2>&1; echo "parity rc=$?"; tail -12 $O/r02w_parity.log
for fs in 0 1 0 1; do
  MM2GB_FREE_SWEEP=$fs timeout 300 python bench.py --steps 3 --warmup 1 --no-pcie --no-e2e --no-post --no-bins --cpu-seconds 0 2>/dev/null | tail -1 > $O/r02w_ab_$fs.json
  python -c "
import json; d=json.load(open('$O/r02w_ab_$fs.json')); print('free_sweep=$fs', round(d['value']/1e12,3), d['ms_per_step'], d['roofline']['kernel_ms'])"
done
for fs in 0 1; do
  MM2GB_FREE_SWEEP=$fs timeout 300 python bench.py --steps 2 --warmup 1 --no-pcie --no-e2e --no-post --cpu-seconds 0 2>/dev/null | tail -1 > $O/r02w_bins_$fs.json
  python -c "
import json; d=json.load(open('$O/r02w_bins_$fs.json')); print('free_sweep=$fs bins', json.dumps(d.get('bins'))[:900])"
done
