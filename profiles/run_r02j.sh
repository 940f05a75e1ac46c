#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
for n in tiny dense_tied; do timeout 20 python -u profiles/rmq_dbg_tmp.py $n 2>&1 | grep -v "amdgpu.ids" | tail -1; echo "-- $n"; done
timeout 120 python -u -m pytest tests/test_gpu_rmq.py -x -q > $O/r02j_rmq.log 2>&1; echo "rmq rc=$?"; tail -5 $O/r02j_rmq.log | cut -c1-200
timeout 200 python -u -m pytest tests/test_gpu_e2e_host.py -x -q > $O/r02j_e2e.log 2>&1; echo "e2e rc=$?"; tail -5 $O/r02j_e2e.log | cut -c1-300
