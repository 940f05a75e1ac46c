#!/bin/bash
# HEAD check after the host-RMQ tie fix: whole -m gpu suite (all failures shown), smoke, mapper rate with re-chaining on host threads and
# on the device, RMQ kernel rate
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
python -m pytest tests -m gpu -q --durations=15 > $O/r02v_gputest.log 2>&1; echo "gputest rc=$?"; tail -25 $O/r02v_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python profiles/rmq_rate.py --out $O/r02v_rmq_rate.json > $O/r02v_rmq_rate.log 2>&1; echo "rmq rate rc=$?"; tail -3 $O/r02v_rmq_rate.log | cut -c1-600
python profiles/mapper_rate.py 3000 > $O/r02v_mapper_rate_3000.json 2> $O/r02v_mapper_rate_3000.err; echo "mapper 3000 rc=$?"; cut -c1-900 $O/r02v_mapper_rate_3000.json
MAPPER_RATE_NO_REF=1 MAPPER_RATE_RECHAIN_DEVICE=1 python profiles/mapper_rate.py 3000 > $O/r02v_mapper_rate_3000_dev.json 2> $O/r02v_mapper_rate_3000_dev.err; echo "mapper 3000 dev rc=$?"; cut -c1-900 $O/r02v_mapper_rate_3000_dev.json
python profiles/mapper_rate.py 400 > $O/r02v_mapper_rate_400.json 2> $O/r02v_mapper_rate_400.err; echo "mapper 400 rc=$?"; cut -c1-900 $O/r02v_mapper_rate_400.json
