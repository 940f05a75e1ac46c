#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
for v in default trk default trk; do
  if [ $v = trk ]; then export MM2GB_LIB_PATH=$GRAFT_REPO_ROOT/mm2-gb_amd/variants/trk/libmm2gb_chain.so; else unset MM2GB_LIB_PATH; fi
  timeout 300 python bench.py --cpu-seconds 0 --no-pcie --no-bins --no-e2e --no-post --steps 5 --warmup 2 > $O/r02l_bench_$v.json 2> $O/r02l_bench_$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r02l_bench_$v.json")); print("$v", d["stage_ms"], "%.4f T pairs/s" % (d["value"]/1e12))
PY
done
unset MM2GB_LIB_PATH
timeout 600 python profiles/rmq_rate.py --reads 2000 --out $O/r02l_rmq_rate.json | tail -1
