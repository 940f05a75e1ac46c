# A/B of kernel builds on one box, back to back: VARIANTS="name[:ENV=VAL] ..." ; name = main or a library mm2-gb_amd/ab/lib<name>.so
# ANCHORS (default: bench.py's) sets the batch size
for rep in 1 2; do
for spec in ${VARIANTS:-main}; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
  if [ $v = main ]; then unset MM2GB_LIB_PATH; else export MM2GB_LIB_PATH=$PWD/mm2-gb_amd/ab/lib$v.so; fi
  env ${envs//,/ } timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-pcie --no-e2e --no-bins --no-post ${ANCHORS:+--anchors $ANCHORS} 2>/dev/null | tail -1 > gpurun_out/ab_tmp.json
  python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$spec', d['config']['anchors_per_gpu'], round(d['value']/1e12,3), d['roofline']['kernel_ms'], round(d['ms_per_step'],3))"
done
done
