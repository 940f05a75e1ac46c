# A/B of post-pass builds on one box: VARIANTS="name ..." ; name = main or a library mm2-gb_amd/ab/lib<name>.so; prints bench.py's post_pass_device.ms
for rep in 1 2; do
for v in ${VARIANTS:-main}; do
  if [ $v = main ]; then unset MM2GB_LIB_PATH; else export MM2GB_LIB_PATH=$PWD/mm2-gb_amd/ab/lib$v.so; fi
  timeout 300 python bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-pcie --no-e2e --no-bins ${ANCHORS:+--anchors $ANCHORS} 2>/dev/null | tail -1 > gpurun_out/ab_tmp.json
  python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$v', d['config']['anchors_per_gpu'], 'post_pass_device ms', d['post_pass_device']['ms'], 'chains', d['post_pass_device']['chains'])"
done
done
