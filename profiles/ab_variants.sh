# A/B of kernel builds: VARIANTS="name[:ENV=VAL] ..." ; name = main or a library in mm2-gb_amd/variants/lib<name>.so
for spec in ${VARIANTS:-main}; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
  if [ $v = main ]; then unset MM2GB_LIB_PATH; else export MM2GB_LIB_PATH=$PWD/mm2-gb_amd/variants/lib$v.so; fi
  env $envs MM2GB_BENCH_CPU_SECONDS=0 timeout 300 python bench.py --steps 3 --warmup 1 --no-pcie 2>/dev/null | tail -1 > gpurun_out/ab_tmp.json
  python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$spec', round(d['value']/1e12,3), d['roofline']['kernel_ms'], d['plan'])"
done
