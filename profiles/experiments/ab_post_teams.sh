for rep in 1 2; do
for t in ${TEAMS:-0 64 256 640 1500}; do
  MM2GB_POST_TEAM_READS=$t timeout 300 python bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-pcie --no-e2e --no-bins 2>/dev/null | tail -1 > gpurun_out/ab_tmp.json
  python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('team reads $t', 'post_pass_device ms', d['post_pass_device']['ms'])"
done
done
