# kernel time of the default workload at several batch sizes: bash profiles/bench_sizes.sh [ENV=VAL ...]
for a in ${SIZES:-500000000 100000000 20000000}; do env "$@" MM2GB_BENCH_CPU_SECONDS=0 python bench.py --anchors $a --steps 3 --warmup 1 --no-pcie 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['anchors_per_gpu'], round(d['value']/1e12,3), 'T pairs/s  kernel', d['roofline']['kernel_ms'], 'ms')"; done
