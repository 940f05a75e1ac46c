for pct in 0 25 50 100 200; do for a in 500000000 100000000 50000000 20000000; do MM2GB_WHOLE_WG_PCT=$pct MM2GB_BENCH_CPU_SECONDS=0 python bench.py --anchors $a --steps 3 --warmup 1 --no-pcie 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pct', $pct, d['config']['anchors_per_gpu'], round(d['value']/1e12,3), d['roofline']['kernel_ms'])"; done; done
