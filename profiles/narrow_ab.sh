#!/bin/bash
# The narrow-window regime (10-30 kb reads, SURVEY 8d bins[0]) with the edge blocks' window test from a scalar prefix mask (default) against the
# vector compare per source (MM2GB_EDGE=old); the same for the headline workload.   bash profiles/narrow_ab.sh > profiles/r06_narrow_ab.txt
cd "${GRAFT_REPO_ROOT:-.}"
one() { # label, env, bench arguments
  local label=$1 e=$2; shift 2
  for rep in 1 2; do
    env $e timeout 600 python bench.py --no-pcie --no-bins --no-e2e --no-post --no-config2 --cpu-seconds 0 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-44s %s  %.4f T pairs/s  %.3f ms/step  (k_score %.3f ms)' % ('$label', '$e' or 'default', d['value']/1e12, d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
}
one "10-30 kb reads, 100 M anchors" "MM2GB_EDGE=new" --len-lo 10000 --len-hi 30000 --anchors 100000000 --steps 10 --warmup 2
one "10-30 kb reads, 100 M anchors" "MM2GB_EDGE=old" --len-lo 10000 --len-hi 30000 --anchors 100000000 --steps 10 --warmup 2
one "30-100 kb reads, 100 M anchors" "MM2GB_EDGE=new" --len-lo 30000 --len-hi 100000 --anchors 100000000 --steps 10 --warmup 2
one "30-100 kb reads, 100 M anchors" "MM2GB_EDGE=old" --len-lo 30000 --len-hi 100000 --anchors 100000000 --steps 10 --warmup 2
one "100-300 kb reads, 500 M anchors (headline)" "MM2GB_EDGE=new" --steps 10 --warmup 2
one "100-300 kb reads, 500 M anchors (headline)" "MM2GB_EDGE=old" --steps 10 --warmup 2
