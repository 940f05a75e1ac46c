#!/bin/bash
# kernel trace of the device post-pass alone (profiles/post_only.py): per-kernel average durations -> gpurun_out/<tag>/post_kernel_stats.csv
# usage: profiles/post_trace.sh <tag> [post_only.py arguments]
tag=${1:-trace}; shift
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o post -- python3 $root/profiles/post_only.py --runs 3 "$@" > $out/post_trace.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/post_kernel_stats.csv && grep -E "k_post|Name" $out/post_kernel_stats.csv | cut -c1-200
tail -2 $out/post_trace.log
f2=$(find $out/prof -name "*kernel_trace.csv" | head -1); [ -n "$f2" ] && python3 - "$f2" > $out/post_kernel_order.txt <<PY
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "k_post" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=None
for r in rows[-40:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    t0=t0 or s
    print("%-40s start %8.3f ms  dur %8.3f ms" % (r["Kernel_Name"].split("(")[0][-38:], (s-t0)/1e6, (e-s)/1e6))
PY
rm -rf $out/prof
