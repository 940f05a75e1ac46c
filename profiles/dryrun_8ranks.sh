#!/bin/bash
# The 8-GPU command at its REAL per-rank load, on the one GPU there is: MM2GB_DEVICES=0,0,0,0,0,0,0,0 python3 bench.py --gpus 8 (500 M anchors
# per rank, every leg), with the host's memory and the device's memory sampled every 2 s.  No curve is claimed from it: it shows that the
# command the driver will run on an 8-GPU node completes, how long it takes and what it needs.   bash profiles/dryrun_8ranks.sh r05
TAG=${1:-r05}
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
( while true; do
    echo "$(date +%s) host_used_GiB=$(free -g | awk '/^Mem:/{print $3}') host_avail_GiB=$(free -g | awk '/^Mem:/{print $7}') vram_used_B=$(rocm-smi --showmeminfo vram 2>/dev/null | awk '/Used Memory/{print $NF; exit}')"
    sleep 2
  done ) > $O/${TAG}_dryrun8_mem.log 2>&1 &
SAMPLER=$!
t0=$(date +%s)
MM2GB_DEVICES=0,0,0,0,0,0,0,0 python3 bench.py --gpus 8 > $O/${TAG}_bench_8ranks_full_size_one_gpu.json 2> $O/${TAG}_bench_8ranks_full_size_one_gpu.err
rc=$?
t1=$(date +%s)
kill $SAMPLER
echo "rc=$rc wall_seconds=$((t1 - t0))" | tee $O/${TAG}_dryrun8_summary.txt
awk '{for(i=2;i<=NF;i++){split($i,a,"=");if(a[1]=="host_used_GiB"&&a[2]>h)h=a[2];if(a[1]=="vram_used_B"&&a[2]>v)v=a[2]}}END{printf "peak host_used_GiB=%d peak vram_used_GiB=%.1f\n",h,v/1073741824}' $O/${TAG}_dryrun8_mem.log | tee -a $O/${TAG}_dryrun8_summary.txt
cut -c1-600 $O/${TAG}_bench_8ranks_full_size_one_gpu.json
tail -5 $O/${TAG}_bench_8ranks_full_size_one_gpu.err
