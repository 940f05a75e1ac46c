#!/bin/bash
# round 6 at HEAD, on the GPU box: part a -- the whole -m gpu suite, smoke, the judged profile (kernel trace, FETCH / WRITE, SQ / LDS / L2 of k_score),
# the post-pass kernels' counters (profiles/pmc_post_only.sh: the split post-pass), configs[2] counters, the default bench line; part b -- the 8-rank
# command at full size on the one GPU, the mapper and RMQ rates.   bash profiles/run_round6.sh r06 a|b
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; TAG=${1:-r06}; PART=${2:-a}
if [ "$PART" = a ]; then
  timeout 900 python -m pytest tests -m gpu -x -q --durations=12 > $O/${TAG}_gputest.log 2>&1; echo "gputest rc=$?"; tail -16 $O/${TAG}_gputest.log
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  timeout 900 bash profiles/profile_round.sh $TAG
  timeout 900 bash profiles/collect_pmc.sh 500000000 $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG > gpurun_out/${TAG}_pmc.log 2>&1
  python profiles/summarize_round.py $TAG | tail -14
  timeout 1500 bash profiles/pmc_post_only.sh $GRAFT_REPO_ROOT/gpurun_out/pmc_post_$TAG sq1 sq2 fetch write l2 > $O/${TAG}_pmc_post.log 2>&1
  python profiles/summarize_post_only.py gpurun_out/pmc_post_$TAG $TAG > $O/${TAG}_post_summary.log 2>&1; tail -30 $O/${TAG}_post_summary.log
  timeout 300 bash profiles/post_trace.sh ${TAG}_post > /dev/null 2>&1; cp $O/${TAG}_post/post_kernel_stats.csv $O/${TAG}_post_kernel_stats.csv; cp $O/${TAG}_post/post_kernel_order.txt $O/${TAG}_post_kernel_order.txt
  MM2GB_DEBUG_PHASES=1 timeout 200 python3 profiles/post_only.py --runs 2 > $O/${TAG}_post_schedule.txt 2>&1
  timeout 900 bash profiles/profile_config2.sh $TAG > $O/${TAG}_config2.log 2>&1; tail -5 $O/${TAG}_config2.log
  cp profiles/${TAG}_*.csv profiles/${TAG}_*.json profiles/traffic_latest.json profiles/post_traffic_latest.json profiles/config2_counters_latest.json $O/ 2>/dev/null
  timeout 1500 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; echo "bench rc=$?"; cut -c1-300 $O/${TAG}_bench_default.json
else
  timeout 900 bash profiles/dryrun_8ranks.sh $TAG
  timeout 600 python profiles/mapper_rate.py 3000 > $O/${TAG}_mapper_rate_3000.json 2> $O/${TAG}_mapper_rate_3000.err; echo "mapper rc=$?"; tail -c 600 $O/${TAG}_mapper_rate_3000.json
  timeout 600 python profiles/rmq_rate.py > $O/${TAG}_rmq_rate.json 2> $O/${TAG}_rmq_rate.err; echo "rmq rc=$?"; tail -c 400 $O/${TAG}_rmq_rate.json
fi
