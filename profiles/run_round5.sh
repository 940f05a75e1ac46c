#!/bin/bash
# round 5 at HEAD, on the GPU box: the whole -m gpu suite, smoke, the judged profile (kernel trace, FETCH / WRITE, SQ / LDS / L2 of k_score; the
# post-pass kernels' counters), configs[2] counters, the default bench line, the 8-rank command at full size on the one GPU, the mapper and RMQ rates
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; TAG=${1:-r05}
python -m pytest tests -m gpu -x -q --durations=12 > $O/${TAG}_gputest.log 2>&1; echo "gputest rc=$?"; tail -16 $O/${TAG}_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash profiles/profile_round.sh $TAG
bash profiles/collect_pmc.sh 500000000 $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG > gpurun_out/${TAG}_pmc.log 2>&1
python profiles/summarize_round.py $TAG | tail -14
bash profiles/collect_pmc_post.sh $GRAFT_REPO_ROOT/gpurun_out/pmc_post_$TAG > $O/${TAG}_pmc_post.log 2>&1
python profiles/summarize_post.py $TAG > $O/${TAG}_post_summary.log 2>&1; tail -12 $O/${TAG}_post_summary.log
bash profiles/profile_config2.sh $TAG > $O/${TAG}_config2.log 2>&1; tail -5 $O/${TAG}_config2.log
cp profiles/${TAG}_*.csv profiles/${TAG}_*.json profiles/traffic_latest.json profiles/post_traffic_latest.json profiles/config2_counters_latest.json $O/ 2>/dev/null
python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; echo "bench rc=$?"; cut -c1-300 $O/${TAG}_bench_default.json
bash profiles/dryrun_8ranks.sh $TAG
python profiles/mapper_rate.py 3000 > $O/${TAG}_mapper_rate_3000.json 2> $O/${TAG}_mapper_rate_3000.err; echo "mapper rc=$?"; tail -c 600 $O/${TAG}_mapper_rate_3000.json
python profiles/rmq_rate.py > $O/${TAG}_rmq_rate.json 2> $O/${TAG}_rmq_rate.err; echo "rmq rc=$?"; tail -c 400 $O/${TAG}_rmq_rate.json
