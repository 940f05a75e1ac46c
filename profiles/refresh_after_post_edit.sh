cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; TAG=r05
bash profiles/collect_pmc_post.sh $GRAFT_REPO_ROOT/gpurun_out/pmc_post_$TAG > $O/${TAG}_pmc_post.log 2>&1
python profiles/summarize_post.py $TAG > $O/${TAG}_post_summary.log 2>&1; tail -6 $O/${TAG}_post_summary.log
cp profiles/${TAG}_post_counters.json profiles/post_traffic_latest.json $O/ 2>/dev/null
python profiles/mapper_rate.py 3000 > $O/${TAG}_mapper_rate_3000.json 2> $O/${TAG}_mapper_rate_3000.err; echo "mapper rc=$?"; tail -c 500 $O/${TAG}_mapper_rate_3000.json
python profiles/rmq_rate.py > $O/${TAG}_rmq_rate.json 2> $O/${TAG}_rmq_rate.err; echo "rmq rc=$?"; tail -c 300 $O/${TAG}_rmq_rate.json
python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; echo "bench rc=$?"; cut -c1-200 $O/${TAG}_bench_default.json
