#!/bin/bash
# round 4 at HEAD, on the GPU box: the whole -m gpu suite, smoke, the judged profile (kernel trace, FETCH / WRITE, SQ / LDS / L2 of k_score; the
# post-pass kernels' counters), the default bench line, N > 1 on one GPU (2 ranks; 8 ranks dry run), the mapper and RMQ rates
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; TAG=${1:-r04}
python -m pytest tests -m gpu -x -q --durations=12 > $O/${TAG}_gputest.log 2>&1; echo "gputest rc=$?"; tail -16 $O/${TAG}_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash profiles/run_profile_round.sh $TAG 2>&1 | tail -3
bash profiles/collect_pmc_post.sh $GRAFT_REPO_ROOT/gpurun_out/pmc_post_$TAG > $O/${TAG}_pmc_post.log 2>&1
python profiles/summarize_post.py $TAG > $O/${TAG}_post_summary.log 2>&1; tail -12 $O/${TAG}_post_summary.log
cp profiles/${TAG}_post_counters.json profiles/post_traffic_latest.json $O/ 2>/dev/null
MM2GB_DEVICES=0,0,0,0,0,0,0,0 python bench.py --gpus 8 --anchors 60000000 --host-anchors 30000000 --no-post > $O/${TAG}_bench_8ranks_one_gpu.json 2> $O/${TAG}_bench_8ranks_one_gpu.err; echo "bench8 rc=$?"
python profiles/mapper_rate.py 3000 > $O/${TAG}_mapper_rate_3000.json 2> $O/${TAG}_mapper_rate_3000.err; echo "mapper rc=$?"; tail -c 600 $O/${TAG}_mapper_rate_3000.json
python profiles/rmq_rate.py > $O/${TAG}_rmq_rate.json 2> $O/${TAG}_rmq_rate.err; echo "rmq rc=$?"; tail -c 400 $O/${TAG}_rmq_rate.json
