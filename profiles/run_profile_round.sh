#!/bin/bash
# The judged profile of a round, on the GPU box:  bash profiles/run_profile_round.sh r02
# kernel trace + FETCH/WRITE passes of the default bench (profile_round.sh), SQ / LDS / L2 counter passes (collect_pmc.sh),
# then the small summaries under profiles/ (summarize_round.py) and the default bench line itself.
TAG=${1:-r02}
cd "$GRAFT_REPO_ROOT" || exit 1
bash profiles/profile_round.sh $TAG
bash profiles/collect_pmc.sh 500000000 $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG > gpurun_out/${TAG}_pmc.log 2>&1
python profiles/summarize_round.py $TAG
cp profiles/${TAG}_*.csv profiles/${TAG}_*.json profiles/traffic_latest.json gpurun_out/ 2>/dev/null
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
MM2GB_DEVICES=0,0 python bench.py --gpus 2 --anchors 200000000 --host-anchors 100000000 --no-post > gpurun_out/${TAG}_bench_2ranks_one_gpu.json 2> gpurun_out/${TAG}_bench_2ranks_one_gpu.err
cut -c1-300 gpurun_out/${TAG}_bench_default.json
