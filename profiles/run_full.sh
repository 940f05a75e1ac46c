#!/bin/bash
# everything the driver runs at round end, on the GPU box: the whole -m gpu suite with durations, smoke(), the default bench line,
# the N > 1 path on one GPU, and the RCCL rendezvous path at world size 1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
TAG=${1:-r02}
python -m pytest tests -m gpu -x -q --durations=12 > $O/${TAG}_gputest.log 2>&1; echo "gputest rc=$?"; tail -18 $O/${TAG}_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; echo "bench rc=$?"
MM2GB_DEVICES=0,0 python bench.py --gpus 2 --anchors 200000000 --host-anchors 100000000 --no-post > $O/${TAG}_bench_2ranks_one_gpu.json 2> $O/${TAG}_bench_2ranks_one_gpu.err; echo "bench2 rc=$?"
MM2GB_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --anchors 50000000 --cpu-seconds 0 --no-bins --no-e2e --no-post --no-pcie > $O/${TAG}_bench_force_dist.json 2> $O/${TAG}_bench_force_dist.err; echo "force-dist rc=$?"
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench_default.json"))
print(json.dumps(d["roofline"])[:700]); print(d["value"], d["ms_per_step"], d["post_pass_device"]["ms"], d["host_path"]["scores_only"]["seconds"], d["host_path"]["chains_rank0_alone"]["seconds"], d["cpu_baseline"]["value"], d["e2e"])
d = json.load(open("gpurun_out/${TAG}_bench_force_dist.json")); print("force dist:", d["config"]["rendezvous"], d["value"])
PY
python profiles/stream_api_rate.py --out $O/${TAG}s_stream_api_rate.json > $O/${TAG}s_stream_api_rate.log 2>&1; echo "stream rate rc=$?"; grep reads_per_batch $O/${TAG}s_stream_api_rate.log | cut -c1-230
MM2GB_POST=gpu python profiles/stream_api_rate.py --out $O/${TAG}s_stream_api_rate_device_post.json > $O/${TAG}s_stream_api_rate_device_post.log 2>&1; echo "stream rate (device post) rc=$?"; grep reads_per_batch $O/${TAG}s_stream_api_rate_device_post.log | cut -c1-230
