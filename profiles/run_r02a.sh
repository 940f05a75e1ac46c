#!/bin/bash
# round 2, first GPU pass: GPU tests, default bench, the one-GPU check of the N>1 bench path, instruction-rate ubench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=15 > $O/r02a_gputest.log 2>&1; echo "gputest rc=$?"
tail -3 $O/r02a_gputest.log
python bench.py > $O/r02a_bench.json 2> $O/r02a_bench.err; echo "bench rc=$?"
MM2GB_DEVICES=0,0 python bench.py --gpus 2 --anchors 100000000 --host-anchors 50000000 --steps 3 > $O/r02a_bench_2ranks.json 2> $O/r02a_bench_2ranks.err; echo "bench2 rc=$?"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w profiles/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate > $O/r02a_valu_rate.txt 2>&1; echo "ubench rc=$?"
cut -c1-600 $O/r02a_bench.json; echo; cut -c1-400 $O/r02a_bench_2ranks.json; echo; tail -5 $O/r02a_bench_2ranks.err
