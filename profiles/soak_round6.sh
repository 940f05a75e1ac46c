#!/bin/bash
# Fuzz soaks at HEAD (round 6): the RMQ fill with ties weighed (random reads and parameters, random device form per batch), the post-pass with two short runs
# to a wave, and the score kernels' six engine configurations -- each in-suite fuzz under N more seeds.   bash profiles/soak_round6.sh [first seed] [seeds]
cd "${GRAFT_REPO_ROOT:-.}"
first=${1:-6000}; n=${2:-30}
for s in $(seq $first $((first + n - 1))); do
  MM2GB_FUZZ_SEED=$s timeout 900 python -m pytest tests/test_gpu_rmq.py -x -q -k fuzz 2>&1 | tail -1 | sed "s/^/seed $s rmq: /"
done
for s in $(seq $first $((first + n / 2 - 1))); do
  MM2GB_FUZZ_SEED=$s timeout 900 python -m pytest tests/test_gpu_post.py tests/test_gpu_parity.py -x -q -k fuzz 2>&1 | tail -1 | sed "s/^/seed $s post + score: /"
done
