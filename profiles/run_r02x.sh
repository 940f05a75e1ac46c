#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python profiles/block_kinds.py > gpurun_out/r02x_block_kinds.json 2> gpurun_out/r02x_block_kinds.err; echo "rc=$?"; cat gpurun_out/r02x_block_kinds.json; tail -3 gpurun_out/r02x_block_kinds.err
