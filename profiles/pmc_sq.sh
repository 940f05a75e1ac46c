set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_r01v9; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/bench.py" --anchors 500000000 --steps 1 --warmup 1 --cpu-seconds 0 --no-pcie > "$OUT/$name.log" 2>&1; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH
run sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
