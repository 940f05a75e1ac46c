OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_window; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/bench.py" --anchors 500000000 --steps 1 --warmup 1 --cpu-seconds 0 --no-pcie --no-e2e --no-bins --no-post > "$OUT/$name.log" 2>&1; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VMEM_WR
run sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_FLAT TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
python3 - <<PY
import csv, glob, collections
for run in ("sq1", "sq2", "sq3"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % run, recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "k_window" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in sorted(acc): print("k_window", k, "%.4g" % (acc[k] / n[k]), "launches", n[k])
PY
