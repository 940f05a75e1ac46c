# usage: bash profiles/pmc_variant.sh NAME [path/to/lib.so]   -> gpurun_out/pmc_NAME/{sq1,sq2}/...
set -u
NAME=$1; [ $# -ge 2 ] && export MM2GB_LIB_PATH=$2
# KERNEL: which instantiation to report -- batches of up to 150 M anchors launch k_score<0, false, true> (the gang build); fails loudly when no row matches
KERNEL=${KERNEL:-"k_score<0, false,"}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$NAME; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/bench.py" --anchors ${ANCHORS:-500000000} --steps 1 --warmup 1 --cpu-seconds 0 --no-pcie --no-e2e --no-bins --no-post --no-config2 > "$OUT/$name.log" 2>&1; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH
run sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
python3 - <<PY
import csv, glob, collections
for run in ("sq1", "sq2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % run, recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "$KERNEL" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        if not acc: raise SystemExit("pmc_variant.sh: no row of a kernel matching '$KERNEL' in " + f)
        for k in sorted(acc): print("$NAME", k, "%.4g" % (acc[k] / n[k]), "launches", n[k])
PY
