# usage: bash profiles/pmc_quick.sh NAME [path/to/lib.so]: instruction counts of the largest k_score launch of a bare bench step
set -u
NAME=$1; [ $# -ge 2 ] && export MM2GB_LIB_PATH=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$NAME; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d "$OUT/sq" -- python3 "$GRAFT_REPO_ROOT/bench.py" --anchors ${ANCHORS:-500000000} --steps 1 --warmup 0 --cpu-seconds 0 --no-pcie --no-bins --no-post --no-e2e --no-config2 > "$OUT/sq.log" 2>&1
python3 - <<PY
import csv, glob, collections
by = collections.defaultdict(dict)
for f in glob.glob("$OUT/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_score" in r["Kernel_Name"]:
            by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
if not by: raise SystemExit("pmc_quick.sh: no k_score row in the counter files under $OUT/sq")
best = max(by.values(), key=lambda d: d.get("SQ_INSTS_VALU", 0))      # the bench batch's launch (the early-exit launches of the other MODEs count almost nothing)
print("$NAME", {k: "%.4g" % v for k, v in sorted(best.items())})
PY
