#!/bin/bash
# What ending a process that holds 32 engines' memory costs (profiles/ubench/exit_cost.hip): whole program and the part after main.
# usage (GPU box, repo root): bash profiles/exit_cost.sh
cd ${GRAFT_REPO_ROOT:-.}
B=/tmp/exit_cost
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 profiles/ubench/exit_cost.hip -o $B || exit 1
run() {
  python3 - "$@" <<'P'
import subprocess, sys, time, re
t0 = time.time()
out = subprocess.run(["/tmp/exit_cost"] + sys.argv[1:], capture_output=True, text=True).stdout
t1 = time.time()
m = re.search(r"main ends at epoch ([0-9.]+)", out)
print(" ".join(sys.argv[1:]), "| whole %.3f s | after main %.3f s ||" % (t1 - t0, t1 - float(m.group(1)) if m else -1), " ;; ".join(l for l in out.splitlines() if not l.startswith("main ends")))
P
}
run 0.1 1 0.1 1 1 leave
for mode in free leave quick; do run 40 640 10 100 100 $mode; done
for mode in free leave; do run 40 32 10 10 100 $mode; done
for mode in free leave; do run 40 640 0.1 1 100 $mode; done
for mode in free leave; do run 0.1 1 10 100 100 $mode; done
for mode in free leave; do run 0.1 1 0.1 1 100 $mode; done
