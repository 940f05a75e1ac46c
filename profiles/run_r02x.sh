#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 profiles/ubench/lds_region.hip -o /tmp/lds_region 2>/dev/null && timeout 120 /tmp/lds_region > $O/r02x_lds_region.txt 2>&1; echo "rc=$?"; cat $O/r02x_lds_region.txt
