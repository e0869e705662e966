#!/bin/bash
# quick timing of the specialised kernels.  usage: scripts/rtc_quick.sh "<workloads>" "<opt=value ...>" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
WLS=$1; shift
for o in "$@"; do
  echo "== $o"
  timeout -k 10 300 python scripts/kbench.py $WLS $o hash=1 frames=${FRAMES:-48} 2>&1 | grep -v amdgpu.ids
done
