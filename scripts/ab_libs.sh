#!/bin/bash
# Same-box A/B of library builds: alternates kbench runs of each build (frame rates differ by ~2 % between boxes and runs,
# so builds are only compared inside one gpurun call).
# usage: scripts/ab_libs.sh <rounds> "<kbench args>" name=path[:extra kbench args] ...   (path "" = the tree's library)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
N=$1; shift
ARGS=$1; shift
for r in $(seq 1 $N); do
  for v in "$@"; do
    name=${v%%=*}; rest=${v#*=}; path=${rest%%:*}; extra=""
    [ "$rest" != "$path" ] && extra=${rest#*:}
    if [ -n "$path" ]; then export RM_HIP_LIB=$R/$path; else unset RM_HIP_LIB; fi
    printf "%-14s " $name
    timeout -k 10 300 python scripts/kbench.py $ARGS $extra 2>&1 | grep -v amdgpu.ids
  done
done
