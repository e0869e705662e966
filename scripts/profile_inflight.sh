#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_inflight
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 48 --warmup 12 --no-cpu-baseline > $OUT/trace.log 2>&1
cat $OUT/trace/*/*kernel_stats.csv | cut -c1-160 | head -5
tail -1 $OUT/trace.log | cut -c1-300
