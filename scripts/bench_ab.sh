#!/bin/bash
# Same-box A/B of library builds on bench.py lines.  usage: scripts/bench_ab.sh "<workloads>" name=path ...  (path "" = the tree's library)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
WLS=$1; shift
for w in $WLS; do for v in "$@"; do
  name=${v%%=*}; path=${v#*=}
  if [ -n "$path" ]; then export RM_HIP_LIB=$R/$path; else unset RM_HIP_LIB; fi
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s %-12s %9.1f frames/s  alone %.3f ms' % ('$name', '$w', d['value'], d['roofline']['kernel_ms']))"
done; done
