#!/bin/bash
# HBM traffic (FETCH_SIZE + WRITE_SIZE, separate passes) and kernel time of one workload's render kernel.  usage: scripts/traffic_quick.sh <workload> [bench args]
W=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trq_$W; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $W --frames-in-flight 1 --steps 6 --warmup 2 --no-cpu-baseline $@"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- python3 $R/bench.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p2 -- python3 $R/bench.py $ARGS > $OUT/p2.log 2>&1
python3 - $OUT $W <<'PY'
import sys, os, json
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts"))
from make_pmc_json import per_launch
c = {}
for p in ("p1", "p2"):
    d, n = per_launch(sys.argv[1], p)
    c.update(d)
k = [json.loads(l) for l in open(os.path.join(sys.argv[1], "p1.log")) if l.startswith("{")][-1]["roofline"]["kernel_ms"]
print("%s: fetch %.1f MB  write %.1f MB  (kernel %.3f ms under the profiler)" % (sys.argv[2], c.get("FETCH_SIZE", 0) / 1024, c.get("WRITE_SIZE", 0) / 1024, k))
PY
