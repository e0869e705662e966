#!/bin/bash
# What do the waves of one workload's render kernel wait for?  Three PMC passes (each alone, no traces).
# usage: scripts/pmc_waits.sh <label> <bench args...>
set -e
L=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcw_$L
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--frames-in-flight 1 --steps 6 --warmup 2 --no-cpu-baseline $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d $OUT/p1 -- python3 $R/bench.py $ARGS > $OUT/p1.log 2>&1 || echo "pass 1 failed"
rocprofv3 --pmc SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SMEM SQ_IFETCH \
    --output-format csv -d $OUT/p2 -- python3 $R/bench.py $ARGS > $OUT/p2.log 2>&1 || echo "pass 2 failed"
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_TC_STALL SQ_IFETCH_LEVEL \
    --output-format csv -d $OUT/p3 -- python3 $R/bench.py $ARGS > $OUT/p3.log 2>&1 || echo "pass 3 failed"
python3 - $OUT $L <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts"))
from make_pmc_json import per_launch
out, label = sys.argv[1], sys.argv[2]
c = {}
for p in ("p1", "p2", "p3"):
    d, n = per_launch(out, p)
    c.update(d)
print("== %s" % label)
for k in sorted(c):
    print("%-28s %14.6g" % (k, c[k]))
PY
