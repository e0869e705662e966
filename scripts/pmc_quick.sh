#!/bin/bash
# Two PMC passes (each alone, no traces) over one bench.py configuration; prints the render kernel's per-launch means.
# usage: scripts/pmc_quick.sh <label> <bench args...>     e.g. scripts/pmc_quick.sh lean --workload C5 --opt oct_lean=1
set -e
L=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcq_$L
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--frames-in-flight 1 --steps 6 --warmup 2 --no-cpu-baseline $@"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
    --output-format csv -d $OUT/p1 -- python3 $R/bench.py $ARGS > $OUT/p1.log 2>&1 || echo "pass 1 failed"
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM \
    --output-format csv -d $OUT/p2 -- python3 $R/bench.py $ARGS > $OUT/p2.log 2>&1 || echo "pass 2 failed"
python3 - $OUT $L <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts"))
from make_pmc_json import per_launch
out, label = sys.argv[1], sys.argv[2]
c = {}
for p in ("p1", "p2"):
    d, n = per_launch(out, p)
    c.update(d)
print("== %s (%d launches)" % (label, n))
for k in sorted(c):
    print("%-32s %14.6g" % (k, c[k]))
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
    print("lane utilisation %.3f" % (c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])))
PY
