#!/bin/bash
# One PMC pass: scalar-memory and scalar-ALU instructions and wait cycles of the render kernel, per launch.
# usage: scripts/pmc_smem.sh <label> <bench args...>
L=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcs_$L; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/p1 -- python3 $R/bench.py --frames-in-flight 1 --steps 6 --warmup 2 --no-cpu-baseline --no-verify "$@" > $OUT/p1.log 2>&1 || echo "pass failed"
python3 - $OUT $L <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts"))
from make_pmc_json import per_launch
c, n = per_launch(sys.argv[1], "p1")
print("== %s (%d launches)" % (sys.argv[2], n))
for k in sorted(c):
    print("%-22s %14.6g" % (k, c[k]))
PY
