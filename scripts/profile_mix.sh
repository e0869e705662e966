#!/bin/bash
# VALU instruction mix of the bench command (rocprofv3 PMC passes, each alone).
# usage: scripts/profile_mix.sh <tag> [bench args...]
set -e
TAG=${1:-mix}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline $@"
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
for SET in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32" \
           "SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU"; do
  N=$(echo $SET | cut -c1-40 | tr ' ' '_')
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$N -- python3 $R/bench.py $ARGS > $OUT/pmc_$N.log 2>&1 || echo "set failed: $SET"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'render_kernel' not in r.get('Kernel_Name', ''): continue
        tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(tot): print('%-28s %14.4g per launch (%d launches)' % (k, tot[k] / max(n[k], 1), n[k]))
PY
