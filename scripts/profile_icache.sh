#!/bin/bash
# Instruction-cache counters of the render kernel (own PMC pass, no traces).  usage: bash scripts/profile_icache.sh
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_icache
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --frames-in-flight 1 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_kernel" in row.get("Kernel_Name", ""):
            agg[row["Counter_Name"]][0] += float(row["Counter_Value"]); agg[row["Counter_Name"]][1] += 1
for k, (s, n) in sorted(agg.items()):
    print("%-22s %16.1f per launch (%d)" % (k, s / n, n))
PY
