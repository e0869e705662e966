#!/bin/bash
# The round's whole measurement battery in one GPU call (re-run after ANY edit under cpu_raymarcher_amd/csrc: the PMC files
# are stamped with the kernel sources' hash and bench.py refuses stale ones).  Started by scripts/gpu_battery.sh, which
# leaves the commit in .build_commit for the stamps.
#   0. scripts/bin/valu_issue_bench                                          -> profiles/<round>/valu_issue_costs.json
#   1. scripts/profile_round.sh for C3, C2, C5, N4chicken (kernel trace + PMC passes)   -> profiles/<round>/pmc_<W>.{json,txt}
#   2. the bench lines                                                       -> profiles/<round>/bench_*.json
#   3. scripts/shard_overhead.py (N = 2, 4, 8; C3 and C5)                    -> profiles/<round>/shard_overhead.txt
# Everything is copied to gpurun_out/<round>/ as well (gpurun merges only gpurun_out/ back).
# usage: scripts/battery.sh [quick]      (quick: profiles + the C3 / C2 / C5 lines only)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export RM_ROUND=${RM_ROUND:-r03}
P=profiles/$RM_ROUND
O=gpurun_out/$RM_ROUND; mkdir -p $O $P
if [ -x scripts/bin/valu_issue_bench ]; then
  timeout -k 10 300 scripts/bin/valu_issue_bench > $P/valu_issue_costs.json 2> $O/valu_issue_bench.err || echo "valu_issue_bench failed"
fi
for w in C3 C2 C5 N4chicken; do
  scripts/profile_round.sh $w > gpurun_out/profile_$w.log 2>&1 || echo "profile $w failed"
  tail -1 gpurun_out/profile_$w.log
  cp $(ls -S gpurun_out/prof_${RM_ROUND}_$w/trace/*/*kernel_stats.csv | head -1) $P/kernel_stats_$w.csv 2>/dev/null || true
done
# the `valu` block: the instruction mix priced with the measured issue costs needs the kernels' listings (built here: same
# compiler, same sources as the library that travelled)
make -s -C cpu_raymarcher_amd/csrc asm > gpurun_out/asm.log 2>&1 && python scripts/price_valu.py C3 C2 C5 > gpurun_out/price_valu.log 2>&1 || echo "price_valu failed (see gpurun_out/price_valu.log)"
tail -3 gpurun_out/price_valu.log
line() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 400 python bench.py "$@" > $P/bench_$name.json 2> $O/bench_$name.err || echo "bench $name failed"
  python - $P/bench_$name.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-28s %9.1f %-9s kernel alone %.3f ms frac %.4f traffic %s valu %s verified %s" % (sys.argv[1].split("bench_")[-1], d["value"], d["unit"], r.get("kernel_ms", 0), r["frac"], r.get("traffic"),
      (r.get("valu") or {}).get("frac"), d.get("frames_verified")))
PY
}
line C3
line C3_20steps --steps 20 --warmup 5 --no-cpu-baseline
line C3_serial --frames-in-flight 1 --no-cpu-baseline
line C3_reduce_kernels --diagnostics reduce --no-cpu-baseline
line C2 --workload C2
line C5 --workload C5
if [ "$1" != quick ]; then
  line C3_sqrt --opt length=1 --no-cpu-baseline
  for w in N3 N3mixed N4chicken N4screw N4mandelbulb; do line $w --workload $w --no-cpu-baseline; done
  # the same scenes through the device interpreter (the run-time specialiser off) and the specialised code without pruning
  for w in N4chicken N4screw N4mandelbulb; do line ${w}_interpreter --workload $w --opt specialise=0 --no-cpu-baseline; done
  line N4chicken_unpruned --workload N4chicken --opt prune=0 --no-cpu-baseline
  line analytics_sweep --analytics-sweep --steps 60 --no-cpu-baseline
  timeout -k 10 300 python scripts/shard_overhead.py 240 N=2 N=4 N=8 > $P/shard_overhead.txt 2>&1 || true
  timeout -k 10 300 python scripts/shard_overhead.py 120 N=8 WL=C5 >> $P/shard_overhead.txt 2>&1 || true
  timeout -k 10 300 python scripts/shard_overhead.py 240 N=8 GATHER=1 >> $P/shard_overhead.txt 2>&1 || true
  grep -v amdgpu.ids $P/shard_overhead.txt
fi
cp -r $P/* $O/ 2>/dev/null || true
rm -f $O/*.err.copy 2>/dev/null || true
