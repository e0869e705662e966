#!/bin/bash
# The round's whole measurement battery in one GPU call (re-run after ANY edit under cpu_raymarcher_amd/csrc: the PMC files
# are stamped with the kernel sources' hash and bench.py refuses stale ones).
#   1. scripts/profile_r02.sh for C3, C2, C5 (kernel trace + PMC passes)  -> gpurun_out/profiles_r02/pmc_<W>.{json,txt}
#   2. the bench lines                                                       -> gpurun_out/r02/bench_*.json
#   3. scripts/shard_overhead.py (N = 2, 4, 8; C3 and C5)                    -> gpurun_out/r02/shard_overhead.txt
# Afterwards, in the build container: cp gpurun_out/profiles_r02/* gpurun_out/r02/* profiles/r02/ ; make -C csrc asm ;
# python scripts/price_valu.py C3 C2 C5
# usage: scripts/battery_r02.sh [quick]      (quick: profiles + the C3 / C2 / C5 lines only)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r02; mkdir -p $O gpurun_out/profiles_r02
for w in C3 C2 C5; do
  scripts/profile_r02.sh $w > gpurun_out/profile_$w.log 2>&1
  tail -1 gpurun_out/profile_$w.log
  mkdir -p profiles/r02 && cp gpurun_out/profiles_r02/pmc_$w.json gpurun_out/profiles_r02/pmc_$w.txt profiles/r02/   # bench.py reads profiles/r02
  cp $(ls -S gpurun_out/prof_r02_$w/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats_$w.csv 2>/dev/null || true
done
cd $R
# the `valu` block: the instruction mix priced with the measured issue costs needs the kernels' listings (built here: same
# compiler, same sources as the library that travelled)
make -s -C cpu_raymarcher_amd/csrc asm > gpurun_out/asm.log 2>&1 && python scripts/price_valu.py C3 C2 C5 > gpurun_out/price_valu.log 2>&1 || echo "price_valu failed (see gpurun_out/price_valu.log)"
cp profiles/r02/pmc_C3.json profiles/r02/pmc_C2.json profiles/r02/pmc_C5.json gpurun_out/profiles_r02/ 2>/dev/null || true
tail -3 gpurun_out/price_valu.log
line() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 400 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-28s %9.1f %-9s kernel alone %.3f ms frac %.4f traffic %s valu %s" % (sys.argv[1].split("bench_")[-1], d["value"], d["unit"], r.get("kernel_ms", 0), r["frac"], r.get("traffic"),
      (r.get("valu") or {}).get("frac")))
PY
}
line C3
line C3_20steps --steps 20 --warmup 5 --no-cpu-baseline
line C3_serial --frames-in-flight 1 --no-cpu-baseline
line C2 --workload C2
line C5 --workload C5
if [ "$1" != quick ]; then
  line C3_sqrt --opt length=1 --no-cpu-baseline
  for w in N3 N3mixed N4chicken N4screw N4mandelbulb; do line $w --workload $w --no-cpu-baseline; done
  line analytics_sweep --analytics-sweep --steps 60 --no-cpu-baseline
  timeout -k 10 300 python scripts/shard_overhead.py 240 N=2 N=4 N=8 > $O/shard_overhead.txt 2>&1
  timeout -k 10 300 python scripts/shard_overhead.py 120 N=8 WL=C5 >> $O/shard_overhead.txt 2>&1
  cat $O/shard_overhead.txt
fi
