#!/bin/bash
# Kernel time of the SURVEY 8(f) workloads (N3 boxes/tori, N4 operators/Mandelbulb): one bench.py JSON line each
# into gpurun_out/widen_<workload>.json.  Usage: scripts/bench_widen.sh [workload ...]
set -e
WL=${@:-N3 N3mixed N4chicken N4screw N4mandelbulb}
for w in $WL; do
  timeout -k 10 200 python bench.py --workload $w --steps 24 --warmup 2 > gpurun_out/widen_$w.json 2> gpurun_out/widen_$w.err
  python - "$w" <<'PY'
import json, sys
d = json.load(open('gpurun_out/widen_%s.json' % sys.argv[1]))
print('%-14s %8.1f fps  kernel %7.2f ms  sdf/px %.2f' % (sys.argv[1], d['value'], d['roofline']['kernel_ms'], d['avg_sdf_calls_per_pixel']))
PY
done
