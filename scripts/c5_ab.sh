#!/bin/bash
# C5 kernel alone under option sets, then the octree parity tests.  usage: scripts/c5_ab.sh "<opts a>" "<opts b>" ...   (opts: k=v,k=v)
[ $# -eq 0 ] && set -- "oct_lean=1" "oct_lean=0"
for o in "$@"; do
  args=""; for kv in ${o//,/ }; do args="$args --opt $kv"; done
  python bench.py --workload C5 --frames-in-flight 1 --no-cpu-baseline --steps 10 --warmup 3 $args > gpurun_out/c5_ab.json 2>gpurun_out/c5_ab.err || { tail -5 gpurun_out/c5_ab.err; exit 1; }
  python - "$o" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c5_ab.json").read().strip().splitlines()[-1])
print("%-28s %7.1f frames/s  kernel %.3f ms  %s" % (sys.argv[1], d["value"], d["roofline"]["kernel_ms"], d["roofline"].get("kernel")))
PY
done
python -m pytest tests -x -q -m gpu -k "octree or Octree or C5 or c5 or golden or accel" 2>&1 | tail -3
