#!/bin/bash
# one workload's kernel alone under option sets.  usage: scripts/wl_ab.sh <workload> "<k=v,k=v>" ...
W=$1; shift
for o in "$@"; do
  args=""; for kv in ${o//,/ }; do [ "$kv" != "-" ] && args="$args --opt $kv"; done
  python bench.py --workload $W --frames-in-flight 1 --no-cpu-baseline --steps 10 --warmup 3 $args > gpurun_out/wl_ab.json 2>gpurun_out/wl_ab.err || { tail -5 gpurun_out/wl_ab.err; exit 1; }
  python - "$W $o" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/wl_ab.json").read().strip().splitlines()[-1])
print("%-36s %8.1f frames/s  kernel %.3f ms  %s" % (sys.argv[1], d["value"], d["roofline"]["kernel_ms"], d["roofline"].get("kernel")))
PY
done
