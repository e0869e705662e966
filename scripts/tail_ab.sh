#!/bin/bash
# C3 driver configuration (--steps 20 --warmup 5) under bench.py argument sets.  usage: scripts/tail_ab.sh "<args>" ...
[ $# -eq 0 ] && set -- "" "--tail-ramp 0"
for a in "$@"; do
  for rep in 1 2; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s' % '$a', round(d['value'],1), round(d['ms_per_step'],4))"
  done
done
