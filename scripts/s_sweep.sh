#!/bin/bash
# frames/s against the number of frames in flight (bench.py's in-flight options: one persistent workgroup per CU and launch)
for s in "$@"; do
  python bench.py --frames-in-flight $s --steps 60 --warmup 12 --no-cpu-baseline --tail-ramp 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S=%-3s' % '$s', round(d['value'],1), 'frames/s', round(d['ms_per_step'],4), 'ms/frame')"
done
