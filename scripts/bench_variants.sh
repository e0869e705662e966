#!/bin/bash
# bench.py under different in-flight settings (frames/s, ms per step, launch alone).  usage: bash scripts/bench_variants.sh
run() { echo -n "Q=${GPU_MAX_HW_QUEUES:-default(8 in bench.py)} $* : "; timeout -k 10 200 python bench.py --steps 96 --warmup 16 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],3))" || exit 1; }
run
run --frames-in-flight 12
run --frames-in-flight 16
export GPU_MAX_HW_QUEUES=12
run --frames-in-flight 12
export GPU_MAX_HW_QUEUES=16
run --frames-in-flight 16
run --frames-in-flight 12
export GPU_MAX_HW_QUEUES=4
run --frames-in-flight 12
