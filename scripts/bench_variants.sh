#!/bin/bash
# bench.py under different settings (frames/s, ms per step, launch alone).  usage: bash scripts/bench_variants.sh
run() { echo -n "$* : "; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],3))" || exit 1; }
run
run --opt blocks_per_cu=1
run --opt blocks_per_cu=1 --frames-in-flight 16
run --opt blocks_per_cu=1 --frames-in-flight 24
run --opt blocks_per_cu=1 --opt item_px=128 --opt tile_w=8
run --opt blocks_per_cu=1
