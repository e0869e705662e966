#!/bin/bash
# bench.py under different settings (frames/s, ms per step, launch alone).  usage: bash scripts/bench_variants.sh
run() { echo -n "$* : "; timeout -k 10 200 python bench.py --steps 96 --warmup 16 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],3))" || exit 1; }
run
run --opt item_px=256
run --opt item_px=256 --opt tile_w=8
run --opt item_px=256 --opt tile_w=32
run --opt item_px=256 --opt blocks_per_cu=3
run --opt item_px=256 --opt blocks_per_cu=1
run --opt item_px=256 --frames-in-flight 16
run --opt item_px=256 --opt static=75
