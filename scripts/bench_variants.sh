#!/bin/bash
# bench.py under different settings (frames/s, ms per step, launch alone).  usage: bash scripts/bench_variants.sh
run() { echo -n "$* : "; timeout -k 10 200 python bench.py --steps 96 --warmup 16 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],3))" || exit 1; }
run
run --opt list_cap=8
run --opt list_cap=12
run --opt rel=0
run --opt lds_kb=40
run
run --opt list_cap=8
