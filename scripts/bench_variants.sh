run() { echo -n "$* : "; env $ENVV timeout -k 10 200 python bench.py --steps 60 --warmup 8 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" || exit 1; }
ENVV="A=1"
run
run --opt static=0 --opt blocks_per_cu=2
run --opt static=0 --opt blocks_per_cu=2 --frames-in-flight 8
run --opt static=75 --opt blocks_per_cu=2 --frames-in-flight 8
run --opt static=0 --opt blocks_per_cu=3 --frames-in-flight 8
run --frames-in-flight 8
ENVV="GPU_MAX_HW_QUEUES=8"
echo "GPU_MAX_HW_QUEUES=8"
run --frames-in-flight 8
run --opt static=0 --opt blocks_per_cu=2 --frames-in-flight 8
run --opt static=0 --opt blocks_per_cu=2 --frames-in-flight 12
