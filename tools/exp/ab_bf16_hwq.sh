S="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-microbench --dtype bf16"
for r in 1 2 3; do for q in 2 4; do GPU_MAX_HW_QUEUES=$q python bench.py $S 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bf16 Q=$q', d['value'], d['ms_per_step'], d['median_ms_per_step'])"; done; done
