S="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-microbench"
for r in 1 2 3; do for te in 13 1000 7; do python bench.py $S --time-every $te 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('time-every $te', d['value'], d['ms_per_step'], d['median_ms_per_step'], [round(x,2) for x in d['per_step_ms'][:3]], round(d['per_step_ms'][13],2), round(d['per_step_ms'][14],2))"; done; done
