#!/bin/bash
# alternating A/B of one environment switch on one box: tools/exp/ab_env2.sh "VAR=VALUE" [rounds]
S="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-microbench"
for r in $(seq 1 ${2:-3}); do
  for v in base alt; do
    if [ $v = alt ]; then e="$1"; else e="ODW_AB_BASE=1"; fi
    env $e python bench.py $S 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', $r, d['value'], d['ms_per_step'], 'median', d['median_ms_per_step'], 'host', d['host_ms_per_step'], d['ms_per_step_by_labels'])"
  done
done
