#!/bin/bash
# alternating A/B on one box: device-resident lists (default) / without the eager contrastive backward / host lists
S="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-microbench"
for r in 1 2 3; do
  for v in dev noeager host; do
    case $v in
      dev) env="" ;;
      noeager) env="ODW_NO_EAGER_CONTRA=1" ;;
      host) env="ODW_HOST_LISTS=1" ;;
    esac
    env $env python bench.py $S 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', $r, d['value'], d['ms_per_step'], 'median', d['median_ms_per_step'], 'host', d['host_ms_per_step'], d['ms_per_step_by_labels'])"
  done
done
