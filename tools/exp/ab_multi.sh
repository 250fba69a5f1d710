#!/bin/bash
# several environment variants, alternating, on one box: tools/exp/ab_multi.sh rounds "A=1 B=2" "C=3" ...
S="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-microbench"
R=$1; shift
for r in $(seq 1 $R); do
  for e in "$@"; do
    env $e python bench.py $S 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-46s' % '$e', $r, d['value'], d['ms_per_step'], 'median', d['median_ms_per_step'], 'host', d['host_ms_per_step'], d['ms_per_step_by_labels'], 'allocs', d['device_allocs_in_timed_region'])"
  done
done
