#!/bin/bash
# alternating runs of several environment settings on one box: tools/exp/ab_multi2.sh rounds "A=1" "B=2 C=3" ...   ("-" = no setting)
S="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-microbench"
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for e in "$@"; do
    if [ "$e" = "-" ]; then ee="ODW_AB_BASE=1"; else ee="$e"; fi
    env $ee python bench.py $S 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-40s' % '$e', $r, d['value'], d['ms_per_step'], 'median', d['median_ms_per_step'], d['ms_per_step_by_labels'])"
  done
done
