#!/bin/bash
# usage: tools/exp/ab_env.sh ROUNDS "ENV1=.." "ENV2=.." ...   -- alternating short bench runs on one box
rounds=$1; shift
S="--no-cpu-baseline --no-secondary --no-microbench --steps 30 --warmup 6"
for r in $(seq $rounds); do
for v in "$@"; do
  echo "$v: $(env $v python bench.py $S 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["median_ms_per_step"], d["value"])')"
done; done
