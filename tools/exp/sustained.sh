#!/bin/bash
# step time against the length of the timed region (sustained clocks), with the clocks sampled during the longest run
S="--gpus 1 --warmup 5 --no-cpu-baseline --no-secondary --no-microbench"
for n in 20 100 300 20; do
  python bench.py $S --steps $n 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['per_step_ms']; print('steps $n', d['value'], d['ms_per_step'], 'first 10', round(sum(p[:10])/10,3), 'last 10', round(sum(p[-10:])/10,3))"
done
python bench.py $S --steps 2000 > /tmp/b.json 2>/dev/null &
pid=$!
for i in $(seq 1 60); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -E 's/.*sclk clock level: [^(]*\(([0-9]+)Mhz\).*/sclk \1/; s/.*Power \(W\): ([0-9.]+).*/W \1/' | tr '\n' ' '; echo
  sleep 1
  kill -0 $pid 2>/dev/null || break
done
wait $pid
tail -1 /tmp/b.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['per_step_ms']; print('steps 2000', d['value'], d['ms_per_step'], 'first 10', round(sum(p[:10])/10,3), 'steps 100-110', round(sum(p[100:110])/10,3), 'last 10', round(sum(p[-10:])/10,3))"
