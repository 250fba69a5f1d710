S="--gpus 1 --warmup 5 --no-cpu-baseline --no-secondary --no-microbench --steps 1200"
ODW_ACT_STREAM=1 ODW_PRIO=-1,0,-1 python bench.py $S > /tmp/b.json 2>/dev/null &
pid=$!
for i in $(seq 1 70); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -E 's/.*sclk clock level: [^(]*\(([0-9]+)Mhz\).*/sclk \1/; s/.*Power \(W\): ([0-9.]+).*/W \1/' | tr '\n' ' '; echo
  sleep 1
  kill -0 $pid 2>/dev/null || break
done
wait $pid
tail -1 /tmp/b.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('slow mode', d['value'], d['ms_per_step'])"
