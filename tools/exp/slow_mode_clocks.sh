S="--gpus 1 --steps 300 --warmup 5 --no-cpu-baseline --no-secondary --no-microbench"
for e in "ODW_X=1" "ODW_ACT_STREAM=1 ODW_PRIO=-1,0,-1"; do
  echo "== $e"
  env $e python bench.py $S > /tmp/b.json 2>/dev/null &
  pid=$!
  sleep 45
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | head -6 | tr '\n' ';'; echo; sleep 1; done
  wait $pid
  tail -1 /tmp/b.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
