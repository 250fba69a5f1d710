S="--no-cpu-baseline --no-secondary --no-microbench --steps 30 --warmup 6"
for r in 1 2 3; do
for v in "X=1" "ODW_STACKED_BATCHED=1" "ODW_NO_OVERLAP=1" "ODW_NO_EARLY_BWD=1"; do
  echo "$v: $(env $v python bench.py $S 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["median_ms_per_step"], d["value"])')"
done; done
