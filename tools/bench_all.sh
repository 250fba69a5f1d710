#!/bin/bash
# every bench line of a round into gpurun_out/<tag>/ (copied to profiles/rNN/ afterwards)
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/bench_$tag
mkdir -p $out
cd $root
python bench.py 2>/dev/null | tail -1 > $out/bench_default.json
S="--no-cpu-baseline --no-secondary --no-microbench"
python bench.py $S --proposals 500 --size 300 2>/dev/null | tail -1 > $out/bench_c1.json
python bench.py $S --proposals 4000 --size 800 --classes 81 2>/dev/null | tail -1 > $out/bench_c4.json
python bench.py $S --arch r50 2>/dev/null | tail -1 > $out/bench_r50.json
python bench.py $S --global-batch 2 2>/dev/null | tail -1 > $out/bench_globalbatch2.json
python bench.py $S --global-batch 8 --steps 8 --warmup 3 2>/dev/null | tail -1 > $out/bench_globalbatch8.json
python bench.py $S --pooler ROIAlign 2>/dev/null | tail -1 > $out/bench_roialign.json
python bench.py $S --dtype bf16x3 --steps 6 --warmup 2 2>/dev/null | tail -1 > $out/bench_bf16x3.json
python bench.py $S --dtype bf16x2 --steps 6 --warmup 2 2>/dev/null | tail -1 > $out/bench_bf16x2.json
for f in $out/*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], d["dtype"], d["value"], d["ms_per_step"], "median", d.get("median_ms_per_step"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
