#!/bin/bash
# round 6: the driver's command (twice) + the secondary configurations, into gpurun_out/bench_r06/
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/bench_r06
mkdir -p $out
cd $root
python bench.py --gpus 1 --steps 20 --warmup 5 2>$out/driver_cmd.err | tail -1 > $out/bench_driver_cmd.json
python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_driver_cmd_b.json
S="--no-cpu-baseline --no-secondary --no-microbench"
python bench.py $S 2>/dev/null | tail -1 > $out/bench_default_50.json
python bench.py $S --first-image 0 --rotate 1 2>/dev/null | tail -1 > $out/bench_img0.json
python bench.py $S --first-image 2 --rotate 1 2>/dev/null | tail -1 > $out/bench_img2.json
ODW_HOST_LISTS=1 python bench.py $S --steps 20 2>/dev/null | tail -1 > $out/bench_host_lists.json
python bench.py $S --proposals 500 --size 300 2>/dev/null | tail -1 > $out/bench_c1.json
python bench.py $S --proposals 4000 --size 800 --classes 81 2>/dev/null | tail -1 > $out/bench_c4.json
python bench.py $S --arch r50 2>/dev/null | tail -1 > $out/bench_r50.json
python bench.py $S --global-batch 8 --steps 8 --warmup 3 2>$out/b8.err | tail -1 > $out/bench_globalbatch8.json
python bench.py $S --pooler ROIAlign 2>/dev/null | tail -1 > $out/bench_roialign.json
for f in $out/*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], d["dtype"], d["value"], d["ms_per_step"], "median", d.get("median_ms_per_step"), "max", max(d["per_step_ms"]), "host", d.get("host_ms_per_step"), "by labels", d.get("ms_per_step_by_labels"), "allocs", d.get("device_allocs_in_timed_region"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
