#!/bin/bash
# The driver's exact bench command, N times on one box, each with a per-step host timeline (bench.py --timeline):
# which steps are slower than 1.25 x the median and what the host / the allocator / the planner did in them.
#   tools/tail_repro.sh [runs] [tag]      ->  gpurun_out/<tag>_run<i>.json (bench line) + <tag>_timeline<i>.json
N=${1:-5}; TAG=${2:-tail}
mkdir -p gpurun_out
for i in $(seq 1 $N); do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-microbench \
      --timeline gpurun_out/${TAG}_timeline$i.json > gpurun_out/${TAG}_run$i.json 2> gpurun_out/${TAG}_err$i.log
  python - <<EOF
import json
b = json.loads(open("gpurun_out/${TAG}_run$i.json").read().strip().splitlines()[-1])
print("run $i: value %.0f  mean %.3f  median %.3f  max %.2f  allocs %s" % (b["value"], b["ms_per_step"], b["median_ms_per_step"], max(b["per_step_ms"]), b["device_allocs_in_timed_region"]))
print("   per_step", b["per_step_ms"])
t = json.load(open("gpurun_out/${TAG}_timeline$i.json"))
for s in t["steps"]:
    if s["step"] in t["slow_steps"]:
        print("   SLOW", json.dumps(s))
EOF
done
