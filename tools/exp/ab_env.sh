# A/B of one environment switch on the default bench, alternating runs on one box: usage ab_env.sh NAME=VALUE [runs]
sw=$1; n=${2:-2}
mkdir -p gpurun_out/ab
B="python bench.py --no-cpu-baseline --no-secondary --no-microbench --steps 40"
for i in $(seq 1 $n); do
$B > gpurun_out/ab/on_$i.json 2>/dev/null
env $sw $B > gpurun_out/ab/off_$i.json 2>/dev/null
done
for i in $(seq 1 $n); do for f in on_$i off_$i; do python -c "
import json
d=json.loads(open('gpurun_out/ab/$f.json').read().strip().splitlines()[-1])
print('$f (off = $sw)', d['value'], d['ms_per_step'], 'median', d['median_ms_per_step'])"; done; done
