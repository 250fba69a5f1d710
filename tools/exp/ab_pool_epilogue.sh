mkdir -p gpurun_out/poolep
python -m pytest tests/test_conv_planes2_gpu.py tests/test_conv_gpu.py -m gpu -x -q > gpurun_out/poolep/t.log 2>&1; tail -5 gpurun_out/poolep/t.log
python -m pytest tests/test_e2e_gpu.py tests/test_timed_step_gpu.py -m gpu -x -q > gpurun_out/poolep/e2e.log 2>&1; tail -3 gpurun_out/poolep/e2e.log
B="python bench.py --no-cpu-baseline --no-secondary --no-microbench"
for k in a b; do
$B > gpurun_out/poolep/on_$k.json 2>/dev/null
ODW_CONV_POOL_EPILOGUE=0 $B > gpurun_out/poolep/off_$k.json 2>/dev/null
done
for f in on_a off_a on_b off_b; do python -c "
import json
d=json.loads(open('gpurun_out/poolep/$f.json').read().strip().splitlines()[-1]); r=d['roofline']['graph_regions']
print('$f', d['value'], d['ms_per_step'], d['median_ms_per_step'], r['VGG body forward (HIP graph)']['avg_launch_ms'])"; done
