mkdir -p gpurun_out/abseg
B="python bench.py --no-cpu-baseline --no-secondary --no-microbench --first-image 0 --rotate 1"
$B > gpurun_out/abseg/seg8_a.json 2>/dev/null
ODW_CONV_PLANES2=0 $B > gpurun_out/abseg/seg8_p2off_a.json 2>/dev/null
$B > gpurun_out/abseg/seg8_b.json 2>/dev/null
ODW_CONV_PLANES2=0 $B > gpurun_out/abseg/seg8_p2off_b.json 2>/dev/null
touch od_wscl_amd/csrc/gemm_bf16.hip
ODW_EXTRA_FLAGS=-DODW_MAX_SEG=4 python -c "from od_wscl_amd import _build; _build.build()"
$B > gpurun_out/abseg/seg4_a.json 2>/dev/null
$B > gpurun_out/abseg/seg4_b.json 2>/dev/null
for f in seg8_a seg8_b seg8_p2off_a seg8_p2off_b seg4_a seg4_b; do python -c "
import json
d=json.loads(open('gpurun_out/abseg/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['median_ms_per_step'])"; done
