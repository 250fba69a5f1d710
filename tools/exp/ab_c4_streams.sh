S="--no-cpu-baseline --no-secondary --no-microbench --proposals 4000 --size 800 --classes 81 --steps 30"
for r in 1 2; do for e in "ODW_X=1" "ODW_ACT_STREAM=0" "ODW_ACT_STREAM=0 ODW_PRIO=-1,0,-1" "ODW_ACT_STREAM=1 GPU_MAX_HW_QUEUES=3"; do env $e python bench.py $S 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C4 %-40s' % '$e', d['value'], d['ms_per_step'], d['median_ms_per_step'], d['device_allocs_in_timed_region'], d['roofline']['frac'])"; done; done
