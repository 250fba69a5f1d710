mkdir -p gpurun_out/r05f
(time python -m pytest tests/test_trajectory_gpu.py -m gpu -x -q -s) > gpurun_out/r05f/traj.log 2>&1; grep -A1 "TRAJ summary\|passed\|failed\|real" gpurun_out/r05f/traj.log | cut -c1-400
python -m pytest tests/test_conv_gpu.py tests/test_split_gpu.py tests/test_conv_planes2_gpu.py tests/test_timed_step_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q > gpurun_out/r05f/t1.log 2>&1; tail -4 gpurun_out/r05f/t1.log
bash tools/prof.sh r05f_prof --no-secondary --no-microbench --no-cpu-baseline > gpurun_out/r05f/prof.log 2>&1
ODW_PREP_ASYNC=0 python bench.py --no-cpu-baseline --no-secondary --no-microbench > gpurun_out/r05f/bench_prep0.json 2>/dev/null
python bench.py --no-cpu-baseline --no-secondary --no-microbench > gpurun_out/r05f/bench_prep1.json 2>/dev/null
ODW_PREP_ASYNC=0 python bench.py --no-cpu-baseline --no-secondary --no-microbench > gpurun_out/r05f/bench_prep0_b.json 2>/dev/null
python bench.py --no-cpu-baseline --no-secondary --no-microbench > gpurun_out/r05f/bench_prep1_b.json 2>/dev/null
