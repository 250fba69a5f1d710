mkdir -p gpurun_out/r05c
python -m pytest tests/test_gemm_gpu.py tests/test_roi_pool_pin.py tests/test_pair_gpu.py -m gpu -x -q > gpurun_out/r05c/t1.log 2>&1; tail -4 gpurun_out/r05c/t1.log
python -m pytest tests/test_trajectory_gpu.py -m gpu -x -q -s > gpurun_out/r05c/traj.log 2>&1; tail -6 gpurun_out/r05c/traj.log
python -m pytest tests/test_e2e_gpu.py -m gpu -x -q > gpurun_out/r05c/e2e.log 2>&1; tail -4 gpurun_out/r05c/e2e.log
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r05c/bench.json 2> gpurun_out/r05c/bench.err; tail -c 300 gpurun_out/r05c/bench.err
python bench.py --no-cpu-baseline --no-secondary --no-microbench --first-image 2 --rotate 1 > gpurun_out/r05c/bench_img2.json 2>/dev/null
