mkdir -p gpurun_out/r05d
python -m pytest tests/test_conv_planes2_gpu.py -m gpu -x -q > gpurun_out/r05d/planes2.log 2>&1; tail -15 gpurun_out/r05d/planes2.log
python -m pytest tests/test_pair_gpu.py tests/test_gemm_gpu.py tests/test_conv_gpu.py tests/test_split_gpu.py -m gpu -x -q > gpurun_out/r05d/t1.log 2>&1; tail -4 gpurun_out/r05d/t1.log
python -m pytest tests/test_e2e_gpu.py -m gpu -x -q > gpurun_out/r05d/e2e.log 2>&1; tail -4 gpurun_out/r05d/e2e.log
ODW_TRAJ_LR=2e-5 python -m pytest tests/test_trajectory_gpu.py -m gpu -x -q -s > gpurun_out/r05d/traj_2e-5.log 2>&1; grep "TRAJ summary" gpurun_out/r05d/traj_2e-5.log
ODW_TRAJ_LR=5e-5 python -m pytest tests/test_trajectory_gpu.py -m gpu -x -q -s > gpurun_out/r05d/traj_5e-5.log 2>&1; grep "TRAJ summary" gpurun_out/r05d/traj_5e-5.log
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r05d/bench.json 2> gpurun_out/r05d/bench.err; tail -c 300 gpurun_out/r05d/bench.err
ODW_CONV_PLANES2=0 python bench.py --no-cpu-baseline --no-secondary --no-microbench > gpurun_out/r05d/bench_p2off.json 2>/dev/null
python bench.py --no-cpu-baseline --no-secondary --no-microbench > gpurun_out/r05d/bench_b.json 2>/dev/null
ODW_CONV_PLANES2=0 python bench.py --no-cpu-baseline --no-secondary --no-microbench > gpurun_out/r05d/bench_p2off_b.json 2>/dev/null
