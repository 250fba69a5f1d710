mkdir -p gpurun_out/r05b
python -m pytest tests/test_roi_pool_pin.py -m gpu -x -q > gpurun_out/r05b/pin.log 2>&1; tail -5 gpurun_out/r05b/pin.log
python -m pytest tests/test_trajectory_gpu.py -m gpu -x -q -s > gpurun_out/r05b/traj.log 2>&1; tail -8 gpurun_out/r05b/traj.log
bash tools/prof.sh r05b_img2 --no-secondary --no-microbench --no-cpu-baseline --first-image 2 --rotate 1 > gpurun_out/r05b/prof2.log 2>&1
bash tools/prof.sh r05b_img0 --no-secondary --no-microbench --no-cpu-baseline --first-image 0 --rotate 1 > gpurun_out/r05b/prof0.log 2>&1
