mkdir -p gpurun_out/r05e
python -m pytest tests/test_conv_planes2_gpu.py -m gpu -x -q 2>&1 | tail -3
for lr in 5e-6 1e-5; do
ODW_TRAJ_REPORT=1 ODW_TRAJ_LR=$lr python -m pytest tests/test_trajectory_gpu.py -m gpu -x -q -s > gpurun_out/r05e/traj_$lr.log 2>&1; grep -A1 "TRAJ summary" gpurun_out/r05e/traj_$lr.log | cut -c1-600
done
