mkdir -p gpurun_out/full
(time python -m pytest tests/ -x -q -m gpu) > gpurun_out/full/pytest.log 2>&1; tail -15 gpurun_out/full/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full/smoke.log 2>&1; tail -2 gpurun_out/full/smoke.log
