mkdir -p gpurun_out/ra
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "roi_align" > gpurun_out/ra/t.log 2>&1; tail -5 gpurun_out/ra/t.log
python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "align" > gpurun_out/ra/e2e.log 2>&1; tail -3 gpurun_out/ra/e2e.log
python tools/microbench.py 2>/dev/null | grep roi_align
ODW_ROI_ALIGN_ATOMIC=1 python tools/microbench.py 2>/dev/null | grep roi_align_bwd
