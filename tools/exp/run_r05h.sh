mkdir -p gpurun_out/r05h
python tools/bwd2_report.py > gpurun_out/r05h/bwd2_report.txt 2> gpurun_out/r05h/bwd2.err; tail -40 gpurun_out/r05h/bwd2_report.txt; tail -5 gpurun_out/r05h/bwd2.err
