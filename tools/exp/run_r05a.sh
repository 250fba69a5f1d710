mkdir -p gpurun_out/r05a
python bench.py > gpurun_out/r05a/bench_default.json 2> gpurun_out/r05a/bench_default.err
tail -c 600 gpurun_out/r05a/bench_default.err
bash tools/prof.sh r05a_prof --no-secondary --no-microbench --no-cpu-baseline > gpurun_out/r05a/prof.log 2>&1
tail -5 gpurun_out/r05a/prof.log
