mkdir -p gpurun_out/r05final
python bench.py > gpurun_out/r05final/bench_default.json 2> gpurun_out/r05final/bench_default.err
bash tools/prof.sh r05final_prof --no-secondary --no-microbench --no-cpu-baseline > gpurun_out/r05final/prof.log 2>&1
python bench.py --no-cpu-baseline --no-secondary --no-microbench --first-image 0 --rotate 1 > gpurun_out/r05final/bench_img0.json 2>/dev/null
python bench.py --no-cpu-baseline --no-secondary --no-microbench --first-image 2 --rotate 1 > gpurun_out/r05final/bench_img2.json 2>/dev/null
python bench.py --no-cpu-baseline --no-secondary --no-microbench --proposals 4000 --classes 81 --size 800 > gpurun_out/r05final/bench_c4.json 2>/dev/null
python bench.py --no-cpu-baseline --no-secondary --no-microbench --proposals 500 --size 300 > gpurun_out/r05final/bench_c1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-secondary --no-microbench --arch r50 > gpurun_out/r05final/bench_r50.json 2>/dev/null
python bench.py --no-cpu-baseline --no-secondary --no-microbench --pooler ROIAlign > gpurun_out/r05final/bench_roialign.json 2>/dev/null
python bench.py --no-cpu-baseline --no-secondary --no-microbench --global-batch 8 > gpurun_out/r05final/bench_globalbatch8.json 2>/dev/null
python tools/microbench.py > gpurun_out/r05final/microbench_ops_r05.json 2>/dev/null
ls -la gpurun_out/r05final
