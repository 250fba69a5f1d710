#!/bin/bash
# one measurement round of pairwise_sim_panel_kernel on the GPU box: correctness tests, phase timeline, rocprofv3 durations
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "pairwise or sim" 2>&1 | tail -3
tools/exp/pairwise_timeline.bin 4000 | head -21
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pw
rocprofv3 --kernel-trace --output-format csv -d /tmp/pw -o pw -- python $root/tools/pairwise_bench.py > /tmp/pw.log 2>&1
tail -4 /tmp/pw.log
python $root/tools/kernel_times.py $(find /tmp/pw -name "*kernel_trace.csv") pairwise_sim_panel 55
