#!/bin/bash
# rocprofv3 kernel-trace of a bench run.  Writes into gpurun_out/<name>/:
#   bench_kernel_stats.csv   rocprofv3 --stats summary (whole process, incl. warm-up / MIOpen find)
#   steady_state.csv         per-kernel summary restricted to the timed region (tools/trace_summary.py)
#   wall.csv                 wall-time attribution per kernel with overlap across streams (tools/trace_wall.py)
#   bench_line.txt           the bench's JSON line
# usage: tools/prof.sh <name> <bench args...>
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o bench -- python $root/bench.py "$@" > /tmp/prof_$name.log 2>&1
mkdir -p $root/gpurun_out/$name
find /tmp/prof_$name -name "*kernel_stats.csv" -exec cp {} $root/gpurun_out/$name/ \;
grep '^{"metric"' /tmp/prof_$name.log > $root/gpurun_out/$name/bench_line.txt
trace=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1)
python $root/tools/trace_summary.py $trace $root/gpurun_out/$name/bench_line.txt > $root/gpurun_out/$name/steady_state.csv
python $root/tools/trace_step.py $trace $root/gpurun_out/$name/bench_line.txt > $root/gpurun_out/$name/one_step.txt
python $root/tools/trace_gaps.py $trace $root/gpurun_out/$name/bench_line.txt > $root/gpurun_out/$name/gaps.csv
python $root/tools/trace_wall.py $trace $root/gpurun_out/$name/bench_line.txt > $root/gpurun_out/$name/wall.csv
head -45 $root/gpurun_out/$name/steady_state.csv | cut -c1-200
