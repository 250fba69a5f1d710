#!/bin/bash
# timing experiments on pairwise_sim_panel_kernel (an ODW_EXPERIMENTS build on the GPU box: results are WRONG by design)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
ODW_EXTRA_FLAGS="-DODW_EXPERIMENTS" python -c "from od_wscl_amd import _build; _build.build(force=True)" 
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2 3 4 7 8 15; do
  rm -rf /tmp/pw
  ODW_PAIRWISE_DBG=$dbg rocprofv3 --kernel-trace --output-format csv -d /tmp/pw -o pw -- python $root/tools/pairwise_bench.py > /tmp/pw.log 2>&1
  echo "dbg=$dbg"; python $root/tools/kernel_times.py $(find /tmp/pw -name "*kernel_trace.csv") pairwise_sim_panel 55
done
