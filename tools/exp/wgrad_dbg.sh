#!/bin/bash
# timing experiments on conv_wgrad_halo_kernel (an ODW_EXPERIMENTS build on the GPU box: results are WRONG by design)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
ODW_EXTRA_FLAGS="-DODW_EXPERIMENTS" python -c "from od_wscl_amd import _build; _build.build(force=True)"
for dbg in 0 1 2 4 3 5 6 7; do
  echo "dbg=$dbg"; ODW_WH_DBG=$dbg tools/exp/wgrad_kernels.sh | grep "32768\|131072\|halo" | tail -2
done
