#!/bin/bash
# PMC counters of the convolution weight-gradient kernel (separate --pmc passes; kernel-trace only).
# usage: tools/pmc_wgrad.sh <conv3|conv4|conv5> > gpurun_out/pmc_wgrad_<layer>.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export ODW_WGRAD_LAYER=${1:-conv4}
python $root/tools/wgrad_one.py | tail -1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" ; do
  rm -rf /tmp/pmc_out
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_out -o g -- python $root/tools/wgrad_one.py > /dev/null 2>&1
  f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for r in rows:
    k = r["Kernel_Name"]
    if "wgrad_halo" in k or "gemm_tn" in k:
        key = r["Counter_Name"]
        agg[key] += float(r["Counter_Value"]); n[key] += 1
for k in sorted(agg): print("%-36s %16.0f  (per launch, %d launches)" % (k, agg[k] / max(n[k], 1), n[k]))
PY
done
