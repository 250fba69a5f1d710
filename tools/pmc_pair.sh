#!/bin/bash
# PMC counters of the standalone shared clean + DropBlock fc6 forward (gemm_nt_cm_kernel) (separate passes; no tracing domains besides kernel-trace)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE" ; do
  rm -rf /tmp/pmc_out
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_out -o g -- python $root/tools/pair_one.py > /dev/null 2>&1
  f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for r in rows:
    if "gemm_nt_cm" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in agg: print("%-28s %16.0f  (per launch, %d launches)" % (k, agg[k] / max(n[k], 1), n[k]))
PY
done
