#!/bin/bash
# PMC counters of pairwise_sim_panel_kernel at P = 4000 (separate passes; kernel-trace only besides --pmc)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/pw_one.py <<PY
import sys, torch
sys.path.insert(0, "$root")
from od_wscl_amd import _lib as L
lib = L.lib()
P = 4000
E = torch.nn.functional.normalize(torch.randn(P, 128, device="cuda"), dim=1).contiguous()
S = torch.empty(P, P, device="cuda")
for _ in range(10):
    L.check(lib.odw_pairwise_sim_ws(L.ptr(E), P, 128, L.ptr(S), None, 0, L.stream()), "ps")
torch.cuda.synchronize()
PY
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE" ; do
  rm -rf /tmp/pmc_out
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_out -o g -- python /tmp/pw_one.py > /tmp/pmc.log 2>&1
  f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "set [$set] failed:"; tail -3 /tmp/pmc.log; continue; fi
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for r in rows:
    if "pairwise" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in agg: print("%-28s %16.0f  (per launch, %d launches)" % (k, agg[k] / max(n[k], 1), n[k]))
PY
done
