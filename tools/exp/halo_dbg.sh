#!/bin/bash
# timing experiments of conv3x3_halo_kernel (needs a build with ODW_EXTRA_FLAGS=-DODW_EXPERIMENTS; WRONG results by design):
# ODW_HALO_DBG = 0 real, 1 no DMA, 2 no LDS reads, 3 no MFMA, 4 no stores, 5 no weight DMA, 6 no patch DMA
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for layer in "$@"; do for dbg in 0 1 2 3 4 5 6; do
  rm -rf /tmp/o
  ODW_CONV_LAYER=$layer ODW_HALO_DBG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/o -o t -- python $root/tools/conv_one.py > /dev/null 2>&1
  python - $layer $dbg <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/o/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "halo" in r["Name"]:
        print("%s dbg=%s  calls %s  avg %.1f us" % (sys.argv[1], sys.argv[2], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done; done
