#!/bin/bash
# conv3x3_halo_kernel: kernel time against K (input channels) at a fixed output shape -> fixed cost per launch + cost per K step
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for layer in "$@"; do for cin in 64 128 256 512 1024 1536; do
  rm -rf /tmp/o
  ODW_CONV_LAYER=$layer ODW_CONV_CIN=$cin rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/o -o t -- python $root/tools/conv_one.py > /dev/null 2>&1
  python - $layer $cin <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/o/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "halo" in r["Name"] or "splitk" in r["Name"]:
        print("%s cin=%s  %-28s calls %s  avg %.1f us" % (sys.argv[1], sys.argv[2], r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done; done
