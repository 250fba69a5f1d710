#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace) of the convolution weight gradient at the bench's layer shapes
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/wg
rocprofv3 --kernel-trace --output-format csv -d /tmp/wg -o wg -- python $root/tools/wgrad_bench.py > /tmp/wg.log 2>&1
grep -o "[0-9 ]*->.*dil [0-9].*x[0-9 ]*\|TN .*" /tmp/wg.log | paste - -
python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/wg/**/*kernel_trace.csv",recursive=True)[0]
d=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "halo" in n or "tn_bf16" in n:
        d.setdefault((n[:50], r.get("Grid_Size_X",""), r.get("Grid_Size_Y","")),[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items():
    v=sorted(v); print(k, len(v), "median %.1f us"%v[len(v)//2])
PY
