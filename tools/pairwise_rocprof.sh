#!/bin/bash
# rocprofv3 kernel durations of pairwise_sim (the operator _C.pairwise_sim runs: odw_pairwise_sim_ws) per P, next to the
# graph-replayed HIP-event figure bench.py reports (roofline.kernels) -- the two must agree.
#   tools/pairwise_rocprof.sh > gpurun_out/pairwise_r04_times.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/pw_prof.py <<PY
import sys, torch
sys.path.insert(0, "$root")
from od_wscl_amd import _lib as L
lib = L.lib()
for P in (2000, 4000, 8000):
    E = torch.nn.functional.normalize(torch.randn(P, 128, device="cuda"), dim=1).contiguous()
    S = torch.empty(P, P, device="cuda")
    wsb = lib.odw_pairwise_sim_workspace(P, 128)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    g = torch.cuda.CUDAGraph()
    f = lambda: L.check(lib.odw_pairwise_sim_ws(L.ptr(E), P, 128, L.ptr(S), L.ptr(ws), wsb, L.stream()), "ps")
    for _ in range(5): f()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(30): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    print("EVENTS P=%d %.2f us per launch (graph of 30, HIP events)" % (P, a.elapsed_time(b) / 30 * 1e3), flush=True)
PY
rm -rf /tmp/pw_prof_out
rocprofv3 --kernel-trace --output-format csv -d /tmp/pw_prof_out -o t -- python /tmp/pw_prof.py 2>/dev/null | grep EVENTS
python - <<'PY'
import csv, glob, statistics
f = glob.glob("/tmp/pw_prof_out/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pairwise_sim" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
g = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3 for i in range(len(rows) - 1)]
per = len(d) // 3                      # 5 + 30 + 30 + 30 launches per P
for k, P in enumerate((2000, 4000, 8000)):
    seg = d[k * per:(k + 1) * per][-30:]
    gaps = g[k * per:(k + 1) * per - 1][-29:]
    print("ROCPROF P=%d kernel %s: median %.2f us  min %.2f  max %.2f   gap to the next launch inside the graph: median %.2f us"
          % (P, __import__("re").search(r"pairwise_sim_\w+", rows[k * per]["Kernel_Name"]).group(0), statistics.median(seg), min(seg), max(seg), statistics.median(gaps)))
PY
