"""Idle gaps of the GPU inside the bench's timed region, from a rocprofv3 kernel trace.

usage: trace_gaps.py <kernel_trace.csv> <bench_line.txt>
Groups gaps by (kernel before, kernel after); prints per-step totals.  (Under rocprofv3 the host is slower,
so absolute gap sizes are inflated; the ranking shows where the host cannot keep the queue full.)"""
import csv
import json
import sys

trace, line = sys.argv[1], sys.argv[2]
info = json.loads(open(line).read().strip().splitlines()[-1])
steps, ms = info["steps"], info["ms_per_step"]
rows = list(csv.DictReader(open(trace)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t_end = max(k[1] for k in ks)
t0 = t_end - steps * ms * 1e6
ks = [k for k in ks if k[0] >= t0]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    return n[:60]


agg = {}
busy_end = ks[0][1]
tot = 0
for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
    busy_end = max(busy_end, e0)
    gap = s1 - busy_end
    if gap > 0:
        tot += gap
        a = agg.setdefault((short(n0), short(n1)), [0, 0])
        a[0] += 1
        a[1] += gap
print("# idle %.3f ms/step in %d gaps/step" % (tot / 1e6 / steps, sum(a[0] for a in agg.values()) / steps))
print("gap_ms_per_step,count_per_step,avg_us,before -> after")
for (a, b), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%.4f,%.2f,%.1f,%s -> %s" % (t / 1e6 / steps, c / steps, t / c / 1e3, a, b))
