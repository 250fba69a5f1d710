"""Summarise a rocprofv3 kernel trace over the bench's timed region only.

usage: trace_summary.py <kernel_trace.csv> <bench_line.txt>
The timed region = the last steps*ms_per_step milliseconds before the last kernel of the run
(bench.py prints steps and ms_per_step); warm-up, MIOpen find-mode kernels and the CPU baseline
are outside it.  Output CSV: kernel, calls/step, avg_us, total_ms/step, share."""
import csv
import json
import sys

trace, line = sys.argv[1], sys.argv[2]
info = json.loads(open(line).read().strip().splitlines()[-1])
steps, ms = info["steps"], info["ms_per_step"]
rows = list(csv.DictReader(open(trace)))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ks.sort()
t_end = max(k[1] for k in ks)
t0 = t_end - steps * ms * 1e6
agg = {}
for s, e, n in ks:
    if s >= t0:
        a = agg.setdefault(n, [0, 0])
        a[0] += 1
        a[1] += e - s
tot = sum(a[1] for a in agg.values())
print("# timed region: %d steps x %.3f ms; GPU busy %.3f ms/step (%.1f%% of wall); %d launches/step"
      % (steps, ms, tot / 1e6 / steps, 100.0 * tot / 1e6 / steps / ms, sum(a[0] for a in agg.values()) / steps))
print("kernel,calls_per_step,avg_us,ms_per_step,share")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('"%s",%.2f,%.2f,%.4f,%.4f' % (n[:150].replace('"', "'"), c / steps, t / c / 1e3, t / 1e6 / steps, t / tot))
