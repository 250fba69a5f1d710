"""Per-launch durations of the kernels matching a substring in a rocprofv3 kernel trace, in launch order, summarised in
`chunk`-sized groups (a microbench that launches the same kernel at several sizes: one group per size).
usage: kernel_times.py <kernel_trace.csv> <substring> <chunk>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
n = int(sys.argv[3])
for i in range(0, len(d), n):
    c = sorted(d[i:i + n])
    print("launches %d-%d: median %.2f us  min %.2f  max %.2f" % (i, i + len(c) - 1, c[len(c) // 2], c[0], c[-1]))
