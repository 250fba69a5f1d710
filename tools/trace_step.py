"""One steady-state step of the bench as a kernel sequence: start offset, duration, idle gap before (us), HIP queue.
usage: trace_step.py <kernel_trace.csv> <bench_line.txt>   (the step = from one sgd_kernel triple to the next)"""
import csv, json, sys
trace, line = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(trace)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
queues = {}         # rocprofv3's Queue_Id -> a short letter in order of first use (A = the step's stream)
# step boundaries: the maxpool_fwd-free marker = first nchw_to_nhwc of a forward (begin of backbone forward)
marks = [i for i, k in enumerate(ks) if "nchw_to_nhwc_kernel" in k[2]]
# forward has one nchw_to_nhwc (image), backward another (dfeat): take every second one
starts = marks[::2]
a, b = starts[-4], starts[-2]           # two steps: the boundary between them is in the middle
t0 = ks[a][0]
prev_end = ks[a][0]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    return n[:70]
print("# step of %.3f ms, %d launches" % ((ks[b][0] - t0) / 1e6, b - a))
for s, e, n, q in ks[a:b]:
    ql = queues.setdefault(q, chr(ord("A") + len(queues)))
    print("%9.1f %8.1f %7.1f  %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, max(0, s - prev_end) / 1e3, ql, short(n)))
    prev_end = max(prev_end, e)
