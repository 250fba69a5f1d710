"""Wall-time attribution of the bench's timed region, per kernel, with overlap across streams taken into account.

usage: trace_wall.py <kernel_trace.csv> <bench_line.txt>

The GPU runs kernels of several HIP streams at once (the step's stream, the side stream of the head's update and of the shadow
refresh, the branches of the captured body graphs), so the per-kernel SUM of durations (trace_summary.py) exceeds the wall time
and says nothing about which kernel a faster version of would shorten the step.  This tool cuts the timed region at every kernel
start / end and attributes each slice of wall time
  * `alone_ms`   to the one kernel running in it,
  * `shared_ms`  in equal parts to the kernels running in it (two or more),
  * idle slices to "(idle)".
Per kernel: calls/step, sum of durations, alone, shared, per step.  `alone + shared` over all kernels + idle = the wall time of a
step.  A kernel whose time is mostly `shared` is (at least partly) hidden under another stream's work.
Second table: the same per HIP queue (rocprofv3's Queue_Id), to see which stream paces the step."""
import csv
import json
import sys

trace, line = sys.argv[1], sys.argv[2]
info = json.loads(open(line).read().strip().splitlines()[-1])
steps, ms = info["steps"], info["ms_per_step"]
rows = list(csv.DictReader(open(trace)))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
t_end = max(k[1] for k in ks)
t0 = t_end - steps * ms * 1e6
ks = [k for k in ks if k[1] > t0]
ev = []
for i, (s, e, n, q) in enumerate(ks):
    ev.append((max(s, t0), 1, i))
    ev.append((e, 0, i))
ev.sort()
alone, shared, total, calls = {}, {}, {}, {}
qalone, qshared = {}, {}
running = set()
idle = 0.0
prev = t0
for t, kind, i in ev:
    dt = t - prev
    if dt > 0:
        if not running:
            idle += dt
        elif len(running) == 1:
            j = next(iter(running))
            alone[ks[j][2]] = alone.get(ks[j][2], 0.0) + dt
            qalone[ks[j][3]] = qalone.get(ks[j][3], 0.0) + dt
        else:
            for j in running:
                shared[ks[j][2]] = shared.get(ks[j][2], 0.0) + dt / len(running)
                qshared[ks[j][3]] = qshared.get(ks[j][3], 0.0) + dt / len(running)
    prev = t
    if kind:
        running.add(i)
    else:
        running.discard(i)
for s, e, n, q in ks:
    total[n] = total.get(n, 0.0) + (e - max(s, t0))
    calls[n] = calls.get(n, 0) + 1
f = 1e6 * steps
wall = (t_end - t0) / f
print("# timed region %d steps x %.3f ms; idle %.3f ms/step; alone %.3f; shared %.3f"
      % (steps, ms, idle / f, sum(alone.values()) / f, sum(shared.values()) / f))
print("kernel,calls_per_step,sum_ms,alone_ms,shared_ms,wall_ms")
names = sorted(total, key=lambda n: -(alone.get(n, 0.0) + shared.get(n, 0.0)))
for n in names:
    a, sh = alone.get(n, 0.0) / f, shared.get(n, 0.0) / f
    print('"%s",%.2f,%.4f,%.4f,%.4f,%.4f' % (n[:110].replace('"', "'"), calls[n] / steps, total[n] / f, a, sh, a + sh))
print("# per queue: queue, alone_ms, shared_ms")
for q in sorted(set(qalone) | set(qshared)):
    print("# queue %s  %.3f  %.3f" % (q, qalone.get(q, 0.0) / f, qshared.get(q, 0.0) / f))
