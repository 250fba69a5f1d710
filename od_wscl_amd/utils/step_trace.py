"""Per-step HOST timeline of the training step (bench.py --timeline): where the host thread spends a step's wall time.

A step whose GPU time is 9 ms can take 18 ms when the host blocks -- in a device -> host read, in a hipMalloc of the
caching allocator, in a planner miss that walks the kernel table, in a pinned-ring growth.  `mark(name)` closes the
interval since the previous mark and books it under `name`; counters (`count`) collect the events that are known
causes of a slow step.  Disabled (the default) every call is one attribute test."""
import time

import torch


class StepTrace(object):
    def __init__(self):
        self.enabled = False
        self.gpu_events = False     # also record a HIP event on the current stream at every mark: GPU time between marks
        self.steps = []
        self.cur = None
        self._t = 0.0

    def begin(self, **tags):
        if not self.enabled:
            return
        self.cur = {"tags": tags, "host_ms": {}, "counts": {}, "_ev": []}
        if self.gpu_events:
            self._event("begin")
        st = torch.cuda.memory_stats() if torch.cuda.is_available() else {}
        self.cur["_alloc0"] = (st.get("num_device_alloc", 0), st.get("num_alloc_retries", 0),
                               st.get("reserved_bytes.all.current", 0))
        self._t0 = self._t = time.perf_counter()

    def mark(self, name):
        if self.cur is None:
            return
        t = time.perf_counter()
        h = self.cur["host_ms"]
        h[name] = h.get(name, 0.0) + (t - self._t) * 1e3
        if self.gpu_events:
            self._event(name)
            t = time.perf_counter()
        self._t = t

    def _event(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.cur["_ev"].append((name, e))

    def count(self, name, n=1):
        if self.cur is None:
            return
        c = self.cur["counts"]
        c[name] = c.get(name, 0) + n

    def note(self, name, value):
        if self.cur is not None:
            self.cur["tags"][name] = value

    def end(self):
        if self.cur is None:
            return
        self.mark("rest")
        cur, self.cur = self.cur, None
        cur["host_total_ms"] = (time.perf_counter() - self._t0) * 1e3
        st = torch.cuda.memory_stats() if torch.cuda.is_available() else {}
        a0 = cur.pop("_alloc0")
        cur["counts"]["device_allocs"] = st.get("num_device_alloc", 0) - a0[0]
        cur["counts"]["alloc_retries"] = st.get("num_alloc_retries", 0) - a0[1]
        cur["reserved_mb"] = round(st.get("reserved_bytes.all.current", 0) / 2 ** 20, 1)
        cur["reserved_delta_mb"] = round((st.get("reserved_bytes.all.current", 0) - a0[2]) / 2 ** 20, 1)
        self.steps.append(cur)

    def report(self, per_step_gpu_ms=None, slow=1.25):
        """List of per-step records (rounded), `gpu_ms` attached, plus the indices of the steps slower than `slow` x the median."""
        import numpy as np
        out = []
        for i, s in enumerate(self.steps):
            r = {"step": i, "host_total_ms": round(s["host_total_ms"], 3), "reserved_mb": s["reserved_mb"],
                 "reserved_delta_mb": s["reserved_delta_mb"],
                 "host_ms": {k: round(v, 3) for k, v in s["host_ms"].items()}, "counts": s["counts"]}
            r.update(s["tags"])
            ev = s.get("_ev") or []
            if len(ev) > 1:
                # GPU time between the events recorded at consecutive marks, on the stream the step launches on: the
                # interval named X = the stream's work queued between the previous mark and mark X (plus any wait for it)
                g = {}
                for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
                    g[n1] = round(g.get(n1, 0.0) + e0.elapsed_time(e1), 3)
                r["gpu_between_marks_ms"] = g
            if per_step_gpu_ms is not None and i < len(per_step_gpu_ms):
                r["gpu_ms"] = round(float(per_step_gpu_ms[i]), 3)
            out.append(r)
        slow_steps = []
        if per_step_gpu_ms is not None and len(per_step_gpu_ms):
            med = float(np.median(per_step_gpu_ms))
            slow_steps = [i for i, v in enumerate(per_step_gpu_ms) if v > slow * med]
        return {"steps": out, "slow_steps": slow_steps}


step_trace = StepTrace()
