"""HIP-event timing of named kernel launches on the stream they are launched on.

bench.py uses it for the `roofline` object: achieved = algorithmic FLOPs (or bytes) of one
launch / its average duration, measured live inside the timed region."""
import contextlib

import torch


class KernelTimer(object):
    def __init__(self):
        self.enabled = False
        self.reset()

    def reset(self):
        self.records = {}       # name -> list of (start_event, end_event, flops, bytes)

    @contextlib.contextmanager
    def region(self, name, flops=0.0, nbytes=0.0):
        if not self.enabled or name is None:
            yield
            return
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record(torch.cuda.current_stream())
        try:
            yield
        finally:
            e.record(torch.cuda.current_stream())
            self.records.setdefault(name, []).append((s, e, flops, nbytes))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [s.elapsed_time(e) for s, e, _, _ in recs]
            out[name] = dict(launches=len(recs), avg_ms=sum(ms) / len(ms), total_ms=sum(ms),
                             flops=sum(r[2] for r in recs) / len(recs), bytes=sum(r[3] for r in recs) / len(recs))
        return out

    def roofline(self, dtype, mfma_peaks, hbm_peak_gbps, dominant=None):
        summ = self.summary()
        if not summ:
            return None
        name = dominant or max(summ, key=lambda k: summ[k]["total_ms"])
        r = summ[name]
        if r["flops"] > 0:
            achieved = r["flops"] / (r["avg_ms"] * 1e-3) / 1e12
            peak = mfma_peaks[dtype]
            return {"kernel": name, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
                    "avg_launch_ms": round(r["avg_ms"], 4), "launches": r["launches"],
                    "flops_per_launch": r["flops"],
                    "others": {k: round(v["total_ms"], 3) for k, v in summ.items() if k != name}}
        achieved = r["bytes"] / (r["avg_ms"] * 1e-3) / 1e9
        return {"kernel": name, "bound": "hbm", "achieved": round(achieved, 1), "peak": hbm_peak_gbps,
                "unit": "GB/s", "frac": round(achieved / hbm_peak_gbps, 4), "traffic": None,
                "avg_launch_ms": round(r["avg_ms"], 4), "launches": r["launches"], "bytes_per_launch": r["bytes"]}


kernel_timer = KernelTimer()
