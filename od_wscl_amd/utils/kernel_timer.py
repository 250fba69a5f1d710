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
        """The `roofline` object of bench.py for the dominant KERNEL SYMBOL (keys not starting with
        "layer/"): achieved = algorithmic FLOPs of its launches / their measured duration."""
        summ = self.summary()
        kern = {k: v for k, v in summ.items() if not k.startswith("layer/")}
        if not kern:
            return None
        name = dominant if dominant in kern else max(kern, key=lambda k: kern[k]["total_ms"])
        r = kern[name]
        layers = {k[6:]: {"ms_per_launch": round(v["avg_ms"], 4), "TFLOP/s": round(v["flops"] / (v["avg_ms"] * 1e-3) / 1e12, 1),
                          "launches": v["launches"]} for k, v in summ.items() if k.startswith("layer/")}
        achieved = r["flops"] / (r["avg_ms"] * 1e-3) / 1e12
        peak = mfma_peaks[dtype]
        return {"kernel": name, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": None, "avg_launch_ms": round(r["avg_ms"], 4),
                "launches": r["launches"], "flops_per_launch": r["flops"],
                "other_kernels_ms": {k: round(v["total_ms"], 3) for k, v in kern.items() if k != name},
                "layers": layers}


kernel_timer = KernelTimer()
