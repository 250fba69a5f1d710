"""HIP-event timing of named kernel launches on the stream they are launched on.

bench.py uses it for the `roofline` object: achieved = algorithmic FLOPs (or bytes) of one
launch / its average duration, measured live inside the timed region."""
import torch


class _Region(object):
    """Context manager of one timed launch (a plain class: contextlib's generator wrapper costs more than the
    two event records it brackets)."""
    __slots__ = ("timer", "name", "flops", "nbytes", "start", "layer")

    def __init__(self, timer, name, flops, nbytes):
        self.timer, self.name, self.flops, self.nbytes = timer, name, flops, nbytes
        self.layer = timer.layer

    def __enter__(self):
        self.start = self.timer.event()
        self.start.record()
        return self

    def __exit__(self, *exc):
        end = self.timer.event()
        end.record()
        rec = (self.start, end, self.flops, self.nbytes)
        self.timer.records.setdefault(self.name, []).append(rec)
        if self.layer is not None:       # the same two events also feed the per-layer view
            self.timer.records.setdefault("layer/" + self.layer, []).append(rec)
        return False


class _NoRegion(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_REGION = _NoRegion()


class KernelTimer(object):
    def __init__(self):
        self.enabled = False      # build-time switch (engine.build_training_step, ODW_NO_TIMER)
        self.active = True        # per-step switch: bench.py times 1 step in `sample_every` (the event pairs cost
        self.layer = None         # ~5 % of the step when every launch of every step carries them)
        self.tally = None         # a float while a counting pass runs (region() adds its FLOPs and records nothing)
        self.reset()

    def reset(self, prealloc=0):
        """prealloc: HIP events created (and recorded once, which is what creates them) ahead of the timed region -- a
        timed step brackets ~110 launches, and creating its 220 events on the fly cost it about half a millisecond."""
        self.records = {}       # name -> list of (start_event, end_event, flops, bytes)
        self.timed_steps = 0    # steps whose launches carried events (set by the caller: flops_per_timed_step)
        self.pool = []
        if prealloc and torch.cuda.is_available():
            for _ in range(int(prealloc)):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                self.pool.append(e)
            torch.cuda.synchronize()

    def event(self):
        return self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)

    def region(self, name, flops=0.0, nbytes=0.0):
        """with kernel_timer.region(symbol, flops=...): <one launch on torch's current stream>"""
        if self.tally is not None:       # counting pass (a HIP-graph capture's warm-up): FLOPs only, no events
            self.tally += flops
            return _NO_REGION
        if not self.enabled or not self.active or name is None:
            return _NO_REGION
        return _Region(self, name, flops, nbytes)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [s.elapsed_time(e) for s, e, _, _ in recs]
            out[name] = dict(launches=len(recs), avg_ms=sum(ms) / len(ms), total_ms=sum(ms),
                             flops=sum(r[2] for r in recs) / len(recs), bytes=sum(r[3] for r in recs) / len(recs))
        return out

    def hbm_entries(self, hbm_peak_gbps):
        """Regions that carry algorithmic BYTES (the HBM-bound kernels of the step: ROI pooling forward / backward):
        name -> average launch time, bytes, achieved GB/s and fraction of the HBM roofline."""
        out = {}
        for name, v in self.summary().items():
            if v["bytes"] > 0 and not name.startswith("layer/"):
                gbps = v["bytes"] / (v["avg_ms"] * 1e-3) / 1e9
                out[name] = {"bound": "hbm", "avg_launch_us": round(v["avg_ms"] * 1e3, 2), "launches": v["launches"],
                             "algorithmic_bytes": int(v["bytes"]), "achieved_GBps": round(gbps, 1),
                             "peak_GBps": hbm_peak_gbps, "frac": round(gbps / hbm_peak_gbps, 4)}
        return out

    def flops_per_timed_step(self):
        """MFMA work issued per step: the sum of 2MNK over every timed GEMM / convolution launch divided by the number
        of steps that carried events (regions are recorded once per launch; "layer/" entries are views of the same)."""
        total, steps = 0.0, 0
        for name, recs in self.records.items():
            if name.startswith("layer/"):
                continue
            total += sum(r[2] for r in recs)
        steps = getattr(self, "timed_steps", 0)
        return total / steps if steps else None

    def roofline(self, dtype, mfma_peaks, hbm_peak_gbps, dominant=None):
        """The `roofline` object of bench.py for the dominant KERNEL SYMBOL (keys not starting with
        "layer/"): achieved = algorithmic FLOPs of its launches / their measured duration."""
        summ = self.summary()
        kern = {k: v for k, v in summ.items() if not k.startswith("layer/")}
        if not kern:
            return None
        # a region named "<symbol> split-K+reduce" brackets two kernels (the GEMM and its reduction pass): it is
        # reported with the others but it is not a kernel symbol, so it cannot be "the dominant kernel"
        single = {k: v for k, v in kern.items() if " " not in k.split(">")[-1] and v["flops"] > 0} or kern
        name = dominant if dominant in kern else max(single, key=lambda k: single[k]["total_ms"])
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
