"""HIP-event timing of named kernel launches on the stream they are launched on.

bench.py uses it for the `roofline` object: achieved = algorithmic FLOPs (or bytes) of one
launch / its average duration, measured live inside the timed region."""
import torch


class _Region(object):
    """Context manager of one timed launch (a plain class: contextlib's generator wrapper costs more than the
    two event records it brackets)."""
    __slots__ = ("timer", "name", "flops", "nbytes", "start", "layer", "alg", "shape")

    def __init__(self, timer, name, flops, nbytes, alg, shape):
        self.timer, self.name, self.flops, self.nbytes = timer, name, flops, nbytes
        self.alg, self.shape = (flops if alg is None else alg), shape
        self.layer = timer.layer

    def __enter__(self):
        self.start = self.timer.event()
        self.start.record()
        return self

    def __exit__(self, *exc):
        end = self.timer.event()
        end.record()
        rec = (self.start, end, self.flops, self.nbytes, self.alg, self.shape)
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
        self.tally_alg = 0.0      # the same pass's FLOPs of the REFERENCE's arithmetic (one fp32 product per plane-product group)
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

    def region(self, name, flops=0.0, nbytes=0.0, alg=None, shape=None):
        """with kernel_timer.region(symbol, flops=...): <one launch on torch's current stream>
        flops = MFMA work ISSUED (every plane product counted); alg = the FLOPs of the reference's arithmetic for the same
        result (one fp32 product where a split mode issues three or six; default: = flops); shape = the launch shape as a
        string (the PMC traffic table of bench.py is keyed on it)."""
        if self.tally is not None:       # counting pass (a HIP-graph capture's warm-up): FLOPs only, no events
            if flops > 0:                # (a byte-counted region -- nbytes > 0, flops == 0: its `alg` is BYTES of the reference
                self.tally += flops      # operator, not FLOPs -- must not leak into a graph region's algorithmic FLOPs)
                self.tally_alg += flops if alg is None else alg
            return _NO_REGION
        if not self.enabled or not self.active or name is None:
            return _NO_REGION
        return _Region(self, name, flops, nbytes, alg, shape)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [r[0].elapsed_time(r[1]) for r in recs]
            by_shape = {}
            for r, t in zip(recs, ms):
                if r[5] is not None:
                    by_shape[r[5]] = by_shape.get(r[5], 0.0) + t
            out[name] = dict(launches=len(recs), avg_ms=sum(ms) / len(ms), total_ms=sum(ms),
                             flops=sum(r[2] for r in recs) / len(recs), bytes=sum(r[3] for r in recs) / len(recs),
                             alg=sum(r[4] for r in recs) / len(recs),
                             top_shape=max(by_shape, key=by_shape.get) if by_shape else None)
        return out

    def hbm_entries(self, hbm_peak_gbps):
        """Regions that carry algorithmic BYTES (the HBM-bound kernels of the step: ROI pooling forward / backward):
        name -> average launch time, bytes, achieved GB/s and fraction of the HBM roofline."""
        out = {}
        for name, v in self.summary().items():
            if v["bytes"] > 0 and not name.startswith("layer/"):
                gbps = v["bytes"] / (v["avg_ms"] * 1e-3) / 1e9
                # bytes = what this kernel reads and writes by construction; survey_8d_bytes = the bytes of the reference
                # operator it stands for (fp32 output + int32 argmax + the map once)
                ref = v["alg"] if (v["flops"] == 0 and v["alg"] > 0) else 0.0        # byte regions pass flops = 0: alg = reference bytes
                out[name] = {"bound": "hbm", "avg_launch_us": round(v["avg_ms"] * 1e3, 2), "launches": v["launches"],
                             "algorithmic_bytes": int(v["bytes"]), "achieved_GBps": round(gbps, 1),
                             "peak_GBps": hbm_peak_gbps, "frac": round(gbps / hbm_peak_gbps, 4)}
                if ref > 0:
                    out[name]["survey_8d_bytes"] = int(ref)
                    out[name]["frac_vs_survey_8d"] = round(ref / (v["avg_ms"] * 1e-3) / 1e9 / hbm_peak_gbps, 4)
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
        """The `roofline` object of bench.py for the dominant KERNEL SYMBOL: the symbol with the largest total time over
        ALL its launches of the timed steps, the launches that run as split-K partial products included (their regions
        bracket the GEMM and its short reduction pass: "<symbol> split-K+reduce" -- counted under the symbol, the pass's
        few microseconds with them, which makes the figure slightly pessimistic).
        `frac` = FLOPs of the REFERENCE's arithmetic for those launches (one fp32 product per result) / their measured time /
        peak; `frac_issued` = the MFMA work actually issued (every bf16 plane product counted) on the same scale."""
        summ = self.summary()
        kern = {k: v for k, v in summ.items() if not k.startswith("layer/")}
        if not kern:
            return None

        def symbol_of(name):            # "sym<...> split-K+reduce" -> "sym<...>"
            head, sep, tail = name.rpartition(">")
            return head + sep if sep and tail.startswith(" ") else name

        sym = {}
        for k, v in kern.items():
            if v["flops"] <= 0 or "HIP graph" in k:
                continue            # byte-counted regions (ROI pooling) and whole-graph replays are not kernel symbols
            a = sym.setdefault(symbol_of(k), dict(total_ms=0.0, launches=0, flops=0.0, alg=0.0, split_ms=0.0, shapes={}))
            a["total_ms"] += v["total_ms"]
            a["launches"] += v["launches"]
            a["flops"] += v["flops"] * v["launches"]
            a["alg"] += v["alg"] * v["launches"]
            if k != symbol_of(k):
                a["split_ms"] += v["total_ms"]
        if not sym:
            return None
        for name, recs in self.records.items():      # the launch shape a symbol spends most of its time in
            if name.startswith("layer/") or symbol_of(name) not in sym:
                continue
            sh = sym[symbol_of(name)]["shapes"]
            for r in recs:
                if r[5] is not None:
                    sh[r[5]] = sh.get(r[5], 0.0) + r[0].elapsed_time(r[1])
        name = dominant if dominant in sym else max(sym, key=lambda k: sym[k]["total_ms"])
        peak = mfma_peaks[dtype]

        def entry(a):
            t = a["total_ms"] * 1e-3
            top = max(a["shapes"], key=a["shapes"].get) if a["shapes"] else None
            return {"total_ms": round(a["total_ms"], 3), "launches": a["launches"],
                    "avg_launch_ms": round(a["total_ms"] / a["launches"], 4),
                    "of_which_split_k_with_reduce_ms": round(a["split_ms"], 3),
                    "achieved": round(a["alg"] / t / 1e12, 2), "frac": round(a["alg"] / t / 1e12 / peak, 4),
                    "issued": round(a["flops"] / t / 1e12, 2), "frac_issued": round(a["flops"] / t / 1e12 / peak, 4),
                    "top_shape": top, "top_shape_ms": round(a["shapes"][top], 3) if top else None}

        layers = {k[6:]: {"ms_per_launch": round(v["avg_ms"], 4), "TFLOP/s": round(v["alg"] / (v["avg_ms"] * 1e-3) / 1e12, 1),
                          "issued_TFLOP/s": round(v["flops"] / (v["avg_ms"] * 1e-3) / 1e12, 1),
                          "launches": v["launches"]} for k, v in summ.items() if k.startswith("layer/")}
        e = entry(sym[name])
        out = {"kernel": name, "bound": "mfma", "achieved": e["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": e["frac"],
               "achieved_issued": e["issued"], "frac_issued": e["frac_issued"], "traffic": None,
               "avg_launch_ms": e["avg_launch_ms"], "launches": e["launches"], "total_ms": e["total_ms"],
               "of_which_split_k_with_reduce_ms": e["of_which_split_k_with_reduce_ms"],
               "flops_per_launch": sym[name]["alg"] / sym[name]["launches"],
               "issued_flops_per_launch": sym[name]["flops"] / sym[name]["launches"],
               "top_shape": e["top_shape"], "top_shape_ms": e["top_shape_ms"],
               "symbols": {k: entry(v) for k, v in sorted(sym.items(), key=lambda kv: -kv[1]["total_ms"])},
               # replays of captured launch sequences (the body's forward / backward): one region each, the MFMA work of
               # the launches inside counted during the capture's warm-up pass
               "graph_regions": {k: entry(dict(total_ms=v["total_ms"], launches=v["launches"], flops=v["flops"] * v["launches"],
                                               alg=v["alg"] * v["launches"], split_ms=0.0, shapes={}))
                                 for k, v in kern.items() if "HIP graph" in k and v["flops"] > 0},
               "regions_ms": {k: round(v["total_ms"], 3) for k, v in kern.items()},
               "layers": layers}
        return out


kernel_timer = KernelTimer()
