"""One training step of the hot path (the caller of GeneralizedRCNN.forward in the reference:
engine/trainer.py:79-120 -- forward, sum of losses, backward, gradient all-reduce, SGD step;
optimiser semantics of solver/build.py:10-24).

MI355X-first structure:
  * every trainable parameter lives in ONE flat fp32 buffer, its gradient in a second and its
    momentum in a third (611 MB each for VGG16/VOC out of 288 GB of HBM): the optimiser is
    three launches of one fused kernel (GEMM weights / other weights / biases) instead of 50
    parameter groups, and the data-parallel exchange is an RCCL all-reduce of contiguous
    slices of the gradient buffer -- no bucketing copies.
  * the big Linear weights get their gradient written by the wgrad GEMM straight into the
    flat buffer ("fresh" flag = overwrite on first touch, accumulate afterwards), so the
    411 MB fc6 gradient is never zero-filled nor copied.
  * the fused SGD kernel also refreshes the bf16 shadow weights the matrix cores read.
  * the head's parameters (fc6/fc7, Sim_Net, predictor: 99 % of the bytes) are stepped on a SECOND HIP stream as
    soon as their gradients are final -- when d(loss)/d(pooled) arrives at the ROIPool node -- so their
    all-reduce (N>1), SGD pass and shadow refresh overlap the backbone's backward, whose small
    convolutions leave most CUs idle; only the backbone's 15 M parameters are stepped after backward.
"""
import os

import torch
import torch.distributed as dist

from . import _lib as L
from . import precision
from . import synthetic
from .layers import linear as linear_layer
from .modeling.detector import build_detection_model
from .utils.kernel_timer import kernel_timer  # noqa: F401  (re-exported for bench.py)
from .utils.step_trace import step_trace


def load_formula_weights(model, seed, overrides=None):
    shapes = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
    sd = synthetic.init_state_dict(shapes, seed, overrides=overrides)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(sd[n]))
        bufs = [(n, tuple(b.shape)) for n, b in model.named_buffers()]
        if bufs:                               # frozen batch-norm statistics of the ResNet bodies
            bd = synthetic.init_buffers(bufs, seed)
            for n, b in model.named_buffers():
                b.copy_(torch.from_numpy(bd[n]))


def lr_factor(cfg, iteration):
    """WarmupMultiStepLR.get_lr / base_lr (solver/lr_scheduler.py:14-56) after `iteration` scheduler steps -- the
    trainer steps the scheduler BEFORE the forward of (1-based) iteration i, so step i runs at lr_factor(cfg, i)
    (engine/trainer.py:199-204)."""
    from bisect import bisect_right
    s = cfg.SOLVER
    warm = 1.0
    if iteration < s.WARMUP_ITERS:
        if s.WARMUP_METHOD == "constant":
            warm = s.WARMUP_FACTOR
        elif s.WARMUP_METHOD == "linear":
            alpha = float(iteration) / s.WARMUP_ITERS
            warm = s.WARMUP_FACTOR * (1 - alpha) + alpha
        else:
            raise ValueError("Only 'constant' or 'linear' warmup_method accepted, got %s" % s.WARMUP_METHOD)
    return warm * s.GAMMA ** bisect_right(list(s.STEPS), iteration)


def momentum_correction(cur_lr, new_lr, threshold=1.1, eps=1e-10):
    """update_momentum (engine/trainer.py:38-51): when the learning rate jumps by more than 10 % the momentum
    buffers are rescaled by new_lr / cur_lr; returns that factor or None."""
    if not (cur_lr > 1e-7 and cur_lr != new_lr):
        return None
    ratio = max(new_lr / max(cur_lr, eps), cur_lr / max(new_lr, eps))
    return new_lr / cur_lr if ratio > threshold else None


def all_reduce_flat(flat, world, chunk_elems=64 * 1024 * 1024, group=None):
    """Sum-all-reduce a flat gradient buffer in place over `world` ranks as a few large contiguous
    collectives (256 MB of fp32 each by default: large messages for RCCL's rings over xGMI, and
    no bucketing copies because the buffer is already contiguous).  The 1/world of the mean is
    folded into the fused SGD kernel (grad_scale).  No-op for world == 1."""
    if world <= 1:
        return
    works = []
    n = flat.numel()
    for s in range(0, n, chunk_elems):
        works.append(dist.all_reduce(flat[s:min(n, s + chunk_elems)], async_op=True, group=group))
    for w in works:
        w.wait()


class GradExchange(object):
    """The data-parallel exchange of ONE flat gradient buffer, piece by piece, as the pieces become final.

    The reference wraps the model in DistributedDataParallel (tools/train_net.py:50-55), whose buckets are all-reduced
    as autograd fills them.  Here the gradients already live in one flat buffer, so a "bucket" is just a range of it:
    `ready(lo, hi)` says that flat[lo:hi) will not change any more in this step and starts its sum-all-reduce at once
    (on the side stream, behind an event recorded on the producing stream); `finish(lo, hi)` exchanges whatever part
    of [lo, hi) was not announced and waits for everything.  Pieces larger than `chunk` are cut so that several
    collectives are in flight.  dtype "bf16": the piece is rounded to bf16 into a persistent staging buffer, summed in
    bf16 on the wire (half the xGMI bytes), and written back as fp32 -- the optimiser still accumulates in fp32.
    The mean's 1/world stays folded into the SGD kernel.  world == 1: every call is a no-op."""

    def __init__(self, flat, world, dtype="fp32", chunk_elems=32 * 1024 * 1024, side=None, group=None):
        if dtype not in ("fp32", "bf16"):
            raise ValueError("GradExchange: dtype %r (fp32 | bf16)" % (dtype,))
        self.flat, self.world, self.dtype, self.chunk, self.side, self.group = flat, world, dtype, chunk_elems, side, group
        self.stage = torch.empty_like(flat, dtype=torch.bfloat16) if (dtype == "bf16" and world > 1) else None
        self.done, self.works = [], []
        # bench.py: (stream tag, start event, end event) around every finish() -- the time a stream is blocked on collectives
        self.measure, self.marks = False, []

    def begin(self):
        self.done, self.works = [], []

    def _issue(self, lo, hi):
        for a in range(lo, hi, self.chunk):
            b = min(hi, a + self.chunk)
            piece = self.flat[a:b]
            if self.stage is not None:
                st = self.stage[a:b]
                st.copy_(piece)
                self.works.append((dist.all_reduce(st, async_op=True, group=self.group), a, b))
            else:
                self.works.append((dist.all_reduce(piece, async_op=True, group=self.group), a, b))

    def ready(self, lo, hi, stream=None):
        """flat[lo:hi) is final: exchange it now (asynchronously), behind the producing stream's work, on `stream`
        (default: the side stream; pieces announced while that stream is busy with the head's update -- the backbone's
        runs of layers -- name their own)."""
        if self.world <= 1 or hi <= lo:
            return
        for a, b in self.done:
            if lo < b and a < hi:
                # a range announced twice in one step = a producer that wrote into it AFTER its first announcement (e.g. a
                # weight-gradient batch flushed twice): the first collective summed a partial gradient.  Never silently.
                raise RuntimeError("GradExchange.ready: [%d, %d) overlaps [%d, %d), already handed to the all-reduce in this "
                                   "step" % (lo, hi, a, b))
        self.done.append((lo, hi))
        side = stream if stream is not None else self.side
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._issue(lo, hi)
        else:
            self._issue(lo, hi)

    def pending(self, lo, hi):
        """The parts of [lo, hi) no ready() call covered, as a sorted list of ranges."""
        out, pos = [], lo
        for a, b in sorted(self.done):
            a, b = max(a, lo), min(b, hi)
            if b <= a:
                continue
            if a > pos:
                out.append((pos, a))
            pos = max(pos, b)
        if pos < hi:
            out.append((pos, hi))
        return out

    def finish(self, lo, hi):
        """Exchange what is left of [lo, hi) and wait for every collective issued since begin() (the caller runs this
        on the stream that consumes the gradients)."""
        if self.world <= 1:
            return
        if self.measure:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for a, b in self.pending(lo, hi):
            self.done.append((a, b))
            self._issue(a, b)
        for w, a, b in self.works:
            w.wait()
            if self.stage is not None:
                self.flat[a:b].copy_(self.stage[a:b])
        self.works = []
        if self.measure:
            e1.record()
            on_side = self.side is not None and torch.cuda.current_stream() == self.side
            self.marks.append(("side" if on_side else "main", e0, e1))


def reduce_loss_dict(loss_dict, world=None, dst=0, group=None):
    """Logging reduce of the reference's trainer (engine/trainer.py:14-36): the per-rank loss values summed onto rank
    `dst` in ONE small collective (the dictionary's values stacked in sorted-key order) and divided by the world size
    there; other ranks get the un-normalised partial sums back, like the reference.  world < 2: the input itself."""
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world < 2:
        return loss_dict
    with torch.no_grad():
        names = sorted(loss_dict.keys())
        stacked = torch.stack([loss_dict[k].detach().float().reshape(()) for k in names], dim=0)
        dist.reduce(stacked, dst=dst, group=group)
        if dist.get_rank() == dst:
            stacked /= world
        return {k: v for k, v in zip(names, stacked)}


class FlatSGD(object):
    """Parameters, gradients and momenta as three flat fp32 buffers + the fused SGD kernel."""

    def __init__(self, cfg, model, world=1):
        self.cfg, self.world = cfg, world
        self.momentum = cfg.SOLVER.MOMENTUM
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        # the 8 predictor heads are evaluated as ONE GEMM (roi_weak_predictors.py): laid out back to back they ARE one
        # (5C+12C) x 4096 matrix and one bias vector -- no torch.cat per step, one weight-gradient launch, one shadow
        pred = getattr(getattr(model, "roi_heads", None), "predictor", None)
        pred_w, pred_b = ([], [])
        if hasattr(pred, "set_fused") and os.environ.get("ODW_NO_PRED_FUSE") != "1":
            heads = [getattr(pred, h) for h in pred.head_names]
            if all(h.weight.requires_grad and h.bias.requires_grad and h.weight.numel() % 4 == 0 for h in heads):
                pred_w, pred_b = [h.weight for h in heads], [h.bias for h in heads]
        pred_ids = set(id(p) for p in pred_w + pred_b)
        name_of = {id(p): n for n, p in named}
        gemm_w = [(n, p) for n, p in named if self._is_gemm_weight(model, n, p)] + [(name_of[id(p)], p) for p in pred_w]
        gemm_ids = set(id(p) for _, p in gemm_w)
        other_w = [(n, p) for n, p in named if id(p) not in gemm_ids and "bias" not in n]
        biases = [(n, p) for n, p in named if "bias" in n and id(p) not in pred_ids] + [(name_of[id(p)], p) for p in pred_b]
        order = gemm_w + other_w + biases
        # weights keep 16-byte-aligned slices (GEMM / conv operands); biases are packed back to back (only ever read
        # as scalars or hit by atomics), which also makes the predictor's 8 bias vectors one contiguous vector
        sizes = [p.numel() if "bias" in n else (p.numel() + 3) // 4 * 4 for n, p in order]
        sizes[-1] += (-sum(sizes)) % 4
        total = sum(sizes)
        dev = order[0][1].device
        self.flat_p = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self.slices = {}
        for (n, p), sz in zip(order, sizes):
            view = self.flat_p[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_g[off:off + p.numel()].view(p.shape)
            self.slices[n] = (off, p.numel())
            off += sz
        n_gemm = sum(s for s, _ in zip(sizes, gemm_w))
        n_other = sum(sizes[len(gemm_w):len(gemm_w) + len(other_w)])
        self.regions = [  # (start, length, lr, wd)
            (0, n_gemm, cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY),
            (n_gemm, n_other, cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY),
            (n_gemm + n_other, total - n_gemm - n_other, cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR,
             cfg.SOLVER.WEIGHT_DECAY_BIAS)]
        self.gemm_params = [p for _, p in gemm_w]
        self.n_gemm = n_gemm
        # (ODW_PRIO = "step,optimiser,contrastive" stream priorities, an experiment knob.  Default 0,0,0 since the end of round 6:
        # EVERY stream of the process at normal priority, so that all of them share the runtime's pool of GPU_MAX_HW_QUEUES = 4
        # hardware queues.  A stream of another priority gets a queue of its own, and with five queues busy at once the whole
        # step runs at two thirds of its speed -- 12.3 against 8.3 ms, body forward included (profiles/r06/ab_actstream*.txt).)
        prio = [int(v) for v in os.environ.get("ODW_PRIO", "0,0,0").split(",")]
        self.side = torch.cuda.Stream(device=dev, priority=prio[1]) if dev.type == "cuda" else None
        odw = getattr(cfg, "ODW", None)
        self.exchange = GradExchange(self.flat_g, world, dtype=getattr(odw, "GRAD_EXCHANGE", "fp32"), side=self.side)
        self.wgrad_slices = int(getattr(odw, "WGRAD_SLICES", 4)) if world > 1 else 1
        self.early_done = False
        self.first = True
        self.lr_scale = 1.0
        self.total = total
        # SOLVER.ITER_SIZE bookkeeping, owned by the optimiser like the reference's (engine/trainer.py:86-120): gradients are
        # zeroed by the optimiser step, not by a position in the iteration count, and the schedule advances by scheduler
        # steps taken, not by the iteration index (skipped batches and resumed runs keep both consistent)
        self.grads_clean = True         # nothing accumulated since the last step() / construction / resume()
        self.sched_steps = 0            # WarmupMultiStepLR.last_epoch: scheduler steps taken so far
        # bf16 shadows of the GEMM weights: one flat buffer the SGD kernel refreshes in the same pass
        self.flat_w16 = None
        self.shadows = []
        if gemm_w:
            from . import gemm

            mode = precision.get_precision()
            single_bwd = not precision.bwd_split()      # "bf16" / "bf16x2f": the backward reads one bf16 plane of W

            def managed_shadow(weight, o, cm=None):
                sh = gemm.Shadow(weight)
                if (cm is not None and mode == "bf16x2f" and os.environ.get("ODW_NO_PAIR") != "1"
                        and cm[0] % 64 == 0 and 1 <= cm[1] <= 64):
                    # the first head Linear: also as cell-major planes [hi | mid] for the shared clean + DropBlock forward
                    sh.cm = tuple(cm)
                    sh.w_cm = torch.empty((weight.shape[0], 2 * weight.shape[1]), dtype=torch.bfloat16, device=dev)
                n_out, k_in = weight.shape
                r64 = lambda v: (v + 63) // 64 * 64
                if single_bwd:      # W (bf16) is a slice of the flat shadow the SGD kernel rewrites, W^T refreshed in place
                    w16 = self.flat_w16[o:o + weight.numel()].view(weight.shape)
                    sh.bwd2 = gemm.bwd2_layer(weight)       # (ODW_BWD2=1, a measurement mode: small-K layers' backward on two planes)
                    t_bwd = len(precision.patterns("gemm")[1]) if sh.bwd2 else 1
                    sh.wt = torch.empty((k_in, t_bwd * r64(n_out)), dtype=torch.bfloat16, device=dev)
                    if mode == "bf16":
                        sh.w = w16
                    else:           # "bf16x2f": the forward operand = the bf16 planes of the fp32 master, re-split in place
                        sh.w16 = w16
                        # (a weight kept as cell-major planes has no channel-major forward copy: every forward reads w_cm)
                        sh.w = None if sh.w_cm is not None else torch.empty(
                            (n_out, len(precision.patterns("gemm")[1]) * r64(k_in)), dtype=torch.bfloat16, device=dev)
                    sh.managed = True
                    sh.mode = mode
                # ("bf16x3" / "bf16x2": the plane layouts are rebuilt from the fp32 master after each step, Shadow.refresh)
                # Linears whose gradient is large enough for its read-modify-write to matter get ONE weight-gradient
                # GEMM per step over all their evaluations (gemm.WgradBatch)
                sh.batch = gemm.WgradBatch() if (weight.numel() >= (8 << 20) and os.environ.get("ODW_NO_WGRAD_BATCH") != "1") else None
                if sh.batch is not None:
                    sh.batch.split = bool(sh.bwd2)
                if sh.batch is not None and self.world > 1 and os.environ.get("ODW_NO_OVERLAP") != "1":
                    # its gradient is produced by ONE GEMM per step (the batch): hand it to the exchange as it retires,
                    # the largest ones (fc6: 411 MB) in row blocks
                    weight._odw_grad_ready = self._on_grad_ready
                    weight._odw_flat_offset = o
                    weight._odw_slice_rows = (-(-n_out // self.wgrad_slices) + 255) // 256 * 256 if weight.numel() >= (64 << 20) else 0
                self.shadows.append(sh)
                return sh

            if single_bwd:
                self.flat_w16 = torch.empty(n_gemm, dtype=torch.bfloat16, device=dev)
            for n, p in gemm_w:
                if id(p) in pred_ids:
                    continue
                mod = model.get_submodule(n.rsplit(".", 1)[0])
                mod._shadow = managed_shadow(p, self.slices[n][0], getattr(mod, "cm_layout", None))
            if pred_w:
                ow, nw = self.slices[name_of[id(pred_w[0])]][0], sum(p.numel() for p in pred_w)
                ob, nb = self.slices[name_of[id(pred_b[0])]][0], sum(p.numel() for p in pred_b)
                w_cat = self.flat_p[ow:ow + nw].view(-1, pred_w[0].shape[1]).detach().requires_grad_(True)
                b_cat = self.flat_p[ob:ob + nb].detach().requires_grad_(True)
                w_cat.grad = self.flat_g[ow:ow + nw].view_as(w_cat)
                b_cat.grad = self.flat_g[ob:ob + nb]
                pred.set_fused(w_cat, b_cat, managed_shadow(w_cat, ow))
                self.gemm_params.append(w_cat)
            self._refresh_shadows(initial=True)

    def _refresh_shadows(self, initial=False):
        from . import gemm
        if self.flat_w16 is None:           # "bf16x3" / "bf16x2": invalidate, the next forward re-splits the fp32 master
            for sh in self.shadows:
                sh.version = -1
                sh.w = sh.wt = None
            return
        if initial:
            L.check(L.lib().odw_f32_to_bf16(L.ptr(self.flat_p), L.ptr(self.flat_w16), self.n_gemm, L.stream()),
                    "f32_to_bf16")
        for sh in self.shadows:
            n, k = sh.weight.shape
            if sh.mode == "bf16":
                gemm.transpose_bf16(sh.w, n, k, out=sh.wt)    # in place: same buffer every step, no allocator traffic
            else:                           # "bf16x2f": W^T from the refreshed bf16 plane, forward planes from the master
                if sh.bwd2:                 # (ODW_BWD2: the planes of W^T, from the master)
                    precision.split_cols(sh.weight.detach(), precision.patterns("gemm")[1], (n + 63) // 64 * 64, out=sh.wt)
                else:
                    gemm.transpose_bf16(sh.w16, n, k, out=sh.wt)
                if sh.w_cm is not None:
                    gemm.split_rows_cm(sh.weight.detach(), sh.cm[0], sh.cm[1], out=sh.w_cm)
                else:
                    precision.split_rows(sh.weight.detach(), precision.patterns("gemm")[1], (k + 63) // 64 * 64, out=sh.w)

    @staticmethod
    def _is_gemm_weight(model, name, p):
        if p.dim() != 2:
            return False
        mod = model.get_submodule(name.rsplit(".", 1)[0])
        return isinstance(mod, linear_layer.Linear)

    def sync_from_params(self, model=None):
        """After weights were loaded into the model (utils/checkpoint.load_checkpoint copies into the flat views):
        rebuild the bf16 shadows the matrix cores read (and, given the model, let its HIP body re-pack the frozen
        layers' copies, which it otherwise packs once)."""
        self.join_side()
        body = getattr(model, "backbone_hip", None) if model is not None else None
        if body is not None and hasattr(body, "invalidate_weights"):
            body.invalidate_weights()
        if self.shadows:
            self._refresh_shadows(initial=True)

    def set_iteration(self, iteration):
        """Learning-rate schedule + momentum correction for (1-based) training iteration `iteration`."""
        f = lr_factor(self.cfg, iteration)
        if f != self.lr_scale:
            base = self.cfg.SOLVER.BASE_LR
            corr = momentum_correction(base * self.lr_scale, base * f)
            if corr is not None and not self.first:
                self.join_side()
                self.flat_m.mul_(corr)
            self.lr_scale = f

    # ---- optimizer / scheduler state in the reference's checkpoint layout (utils/checkpoint.py:41-63 saves
    # optimizer.state_dict() of torch.optim.SGD with one parameter per group, solver/build.py:10-24, and the
    # scheduler's state_dict) ------------------------------------------------------------------------------------
    def _param_order(self, model):
        return [n for n, p in model.named_parameters() if p.requires_grad and n in self.slices]

    def state_dict(self, model):
        """torch.optim.SGD.state_dict() layout: group i / state i = the i-th trainable parameter of
        model.named_parameters(); momentum buffers are views of the flat momentum copied out."""
        s = self.cfg.SOLVER
        self.join_side()
        groups, state = [], {}
        for i, n in enumerate(self._param_order(model)):
            off, numel = self.slices[n]
            bias = "bias" in n
            groups.append({"lr": (s.BASE_LR * s.BIAS_LR_FACTOR if bias else s.BASE_LR) * self.lr_scale,
                           "weight_decay": s.WEIGHT_DECAY_BIAS if bias else s.WEIGHT_DECAY, "momentum": self.momentum,
                           "dampening": 0, "nesterov": False, "params": [i]})
            if not self.first:
                shape = dict(model.named_parameters())[n].shape
                state[i] = {"momentum_buffer": self.flat_m[off:off + numel].view(shape).clone()}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, model, sd):
        """Momentum buffers of a checkpoint written by state_dict() above or by the reference's torch.optim.SGD."""
        self.join_side()
        order = self._param_order(model)
        state = sd.get("state", {})
        if len(sd.get("param_groups", order)) != len(order):
            raise ValueError("optimizer state has %d parameter groups, the model %d trainable parameters"
                             % (len(sd["param_groups"]), len(order)))
        loaded = 0
        for i, n in enumerate(order):
            entry = state.get(i, state.get(str(i)))
            if entry is None or entry.get("momentum_buffer") is None:
                continue
            off, numel = self.slices[n]
            self.flat_m[off:off + numel].copy_(entry["momentum_buffer"].reshape(-1).to(self.flat_m.device))
            loaded += 1
        if loaded:
            self.first = False
        return loaded

    def scheduler_state(self, iteration):
        """WarmupMultiStepLR.state_dict() fields that matter on resume (solver/lr_scheduler.py:14-56).  `last_epoch` is the
        number of SCHEDULER steps taken -- one per SOLVER.ITER_SIZE group (engine/trainer.py:86-91), not one per
        iteration; a caller that never passed iteration numbers to the step gets the group index of `iteration`."""
        s = self.cfg.SOLVER
        iter_size = max(1, int(s.ITER_SIZE))
        pos = self.sched_steps if self.sched_steps > 0 else (int(iteration) + iter_size - 1) // iter_size
        # "unit": what last_epoch counts.  Checkpoints of earlier revisions stored the micro-ITERATION index there (no marker);
        # utils/checkpoint.restore_training_state reconciles those against the iteration count.
        return {"unit": "sched_steps", "last_epoch": int(pos), "milestones": tuple(s.STEPS), "gamma": s.GAMMA, "warmup_factor": s.WARMUP_FACTOR,
                "warmup_iters": s.WARMUP_ITERS, "warmup_method": s.WARMUP_METHOD}

    def resume(self, sched_steps):
        """A run whose scheduler has taken `sched_steps` steps (the checkpoint's scheduler.last_epoch): its learning-rate
        factor (no momentum rescale: the buffers were saved under that factor), and a FRESH accumulation group -- the
        partial gradient sum of an interrupted SOLVER.ITER_SIZE group is not part of a checkpoint."""
        self.sched_steps = max(0, int(sched_steps))
        self.lr_scale = lr_factor(self.cfg, self.sched_steps) if self.sched_steps > 0 else 1.0
        self.grads_clean = True

    def _on_grad_ready(self, weight, r0, r1):
        """gemm.WgradBatch: rows [r0, r1) of `weight`'s gradient are final (called from backward)."""
        if getattr(self, "hold", False):
            return                      # inside a SOLVER.ITER_SIZE group: exchanged once, after its last backward
        k = weight.shape[1]
        o = weight._odw_flat_offset
        self.exchange.ready(o + r0 * k, o + r1 * k)

    def begin_step(self, accumulate=False):
        """Gradient buffer state for a new step: GEMM weights are overwritten by their first wgrad
        launch; everything autograd accumulates into (convs, predictor heads, biases) is zeroed.
        accumulate: a later micro-step of a SOLVER.ITER_SIZE group -- every gradient is added to what is there."""
        if not accumulate:
            for p in self.gemm_params:
                p._odw_fresh = True
            self.flat_g[self.n_gemm:].zero_()
        self.early_done = False
        self.hold = False
        self.exchange.begin()

    def _sgd_region(self, i, paced=0):
        """paced = workgroup cap of an update that runs BESIDE other kernels (head_grads_ready): at full width the
        HBM-bound pass starves whatever shares the GPU with it (the first backbone weight-gradient GEMM took 400 us
        instead of 43) and the overlap bought nothing; on 192-256 workgroups it stretches over the backbone's backward
        and mostly disappears behind it (8.10 -> 7.96 ms/step, three alternating runs each on one box; 128 and fewer
        outlast the backward).  Round 3: the next forward no longer waits for this pass at the end of the step (step()),
        so outlasting the backward costs nothing; 160 workgroups measured 9.42-9.68 ms against 9.57-9.85 at 256 and
        9.56 unpaced (tools/exp/ab_env.sh, alternating, box drift +-0.15 ms)."""
        start, n, lr, wd = self.regions[i]
        if n == 0:
            return
        shadow = self.flat_w16 if (i == 0 and self.flat_w16 is not None) else None
        L.check(L.lib().odw_sgd_momentum_paced(L.ptr(self.flat_p[start:]), L.ptr(self.flat_g[start:]),
                                               L.ptr(self.flat_m[start:]), L.ptr(shadow), n, lr * self.lr_scale, wd,
                                               self.momentum, 1.0 / self.world, 1 if self.first else 0, int(paced),
                                               L.stream()), "sgd_momentum")

    def head_grads_ready(self):
        """Called from backward (tensor hook on the pooled features) once every gradient of region 0 is final:
        all-reduce + SGD + shadow refresh of the head on the side stream, overlapping the backbone's backward.
        (Measured alternative: SGD deferred to overlap the NEXT step's backbone forward instead -- same step time:
        the optimiser pass is 3.4 GB of HBM traffic wherever it runs.)"""
        if (self.side is None or self.n_gemm == 0 or self.early_done or getattr(self, "hold", False)
                or os.environ.get("ODW_NO_OVERLAP") == "1"):
            return
        self.flush_wgrad()
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.exchange.finish(0, self.n_gemm)       # what the weight-gradient GEMMs did not hand over as they retired
            # (paced only without an exchange in front of it: at N > 1 the all-reduce already takes the window)
            self._sgd_region(0, paced=int(os.environ.get("ODW_SGD_PACE", "160")) if self.world == 1 else 0)
            self._refresh_shadows()
        self.early_done = True

    def all_reduce(self):
        """Whatever has not been exchanged yet (the backbone and the biases, or everything)."""
        if self.early_done:
            self.exchange.finish(self.n_gemm, self.total)
        elif self.side is not None and self.exchange.works:
            # pieces were issued on the side stream but the head's early step did not run (ODW_NO_OVERLAP, ITER_SIZE)
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                self.exchange.finish(0, self.total)
            torch.cuda.current_stream().wait_stream(self.side)
        else:
            self.exchange.finish(0, self.total)

    def flush_wgrad(self):
        """Weight-gradient batches whose last registered evaluation never ran its backward (e.g. a loss that does not
        reach every pass): run the GEMM over what was filled."""
        for sh in self.shadows:
            b = getattr(sh, "batch", None)
            if b is not None and b.rows:
                b.flush(sh.weight)

    def step(self):
        self.flush_wgrad()
        if not self.early_done:
            self._sgd_region(0)
            self._refresh_shadows()
        self._sgd_region(1)
        self._sgd_region(2)
        if self.early_done:
            # the next forward reads the refreshed weights -- but not before its first head GEMM, ~2 ms of backbone
            # forward away: the wait is handed to the shadows (gemm.Shadow.refresh) instead of blocking the stream here.
            # (At N > 1 the side stream carries exchange -> update -> refresh back to back after the backward.)
            ev = torch.cuda.Event()
            ev.record(self.side)
            self._head_event = ev
            for sh in self.shadows:
                sh.pending = ev
        self.first = False
        self.grads_clean = True             # (optimizer.zero_grad() of engine/trainer.py:120: the next backward starts a sum)

    _head_event = None

    def join_side(self):
        """Make the current stream wait for the side stream's update of the head (anything that reads or writes the flat
        buffers outside the step: schedules, checkpoints)."""
        if self._head_event is not None:
            torch.cuda.current_stream().wait_event(self._head_event)
            self._head_event = None
            for sh in self.shadows:
                sh.pending = None


def build_training_step(cfg, device, dtype="bf16", world=1, seed=1234, backend="hip"):
    """One training step on the gfx950 kernels: MFMA GEMMs / implicit-GEMM convolutions, fused flat SGD, flat-buffer
    RCCL all-reduce.  dtype = arithmetic precision of the products (od_wscl_amd.precision): "bf16" (throughput) |
    "bf16x3" (fp32-grade, the reference's DTYPE float32 on the bf16 matrix cores; "fp32" is an alias) | "bf16x2" |
    "bf16x2f" (forward as bf16x2 -- the parity bar on losses and selections --, backward on single bf16 planes).
    (The hipBLASLt / MIOpen / torch.optim comparison step lives in tools/torch_baseline.py, outside the product.)"""
    if backend != "hip":
        raise ValueError("od_wscl_amd.engine has one back end (the HIP kernels); the library comparison path is "
                         "tools/torch_baseline.py")
    precision.set_precision({"fp32": "bf16x3", "f32": "bf16x3"}.get(dtype, dtype))
    model = build_detection_model(cfg).to(device)
    load_formula_weights(model, 1)
    model.train()
    fe = model.roi_heads.feature_extractor
    fe.fc6.tag = "fc6"
    fe.fc7.tag = "fc7"
    model.roi_heads.model_sim.mlp[0].tag = "sim0"
    kernel_timer.enabled = os.environ.get("ODW_NO_TIMER") != "1"
    debug = os.environ.get("ODW_DEBUG_SYNC", "").split(",")

    def mark(tag):
        if tag in debug or "all" in debug:
            torch.cuda.synchronize()
            print("[odw] ok", tag, flush=True)

    hip_body = model.hip_body()
    # the static launch sequences of the VGG body as HIP graphs (one pair per input shape); not inside SOLVER.ITER_SIZE
    # groups, whose later micro-steps ACCUMULATE into the gradients (different kernel arguments)
    if hasattr(hip_body, "use_graphs") and max(1, int(cfg.SOLVER.ITER_SIZE)) == 1 and os.environ.get("ODW_NO_GRAPHS") != "1":
        hip_body.use_graphs = True
        # bounded: at most ODW.GRAPH_CACHE shapes stay captured (least recently used evicted), a shape is captured when it
        # comes back, everything else runs eagerly (vgg16_hip.VGGBackboneHip._graph_for)
        # (the ODW_GRAPH_CACHE environment knob documented in vgg16_hip.py wins over the config default)
        hip_body.graph_cache_size = int(os.environ.get("ODW_GRAPH_CACHE") or
                                        getattr(getattr(cfg, "ODW", None), "GRAPH_CACHE", hip_body.graph_cache_size))
    if cfg.MODEL.BACKBONE.CONV_BODY.startswith("VGG16"):
        conv_desc = "od_wscl_amd HIP implicit-GEMM conv3x3 (NHWC, MFMA)"
    else:
        conv_desc = "od_wscl_amd HIP: 1x1 convs on the MFMA GEMM, implicit-GEMM conv3x3, folded frozen BN (NHWC)"
    opt = FlatSGD(cfg, model, world)
    model.roi_heads.head_grads_ready = opt.head_grads_ready
    # N > 1: the body's backward in three runs of layers (conv5 / conv4 / conv3 for VGG16), each run's weight gradients
    # handed to the exchange when its launches are queued -- on a stream of their own: the side stream is busy with the
    # head's update by then.  What is left for the end of the step is the last run (the smallest) and the biases.
    if world > 1 and hasattr(hip_body, "bwd_segments") and os.environ.get("ODW_NO_OVERLAP") != "1":
        from .modeling.backbone.vgg16_hip import backward_segments
        hip_body.bwd_segments = int(os.environ.get("ODW_BWD_SEGMENTS", "3"))
        name_of = {id(p): n for n, p in model.named_parameters()}
        seg_stream = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
        ranges = []
        for hi_li, lo_li in backward_segments(hip_body):
            offs = [opt.slices[name_of[id(hip_body.layers[li].conv.weight)]] for li in range(lo_li, hi_li + 1)]
            ranges.append((min(o for o, _ in offs), max(o + k for o, k in offs)))

        def on_segment_done(k, ranges=ranges):
            if not getattr(opt, "hold", False):
                opt.exchange.ready(ranges[k][0], ranges[k][1], stream=seg_stream)
        hip_body.on_segment_done = on_segment_done
    # the dense losses' backward is queued from inside the loss, before its second host read (loss_fused.early_backward);
    # evaluations of the large Linears that register with their weight-gradient batch after that get reserved columns
    le = getattr(model.roi_heads, "loss_evaluator", None)
    if hasattr(le, "early_backward") and os.environ.get("ODW_NO_EARLY_BWD") != "1":
        le.early_backward = True
        for sh in opt.shadows:                      # (per instance: other models of the process keep reserve = 0)
            if getattr(sh, "batch", None) is not None:
                sh.batch.reserve = 1024
        # (measured and rejected, tools/exp/ab_early.sh: the stacked pass's weight gradient as its own GEMM in the early
        # part -- more cover for the host's small launches, but a second read-modify-write of fc6's 411 MB gradient and
        # a 100 MB transposed copy of its own: 10.6-11.2 ms against 9.9-10.1)

    # SOLVER.ITER_SIZE (config/defaults.py:459-461, engine/trainer.py:86,118-120): gradients are summed over ITER_SIZE
    # consecutive iterations, the optimiser steps after the last of them and the schedule advances once per group
    # (before its first iteration).  The reference's DDP averages every micro-batch's gradients over the ranks as
    # they are produced; the sum of those means is the mean of the sums: ONE exchange per group here.
    iter_size = max(1, int(cfg.SOLVER.ITER_SIZE))
    micro = [0]

    # The step runs on a HIGH-priority stream of its own; the optimiser's side stream (and RCCL's) keep the default, lower
    # priority.  The head's update is 3.4 GB of HBM traffic beside the body's backward: with equal priorities its
    # workgroups took CUs and bandwidth from the convolution kernels (body backward 1.05 ms alone, 1.6-1.9 ms beside it);
    # with the step's launches dispatched first: 9.39-9.41 against 9.64-9.68 ms per step (tools/exp/ab_env.sh 3 X=1
    # ODW_HP_STREAM=0).  The caller's stream waits for the step's at the end, so nothing changes for code around it.
    hp_stream = [None]
    # Round 6: OFF unless ODW_HP_STREAM=1.  That measurement predates the paced optimiser pass (ODW_SGD_PACE) and the device-resident
    # loss; with both, the two stream hops per step (caller -> step stream -> caller) cost more than the priority buys: alternating
    # runs on two boxes (profiles/r06/ab_hp.txt, ab_hp_pace.txt) 8.29-8.31 against 8.36 ms per step with two hardware queues,
    # 8.59 against 8.77 with the runtime's four -- and without the extra stream the queue count no longer matters (2 / 3 / 4 queues:
    # 8.30 / 8.29 / 8.31), so the runtime's default stays.  (At N > 1 it was never on: RCCL's kernels sit on default-priority
    # streams, and a step that always dispatches first could keep them off the CUs.)
    hp_env = os.environ.get("ODW_HP_STREAM")
    use_hp = torch.device(device).type == "cuda" and hp_env == "1"

    def step(images, targets, rois, rand, iteration=None):
        if not use_hp:
            return _step(images, targets, rois, rand, iteration)
        if hp_stream[0] is None:
            hp_stream[0] = torch.cuda.Stream(device=device, priority=int(os.environ.get("ODW_PRIO", "0,0,0").split(",")[0]))
        hp, cur = hp_stream[0], torch.cuda.current_stream()
        hp.wait_stream(cur)
        with torch.cuda.stream(hp):
            out = _step(images, targets, rois, rand, iteration)
        cur.wait_stream(hp)
        for d in out:                                   # (the caller reads the loss / accuracy scalars on ITS stream)
            for v in d.values():
                if torch.is_tensor(v):
                    v.record_stream(cur)
        return out

    def _step(images, targets, rois, rand, iteration=None):
        # Position in the SOLVER.ITER_SIZE group, by the (1-based) iteration INDEX like the reference (trainer.py:86,118:
        # the scheduler steps when (iteration - 1) % iter_size == 0, the optimiser when iteration % iter_size == 0) -- a
        # skipped batch (trainer.py:80-82) moves neither.  Whether this backward starts a gradient sum or adds to one
        # is the optimiser's state (zeroed by its last step, trainer.py:119-120), not a function of the index: a skipped
        # first iteration does not leave the previous group's sum in place, a skipped last one keeps accumulating.
        k = (iteration - 1) % iter_size if iteration is not None else micro[0] % iter_size
        micro[0] += 1
        last = k == iter_size - 1
        sched_before = (opt.sched_steps, opt.lr_scale)
        if iteration is not None and k == 0:        # WarmupMultiStepLR + update_momentum (lr_scheduler.py, trainer.py:38-51)
            opt.sched_steps += 1
            opt.set_iteration(opt.sched_steps)
        accumulate = not opt.grads_clean
        opt.begin_step(accumulate=accumulate)
        opt.grads_clean = False
        opt.hold = not last                 # the head's early exchange + update waits for the group's last backward
        hip = getattr(model, "backbone_hip", None)
        if hip is not None:
            hip.accumulate = accumulate
        step_trace.mark("begin_step")
        try:
            losses, accs = model(images, targets, rois, rand=rand)
            mark("forward")
            step_trace.mark("loss_tail")
            finish = getattr(losses, "finish_backward", None)
            if finish is not None:          # the dense losses' backward already ran inside the loss (early_backward)
                finish()
            else:
                loss = getattr(losses, "total", None)
                if loss is None:
                    loss = sum(losses.values())
                loss.backward()
        except BaseException:
            # A step that dies mid-way (out of memory, the > 2048 pseudo-GT error of the fused loss after its early backward
            # has already written gradients) must not leave a half-built sum behind: a caller that skips the batch or retries
            # (engine/trainer.py:80-82 skips bad batches) would otherwise ADD its next backward to the partial gradients.
            # An ITER_SIZE == 1 step starts fresh again.  Inside a SOLVER.ITER_SIZE group the sum so far is dropped with the
            # failed micro-step; the optimiser still steps at the group's last INDEX (the position is a function of the
            # iteration number, like the reference's trainer.py:86,118), over the micro-steps that follow the failure.
            # With the early update on (head_grads_ready), the head's SGD pass of a step that fails AFTER the pooling node's
            # backward has already been applied; a failure before that point (the fused loss, the head's backward) has not.
            opt.grads_clean = True
            opt.hold = False
            for sh in opt.shadows:
                b = getattr(sh, "batch", None)
                if b is not None:
                    b.reset()
            # the feature extractor's parked gradients belong to the failed step too: the next forward must not find them
            # ("the gradient of the previous step's sampled-row views was never folded")
            holder = getattr(fe, "_grad_holder", None)
            if holder is not None:
                holder.pending, holder.dyn_extra = [], None
                fe._grad_holder = None
            # the scheduler step taken for this (failed) iteration (a momentum rescale at a learning-rate jump, trainer.py:38-51,
            # is not undone: it belongs to the schedule position the retried iteration reaches again)
            opt.sched_steps, opt.lr_scale = sched_before
            raise
        mark("backward")
        step_trace.mark("backward_launch")
        if last:
            opt.all_reduce()
            opt.step()
        else:
            opt.flush_wgrad()
        mark("optimizer")
        step_trace.mark("optimizer_launch")
        if "loss" in debug:
            print("[odw] losses", {k: round(float(v), 5) for k, v in losses.items()}, flush=True)
        return losses, accs

    step.model, step.optimizer = model, opt
    mode = precision.get_precision()
    info = {"gemm_backend": "od_wscl_amd HIP MFMA gemm_nt_bf16 (%s operands, fp32 acc)" % mode,
            "conv_backend": conv_desc + " " + mode, "optimizer": "od_wscl_amd fused flat SGD", "precision": mode}
    return step, info
