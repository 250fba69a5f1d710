"""The contrastive branch of the OD-WSCL loss with its control flow on the device (round 6).

Reference: roi_heads/weak_head/loss.py:281-347 -- loop 1 (IoU sampling, the drop / noise views of the sampled rows
through fc6 / fc7 / Sim_Net, the class banks), loop 2 (object discovery against the banks), SupConLossV2 over
[banks; discoveries].  loss_fused.RoIRegLossFused (rounds 2-5) shaped every tensor of this branch from two blocking
host reads per step.  Here NOTHING is read back: csrc/loss_lists.hip turns the selection kernels' counts into device
lists with device lengths, every tensor is allocated for the capacity of its extent and every kernel reads the live
extent when it starts (od_wscl_amd/dyn.py).  The host queues the whole step without waiting for the GPU once.

The branch is ONE autograd node (`_ContrastiveFn`): its forward ran eagerly (no graph), its backward walks the chain
SupCon -> Sim_Net -> fc7 -> fc6 for the stacked views and for the re-attached clean rows by hand -- the same
kernels in the same order as gemm._backward_single_plane / gemm._ReuseLinear, with device extents -- and leaves the
two row sets' input gradients in the pooling node's side buffer (fc_extractor._PoolStack.backward scatters it).

Values are the reference's and the host-list path's: same draws (the per-row draw table reproduces the stacked
passes' segment keys), same summation orders; tests/test_dyn_gpu.py compares list by list with the host assembly."""
import ctypes

import numpy as np
import torch

from .... import _lib as L
from .... import dyn
from .... import gemm
from .... import precision as P
from ....utils.kernel_timer import kernel_timer
from ....utils.step_trace import step_trace

GW = 16                      # ints per group row of the host table (csrc/loss_lists.hip: kGW)
GT_MAX = 2048                # pseudo-GT boxes od_assign holds per branch

import os as _os
_DEBUG = _os.environ.get("ODW_DEBUG_DYN") == "1"
_EXACT = False              # tests/test_e2e_gpu.py sets it: hints = the live extents (read back, blocking)


def _dbg(tag, **kw):
    """ODW_DEBUG_DYN=1: synchronise and report after every stage of the branch (localises a faulting launch)."""
    if _DEBUG:
        torch.cuda.synchronize()
        print("[odw dyn] ok", tag, {k: (v.tolist() if torch.is_tensor(v) else v) for k, v in kw.items()}, flush=True)


class HintReader(object):
    """The extents of EARLIER steps, read back without ever blocking: after a step's lists exist their scalars are copied
    to pinned memory on a side stream; a later step looks whether the copy has finished (event query) and, if so, takes
    the values as hints for its plans -- and raises if one of them overflowed its capacity."""

    def __init__(self, device):
        self.device = device
        self.stream = torch.cuda.Stream(device=device)
        self.slots = [None] * 4
        self.k = 0
        self.hints = {}
        self.sticky = torch.zeros(2, dtype=torch.int32, device=device)        # set by odw_loss_lists_b, never cleared

    def post(self, scal_a, scal_b, caps, key=None):
        if _os.environ.get("ODW_NO_HINT_POST") == "1":          # (measurement: what the read-back stream costs)
            return
        self.calls = getattr(self, "calls", 0) + 1
        if self.calls > 8 and self.calls % 4:                   # every step while the hints settle, then one step in four: the
            return                                              # copy's stream hop is ~0.03 ms of a step and the extents drift slowly
        k = self.k
        self.k = (k + 1) % len(self.slots)
        slot = self.slots[k]
        if slot is None:
            slot = self.slots[k] = {"buf": torch.zeros(34, dtype=torch.int32).pin_memory(), "ev": None, "caps": None,
                                    "dev": torch.zeros(34, dtype=torch.int32, device=self.device)}
        elif slot["ev"] is not None and not slot["ev"].query():
            return                      # four posts behind and still in flight: skip this step's copy, never wait
        # the step's scalars go to a PERSISTENT device slot on the step's stream first; the side stream copies that to the
        # host.  (Copying the step's own tensors from the side stream would need record_stream on them, and a block so
        # marked is parked until the copy's event has passed -- with the host steps ahead of the GPU, for steps.)
        d = slot["dev"]
        d[:16].copy_(scal_a)
        d[16:32].copy_(scal_b)
        d[32:34].copy_(self.sticky)
        ready = torch.cuda.Event()
        ready.record()
        self.stream.wait_event(ready)
        with torch.cuda.stream(self.stream):
            slot["buf"].copy_(d, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        slot["ev"], slot["caps"], slot["seen"], slot["key"] = ev, caps, False, key

    def poll(self, key=None):
        """Take over whatever has arrived; raise on an overflow (loud, one or two steps late: the step that overflowed
        computed on truncated lists).  Hints are kept per `key` -- the number of (image, class) groups of the step, which the
        host knows and which the extents scale with (an image with three labels samples ~3 x the rows of one with one)."""
        for slot in self.slots:
            if slot is None or slot["ev"] is None or slot.get("seen") or not slot["ev"].query():
                continue
            slot["seen"] = True
            v = slot["buf"].numpy()
            if v[4] or v[16 + 3] or v[32]:
                raise RuntimeError("RoIRegLossFused (device lists): a step's selection did not fit its buffers -- %d sampled rows, "
                                   "%d contrastive rows, %d re-attached rows for capacities %s; raise ODW.MAX_SAMPLED_ROWS"
                                   % (v[0], v[16], v[17], slot["caps"]))
            if v[16 + 9] or v[33]:
                raise RuntimeError("RoIRegLossFused: more than %d pseudo-GT boxes in one branch (od_assign's limit)" % GT_MAX)
            self.hints[slot.get("key")] = {"E1": int(v[0]), "V": int(v[1]), "N": int(v[16]), "A": int(v[17])}
        h = self.hints.get(key)
        if h is None and self.hints:          # no step with this many groups yet: scale the nearest one's
            k0 = min(self.hints, key=lambda k: abs((k or 1) - (key or 1)))
            f = float(key or 1) / float(k0 or 1)
            h = {n: int(v * f) for n, v in self.hints[k0].items()}
        return h or {}


def _bucket(n, step=128):
    """hints move in steps: the plan (kernel variant, K slices) then changes rarely from step to step"""
    return max(step, (int(n) + step - 1) // step * step)


class DeviceContrastive(object):
    """One step's contrastive branch.  build() queues its forward (lists, views, embeddings, banks) up to the point where
    object discovery can run; finish() the rest (lists B, SupCon); the returned loss carries the backward."""

    def __init__(self, owner, fe, model_sim, stacked, sizes, offs, pos_host, C, device, staging, hints):
        self.owner, self.fe, self.model_sim, self.stacked = owner, fe, model_sim, stacked
        self.sizes, self.offs, self.pos_host, self.C, self.device = sizes, offs, pos_host, C, device
        self.staging, self.hints = staging, hints
        self.sum_p, self.max_p, self.n_img = sum(sizes), max(sizes), len(sizes)
        self.maxpos = max(1, max(len(p) for p in pos_host))
        groups = [(idx, ci, c) for idx in range(self.n_img) for ci, c in enumerate(pos_host[idx])]
        self.groups = groups
        self.G = len(groups)
        # Capacities: a function of the batch SHAPE only (images, proposals, classes) -- not of how many labels this batch's
        # images carry -- so that every step asks the caching allocator for the same block sizes (sizes that changed with the
        # label count fragmented the pool: 800 MB blocks were still being cut and re-allocated 15 steps into a run).  Sampled
        # rows: ODW.MAX_SAMPLED_ROWS per image (or every proposal of every class, if that is fewer); a step that needs more
        # sets the sticky overflow flag and HintReader.poll raises.  The launches' grids are bounded, so a generous capacity
        # costs memory, not time.
        per_img = min(int(getattr(owner, "max_sampled_rows", 4096)), (C - 1) * self.max_p)
        self.E_cap = (max(32, per_img * self.n_img) + 31) // 32 * 32         # (V_cap = 2 E_cap: a multiple of 64)
        self.V_cap = 2 * self.E_cap
        self.A_cap = r64up(self.sum_p)
        self.N_cap = 7 * self.E_cap
        h = hints.poll(self.G)
        self.h_E1 = min(self.E_cap, _bucket(h.get("E1", 256), 64))
        self.h_V = min(self.V_cap, _bucket(h.get("V", 512)))
        self.h_A = min(self.A_cap, _bucket(h.get("A", 512)))
        self.h_N = min(self.N_cap, _bucket(h.get("N", 2048)))

    # ------------------------------------------------------------------------------------------------- forward, part 1
    def build(self, counts, rows, sim_feature):
        """lists A -> the stacked drop / noise views -> fc6 -> fc7 -> Sim_Net -> embeddings -> the class banks."""
        lib, dev, fe = L.lib(), self.device, self.fe
        rand = fe.rand
        G, E_cap, V_cap = self.G, self.E_cap, self.V_cap
        # ---- host table: one row per group (keys drawn in the reference's order: drop mask, fc6, fc7 of the drop view; noise,
        # fc6, fc7 of the noise view -- fc_extractor.sampled_row_views)
        grp = np.zeros((G, GW), dtype=np.int64)
        keys = np.zeros((G, 4), dtype=np.int64)
        for g, (idx, ci, c) in enumerate(self.groups):
            kd = rand.key()
            k6d, k7d = rand.key(), rand.key()
            kn = rand.key()
            k6n, k7n = rand.key(), rand.key()
            grp[g, :12] = (idx, ci, c, self.offs[idx]) + k6d + k7d + k6n + k7n
            keys[g] = kd + kn
        cls_order = sorted(range(G), key=lambda g: (self.groups[g][2], g))
        grp_d, keys_d, order_d = self.staging.upload([grp.reshape(-1).astype(np.uint32).view(np.int32),
                                                      keys.reshape(-1).astype(np.uint32).view(np.int32),
                                                      np.asarray(cls_order, dtype=np.int32)])
        self.grp_d, self.keys_d, self.order_d = grp_d, keys_d, order_d
        # ---- lists A (one int32 block + the two draw tables)
        n_cls1 = self.C - 1
        ia = torch.empty(16 + (G + 1) + E_cap + 3 * E_cap + 2 * n_cls1 + 8, dtype=torch.int32, device=dev)
        o = 0
        self.scal_a = ia[o:o + 16]; o += 16
        self.e0 = ia[o:o + G + 1]; o += G + 1
        self.roi_index = ia[o:o + E_cap]; o += E_cap
        self.bank_index = ia[o:o + 3 * E_cap]; o += 3 * E_cap
        self.bank_off = ia[o:o + n_cls1]; o += n_cls1
        self.bank_cnt = ia[o:o + n_cls1]; o += n_cls1
        tabs = torch.empty((2, V_cap, 4), dtype=torch.int32, device=dev)
        self.tab6, self.tab7 = tabs[0], tabs[1]
        L.check(lib.odw_loss_lists_a(L.ptr(grp_d), L.ptr(order_d), G, L.ptr(counts), L.ptr(rows), self.maxpos, rows.shape[-1],
                                     self.sum_p, n_cls1, E_cap, L.ptr(self.scal_a), L.ptr(self.e0), L.ptr(self.roi_index),
                                     L.ptr(self.bank_index), L.ptr(self.bank_off), L.ptr(self.bank_cnt), L.ptr(self.tab6),
                                     L.ptr(self.tab7), L.stream()), "loss_lists_a")
        if _EXACT:          # (tests) plans made for the live extents: a blocking read, what the host-list path pays every step
            va = self.scal_a.cpu().numpy()
            self.h_E1, self.h_V = max(1, int(va[0])), max(1, int(va[1]))
        self.dE1 = dyn.Dyn(self.scal_a[0:1], E_cap, self.h_E1)
        self.dV = dyn.Dyn(self.scal_a[1:2], V_cap, self.h_V)
        self.dV64 = dyn.Dyn(self.scal_a[2:3], V_cap, r64up(self.h_V))
        self.dBank = dyn.Dyn(self.scal_a[3:4], 3 * E_cap, 3 * self.h_E1)
        _dbg("lists_a", scal=self.scal_a, e0=self.e0, caps=(E_cap, V_cap, self.A_cap, self.N_cap), hints=(self.h_E1, self.h_V, self.h_A, self.h_N))
        # ---- the views, written as fc6's operand (cell-major planes for the forward, the channel-major hi plane for the backward)
        planes_cm = self.stacked._odw_planes_cm
        res = fe.pooler.output_size
        S = res[0] * res[1]
        CS = self.stacked.shape[1]
        Cch = CS // S
        self.S, self.CS, self.Cch = S, CS, Cch
        self.gamma = float(fe.sim_drop.drop_prob)
        self.x_cm = torch.empty((V_cap, 2 * CS), dtype=torch.bfloat16, device=dev)
        self.x_hi = torch.empty((V_cap, CS), dtype=torch.bfloat16, device=dev)
        self.keep_sum = torch.empty(G, dtype=torch.float32, device=dev)
        L.check(lib.odw_rows_views_cm_grouped(L.ptr(planes_cm), planes_cm.stride(0), CS, G, E_cap, L.ptr(self.scal_a[0:1]),
                                              L.ptr(self.e0), L.ptr(keys_d), L.ptr(self.roi_index), Cch, S, self.gamma,
                                              L.ptr(self.keep_sum), L.ptr(self.x_cm), self.x_cm.stride(0), CS, L.ptr(self.x_hi),
                                              self.x_hi.stride(0), L.stream()), "rows_views_cm_grouped")
        _dbg("views")
        # ---- fc6 / fc7 / Sim_Net over the views ("bf16x2f" forward: two planes per operand, three plane products)
        pa, pb = P.patterns("gemm")
        T = len(pa)
        fc6, fc7 = fe.fc6, fe.fc7
        sim0, sim2 = self.model_sim.mlp[0], self.model_sim.mlp[2]
        self.sh6, self.sh7 = fc6._get_shadow().refresh(), fc7._get_shadow().refresh()
        self.shs0, self.shs2 = sim0._get_shadow().refresh(), sim2._get_shadow().refresh()
        n6, n7, ns0, ns2 = fc6.weight.shape[0], fc7.weight.shape[0], sim0.weight.shape[0], sim2.weight.shape[0]
        if self.sh6.w_cm is None or self.sh7.w is None:
            raise RuntimeError("DeviceContrastive: fc6 must keep cell-major planes and fc7 channel-major planes (precision bf16x2f)")
        self.h6 = torch.empty((V_cap, n6), dtype=torch.float32, device=dev)
        dyn.gemm_nt_cm(self.x_cm, self.sh6.w_cm, n6, Cch, S, self.h6, self.dV, bias=fc6.bias, relu=True, drop_p=0.5,
                       row_tab=self.tab6, tag="fc6_fwd")

        _dbg("fc6 views")

        def linear(x, layer, sh, n_out, relu, drop_p, tab, tag):
            kp = dyn.r64(x.shape[1])
            xs = dyn.split_rows(x, pa, kp, self.dV)
            y = torch.empty((V_cap, n_out), dtype=torch.float32, device=dev)
            dyn.gemm_nt(xs, sh.w, V_cap, n_out, T * kp, y, bias=layer.bias, relu=relu, drop_p=drop_p, row_tab=tab, m=self.dV,
                        planes=T, tag=tag)
            return y
        self.h7 = linear(self.h6, fc7, self.sh7, n7, True, 0.5, self.tab7, "fc7_fwd")
        self.hs = linear(self.h7, sim0, self.shs0, ns0, True, 0.0, None, "sim0_fwd")
        es = linear(self.hs, sim2, self.shs2, ns2, False, 0.0, None, None)
        self.emb, self.norm_v = dyn.l2norm(es, self.dV)
        _dbg("embeddings")
        # ---- class banks (Q2) gathered from [proposal embeddings; view embeddings]
        self.sim_feature = sim_feature
        self.bank = dyn.gather_rows2(sim_feature, self.emb, self.sum_p, self.bank_index, self.dBank)
        _dbg("bank")
        # ---- the large Linears' weight-gradient batches: the views' and the re-attached rows' column blocks are registered NOW
        # (before the dense losses' early backward allocates the batch buffers), by capacity
        self.slots = {}
        for name, sh in (("fc6", self.sh6), ("fc7", self.sh7), ("sim0", self.shs0)):
            b = getattr(sh, "batch", None)
            if b is not None and sh.weight.requires_grad:
                self.slots[name] = (b.register(V_cap), b.register(self.A_cap))
        return self.bank, self.bank_off, self.bank_cnt

    # ------------------------------------------------------------------------------------------------- forward, part 2
    def finish(self, fresh_idx, fresh_cnt, gt_cnt, final_score, colstat_flat, cs_ld, cs_off, img_off, n_pos, pos_cls, temp, lmda=1.0):
        """lists B -> the re-attached clean rows' embeddings -> SupCon.  Returns lmda x the loss (autograd-connected)."""
        lib, dev = L.lib(), self.device
        G, E_cap, A_cap, N_cap = self.G, self.E_cap, self.A_cap, self.N_cap
        # column offset of the views' block in the weight-gradient batches of fc6 / fc7 (behind the stacked pass's static block)
        # and of Sim_Net's first Linear (no static block: its pass over all proposals runs without autograd)
        p64 = self.sh6.batch.offset(self.slots["fc6"][0]) if "fc6" in self.slots else 0
        if "fc7" in self.slots and self.sh7.batch.offset(self.slots["fc7"][0]) != p64:
            raise RuntimeError("DeviceContrastive: fc6 and fc7 lay their weight-gradient batches out differently")
        if "sim0" in self.slots and self.shs0.batch.offset(self.slots["sim0"][0]) != 0:
            raise RuntimeError("DeviceContrastive: Sim_Net's weight-gradient batch holds a block in front of the views'")
        self.p64 = p64
        ib = torch.empty(16 + 2 * N_cap + A_cap + (A_cap + E_cap), dtype=torch.int32, device=dev)
        o = 0
        self.scal_b = ib[o:o + 16]; o += 16
        self.feat_index = ib[o:o + N_cap]; o += N_cap
        self.labels = ib[o:o + N_cap]; o += N_cap
        self.act_rows = ib[o:o + A_cap]; o += A_cap
        self.roi_index_all = ib[o:o + A_cap + E_cap]; o += A_cap + E_cap
        self.weights = torch.empty(N_cap, dtype=torch.float32, device=dev)
        L.check(lib.odw_loss_lists_b(L.ptr(self.grp_d), L.ptr(self.order_d), G, L.ptr(img_off), L.ptr(n_pos), L.ptr(pos_cls),
                                     self.n_img, self.maxpos, fresh_idx.shape[-1], self.sum_p, L.ptr(self.scal_a), L.ptr(self.e0),
                                     L.ptr(self.roi_index), L.ptr(self.bank_index), L.ptr(self.bank_off), L.ptr(self.bank_cnt),
                                     L.ptr(fresh_idx), L.ptr(fresh_cnt), L.ptr(gt_cnt), GT_MAX, L.ptr(final_score),
                                     final_score.shape[1], L.ptr(colstat_flat), cs_ld, cs_off, N_cap, A_cap, E_cap, p64,
                                     L.ptr(self.scal_b), L.ptr(self.feat_index), L.ptr(self.labels), L.ptr(self.weights),
                                     L.ptr(self.act_rows), L.ptr(self.roi_index_all), L.ptr(self.hints.sticky), L.stream()),
                "loss_lists_b")
        _dbg("lists_b", scal=self.scal_b)
        if _EXACT:
            vb = self.scal_b.cpu().numpy()
            self.h_N, self.h_A = max(1, int(vb[0])), max(1, int(vb[1]))
        self.dN = dyn.Dyn(self.scal_b[0:1], N_cap, self.h_N)
        self.dA = dyn.Dyn(self.scal_b[1:2], A_cap, self.h_A)
        self.dE = dyn.Dyn(self.scal_b[2:3], A_cap + E_cap, self.h_A + self.h_E1)
        self.dA64 = dyn.Dyn(self.scal_b[8:9], A_cap, r64up(self.h_A))
        self.hints.post(self.scal_a, self.scal_b, (E_cap, N_cap, A_cap), key=self.G)
        # ---- embeddings of the clean rows the loss references: their Sim_Net outputs exist (the no-autograd evaluation over
        # all proposals kept them); only the normalisation is redone (Sim_Net.forward's reuse branch)
        kept = self.model_sim.kept
        e_raw = dyn.gather_rows(kept[1], self.act_rows, self.dA)
        self.e_act, self.norm_a = dyn.l2norm(e_raw, self.dA)
        features = dyn.gather_rows2(self.e_act, self.emb, A_cap, self.feat_index, self.dN)
        loss, self.dF = dyn.supcon(features, self.labels, self.weights, temp, self.dN)
        _dbg("supcon", loss=loss)
        step_trace.note("device_lists", True)
        self.lmda = float(lmda)
        self.loss = loss
        return _ContrastiveFn.apply(self.stacked, self.fe.fc6.weight, loss, self)

    def held_batches(self):
        """The weight-gradient batches that also hold a block of ANOTHER evaluation (the stacked pass: fc6, fc7)."""
        out = []
        for name, sh in (("fc6", self.sh6), ("fc7", self.sh7), ("sim0", self.shs0)):
            b = getattr(sh, "batch", None)
            if name in self.slots and b is not None and len(b.rows) > 2:
                out.append((b, sh.weight, getattr(getattr(self.fe, name, None), "tag", None) if name != "sim0" else None))
        return out

    def backward_now(self, act_stream=None):
        """The branch's backward queued at once, with the unit weight of the reference's plain sum of losses
        (engine/trainer.py:102) -- what loss_fused's early backward does for the dense losses.  Everything the branch's
        gradients depend on exists as soon as SupCon has run; queued here, ahead of the dense losses' backward, the views'
        and the re-attached rows' column blocks of the weight-gradient batches are in place when the stacked pass's block
        arrives, so fc6's 411 MB gradient is final -- and, at N > 1 ranks, on the wire -- ~1 ms earlier than when this ran
        behind it.  The autograd node then has nothing left to do."""
        one = _ONE.get(str(self.device))
        if one is None:
            one = _ONE[str(self.device)] = torch.ones(1, dtype=torch.float32, device=self.device)
        self.backward(one, act_stream=act_stream)
        self.done = True

    # ------------------------------------------------------------------------------------------------------- backward
    def backward(self, g, act_stream=None):
        """d(loss)/d(everything) for an incoming gradient g (a device scalar) of the SupCon value.
        act_stream: the re-attached clean rows' chain runs there, beside the views' chain on the current stream -- the two
        share nothing but the weight-gradient batches (disjoint column blocks, flushed after both) and the side buffer
        (disjoint rows)."""
        dev, fe = self.device, self.fe
        V_cap, A_cap, E_cap = self.V_cap, self.A_cap, self.E_cap
        fc6, fc7 = fe.fc6, fe.fc7
        sim0, sim2 = self.model_sim.mlp[0], self.model_sim.mlp[2]
        h6c, h7c = fe._clean_acts              # fc6 / fc7 outputs of the stacked clean + DropBlock pass (rows [0, P) = clean)
        ksc = self.model_sim.kept               # Sim_Net's two outputs over all proposals
        planes_hi = self.stacked._odw_planes    # (2P x >= K) bf16: what fc6's single-plane backward reads of the clean rows
        # ---- d(features) scattered to its two sources
        d_act = dyn.zero_rows(torch.empty((A_cap, 128), dtype=torch.float32, device=dev), self.dA)
        d_emb = dyn.zero_rows(torch.empty((V_cap, 128), dtype=torch.float32, device=dev), self.dV)
        dyn.scatter_rows2(self.dF, self.feat_index, self.dN, A_cap, d_act, d_emb, scale=g.reshape(1), alpha=self.lmda)
        holder = fe._grad_holder
        if holder is None or holder.kind != "extra" or holder.done:
            raise RuntimeError("DeviceContrastive.backward: the pooling node's side buffer is gone (its backward already ran)")
        extra = torch.empty((A_cap + E_cap, self.CS), dtype=torch.float32, device=dev)
        scal_b = self.scal_b
        # Sim_Net's second Linear (128 x 4096: too small for a registered batch) gets ONE weight-gradient product over both row
        # sets as well: [views | re-attached rows] column blocks, the second at the device offset r64(V)
        w2 = sim2.weight
        s2_dzt = s2_xt = None
        if w2.requires_grad:
            s2_cols = dyn.r64(V_cap) + dyn.r64(A_cap)
            s2_dzt = torch.empty((w2.shape[0], s2_cols), dtype=torch.bfloat16, device=dev)
            s2_xt = torch.empty((w2.shape[1], s2_cols), dtype=torch.bfloat16, device=dev)
        cur = torch.cuda.current_stream(dev)
        held0 = None
        if act_stream is not None:
            for name, sh in (("fc6", self.sh6), ("fc7", self.sh7), ("sim0", self.shs0)):      # (allocated HERE, on this stream)
                if name in self.slots:
                    sh.batch.buffers(sh.weight.shape[0], sh.weight.shape[1], dev)
            b0 = self.shs0.batch if "sim0" in self.slots else None
            if b0 is not None and not b0.hold:
                b0.hold, held0 = True, b0

        def chain(d_e, e_norm, norm, m, m64, which):
            """Sim_Net -> fc7 -> fc6 backward of one row set.  which = "views": saved activations are this object's, the column
            blocks of the weight-gradient batches start at their static offsets; "act": activations are rows `act_rows` of the
            stacked pass's, the blocks start at device offsets (behind the views' blocks)."""
            act = which == "act"
            rows = self.act_rows if act else None
            cap = m.cap

            def layer(dy, layer_, sh, x, y, scale, name, need_dx=True, dx_out=None):
                """backward of y = dropout(relu(x W^T + b)): dz, db; dW via the layer's batch (or directly); dx."""
                w = layer_.weight
                n_out, k_in = w.shape
                n8 = dyn.r64(n_out)
                dz = torch.empty((cap, n8), dtype=torch.bfloat16, device=dev)
                db = None
                if layer_.bias is not None and layer_.bias.requires_grad:
                    if layer_.bias.grad is None:
                        layer_.bias.grad = torch.zeros_like(layer_.bias)
                    db = layer_.bias.grad
                slot = self.slots.get(name)
                batch = sh.batch if slot is not None else None
                if batch is not None:
                    dzt_all, xt_all = batch.buffers(n_out, k_in, dev)
                    sl = slot[1 if act else 0]
                    if act:         # behind the views' block: a device offset
                        col = scal_b[4:5] if name in ("fc6", "fc7") else scal_b[6:7]
                        dyn.bwd_prep(dy, y, n_out, scale, dz, dzt_all, db, m, tcol_off=col, y_rows=rows if y is not None else None)
                        dyn.transpose(x, k_in, xt_all, m, col_off=col, src_rows=rows)
                    else:
                        off = batch.offset(sl)
                        dyn.bwd_prep(dy, y, n_out, scale, dz, dzt_all[:, off:], db, m)
                        dyn.transpose(x, k_in, xt_all[:, off:], m)
                    batch.done[sl] = True
                    batch.filled += 1
                    if name in ("fc6", "fc7"):
                        batch.dyn_k = dyn.Dyn(scal_b[5:6], sum(batch.rows), self.p64 + dyn.r64(self.h_V) + dyn.r64(self.h_A))
                    else:
                        batch.dyn_k = dyn.Dyn(scal_b[7:8], sum(batch.rows), dyn.r64(self.h_V) + dyn.r64(self.h_A))
                    if batch.filled == len(batch.rows) and not batch.hold:
                        batch.flush(w, layer_.tag)
                elif w.requires_grad and name == "sim2":
                    col = scal_b[6:7] if act else None
                    dyn.bwd_prep(dy, y, n_out, scale, dz, s2_dzt, db, m, tcol_off=col, y_rows=rows if y is not None else None)
                    dyn.transpose(x, k_in, s2_xt, m, col_off=col, src_rows=rows)
                elif w.requires_grad:
                    dzt = torch.empty((n_out, dyn.r64(cap)), dtype=torch.bfloat16, device=dev)
                    xt = torch.empty((k_in, dyn.r64(cap)), dtype=torch.bfloat16, device=dev)
                    dyn.bwd_prep(dy, y, n_out, scale, dz, dzt, db, m, y_rows=rows if y is not None else None)
                    dyn.transpose(x, k_in, xt, m, src_rows=rows)
                    fresh = w.grad is None or getattr(w, "_odw_fresh", False)
                    if w.grad is None:
                        w.grad = torch.empty_like(w)
                    w._odw_fresh = False
                    dyn.gemm_nt(dzt, xt, n_out, k_in, dyn.r64(cap), w.grad, accumulate=not fresh, k=m64,
                                tag=layer_.tag and layer_.tag + "_wgrad")
                else:
                    dzt = torch.empty((n_out, dyn.r64(cap)), dtype=torch.bfloat16, device=dev)
                    dyn.bwd_prep(dy, y, n_out, scale, dz, dzt, db, m, y_rows=rows if y is not None else None)
                _dbg("  layer %s %s wgrad" % (name, which))
                if not need_dx:
                    return None
                dx = dx_out if dx_out is not None else torch.empty((cap, k_in), dtype=torch.float32, device=dev)
                dyn.gemm_nt(dz, sh.wt, cap, k_in, n_out, dx, m=m, tag=layer_.tag and layer_.tag + "_dgrad")
                return dx

            dz_e = dyn.l2norm_bwd(d_e, e_norm, norm, m)
            if act:
                d_hs = layer(dz_e, sim2, self.shs2, ksc[0], None, 1.0, "sim2")
                d_h7 = layer(d_hs, sim0, self.shs0, h7c, ksc[0], 1.0, "sim0")
                d_h6 = layer(d_h7, fc7, self.sh7, h6c, h7c, 2.0, "fc7")
                layer(d_h6, fc6, self.sh6, planes_hi, h6c, 2.0, "fc6", dx_out=extra)           # rows [0, A) of the side buffer
            else:
                d_hs = layer(dz_e, sim2, self.shs2, self.hs, None, 1.0, "sim2")
                d_h7 = layer(d_hs, sim0, self.shs0, self.h7, self.hs, 1.0, "sim0")
                d_h6 = layer(d_h7, fc7, self.sh7, self.h6, self.h7, 2.0, "fc7")
                dxv = layer(d_h6, fc6, self.sh6, self.x_hi, self.h6, 2.0, "fc6")
                L.check(L.lib().odw_rows_views_bwd_store_grouped(L.ptr(dxv), 1, dxv.stride(0), self.G, E_cap, L.ptr(self.scal_a[0:1]),
                                                                 L.ptr(self.e0), L.ptr(self.keys_d), L.ptr(self.keep_sum), self.Cch,
                                                                 self.S, self.gamma, L.ptr(scal_b[1:2]), L.ptr(extra), L.stream()),
                        "rows_views_bwd_store_grouped")

        _dbg("scatter")
        if act_stream is not None:
            fork = torch.cuda.Event()
            fork.record(cur)
            act_stream.wait_event(fork)
            with torch.cuda.stream(act_stream):
                chain(d_act, self.e_act, self.norm_a, self.dA, self.dA64, "act")
                met = torch.cuda.Event()
                met.record(act_stream)
            chain(d_emb, self.emb, self.norm_v, self.dV, self.dV64, "views")
            cur.wait_event(met)
        else:
            chain(d_emb, self.emb, self.norm_v, self.dV, self.dV64, "views")
            _dbg("bwd views")
            chain(d_act, self.e_act, self.norm_a, self.dA, self.dA64, "act")
        _dbg("bwd act")
        if s2_dzt is not None:
            fresh = w2.grad is None or getattr(w2, "_odw_fresh", False)
            if w2.grad is None:
                w2.grad = torch.empty_like(w2)
            w2._odw_fresh = False
            dyn.gemm_nt(s2_dzt, s2_xt, w2.shape[0], w2.shape[1], s2_dzt.shape[1], w2.grad, accumulate=not fresh,
                        k=dyn.Dyn(scal_b[7:8], s2_dzt.shape[1], dyn.r64(self.h_V) + dyn.r64(self.h_A)), tag=None)
        if held0 is not None:
            held0.hold = False
            if held0.rows and held0.filled == len(held0.rows):
                held0.flush(sim0.weight, sim0.tag)
        holder.dyn_extra = (extra, self.roi_index_all, A_cap + E_cap, self.dE)
        holder.pending = []
        self.extra = extra

    # ------------------------------------------------------------------------------------------------ tests' trace
    def fill_trace(self, tr, rows, counts, inst_idx, inst_cnt, fresh_idx, fresh_cnt):
        """(tests) the index sets of the step as the host-list path reports them -- blocking reads, never in a timed step."""
        counts_h = counts.cpu().numpy()
        for idx in range(self.n_img):
            for ci, c in enumerate(self.pos_host[idx]):
                tr["iou_samples_%d_%d" % (idx, c)] = rows[idx, ci, :int(counts_h[idx][ci])].long()
        inst_h, fresh_h = inst_cnt.cpu().numpy(), fresh_cnt.cpu().numpy()
        for idx in range(self.n_img):
            for i in range(3):
                for ci, c in enumerate(self.pos_host[idx]):
                    tr["pgt_instance_%d_%d_%d" % (idx, i, c)] = inst_idx[idx, i, ci, :int(inst_h[idx][i][ci])].long().clone()
                    tr["sim_new_%d_%d_%d" % (idx, i, c)] = fresh_idx[idx, i, ci, :int(fresh_h[idx][i][ci])].long().clone()
        n = int(self.scal_b[0].item())
        tr["supcon_weights"] = self.weights[:n].clone()
        tr["supcon_n"] = n


_ONE = {}


def r64up(n):
    return (int(n) + 63) // 64 * 64


class _ContrastiveFn(torch.autograd.Function):
    """The contrastive branch as one autograd node: `stacked` (the pooling node's operand handle) and fc6's weight are its
    graph inputs -- the first orders it before the pooling node, whose side buffer it fills; the second keeps it in the graph
    when the feature map takes no gradient.  Parameter gradients are written in place like every Linear of the head."""

    @staticmethod
    def forward(ctx, stacked, w6, loss, branch):
        ctx.branch = branch
        return loss.reshape(()) * branch.lmda           # (sim_lmda x SupCon, loss.py:347; backward scales by the same)

    @staticmethod
    def backward(ctx, g):
        b, ctx.branch = ctx.branch, None
        if not getattr(b, "done", False):       # (backward_now already ran it with the unit weight)
            b.backward(g)
        return None, None, None, None
