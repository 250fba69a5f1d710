"""Full-size parity (BASELINE.json configs C1 / C2 / C4 / C5 -- every committed e2e golden is <= 128 px / <= 64 proposals):

  * one training step of the product (HIP body, fused loss) against the CPU oracle (oracle/hotpath_ref.py) on the same
    formula-generated inputs at P = 500 @ 300 px, P = 2000 @ 600 px (608^2), P = 4000 / 81 classes @ 800 px and @ 688 px
    (two scales of the COCO config's multi-scale training), the same shape with SEVERAL images -- 2 x 4000 @ 576 px and
    3 x 4000 @ 480 px, one label each, so that the contrastive set holds 2-3 classes and loss_sim > 0 --, the R-50-C5 body
    at P = 2000 @ 600 px (config 5), and the reference's single-GPU setup -- 8 images of 2000 proposals on one device
    (README.md:99-100) -- in BOTH parity modes: "bf16x2f" (what bench.py times) and "bf16x3" (fp32-grade);
  * the fused pooling kernels of the training step bit-exact against the C oracle's ROIPool at the C2 shape.

How the selections are compared.  At these sizes thousands of threshold decisions are taken per step and the closest one
sits within rounding noise of its threshold IN THE ORACLE ITSELF (similarity-threshold gaps ~1e-6).  The oracle therefore
records, for every object-discovery iteration, the signed distance of every proposal from each comparison that decides
its candidacy (`dec/*` entries of its trace).  A proposal whose every distance exceeds TOL[mode] must be decided exactly
like the oracle; only proposals closer than that may differ, and then the product's result must be EXACTLY what the
oracle's own discovery tail (NMS in score order, top-1 fallback, set difference: hotpath_ref.discover) produces from
the candidate set with those proposals flipped -- so a flip is accepted only if everything downstream of it is
bit-exact.  The pseudo labels are compared with the oracle's od_layer applied to the accepted instances.  When no
decision flipped (the usual case) this is plain equality of every index list and the losses are held to 1e-3."""
import itertools
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from conftest import weights_for  # noqa: E402

pytestmark = pytest.mark.gpu

SEEDS = {"c1": 303, "c2": 301, "c4": 305, "c5": 300, "c4s": 306, "b8": 302, "c4b": 302, "c4m": 310}
MODES = ("bf16x2f", "bf16x3")
# distance from a threshold below which a decision may legitimately differ from the fp32 oracle's: the deviation of the
# product's similarity values from the oracle's is <= ~2e-7 in "bf16x3" (fp32 re-association) and <= ~2e-5 in "bf16x2f"
# (two-plane products: 2^-16 per term) -- printed by the test as `sim deviation`
TOL = {"bf16x3": 1e-5, "bf16x2f": 1e-4}
# relative gap of two class scores below which NMS may visit them in the other order (the scores are softmax outputs:
# their relative deviation is the ABSOLUTE deviation of the logits; the worst relative deviation over all assigned
# proposals is printed as `score deviation`, ~1e-4 in both modes because it is dominated by near-cancelling logits); a
# swap is accepted only if the survivors are the same proposals.  Observed: no swap in "bf16x3", one at C4 (gap 1.25e-4)
# in "bf16x2f"
ORDER_TOL = {"bf16x3": 1e-5, "bf16x2f": 5e-4}
GRAD_TOL = {"bf16x3": 1e-3, "bf16x2f": 1e-2}      # observed: 7e-5 / 2.6e-3
# relative L2 error of every gradient TENSOR against the oracle's (a permuted or mis-scattered gradient keeps its norm;
# this does not); observed: see the FULLSIZE report lines (profiles/r04/fullsize_report.txt)
GRAD_L2_TOL = {"bf16x3": 5e-3, "bf16x2f": 2e-2}      # observed: <= 1.74e-3 / <= 7.0e-3
MAX_UNCERTAIN = 12


def _replay_selections(H, tr, trace, boxes, labels, classes, tol, order_tol, nms_thr=0.1):
    """Compare every index selection of the product (`trace`) with the oracle's (`tr`), decision by decision.
    Returns (number of iterations in which a near-threshold proposal was decided differently, report lines)."""
    n_img = len(boxes)
    pgt_index = {}
    for k, v in tr.items():                                   # loop 1: integer IoU tests behind an argmax -> exact
        if k.startswith("iou_samples_"):
            assert k in trace, k
            np.testing.assert_array_equal(trace[k].cpu().numpy(), v.numpy(), err_msg=k)
            idx, c = (int(x) for x in k.split("_")[2:])
            pgt_index[(idx, c)] = v.clone()
    inst_r = [[[torch.zeros(0, dtype=torch.long) for _ in range(classes - 1)] for _ in range(3)] for _ in range(n_img)]
    flips, lines = 0, []
    for k, d in tr.items():                                   # loop 2, in the oracle's iteration order
        if not k.startswith("dec/"):
            continue
        idx, i, c = (int(x) for x in k[4:].split("_"))
        assert d["top_gap"] > tol, ("seed unusable: arg-max gap %.2e of %s" % (d["top_gap"], k))
        sm = d["sim_margin"]
        unsure = sm.abs() <= tol
        below = sm < 0
        for s_neg in d["neg"]:                                 # quirk Q3: True rows test 1 >= s, False rows 0 >= s
            unsure |= torch.where(below, s_neg.abs() <= tol, (1.0 - s_neg).abs() <= tol)
        U = unsure.nonzero().view(-1).tolist()
        assert len(U) <= MAX_UNCERTAIN, ("seed unusable: %d proposals within %.0e of a threshold in %s" % (len(U), tol, k))
        cand_o = set(d["cand"].tolist())
        got_inst = trace["pgt_instance_%d_%d_%d" % (idx, i, c)].cpu()
        got_new = trace["sim_new_%d_%d_%d" % (idx, i, c)].cpu()
        top = torch.tensor(d["top"])
        accepted = None
        for r in range(len(U) + 1):                            # fewest flips first: r = 0 is the oracle's own decision
            for S in itertools.combinations(U, r):
                cand = sorted(cand_o.symmetric_difference(S))
                inst, new = H.discover(boxes[idx], torch.tensor(cand, dtype=torch.long), d["score"], top,
                                       pgt_index[(idx, c)], nms_thr)
                ok = torch.equal(new, got_new) and (torch.equal(inst, got_inst) or _same_up_to_score_ties(inst, got_inst, d["score"], order_tol))
                if ok:
                    accepted = (S, inst)
                    break
            if accepted is not None:
                break
        if accepted is None:
            o_inst = tr["pgt_instance_%d_%d_%d" % (idx, i, c)]
            diff = sorted(set(o_inst.tolist()) ^ set(got_inst.tolist()))
            raise AssertionError("selection differs beyond near-threshold flips in %s: uncertain %s (margins %s); instance "
                                 "lists differ in %s (their margins %s, neg %s, in oracle candidates %s); oracle new %s product new %s"
                                 % (k, U, [float(sm[q]) for q in U], diff, [float(sm[q]) for q in diff],
                                    [[float(sn[q]) for q in diff] for sn in d["neg"]], [q in cand_o for q in diff],
                                    sorted(set(tr["sim_new_%d_%d_%d" % (idx, i, c)].tolist()) ^ set(got_new.tolist())),
                                    [(int(a), int(b), float(d["score"][a]), float(d["score"][b])) for a, b in zip(o_inst.tolist(), got_inst.tolist()) if a != b][:10]
                                    + [("len", len(o_inst), len(got_inst)), ("new equal", torch.equal(tr["sim_new_%d_%d_%d" % (idx, i, c)], got_new))]))
        if accepted[0]:
            flips += 1
            lines.append("%s: flipped %s (margins %s)" % (k, list(accepted[0]), [float(sm[p]) for p in accepted[0]]))
        inst_r[idx][i][c] = got_inst
        pgt_index[(idx, c)] = torch.cat((pgt_index[(idx, c)], got_new)).unique()
    score_dev = 0.0
    for idx in range(n_img):                                   # pseudo labels: the oracle's od_layer on the accepted instances
        lab = H.image_label_vector(classes, labels[idx].unique())
        for i in range(3):
            pseudo, weights, _ = H.od_layer(boxes[idx], tr["dec_source/%d_%d" % (idx, i)], lab, inst_r[idx][i])
            np.testing.assert_array_equal(trace["pseudo_%d_%d" % (idx, i)].cpu().numpy(), pseudo.numpy(), err_msg="pseudo_%d_%d" % (idx, i))
            got_w = trace["weights_%d_%d" % (idx, i)].cpu().numpy()
            np.testing.assert_allclose(got_w, weights.numpy(), rtol=2e-3, atol=1e-7)
            score_dev = max(score_dev, float(np.max(np.abs(got_w - weights.numpy()) / np.maximum(np.abs(weights.numpy()), 1e-12))))
    return flips, lines, score_dev


def _same_up_to_score_ties(a, b, score, tol):
    """Two NMS survivor lists (descending class score) that hold the same proposals and differ only in the order of
    entries whose scores are within `tol` (relative) of each other."""
    if a.numel() != b.numel() or not torch.equal(a.sort()[0], b.sort()[0]):
        return False
    sa, sb = score[a], score[b]
    return bool(((sa - sb).abs() <= tol * sa.abs().clamp(min=1e-30)).all())


@pytest.mark.parametrize("name", ["c1", "c2", "c4", "c4s", "c5", "b8", "c4b", "c4m"])
def test_full_size_step_matches_the_oracle(name):
    import fullsize_seed_scan as S
    from oracle import hotpath_ref as H
    from test_e2e_gpu import build_model
    from od_wscl_amd import precision
    from od_wscl_amd.structures import BoxList, to_image_list
    from od_wscl_amd.utils.device_rand import DeviceRand
    seed = SEEDS[name]
    size, p, classes = S.CASES[name][:3]
    arch = S.arch_of(name)
    batch, boxes, lab, _ = S.inputs(name, seed)
    w_np = weights_for(arch, classes)
    # ---- oracle (CPU, fp32): forward with its decision records, backward for the gradient norms
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    frozen = H.FROZEN if arch == "vgg16" else H.FROZEN_RESNET
    param_names = set(n for n, _ in H.param_shapes(classes, arch))
    sd = {}
    for k, v in w_np.items():
        t = torch.from_numpy(v.copy())
        if k in param_names and not k.startswith(frozen):
            t.requires_grad_(True)
        sd[k] = t
    cfg = dict(nms=0.1, lmda=0.03, thres=0.5, temp=0.2, pooler="ROIPool", sampling_ratio=0, arch=arch,
               scale=0.125 if arch == "vgg16" else 0.0625)
    tr = {"_decisions": True}
    ref_losses, ref_accs = H.forward(batch, boxes, lab, sd, H.Rand(seed), cfg, tr)
    sum(ref_losses.values()).backward()
    ref_grad = {n: sd[n].grad.double().norm().item() for n in sd if sd[n].grad is not None}
    ref_grad_t = {n: sd[n].grad.detach().clone() for n in sd if sd[n].grad is not None}
    E_ref = tr["sim_feature"]
    for t in sd.values():
        t.grad = None
    # ---- product, in each parity mode
    for mode in MODES:
        precision.set_precision(mode)
        model = build_model("ROIPool", w_np, "fused", arch, classes)
        rois = [BoxList(b.cuda(), (size, size), "xyxy") for b in boxes]
        targets = []
        for l in lab:
            t = BoxList(torch.zeros((len(l), 4)).cuda(), (size, size), "xyxy")
            t.add_field("labels", l.cuda())
            targets.append(t)
        trace = {}
        model.roi_heads.loss_evaluator.trace = trace
        emb = []
        hook = model.roi_heads.model_sim.register_forward_hook(lambda m, a, out: emb.append(out.detach()))
        losses, accs = model(to_image_list(batch.cuda()), targets, rois, rand=DeviceRand(seed))
        hook.remove()
        sum(losses.values()).backward()
        E = [e for e in emb if e.shape[0] == E_ref.shape[0]][0].float().cpu()
        offs = np.cumsum([0] + [len(b) for b in boxes])
        sim_dev = 0.0
        for k, d in tr.items():
            if k.startswith("dec/"):
                idx = int(k[4:].split("_")[0])
                a, b = E[offs[idx]:offs[idx + 1]], E_ref[offs[idx]:offs[idx + 1]]
                sim_dev = max(sim_dev, float((a @ a[d["top"]] - b @ b[d["top"]]).abs().max()))
        flips, lines, score_dev = _replay_selections(H, tr, trace, boxes, lab, classes, TOL[mode], ORDER_TOL[mode])
        report = {k: (float(losses[k].detach()), float(ref_losses[k])) for k in ref_losses}
        worst_loss = max(abs(g - r) / max(abs(r), 1e-5) for g, r in report.values())
        worst_grad, worst_l2, worst_l2_name = 0.0, 0.0, ""
        for n, p_ in model.named_parameters():
            if n in ref_grad and ref_grad[n] > 1e-6:
                worst_grad = max(worst_grad, abs(p_.grad.double().norm().item() - ref_grad[n]) / ref_grad[n])
                l2 = float((p_.grad.detach().cpu().double() - ref_grad_t[n].double()).norm()) / ref_grad[n]
                if l2 > worst_l2:
                    worst_l2, worst_l2_name = l2, n
        print("FULLSIZE %s %s seed %d: decisions flipped %d, sim deviation %.2e, score deviation %.2e, worst loss deviation %.2e, "
              "worst gradient-norm deviation %.2e, worst gradient-tensor L2 error %.2e (%s), loss_sim %.4e %s"
              % (name, mode, seed, flips, sim_dev, score_dev, worst_loss, worst_grad, worst_l2, worst_l2_name,
                 float(ref_losses["loss_sim"]), lines))
        assert sim_dev <= 0.5 * TOL[mode], ("TOL[%s] no longer covers the similarity deviation" % mode, sim_dev)
        loss_tol = 1e-3 if flips == 0 else 5e-2        # a flipped pick moves the pseudo labels of its neighbourhood
        for k, (got, ref) in report.items():
            assert abs(got - ref) <= loss_tol * max(abs(ref), 1e-5), (k, got, ref, mode)
        for k in ref_accs:
            assert abs(float(accs[k]) - float(ref_accs[k])) < 1e-6, k
        if flips == 0:
            assert worst_grad <= GRAD_TOL[mode], (mode, worst_grad)
            assert worst_l2 <= GRAD_L2_TOL[mode], (mode, worst_l2_name, worst_l2)
        if name in ("c4b", "c4m"):       # the cases exist for this: a non-zero contrastive loss (and SupCon gradient) at P = 4000 / 81 classes
            assert float(ref_losses["loss_sim"]) > 1e-6 and len(set(int(v) for l in lab for v in l)) >= 2
        del model, losses
        torch.cuda.empty_cache()


@pytest.mark.parametrize("P,C,H,W", [(2000, 512, 76, 76), (300, 128, 38, 50)])
def test_pool_stack_nhwc_is_bit_exact_against_the_oracle(P, C, H, W):
    """roi_pool_stack_forward_nhwc (what the bf16 training step runs) directly against the C oracle's ROIPool
    (csrc/cuda/ROIPool_cuda.cu:17-108 restated): the clean rows of the stacked operand are the pooled maxima, the
    16-bit argmax the first maximal cell, bit for bit; the DropBlock rows are ((x * keep) * numel) / sum of them."""
    from oracle import native
    from od_wscl_amd import _lib as L
    from od_wscl_amd import synthetic
    from od_wscl_amd.utils import rng
    feat32 = torch.from_numpy(rng.normal(21, 1, C * H * W).reshape(1, C, H, W)).bfloat16()       # bf16-valued map
    feat32[0, :, 3, 4] = feat32[0, :, 3, 5]                                                       # ties: first cell wins
    feat32[0, :8, 40:44, 30:36] = 0.0                                                             # a window of +-0.0 maxima:
    feat32[0, :8, 40:44, 31] = -0.0                                                               # -0.0 == +0.0, first wins
    feat32[0, :8, 38:46, 28:38] = torch.minimum(feat32[0, :8, 38:46, 28:38], torch.zeros(()).bfloat16())
    nhwc = feat32[0].permute(1, 2, 0).contiguous().cuda()                                         # (H, W, C) bf16
    boxes = synthetic.make_proposals(21, 0, P, H * 8, W * 8, min_size=4)
    rois = np.concatenate([np.zeros((P, 1), np.float32), boxes], 1)
    rois[:4] = [[0, 0, 0, W * 8 - 1, H * 8 - 1], [0, 5, 5, 5, 5], [0, -30, -30, 9, 9], [0, W * 8 + 50, 10, W * 8 + 90, 40]]
    keep = (torch.from_numpy(rng.uniform(21, 3, P * 49).reshape(P, 49)) > 0.3).float().cuda()
    ksum = keep.sum()
    x = torch.empty((2 * P, C * 49), dtype=torch.bfloat16, device="cuda")
    arg = torch.empty((P, C * 49), dtype=torch.int16, device="cuda")
    lib = L.lib()
    r = torch.from_numpy(rois).cuda()
    ws_bytes = lib.odw_roi_pool_stack_nhwc_workspace(P, 1, C, H, W)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    L.check(lib.odw_roi_pool_stack_forward_nhwc(L.ptr(nhwc), L.ptr(r), 0.125, 1, C, H, W, P, L.ptr(keep), L.ptr(ksum),
                                                L.ptr(x), x.stride(0), L.ptr(arg), L.ptr(ws), ws_bytes, L.stream()), "nhwc")
    out, amax = native.roi_pool_fwd(feat32.float().numpy(), rois, 0.125, 7, 7)
    np.testing.assert_array_equal(x[:P].float().cpu().numpy().reshape(P, C, 7, 7), out)
    got_arg = arg.cpu().numpy().view(np.uint16).astype(np.int64).reshape(P, C, 7, 7)
    got_arg[got_arg == 0xFFFF] = -1
    np.testing.assert_array_equal(got_arg, amax)
    numel = np.float32(float(P) * 49)
    want_aug = (torch.from_numpy(out).reshape(P, C, 49) * keep.cpu()[:, None, :] * numel / ksum.cpu()).bfloat16().float()
    np.testing.assert_array_equal(x[P:].float().cpu().numpy().reshape(P, C, 49), want_aug.numpy())


def _bf16_rn(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


@pytest.mark.parametrize("P,C,H,W", [(2000, 512, 76, 76), (300, 128, 38, 50)])
def test_pool_stack_planes_f32_is_bit_exact_against_the_oracle(P, C, H, W):
    """roi_pool_stack_forward_nhwc_f32 (the pooling of the bench's default mode "bf16x2f": fp32 NHWC map -> bf16 planes
    [hi hi mid] of the stacked operand + fp32 pooled values + 16-bit argmax) against the C oracle's ROIPool
    (csrc/cuda/ROIPool_cuda.cu:17-108 restated): maxima and first-maximum positions bit for bit (ties, +-0.0 windows,
    empty and out-of-image ROIs), planes = the exact plane decomposition of the fp32 results."""
    import ctypes
    from oracle import native
    from od_wscl_amd import _lib as L
    from od_wscl_amd import synthetic
    from od_wscl_amd.utils import rng
    feat = rng.normal(22, 1, C * H * W).reshape(1, C, H, W).astype(np.float32)
    feat[0, :, 3, 4] = feat[0, :, 3, 5]                                  # ties: first cell wins
    feat[0, :8, 38:46, 28:38] = np.minimum(feat[0, :8, 38:46, 28:38] if H > 46 else 0.0, 0.0) if H > 46 else 0.0
    if H > 46:
        feat[0, :8, 40:44, 30:36] = 0.0                                  # a window of +-0.0 maxima: -0.0 == +0.0
        feat[0, :8, 40:44, 31] = -0.0
    nhwc = torch.from_numpy(np.ascontiguousarray(feat[0].transpose(1, 2, 0))).cuda()      # (H, W, C) fp32
    boxes = synthetic.make_proposals(22, 0, P, H * 8, W * 8, min_size=4)
    rois = np.concatenate([np.zeros((P, 1), np.float32), boxes], 1)
    rois[:4] = [[0, 0, 0, W * 8 - 1, H * 8 - 1], [0, 5, 5, 5, 5], [0, -30, -30, 9, 9], [0, W * 8 + 50, 10, W * 8 + 90, 40]]
    keep = (torch.from_numpy(rng.uniform(22, 3, P * 49).reshape(P, 49)) > 0.3).float().cuda()
    ksum = keep.sum()
    K = C * 49
    pat = (0, 0, 1)
    planes = torch.empty((2 * P, 3 * K), dtype=torch.bfloat16, device="cuda")
    pooled = torch.empty((P, K), dtype=torch.float32, device="cuda")
    arg = torch.empty((P, K), dtype=torch.int16, device="cuda")
    lib = L.lib()
    r = torch.from_numpy(rois).cuda()
    ws_bytes = lib.odw_roi_pool_stack_nhwc_f32_workspace(P, 1, C, H, W)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    cpat = (ctypes.c_int * 3)(*pat)
    L.check(lib.odw_roi_pool_stack_forward_nhwc_f32(L.ptr(nhwc), L.ptr(r), 0.125, 1, C, H, W, P, L.ptr(keep), L.ptr(ksum),
                                                    ctypes.cast(cpat, ctypes.c_void_p), 3, L.ptr(planes), planes.stride(0), K,
                                                    L.ptr(pooled), L.ptr(arg), L.ptr(ws), ws_bytes, L.stream()), "planes")
    out, amax = native.roi_pool_fwd(feat, rois, 0.125, 7, 7)
    np.testing.assert_array_equal(pooled.cpu().numpy().reshape(P, C, 7, 7), out)
    got_arg = arg.cpu().numpy().view(np.uint16).astype(np.int64).reshape(P, C, 7, 7)
    got_arg[got_arg == 0xFFFF] = -1
    np.testing.assert_array_equal(got_arg, amax)
    pl = planes.float().cpu().numpy()
    flat = out.reshape(P, K)
    hi = _bf16_rn(flat)
    mid = _bf16_rn((flat - hi).astype(np.float32))
    np.testing.assert_array_equal(pl[:P, :K], hi)
    np.testing.assert_array_equal(pl[:P, K:2 * K], hi)
    np.testing.assert_array_equal(pl[:P, 2 * K:], mid)
    numel = np.float32(float(P) * 49)
    aug = (torch.from_numpy(out).reshape(P, C, 49) * keep.cpu()[:, None, :] * numel / ksum.cpu()).reshape(P, K).numpy()
    ahi = _bf16_rn(aug)
    amid = _bf16_rn((aug - ahi).astype(np.float32))
    np.testing.assert_array_equal(pl[P:, :K], ahi)
    np.testing.assert_array_equal(pl[P:, K:2 * K], ahi)
    np.testing.assert_array_equal(pl[P:, 2 * K:], amid)
    # the layout of the shared clean + DropBlock fc6 forward: the hi plane of both halves for the backward, and the clean
    # rows as the two CELL-MAJOR planes (k' = bin * C + c)
    hi_only = torch.full((2 * P, K), -1.0, dtype=torch.bfloat16, device="cuda")
    cm = torch.full((P, 2 * K), -1.0, dtype=torch.bfloat16, device="cuda")
    pooled2, arg2 = torch.empty_like(pooled), torch.empty_like(arg)
    cpat1 = (ctypes.c_int * 1)(0)
    L.check(lib.odw_roi_pool_stack_forward_nhwc_f32_cm(L.ptr(nhwc), L.ptr(r), 0.125, 1, C, H, W, P, L.ptr(keep), L.ptr(ksum),
                                                       ctypes.cast(cpat1, ctypes.c_void_p), 1, L.ptr(hi_only), K, K, L.ptr(pooled2),
                                                       L.ptr(arg2), L.ptr(cm), 2 * K, K, L.ptr(ws), ws_bytes, L.stream()), "cm")
    assert torch.equal(pooled2, pooled) and torch.equal(arg2, arg)
    np.testing.assert_array_equal(hi_only.float().cpu().numpy(), np.concatenate([hi, ahi]))
    cmn = cm.float().cpu().numpy()
    np.testing.assert_array_equal(cmn[:, :K], hi.reshape(P, C, 49).transpose(0, 2, 1).reshape(P, K))
    np.testing.assert_array_equal(cmn[:, K:], mid.reshape(P, C, 49).transpose(0, 2, 1).reshape(P, K))


@pytest.mark.parametrize("dx_f32,skip_clean", [(False, False), (False, True), (True, False)])
def test_pool_stack_backward_matches_the_oracle_scatter(dx_f32, skip_clean):
    """roi_pool_stack_backward_ws (the pooling backward the training step runs: both halves of the stacked operand's
    gradient + parked extra rows, fixed-point accumulation) directly against the oracle: a float64 scatter-add of
    dX_clean + ((dX_aug * keep) * numel) / sum through the oracle's own arg-max -- and bit-identical from run to run."""
    from oracle import native
    from od_wscl_amd import _lib as L
    from od_wscl_amd import synthetic
    from od_wscl_amd.utils import rng
    P, C, H, W, E = 700, 128, 38, 50, 40
    feat = torch.from_numpy(rng.normal(31, 1, C * H * W).reshape(1, C, H, W)).bfloat16().float()
    boxes = synthetic.make_proposals(31, 0, P, H * 8, W * 8, min_size=4)
    rois = np.concatenate([np.zeros((P, 1), np.float32), boxes], 1)
    _, amax = native.roi_pool_fwd(feat.numpy(), rois, 0.125, 7, 7)                     # (P, C, 7, 7), -1 = empty bin
    arg16 = torch.from_numpy(np.where(amax < 0, 0xFFFF, amax).astype(np.uint16).view(np.int16).reshape(P, C * 49)).cuda()
    keep = (torch.from_numpy(rng.uniform(31, 3, P * 49).reshape(P, 49)) > 0.3).float().cuda()
    ksum = keep.sum()
    dx = torch.from_numpy(rng.normal(31, 4, 2 * P * C * 49).reshape(2 * P, C * 49)) * 1e-3
    dx = dx.cuda() if dx_f32 else dx.cuda().bfloat16()
    if skip_clean:
        dx[:P] = float("nan")                                                          # rows [0, P) are unset in the step
    extra = (torch.from_numpy(rng.normal(31, 5, E * C * 49).reshape(E, C * 49)) * 1e-3).cuda()
    extra_roi = torch.from_numpy((np.arange(E) * 17 % P).astype(np.int32)).cuda()
    r = torch.from_numpy(rois).cuda()
    lib = L.lib()
    outs = []
    for _ in range(2):
        gin = torch.empty((1, C, H, W), device="cuda")
        ws = torch.empty(64, dtype=torch.uint8, device="cuda")
        L.check(lib.odw_roi_pool_stack_backward_ws(L.ptr(dx), 1 if dx_f32 else 0, dx.stride(0), L.ptr(arg16), L.ptr(r), L.ptr(keep),
                                                   L.ptr(ksum), L.ptr(extra), L.ptr(extra_roi), E, 1 if skip_clean else 0, 1, C,
                                                   H, W, P, 7, 7, L.ptr(gin), L.ptr(ws), 64, L.stream()), "stack_bwd")
        outs.append(gin.clone())
    assert torch.equal(outs[0], outs[1])                                               # order-independent accumulation
    d = dx.float().cpu().double().numpy().reshape(2, P, C, 49)
    kp = keep.cpu().double().numpy()
    numel, ssum = np.float32(P * 49), float(ksum)
    g = (0.0 if skip_clean else d[0]) + d[1] * kp[:, None, :] * float(numel) / ssum
    if skip_clean:
        g = np.where(kp[:, None, :] == 0, 0.0, g)
    want = np.zeros((C, H * W))
    a = amax.reshape(P, C, 49)
    for c in range(C):
        ok = a[:, c] >= 0
        np.add.at(want[c], a[:, c][ok], g[:, c][ok])
    ex = extra.cpu().double().numpy().reshape(E, C, 49)
    er = extra_roi.cpu().numpy()
    for e in range(E):
        for c in range(C):
            ok = a[er[e], c] >= 0
            np.add.at(want[c], a[er[e], c][ok], ex[e, c][ok])
    got = outs[0].cpu().double().numpy().reshape(C, H * W)
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max() + 1e-12
