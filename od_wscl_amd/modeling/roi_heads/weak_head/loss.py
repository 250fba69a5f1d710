"""RoIRegLossComputation -- the OD-WSCL loss (wetectron/modeling/roi_heads/weak_head/loss.py:172-411,
contra branch): MIL image loss, IoU sampling, similarity-based object discovery, SupCon
contrastive loss, three refinement branches with box regression.

Same call signature and the same returned dict keys as the reference.  Device work goes to the
gfx950 kernels: box IoU, torchvision-semantics NMS, fused pseudo-label assignment, fused SupCon
forward/backward; only the (few) similarity ROWS that object discovery reads are computed, never
the P x P matrix the reference rebuilds per (image, branch, class) (loss.py:319).
Reference quirks Q1-Q12 are kept."""
import torch
from torch.nn import functional as F

from ... import registry
from ....layers import smooth_l1_loss
from ....structures.boxlist_ops import boxlist_iou
from ....structures.bounding_box import BoxList
from .... import _C
from ..sim_head.sim_loss import SupConLossV2
from .pseudo_label_generator import od_layer, oicr_layer


@torch.no_grad()
def generate_img_label(num_classes, labels, device):
    """utils/utils.py:52-57 (built on the device; the reference builds it on the host)."""
    v = torch.zeros(num_classes, device=device)
    v[labels.long()] = 1
    v[0] = 0
    return v


@torch.no_grad()
def cal_iou(proposal, target_index, iou_thres=1e-5):
    """Indices (ascending) of proposals whose IoU with proposal[target_index] is >= iou_thres
    (utils/utils.py:22-26)."""
    iou = _C.box_iou(proposal.bbox, proposal.bbox[target_index.view(-1)])
    idx = torch.nonzero(torch.ge(iou, iou_thres).max(dim=1)[0]).view(-1)
    return idx, iou[idx]


@torch.no_grad()
def easy_nms(proposals, cluster, source_score, nms_iou=0.1):
    """utils/utils.py:28-33: NMS inside `cluster`, survivors in descending-score order (Q10)."""
    keep = _C.nms_torchvision(proposals.bbox[cluster], source_score[cluster], nms_iou)
    return cluster[keep]


def compute_avg_img_accuracy(labels_per_im, score_per_im, num_classes):
    """Top-k accuracy with k = number of positive classes (loss.py:25-33)."""
    k = max(int(labels_per_im.sum().int().item()), 1)
    return labels_per_im[score_per_im.topk(k)[1]].mean()


@registry.ROI_WEAK_LOSS.register("RoIRegLoss")
class RoIRegLossComputation(object):
    def __init__(self, cfg):
        self.contra = cfg.SOLVER.CONTRA
        self.refine_p = cfg.MODEL.ROI_WEAK_HEAD.OICR_P
        self.cls_agnostic_bbox_reg = cfg.MODEL.CLS_AGNOSTIC_BBOX_REG
        self.od_layer = od_layer(cfg)
        self.oicr_layer = oicr_layer(cfg)
        self.nms = cfg.nms
        self.sim_lmda = cfg.lmda
        self.p_thres = cfg.thres
        self.temp = cfg.temp
        if cfg.nms <= 0:
            raise ValueError("cfg.nms must be > 0 (Q11: the reference breaks on nms <= 0)")
        if cfg.loss != "supconv2":
            raise ValueError("only loss='supconv2' is functional in the reference (SURVEY.md item 5)")
        self.sim_loss = SupConLossV2(self.temp)
        self.trace = None       # set to a dict to record the selected index sets (tests)
        self.max_sampled_rows = int(getattr(getattr(cfg, "ODW", None), "MAX_SAMPLED_ROWS", 4096))

    def __call__(self, class_score, det_score, ref_scores, ref_bbox_preds, sim_feature, clean_pooled_feats,
                 feature_extractor, model_sim, proposals, targets, epsilon=1e-8):
        # selection logic and loss arithmetic always run in fp32 (the reference's DTYPE, defaults.py:559)
        def f32(t):
            if callable(t):             # a deferred evaluation (weak_head: Sim_Net on the clean pass): stays deferred
                return lambda: t().float()
            return t.float()
        return self._call([f32(t) for t in class_score], [f32(t) for t in det_score],
                          [f32(t) for t in ref_scores], [f32(t) for t in ref_bbox_preds], f32(sim_feature),
                          clean_pooled_feats, feature_extractor, model_sim, proposals, targets, epsilon)

    def _neck_embed(self, feature_extractor, model_sim, pooled):
        return model_sim(feature_extractor.forward_neck(pooled)).float()

    def _call(self, class_score, det_score, ref_scores, ref_bbox_preds, sim_feature, clean_pooled_feats,
              feature_extractor, model_sim, proposals, targets, epsilon=1e-8):
        if callable(sim_feature):
            sim_feature = sim_feature()
        sizes = [len(p) for p in proposals]
        n_img = len(sizes)
        class_score = F.softmax(torch.cat(class_score, dim=0), dim=1)
        det = torch.cat(det_score, dim=0)
        final_det = torch.cat([F.softmax(d, dim=0) for d in det.split(sizes)], dim=0)
        final_score = class_score * final_det
        final_list = final_score.split(sizes)
        device = class_score.device
        C = class_score.shape[1]
        ref_split = [r.split(sizes) for r in ref_scores]
        box_split = [b.split(sizes) for b in ref_bbox_preds]
        n_ref = len(ref_scores)
        tr = self.trace

        losses = dict(loss_img=0)
        accs = dict(acc_img=0)
        for i in range(n_ref):
            losses["loss_ref_cls%d" % i] = 0
            losses["loss_ref_reg%d" % i] = 0
            accs["acc_ref%d" % i] = 0

        lab_vecs = [generate_img_label(C, t.get_field("labels").unique(), device) for t in targets]
        pos_classes = [v[1:].eq(1).nonzero(as_tuple=False)[:, 0].tolist() for v in lab_vecs]

        def source(idx, i):
            return final_list[idx] if i == 0 else F.softmax(ref_split[i - 1][idx], dim=1)

        pgt_instance = [[None] * n_ref for _ in range(n_img)]
        if self.contra:
            sim_list = sim_feature.split(sizes)
            pooled_list = clean_pooled_feats.split(sizes)
            new_l = lambda: torch.zeros(0, dtype=torch.long, device=device)
            new_f = lambda: torch.zeros(0, dtype=torch.float, device=device)
            pgt_index = [[new_l() for _ in range(C - 1)] for _ in range(n_img)]
            pgt_collection = [new_f() for _ in range(C - 1)]
            pgt_update = [new_f() for _ in range(C - 1)]
            instance_diff = new_f()
            tops = {}

            # -- loop 1: IoU sampling around each branch's top proposal (loss.py:281-307)
            for idx in range(n_img):
                props = proposals[idx]
                for i in range(n_ref):
                    pscore = source(idx, i)[:, 1:]
                    for c in pos_classes[idx]:
                        top = torch.argmax(pscore[:, c])
                        tops[(idx, i, c)] = top
                        near, _ = cal_iou(props, top, self.p_thres)
                        pgt_index[idx][c] = torch.cat((pgt_index[idx][c], near)).unique()
                for c in pos_classes[idx]:
                    rows = pgt_index[idx][c]
                    col = final_list[idx][:, c + 1]
                    hardness = col[rows] / col.sum()                      # Q12
                    pgt_update[c] = torch.cat((pgt_update[c], sim_list[idx][rows]))
                    instance_diff = torch.cat((instance_diff, hardness))
                    picked = pooled_list[idx][rows]
                    drop = self._neck_embed(feature_extractor, model_sim, feature_extractor.drop_pool(picked))
                    pgt_update[c] = torch.cat((pgt_update[c], drop))
                    instance_diff = torch.cat((instance_diff, hardness))
                    noisy = self._neck_embed(feature_extractor, model_sim, feature_extractor.noise_pool(picked))
                    pgt_update[c] = torch.cat((pgt_update[c], noisy))
                    instance_diff = torch.cat((instance_diff, hardness))
                    pgt_collection[c] = pgt_update[c].clone()             # Q2
                    if tr is not None:
                        tr["iou_samples_%d_%d" % (idx, c)] = rows.clone()

            # -- loop 2: similarity-based object discovery (loss.py:311-345)
            for idx in range(n_img):
                props = proposals[idx]
                E = sim_list[idx]
                for i in range(n_ref):
                    pscore = source(idx, i)[:, 1:]
                    inst = [new_l() for _ in range(C - 1)]
                    for c in pos_classes[idx]:
                        top = tops[(idx, i, c)]
                        with torch.no_grad():
                            row = torch.mv(E, E[top])                     # sim_mat[top] without the P x P matrix
                            thr = torch.mv(pgt_collection[c], E[top]).mean()
                            close = torch.ge(row, thr)
                            if len(pos_classes[idx]) > 1:
                                for nc in pos_classes[idx]:
                                    if nc == c:
                                        continue
                                    nrow = torch.mv(E, E[tops[(idx, i, nc)]])
                                    close = torch.ge(close.float(), nrow)  # Q3: bool (0/1) >= similarity
                            close = close.nonzero(as_tuple=False).view(-1)
                            close = easy_nms(props, close, pscore[:, c], nms_iou=self.nms)
                            if close.nelement() == 0:
                                close = top.view(-1)
                            inst[c] = torch.cat((inst[c], close))
                            known = pgt_index[idx][c]
                            fresh = close[~torch.isin(close, known)].unique()   # == the reference's
                            if fresh.nelement() == 0:                           # unique-counts difference
                                fresh = top.view(-1)
                        pgt_update[c] = torch.cat((pgt_update[c], E[fresh]))
                        pgt_index[idx][c] = torch.cat((known, fresh)).unique()
                        col = final_list[idx][:, c + 1]
                        instance_diff = torch.cat((instance_diff, (col[fresh] / col.sum()).view(-1)))
                        if tr is not None:
                            tr["pgt_instance_%d_%d_%d" % (idx, i, c)] = close.clone()
                            tr["sim_new_%d_%d_%d" % (idx, i, c)] = fresh.clone()
                    pgt_instance[idx][i] = inst
            if tr is not None:
                tr["supcon_weights"] = instance_diff.detach().clone()
                tr["supcon_n"] = int(instance_diff.numel())
            losses["loss_sim"] = self.sim_lmda * self.sim_loss(pgt_update, instance_diff, device)   # Q8

        # -- loop 3: MIL image loss + refinement branches (loss.py:349-400)
        for idx in range(n_img):
            props = proposals[idx]
            lab = lab_vecs[idx]
            img_score = torch.clamp(final_list[idx].sum(dim=0), min=epsilon, max=1 - epsilon)
            losses["loss_img"] = losses["loss_img"] + F.binary_cross_entropy(img_score, lab.clamp(0, 1))
            for i in range(n_ref):
                src = source(idx, i)
                if self.contra:
                    pseudo, weights, targets_reg = self.od_layer(props, src, lab, device, pgt_instance[idx][i], True)
                else:
                    pseudo, weights, targets_reg = self.oicr_layer(props, src, lab, device, True)
                if tr is not None:
                    tr["pseudo_%d_%d" % (idx, i)] = pseudo.clone()
                    tr["weights_%d_%d" % (idx, i)] = weights.clone()
                lam = 3 if i == 0 else 1
                ce = F.cross_entropy(ref_split[i][idx], pseudo, reduction="none")
                losses["loss_ref_cls%d" % i] = losses["loss_ref_cls%d" % i] + lam * torch.mean(ce * weights)
                pos = torch.nonzero(pseudo > 0, as_tuple=False).squeeze(1)
                lab_pos = pseudo[pos]
                if self.cls_agnostic_bbox_reg:
                    cols = torch.tensor([4, 5, 6, 7], device=device).expand(pos.numel(), 4)
                else:
                    cols = 4 * lab_pos[:, None] + torch.tensor([0, 1, 2, 3], device=device)
                reg = lam * torch.sum(smooth_l1_loss(box_split[i][idx][pos[:, None], cols], targets_reg[pos],
                                                     beta=1, reduction=False) * weights[pos, None])
                losses["loss_ref_reg%d" % i] = losses["loss_ref_reg%d" % i] + reg / pseudo.numel()
            with torch.no_grad():
                accs["acc_img"] = accs["acc_img"] + compute_avg_img_accuracy(lab, img_score, C)
                for i in range(n_ref):
                    rs = torch.sum(ref_split[i][idx], dim=0)
                    accs["acc_ref%d" % i] = accs["acc_ref%d" % i] + compute_avg_img_accuracy(lab[1:], rs[1:], C)

        for k in losses:
            if "sim" not in k:
                losses[k] = losses[k] / n_img
        for k in accs:
            accs[k] = accs[k] / n_img
        return losses, accs


def make_roi_weak_loss_evaluator(cfg):
    """`RoIRegLoss` (the reference's registry key) resolves to the device-side implementation
    (loss_fused.RoIRegLossFused) unless cfg.ODW.LOSS_IMPL == "loops" asks for the straight-line
    restatement above; both reproduce the reference and are tested against the same golden vectors."""
    from . import loss_fused  # noqa: F401  (registers RoIRegLossFused)
    name = cfg.MODEL.ROI_WEAK_HEAD.LOSS
    impl = cfg.ODW.LOSS_IMPL if "ODW" in cfg else "fused"
    if name == "RoIRegLoss" and impl == "fused":
        name = "RoIRegLossFused"
    return registry.ROI_WEAK_LOSS[name](cfg)
