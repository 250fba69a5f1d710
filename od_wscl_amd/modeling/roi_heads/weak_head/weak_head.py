"""ROIWeakRegHead (wetectron/modeling/roi_heads/weak_head/weak_head.py:72-157), training path:
clean pass -> Sim_Net embedding -> DropBlock pass -> predictor -> OD-WSCL loss."""
import torch
from torch import nn

from ... import registry
from ..sim_head.sim_net import Sim_Net
from ..box_head.inference import make_roi_box_post_processor
from .loss import make_roi_weak_loss_evaluator
from .roi_weak_predictors import make_roi_weak_predictor


class ROIWeakRegHead(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        name = cfg.MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR
        self.feature_extractor = registry.ROI_BOX_FEATURE_EXTRACTORS[name](cfg, in_channels)
        self.predictor = make_roi_weak_predictor(cfg, self.feature_extractor.out_channels)
        self.loss_evaluator = make_roi_weak_loss_evaluator(cfg)
        self.HEUR = cfg.MODEL.ROI_WEAK_HEAD.REGRESS_HEUR
        self.DB_METHOD = cfg.DB.METHOD
        self.model_sim = Sim_Net(cfg, self.feature_extractor.out_channels)
        self.weak_post_processor = make_roi_box_post_processor(cfg, regression=False)
        self.strong_post_processor = make_roi_box_post_processor(cfg, regression=True)

    head_grads_ready = None      # engine.FlatSGD installs its early-step callback here

    def _on_pooled_grad(self, grad):
        self.head_grads_ready()
        return grad

    def set_rand(self, rand):
        """Install the counter-based random source for this step (dropout / DropBlock / noise)."""
        self.feature_extractor.rand = rand

    def go_through_cdb(self, pooled, proposals):
        if not self.training or self.DB_METHOD == "none":
            return pooled
        if self.DB_METHOD == "dropblock":
            return self.feature_extractor.forward_dropblock(pooled, proposals)
        raise ValueError("DB.METHOD %r is outside the OD-WSCL hot path" % self.DB_METHOD)

    def testing_forward(self, cls_score, det_score, proposals, ref_scores=None, ref_bbox_preds=None):
        """weak_head.py:124-145: scores already softmax-ed by the predictor's eval branch."""
        if self.HEUR == "WSDDN":
            return self.weak_post_processor(cls_score * det_score, proposals)
        if self.HEUR == "CLS-AVG":
            return self.weak_post_processor(torch.mean(torch.stack(ref_scores), dim=0), proposals)
        if self.HEUR == "AVG":
            final_score = torch.mean(torch.stack(ref_scores), dim=0)
            final_regression = torch.mean(torch.stack(ref_bbox_preds), dim=0)
            return self.strong_post_processor((final_score, final_regression), proposals, softmax_on=False)
        raise ValueError("REGRESS_HEUR %r is outside the OD-WSCL path" % self.HEUR)

    def forward(self, features, proposals, targets=None, model_cdb=None, iteration=None):
        fe = self.feature_extractor
        if not self.training:
            clean_feats, clean_pooled = fe.forward(features, proposals)
            cls, det, refs, boxes = self.predictor(clean_feats, proposals)
            return clean_feats, self.testing_forward(cls, det, proposals, refs, boxes), {}, {}
        if fe.rand is not None and self.DB_METHOD in ("dropblock", "none"):
            # clean pass + DropBlock pass as one stacked fc6/fc7 evaluation (same draws, same order)
            if self.head_grads_ready is not None and features[0].requires_grad:
                # every gradient of the head (fc6/fc7, Sim_Net, predictor) is final once the ROI pooling node has
                # produced d(loss)/d(features): the optimiser / all-reduce of those 600 MB can start while the
                # backbone is still in backward.  (Hooked after the pooling backward, not before it: that kernel is
                # as HBM-bound as the optimiser and the two would only slow each other down.)
                features[0].register_hook(self._on_pooled_grad)
            if self.DB_METHOD == "dropblock" and fe.can_pool_stack(features):
                # pooling writes the stacked bf16 fc6 operand itself; it stands in for `clean_pooled` in the loss
                clean_feats, aug_feats, clean_pooled = fe.forward_pool_clean_and_aug(features, proposals)
            else:
                clean_pooled = fe.forward_pooler(features, proposals)
                clean_feats, aug_feats = fe.forward_clean_and_aug(clean_pooled)
            if getattr(fe, "sparse_clean", False) and clean_pooled.dim() == 2:
                # all P embeddings drive the selection; the rows the loss differentiates are re-evaluated
                # (fe.recompute_clean_rows).  Deferred: the fused loss launches this evaluation BEHIND its first
                # selection kernel, so that it runs while the host reads that kernel's result (loss_fused.py)
                def sim_feature(model_sim=self.model_sim, clean_feats=clean_feats):
                    with torch.no_grad():
                        return model_sim(clean_feats, keep=True)       # its two Linear outputs are reused, not recomputed
            else:
                sim_feature = self.model_sim(clean_feats)
        else:
            clean_feats, clean_pooled = fe.forward(features, proposals)
            sim_feature = self.model_sim(clean_feats)
            aug_feats = fe.forward_neck(self.go_through_cdb(clean_pooled, proposals))
        cls, det, refs, boxes = self.predictor(aug_feats, proposals)
        loss, acc = self.loss_evaluator([cls], [det], refs, boxes, sim_feature, clean_pooled,
                                        self.feature_extractor, self.model_sim, proposals, targets)
        return aug_feats, proposals, loss, acc


def build_roi_weak_head(cfg, in_channels):
    if not cfg.MODEL.ROI_WEAK_HEAD.REGRESS_ON:
        raise NotImplementedError("only the regression head (ROIWeakRegHead) is on the OD-WSCL hot path")
    return ROIWeakRegHead(cfg, in_channels)
