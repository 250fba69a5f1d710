"""CPU restatement of OD-WSCL's proposal-feature hot path -- TEST INFRASTRUCTURE ONLY.

Plain PyTorch-CPU fp32 + the C oracle for ROIPool/ROIAlign/NMS.  This is the
checker the HIP path is compared against and the "port" that bench.py times as
cpu_baseline; the product (od_wscl_amd/) never imports it.

It follows the reference function by function (paths relative to
/root/reference/wetectron), INCLUDING the quirks Q1-Q12 listed in SURVEY.md
s8a -- they are reproduced, not fixed.  Randomness (dropout, DropBlock centres,
the noise view) is injected through `Rand`, a counter-based generator that the
golden-vector script also injects into the imported reference, consumed in the
reference's call order.

Pinned by tests/golden/e2e_*.npz (outputs of the imported reference on the
same formula-generated inputs; see tests/golden/make_golden.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import rng_ref as _rng          # the oracle's own restatement of the generator (not the product's)
from . import native

# modeling/backbone/vgg16.py:86-93 'VGG16-OICR': conv channels, 'M' maxpool, 'I' identity
# (pool4 removed), trailing three convs dilated by 2.  (cout, dilation) or a marker.
VGG16_OICR = [(64, 1), (64, 1), "M", (128, 1), (128, 1), "M", (256, 1), (256, 1), (256, 1), "M",
              (512, 1), (512, 1), (512, 1), "I", (512, 2), (512, 2), (512, 2)]


def vgg_conv_indices():
    """features.N index of each conv (vgg16.py:58-83: conv,relu per entry; M and I take one slot)."""
    idx, out = 0, []
    for v in VGG16_OICR:
        if isinstance(v, tuple):
            out.append(idx)
            idx += 2
        else:
            idx += 1
    return out


RESNET_BLOCKS = {"r50": (3, 4, 6, 3), "r101": (3, 4, 23, 3)}


def resnet_convs(arch):
    """Ordered (prefix, cin, cout, k) of every conv of a *-C5 body in module-registration order
    (backbone/resnet.py:84-126, :258-343: downsample is registered before conv1)."""
    out = [("backbone.body.stem.conv1", 3, 64, 7)]
    cin = 64
    for li, count in enumerate(RESNET_BLOCKS[arch]):
        mid, cout = 64 << li, 256 << li
        for b in range(count):
            pre = "backbone.body.layer%d.%d." % (li + 1, b)
            if b == 0:
                out.append((pre + "downsample.0", cin, cout, 1))
            out += [(pre + "conv1", cin, mid, 1), (pre + "conv2", mid, mid, 3), (pre + "conv3", mid, cout, 1)]
            cin = cout
    return out


def resnet_buffer_shapes(arch):
    """Ordered (name, shape) of the frozen batch-norm buffers (layers/batch_norm.py:12-17)."""
    out = []
    for pre, _cin, cout, _k in resnet_convs(arch):
        bn = pre[:-len("conv1")] + "bn" + pre[-1] if not pre.endswith("downsample.0") else pre[:-1] + "1"
        for f in ("weight", "bias", "running_mean", "running_var"):
            out.append((bn + "." + f, (cout,)))
    return out


def param_shapes(num_classes=21, arch="vgg16"):
    """Ordered (name, shape) of every parameter, reference names."""
    shapes = []
    fe = "roi_heads.feature_extractor.classifier."
    if arch == "vgg16":
        cin = 3
        for i, v in zip(vgg_conv_indices(), [v for v in VGG16_OICR if isinstance(v, tuple)]):
            shapes.append(("backbone.body.features.%d.weight" % i, (v[0], cin, 3, 3)))
            shapes.append(("backbone.body.features.%d.bias" % i, (v[0],)))
            cin = v[0]
        shapes += [(fe + "1.weight", (4096, 512 * 7 * 7)), (fe + "1.bias", (4096,)),
                   (fe + "4.weight", (4096, 4096)), (fe + "4.bias", (4096,))]
    else:
        shapes += [(pre + ".weight", (cout, cin, k, k)) for pre, cin, cout, k in resnet_convs(arch)]
        # roi_box_feature_extractors.py:58-65: Linear(7*7*2048, 2048), Linear(2048, 4096) at positions 0 and 3
        shapes += [(fe + "0.weight", (2048, 7 * 7 * 2048)), (fe + "0.bias", (2048,)),
                   (fe + "3.weight", (4096, 2048)), (fe + "3.bias", (4096,))]
    pr = "roi_heads.predictor."
    for name, n in (("cls_score", num_classes), ("det_score", num_classes), ("ref1", num_classes),
                    ("bbox_pred1", 4 * num_classes), ("ref2", num_classes), ("bbox_pred2", 4 * num_classes),
                    ("ref3", num_classes), ("bbox_pred3", 4 * num_classes)):
        shapes += [(pr + name + ".weight", (n, 4096)), (pr + name + ".bias", (n,))]
    sm = "roi_heads.model_sim.mlp."
    shapes += [(sm + "0.weight", (4096, 4096)), (sm + "0.bias", (4096,)),
               (sm + "2.weight", (128, 4096)), (sm + "2.bias", (128,))]
    return shapes


# features.{0,2,5,7} are frozen: FREEZE_CONV_BODY_AT=2 (vgg16.py:48-55, config/defaults.py:128)
FROZEN = tuple("backbone.body.features.%d." % i for i in (0, 2, 5, 7))
# stem + layer1 are frozen for the ResNets (resnet.py:128-137 with the same FREEZE_CONV_BODY_AT=2)
FROZEN_RESNET = ("backbone.body.stem.", "backbone.body.layer1.")


class Rand(object):
    """Injected randomness: one stream id per logical draw, in reference call order."""

    def __init__(self, seed, first_stream=1 << 20):
        self.s = _rng.Streams(seed, first_stream)

    def uniform(self, shape):
        n = int(np.prod(shape))
        return torch.from_numpy(_rng.uniform(self.s.seed, self.s.take(), n).reshape(shape))

    def normal(self, shape):
        n = int(np.prod(shape))
        return torch.from_numpy(_rng.normal(self.s.seed, self.s.take(), n).reshape(shape))

    def dropout(self, x, p=0.5):
        """F.dropout(x, p, training=True): keep where u >= p, scale 1/(1-p)  (vgg16.py:124,127)."""
        keep = (self.uniform(tuple(x.shape)) >= p).to(x.dtype)
        return x * keep * (1.0 / (1.0 - p))


# --------------------------------------------------------------------------- native ops
class _RoiPoolFn(torch.autograd.Function):
    """layers/roi_pool.py:11-43 on the C oracle (ROIPool_cuda.cu restated)."""

    @staticmethod
    def forward(ctx, feat, rois, ph, pw, scale):
        out, arg = native.roi_pool_fwd(feat.detach().numpy(), rois.numpy(), scale, ph, pw)
        ctx.save = (arg, rois.numpy().copy(), tuple(feat.shape), ph, pw)
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, g):
        arg, rois, shape, ph, pw = ctx.save
        return torch.from_numpy(native.roi_pool_bwd(g.contiguous().numpy(), arg, rois, shape, ph, pw)), None, None, None, None


class _RoiAlignFn(torch.autograd.Function):
    """layers/roi_align.py:11-46 on the C oracle."""

    @staticmethod
    def forward(ctx, feat, rois, ph, pw, scale, sr):
        ctx.save = (rois.numpy().copy(), tuple(feat.shape), ph, pw, scale, sr)
        return torch.from_numpy(native.roi_align_fwd(feat.detach().numpy(), rois.numpy(), scale, ph, pw, sr))

    @staticmethod
    def backward(ctx, g):
        rois, shape, ph, pw, scale, sr = ctx.save
        return torch.from_numpy(native.roi_align_bwd(g.contiguous().numpy(), rois, scale, shape, ph, pw, sr)), None, None, None, None, None


def rois_with_batch_index(boxes_per_image):
    """modeling/poolers.py:85-96."""
    parts = [torch.cat([torch.full((len(b), 1), float(i)), b], dim=1) for i, b in enumerate(boxes_per_image)]
    return torch.cat(parts, dim=0)


def boxlist_iou(a, b):
    """structures/boxlist_ops.py:127-160 (+1 convention), torch ops in the same order."""
    area1 = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    area2 = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    lt = torch.max(a[:, None, :2], b[:, :2])
    rb = torch.min(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (area1[:, None] + area2 - inter)


def nms_tv(boxes, scores, thr):
    return torch.from_numpy(native.nms_tv(boxes.numpy(), scores.numpy(), thr))


def box_encode(gt, prop, weights=(10.0, 10.0, 5.0, 5.0)):
    """modeling/box_coder.py:22-50."""
    ew = prop[:, 2] - prop[:, 0] + 1
    eh = prop[:, 3] - prop[:, 1] + 1
    ex = prop[:, 0] + 0.5 * ew
    ey = prop[:, 1] + 0.5 * eh
    gw = gt[:, 2] - gt[:, 0] + 1
    gh = gt[:, 3] - gt[:, 1] + 1
    gx = gt[:, 0] + 0.5 * gw
    gy = gt[:, 1] + 0.5 * gh
    wx, wy, ww, wh = weights
    return torch.stack((wx * (gx - ex) / ew, wy * (gy - ey) / eh,
                        ww * torch.log(gw / ew), wh * torch.log(gh / eh)), dim=1)


def smooth_l1(x, t, beta=1.0):
    """layers/smooth_l1_loss.py:4-16 with reduction=False."""
    n = (x - t).abs()
    return torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)


# --------------------------------------------------------------------------- model pieces
def backbone_forward(x, sd):
    """VGG_Base.forward (vgg16.py:34-36,58-83): last ReLU dropped."""
    convs = vgg_conv_indices()
    k = 0
    n_conv = len(convs)
    for v in VGG16_OICR:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
        elif v == "I":
            pass
        else:
            i = convs[k]
            x = F.conv2d(x, sd["backbone.body.features.%d.weight" % i], sd["backbone.body.features.%d.bias" % i],
                         padding=v[1], dilation=v[1])
            k += 1
            if k < n_conv:
                x = F.relu(x)
    return x


def _frozen_bn(x, sd, name):
    """FrozenBatchNorm2d.forward (layers/batch_norm.py:19-31), same operation order."""
    scale = sd[name + ".weight"] * sd[name + ".running_var"].rsqrt()
    bias = sd[name + ".bias"] - sd[name + ".running_mean"] * scale
    return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


def resnet_forward(x, sd, arch="r50"):
    """ResNet.forward of a *-C5 body (backbone/resnet.py:139-146, stem :398-403, Bottleneck :350-375) with
    STRIDE_IN_1X1 and the layer4 stride patch of GeneralizedRCNN.__init__ (detector/generalized_rcnn.py:37-45):
    strides 1,2,2,1 on the first block of layer1..4, carried by conv1 and the downsample conv."""
    p = "backbone.body."
    x = F.relu(_frozen_bn(F.conv2d(x, sd[p + "stem.conv1.weight"], stride=2, padding=3), sd, p + "stem.bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, count in enumerate(RESNET_BLOCKS[arch]):
        for b in range(count):
            q = p + "layer%d.%d." % (li + 1, b)
            stride = 2 if (b == 0 and li in (1, 2)) else 1
            y = F.relu(_frozen_bn(F.conv2d(x, sd[q + "conv1.weight"], stride=stride), sd, q + "bn1"))
            y = F.relu(_frozen_bn(F.conv2d(y, sd[q + "conv2.weight"], padding=1), sd, q + "bn2"))
            y = _frozen_bn(F.conv2d(y, sd[q + "conv3.weight"]), sd, q + "bn3")
            if b == 0:
                x = _frozen_bn(F.conv2d(x, sd[q + "downsample.0.weight"], stride=stride), sd, q + "downsample.1")
            x = F.relu(y + x)
    return x


def neck(pooled, sd, rand):
    """forward_neck of either extractor (vgg16.py:159-162 -- classifier.{1,4};
    roi_box_feature_extractors.py:90-97 -- classifier.{0,3}): Linear,ReLU,Dropout,Linear,ReLU,Dropout."""
    fe = "roi_heads.feature_extractor.classifier."
    a, b = ("1", "4") if (fe + "1.weight") in sd else ("0", "3")
    x = pooled.reshape(pooled.shape[0], -1)
    x = rand.dropout(F.relu(F.linear(x, sd[fe + a + ".weight"], sd[fe + a + ".bias"])))
    x = rand.dropout(F.relu(F.linear(x, sd[fe + b + ".weight"], sd[fe + b + ".bias"])))
    return x


def sim_net(x, sd):
    """Sim_Net.forward (roi_heads/sim_head/sim_net.py:25-26)."""
    sm = "roi_heads.model_sim.mlp."
    h = F.relu(F.linear(x, sd[sm + "0.weight"], sd[sm + "0.bias"]))
    return F.normalize(F.linear(h, sd[sm + "2.weight"], sd[sm + "2.bias"]), dim=1)


def dropblock(x, block_size, drop_prob, rand):
    """DropBlock2D.forward in training (modeling/dropblock/drop_block.py:29-71)."""
    gamma = drop_prob / (block_size ** 2)
    mask = (rand.uniform((x.shape[0], x.shape[2], x.shape[3])) < gamma).float()
    block = F.max_pool2d(mask[:, None], kernel_size=block_size, stride=1, padding=block_size // 2)
    if block_size % 2 == 0:
        block = block[:, :, :-1, :-1]
    block = 1 - block.squeeze(1)
    out = x * block[:, None, :, :]
    return out * block.numel() / block.sum()


def noise_pool(x, rand):
    """vgg16.py:177-180."""
    noise = rand.normal(tuple(x.shape))
    return noise * x + x


def predictor(x, sd):
    """MISTPredictor.forward, training branch (roi_weak_predictors.py:158-187): raw logits."""
    pr = "roi_heads.predictor."
    lin = lambda n: F.linear(x, sd[pr + n + ".weight"], sd[pr + n + ".bias"])
    return (lin("cls_score"), lin("det_score"), [lin("ref1"), lin("ref2"), lin("ref3")],
            [lin("bbox_pred1"), lin("bbox_pred2"), lin("bbox_pred3")])


def supcon_v2(pgt_update, instance_diff, temperature):
    """SupConLossV2.forward (roi_heads/sim_head/sim_loss.py:49-80), Q1 included:
    features are class-major, weights stay in append order."""
    feats, labels = [], []
    for c, emb in enumerate(pgt_update):
        if emb.shape[0] != 0:
            feats.append(emb)
            labels.append(torch.full((emb.shape[0],), float(c)))
    features = torch.cat(feats)
    labels = torch.cat(labels)
    w = instance_diff.detach()
    sim = torch.matmul(features, features.T) / temperature
    sim = sim - sim.max(dim=1, keepdim=True)[0].detach()
    logits_mask = torch.ones_like(sim)
    logits_mask.fill_diagonal_(0)
    e = torch.exp(sim)
    label_mask = torch.eq(labels.view(-1, 1), labels.view(-1, 1).T).float()
    log_prob = torch.log((e * logits_mask * label_mask).sum(1) / (e * logits_mask).sum(1))
    return (-log_prob * w).mean(), features, labels, w


def image_label_vector(num_classes, labels):
    """utils/utils.py:52-57."""
    v = torch.zeros(num_classes)
    v[labels.long()] = 1
    v[0] = 0
    return v


@torch.no_grad()
def od_layer(boxes, source_score, labels_vec, pgt_instance, fg_thresh=0.5):
    """od_layer.__call__ (roi_heads/weak_head/pseudo_label_generator.py:135-197), Q5 included."""
    prob = source_score[:, 1:].clone()
    gt_boxes, gt_classes, gt_scores = [], [], []
    for c in labels_vec[1:].eq(1).nonzero(as_tuple=False)[:, 0]:
        c = int(c)
        col = prob[:, c]
        top = int(torch.argmax(col))
        sim_box = pgt_instance[c]
        if sim_box.numel() == 0:
            gt_boxes.append(boxes[top].view(1, 4))
            gt_classes.append(torch.tensor([c + 1]))
            gt_scores.append(col[top].view(1).clone())
        else:
            gt_boxes.append(boxes[sim_box])
            gt_classes.append(torch.full((sim_box.numel(),), c + 1, dtype=torch.long))
            gt_scores.append(col[sim_box].clone())
        prob[top].fill_(0)  # zeroes the WHOLE row of the top proposal (:159,:165)
    P = source_score.shape[0]
    if not gt_boxes:
        return torch.zeros(P, dtype=torch.long), torch.zeros(P), None
    gt_boxes = torch.cat(gt_boxes)
    gt_classes = torch.cat(gt_classes)
    gt_scores = torch.cat(gt_scores)
    overlaps = boxlist_iou(boxes, gt_boxes).numpy()
    max_ov = torch.from_numpy(overlaps.max(axis=1))
    assign = torch.from_numpy(overlaps.argmax(axis=1))     # numpy first-max on host (:176-177)
    pseudo = gt_classes[assign].clone()
    weights = gt_scores[assign].clone()
    pseudo[max_ov.le(fg_thresh)] = 0                       # bg test is <= (:183)
    targets = box_encode(gt_boxes[assign], boxes)
    return pseudo, weights, targets


def topk_accuracy(labels_vec, scores):
    """compute_avg_img_accuracy (roi_heads/weak_head/loss.py:25-33)."""
    k = max(int(labels_vec.sum().int().item()), 1)
    pred = scores.topk(k)[1]
    return labels_vec[pred].mean()


def _note_margin(tr, kind, value):
    """Record how close a data-dependent decision was to flipping (smaller = more fragile).
    The golden cases are chosen so that every margin is far above fp32 re-ordering noise."""
    key = "margin/" + kind
    v = float(value)
    if key not in tr or v < float(tr[key]):
        tr[key] = torch.tensor(v)


def _top2_gap(col):
    v = torch.topk(col.detach(), min(2, col.numel()))[0]
    return ((v[0] - v[1]) / v[0].abs().clamp(min=1e-30)) if v.numel() > 1 else torch.tensor(1.0)


def discover(boxes, cand, score_col, top, pgt_index_c, nms_thr):
    """The tail of one object-discovery iteration (roi_heads/weak_head/loss.py:330-340) given its candidate set:
    NMS inside the candidates in descending class-score order (utils/utils.py:28-33), top-1 fallback, then the
    proposals not yet in the class's index list.  Returns (pgt_instance entry, fresh indices)."""
    with torch.no_grad():
        close = cand[nms_tv(boxes[cand], score_col[cand], nms_thr)]
    if close.numel() == 0:
        close = torch.cat((close, top.view(-1)))
    inst = close.clone()
    both = torch.cat((close, pgt_index_c))
    u, cnt = both.unique(return_counts=True)
    dup = u[cnt > 1]
    u2, cnt2 = torch.cat((close, dup)).unique(return_counts=True)
    close = u2[cnt2 == 1]
    if close.numel() == 0:
        close = torch.cat((close, top.view(-1)))
    return inst, close


def candidates_from_margins(sim_margin, neg, member=None):
    """The candidate set of one discovery iteration from the recorded comparisons (loss.py:319-329): sim >= threshold,
    then per negative class the bool-vs-float comparison of quirk Q3.  `member` (bool tensor, optional) overrides the
    threshold decision of individual proposals (a test replaying a near-threshold flip)."""
    close = torch.ge(sim_margin, 0) if member is None else member.clone()
    for s_neg in neg:
        close = torch.ge(close.to(s_neg.dtype), s_neg)                                 # Q3
    return close.nonzero(as_tuple=False).view(-1)


# --------------------------------------------------------------------------- the loss
def roi_reg_loss(cls_logit, det_logit, ref_logits, bbox_preds, sim_feature, clean_pooled, sd, rand,
                 boxes_per_image, labels_per_image, cfg, trace=None):
    """RoIRegLossComputation.__call__ (roi_heads/weak_head/loss.py:233-411), contra branch."""
    sizes = [len(b) for b in boxes_per_image]
    n_img = len(sizes)
    C = cls_logit.shape[1]
    nms_thr, lmda, thres, temp = cfg["nms"], cfg["lmda"], cfg["thres"], cfg["temp"]
    eps = 1e-8
    tr = trace if trace is not None else {}

    class_score = F.softmax(cls_logit, dim=1)
    det_score = torch.cat([F.softmax(d, dim=0) for d in det_logit.split(sizes)], dim=0)
    final_score = class_score * det_score
    final_list = final_score.split(sizes)
    ref_split = [r.split(sizes) for r in ref_logits]
    box_split = [b.split(sizes) for b in bbox_preds]
    sim_list = sim_feature.split(sizes)
    pooled_list = clean_pooled.split(sizes)
    n_ref = len(ref_logits)

    lab_vecs = [image_label_vector(C, l.unique()) for l in labels_per_image]
    pos_classes = [v[1:].eq(1).nonzero(as_tuple=False)[:, 0] for v in lab_vecs]

    def source(idx, i):
        return final_list[idx] if i == 0 else F.softmax(ref_split[i - 1][idx], dim=1)

    empty_l = lambda: torch.zeros(0, dtype=torch.long)
    pgt_index = [[empty_l() for _ in range(C - 1)] for _ in range(n_img)]
    pgt_collection = [torch.zeros(0) for _ in range(C - 1)]
    pgt_update = [torch.zeros(0) for _ in range(C - 1)]
    instance_diff = torch.zeros(0)

    # ---- loop 1: IoU sampling (loss.py:281-307)
    for idx in range(n_img):
        boxes = boxes_per_image[idx]
        for i in range(n_ref):
            pscore = source(idx, i)[:, 1:].clone()
            for c in pos_classes[idx]:
                c = int(c)
                top = torch.argmax(pscore[:, c])
                _note_margin(tr, "argmax_rel", _top2_gap(pscore[:, c]))
                iou = boxlist_iou(boxes, boxes[top].view(1, 4))
                near = torch.nonzero(torch.ge(iou, thres).max(dim=1)[0]).view(-1)   # utils/utils.py:22-26
                pgt_index[idx][c] = torch.cat((pgt_index[idx][c], near)).unique()
        for c in pos_classes[idx]:
            c = int(c)
            rows = pgt_index[idx][c]
            pgt_update[c] = torch.cat((pgt_update[c], sim_list[idx][rows]))
            hardness = final_list[idx][rows, c + 1] / final_list[idx][:, c + 1].sum()   # Q12
            instance_diff = torch.cat((instance_diff, hardness))
            drop = neck(dropblock(pooled_list[idx][rows], 1, 0.3, rand), sd, rand)
            pgt_update[c] = torch.cat((pgt_update[c], sim_net(drop, sd)))
            instance_diff = torch.cat((instance_diff, hardness))
            noisy = neck(noise_pool(pooled_list[idx][rows], rand), sd, rand)
            pgt_update[c] = torch.cat((pgt_update[c], sim_net(noisy, sd)))
            instance_diff = torch.cat((instance_diff, hardness))
            pgt_collection[c] = pgt_update[c].clone()                                  # Q2
            tr["iou_samples_%d_%d" % (idx, c)] = rows.clone()

    # ---- loop 2: object discovery (loss.py:311-345)
    pgt_instance = [[[empty_l() for _ in range(C - 1)] for _ in range(n_ref)] for _ in range(n_img)]
    for idx in range(n_img):
        boxes = boxes_per_image[idx]
        E = sim_list[idx]
        for i in range(n_ref):
            pscore = source(idx, i)[:, 1:].clone()
            for c in pos_classes[idx]:
                c = int(c)
                top = torch.argmax(pscore[:, c])
                sim_mat = torch.mm(E, E.T)
                thr = torch.mm(E[top].view(1, -1), pgt_collection[c].T).mean()
                _note_margin(tr, "sim_thresh_abs", (sim_mat[top] - thr).abs().min())
                if pos_classes[idx].shape[0] > 1:
                    close = torch.ge(sim_mat[top], thr)
                    for nc in pos_classes[idx][pos_classes[idx] != c]:
                        ntop = torch.argmax(pscore[:, int(nc)])
                        gap = (close.float() - sim_mat[ntop]).abs()
                        if not bool(close[ntop]):      # 0 >= |e|^2 ~ 1 is robustly false; 1 >= |e|^2 is not
                            gap = torch.cat((gap[:ntop], gap[ntop + 1:]))
                        _note_margin(tr, "q3_abs", gap.min())
                        close = torch.ge(close, sim_mat[ntop])                        # Q3 (bool vs float)
                    close = close.nonzero(as_tuple=False).view(-1)
                else:
                    close = torch.ge(sim_mat[top], thr).nonzero(as_tuple=False).view(-1)
                if close.numel() > 1:
                    ss = torch.sort(pscore[:, c][close].detach(), descending=True)[0]
                    _note_margin(tr, "nms_order_rel", ((ss[:-1] - ss[1:]) / ss[:-1].abs().clamp(min=1e-30)).min())
                if tr.get("_decisions"):
                    # everything a test needs to tell a legitimate near-threshold flip from a wrong selection: the signed
                    # distance of every proposal from each comparison that decides its candidacy, the candidates, the
                    # scores NMS orders them by
                    negs = [int(nc) for nc in pos_classes[idx][pos_classes[idx] != c]] if pos_classes[idx].shape[0] > 1 else []
                    tr["dec/%d_%d_%d" % (idx, i, c)] = dict(
                        top=int(top), top_gap=float(_top2_gap(pscore[:, c])), sim_margin=(sim_mat[top] - thr).detach().clone(),
                        neg=[sim_mat[torch.argmax(pscore[:, nc])].detach().clone() for nc in negs],
                        cand=close.clone(), score=pscore[:, c].detach().clone(), pgt_index=pgt_index[idx][c].clone())
                inst, close = discover(boxes, close, pscore[:, c].detach(), top, pgt_index[idx][c], nms_thr)
                pgt_instance[idx][i][c] = torch.cat((pgt_instance[idx][i][c], inst))
                tr["pgt_instance_%d_%d_%d" % (idx, i, c)] = inst.clone()
                tr["sim_new_%d_%d_%d" % (idx, i, c)] = close.clone()
                pgt_update[c] = torch.cat((pgt_update[c], E[close]))
                pgt_index[idx][c] = torch.cat((pgt_index[idx][c], close)).unique()
                hard = final_list[idx][close, c + 1] / final_list[idx][:, c + 1].sum()
                instance_diff = torch.cat((instance_diff, hard.view(-1)))

    raw_sim, feats, flabels, fw = supcon_v2(pgt_update, instance_diff, temp)
    tr["supcon_n"] = torch.tensor(feats.shape[0])
    tr["supcon_labels"] = flabels.clone()
    tr["supcon_weights"] = fw.clone()
    losses = {"loss_img": 0.0, "loss_sim": lmda * raw_sim}
    accs = {"acc_img": 0.0}
    for i in range(n_ref):
        losses["loss_ref_cls%d" % i] = 0.0
        losses["loss_ref_reg%d" % i] = 0.0
        accs["acc_ref%d" % i] = 0.0

    # ---- loop 3: MIL + refinement (loss.py:349-400)
    for idx in range(n_img):
        boxes = boxes_per_image[idx]
        lab = lab_vecs[idx]
        img_score = torch.clamp(final_list[idx].sum(dim=0), min=eps, max=1 - eps)
        losses["loss_img"] = losses["loss_img"] + F.binary_cross_entropy(img_score, lab.clamp(0, 1))
        for i in range(n_ref):
            pseudo, weights, targets = od_layer(boxes, source(idx, i).detach(), lab, pgt_instance[idx][i])
            if tr.get("_decisions"):
                tr["dec_source/%d_%d" % (idx, i)] = source(idx, i).detach().clone()
            tr["pseudo_%d_%d" % (idx, i)] = pseudo.clone()
            tr["weights_%d_%d" % (idx, i)] = weights.clone()
            lam = 3 if i == 0 else 1
            ce = F.cross_entropy(ref_split[i][idx], pseudo, reduction="none")
            losses["loss_ref_cls%d" % i] = losses["loss_ref_cls%d" % i] + lam * torch.mean(ce * weights)
            pos = torch.nonzero(pseudo > 0, as_tuple=False).squeeze(1)
            cols = 4 * pseudo[pos][:, None] + torch.tensor([0, 1, 2, 3])
            reg = lam * torch.sum(smooth_l1(box_split[i][idx][pos[:, None], cols], targets[pos], beta=1.0)
                                  * weights[pos, None])
            losses["loss_ref_reg%d" % i] = losses["loss_ref_reg%d" % i] + reg / pseudo.numel()
        with torch.no_grad():
            accs["acc_img"] = accs["acc_img"] + topk_accuracy(lab, img_score)
            for i in range(n_ref):
                rs = ref_split[i][idx].sum(dim=0)
                accs["acc_ref%d" % i] = accs["acc_ref%d" % i] + topk_accuracy(lab[1:], rs[1:])

    for k in losses:
        if "sim" not in k:                                                            # Q8
            losses[k] = losses[k] / n_img
    for k in accs:
        accs[k] = accs[k] / n_img
    return losses, accs


def forward(images, boxes_per_image, labels_per_image, sd, rand, cfg, trace=None):
    """GeneralizedRCNN.forward (train) -> ROIWeakRegHead.forward
    (modeling/detector/generalized_rcnn.py:57-97, roi_heads/weak_head/weak_head.py:101-122)."""
    arch = cfg.get("arch", "vgg16")
    feat = backbone_forward(images, sd) if arch == "vgg16" else resnet_forward(images, sd, arch)
    rois = rois_with_batch_index(boxes_per_image)
    if cfg.get("pooler", "ROIPool") == "ROIPool":
        pooled = _RoiPoolFn.apply(feat, rois, 7, 7, cfg.get("scale", 0.125))
    else:
        pooled = _RoiAlignFn.apply(feat, rois, 7, 7, cfg.get("scale", 0.125), cfg.get("sampling_ratio", 0))
    clean_feats = neck(pooled, sd, rand)
    sim_feature = sim_net(clean_feats, sd)
    aug_pooled = dropblock(pooled, 3, 0.3, rand)
    aug_feats = neck(aug_pooled, sd, rand)
    cls_logit, det_logit, ref_logits, bbox_preds = predictor(aug_feats, sd)
    if trace is not None:
        trace.update(feat=feat.detach(), pooled=pooled.detach(), sim_feature=sim_feature.detach(),
                     cls_logit=cls_logit.detach(), det_logit=det_logit.detach(),
                     ref_logits=[r.detach() for r in ref_logits], bbox_preds=[b.detach() for b in bbox_preds])
    return roi_reg_loss(cls_logit, det_logit, ref_logits, bbox_preds, sim_feature, pooled, sd, rand,
                        boxes_per_image, labels_per_image, cfg, trace)


def make_state(seed, num_classes=21, overrides=None, requires_grad=True, arch="vgg16"):
    """Formula-initialised parameters (and, for the ResNets, frozen batch-norm buffers) as torch tensors;
    frozen convs never require grad."""
    from od_wscl_amd import synthetic
    raw = synthetic.init_state_dict(param_shapes(num_classes, arch), seed, overrides=overrides)
    frozen = FROZEN if arch == "vgg16" else FROZEN_RESNET
    sd = {}
    for k, v in raw.items():
        t = torch.from_numpy(v)
        if requires_grad and not k.startswith(frozen):
            t.requires_grad_(True)
        sd[k] = t
    if arch != "vgg16":
        for k, v in synthetic.init_buffers(resnet_buffer_shapes(arch), seed).items():
            sd[k] = torch.from_numpy(v)
    return sd
