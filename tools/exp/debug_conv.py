import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd.config import make_defaults
from od_wscl_amd.modeling.backbone import build_backbone
from od_wscl_amd.modeling.backbone.vgg16_hip import VGGBackboneHip
from od_wscl_amd.utils import rng
def rnd(seed, shape, scale=1.0):
    return torch.from_numpy((rng.normal(seed, 1, int(np.prod(shape))) * scale).reshape(shape)).cuda()
cfg = make_defaults(); cfg.merge_from_list(["MODEL.BACKBONE.CONV_BODY", "VGG16-OICR"])
torch.manual_seed(0)
bb = build_backbone(cfg).cuda()
with torch.no_grad():
    for p in bb.parameters():
        p.copy_(p.bfloat16().float())
        if p.dim() == 1: p.normal_(0, 0.05)
hip = VGGBackboneHip(bb.body); hip.debug = {}
x = rnd(7, (1, 3, 64, 96), 50.0)
feat = hip(x)[0]; g = rnd(8, tuple(feat.shape)); feat.backward(g)
class RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t): return t.bfloat16().float()
    @staticmethod
    def backward(ctx, gr): return gr.bfloat16().float()
bb = bb.cpu(); g = g.cpu()
h = x.bfloat16().float().cpu(); pre = {}; convs = []
mods = list(bb.body.features)
for m in mods:
    if isinstance(m, torch.nn.ReLU): m.inplace = False
for i, m in enumerate(mods):
    h = m(h)
    if isinstance(m, torch.nn.Conv2d):
        convs.append(i)
        if i == len(mods) - 1: h = RoundBF16.apply(h)
        if h.requires_grad:
            h.retain_grad(); pre[len(convs) - 1] = h          # pre-activation (conv output)
        # note: inplace relu follows; clone to keep pre-activation grads
    if isinstance(m, torch.nn.ReLU): h = torch.relu(h) if False else h
    if isinstance(m, (torch.nn.ReLU, torch.nn.MaxPool2d)): h = RoundBF16.apply(h)
h.backward(g)
cos = lambda a, b: (a.flatten() @ b.flatten() / (a.norm() * b.norm() + 1e-30)).item()
for li in sorted(hip.debug):
    r = pre[li].grad
    d = hip.debug[li].cpu()
    print("layer", li, "feat idx", convs[li], "cos", round(cos(d, r), 5), "norm ratio", round((d.norm() / r.norm()).item(), 4),
          "mask agree", round(((d != 0) == (r != 0)).float().mean().item(), 5), "nz frac", round((r != 0).float().mean().item(), 3))
