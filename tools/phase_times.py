"""Wall-clock per phase of one hip-backend step (synchronised between phases)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from od_wscl_amd import engine
from od_wscl_amd import precision as ll
from od_wscl_amd.utils.device_rand import DeviceRand
from od_wscl_amd.modeling.detector import build_detection_model
cfg = bench.build_cfg(21); dev = torch.device("cuda", 0)
ll.set_precision("bf16")
model = build_detection_model(cfg).to(dev); engine.load_formula_weights(model, 1); model.train()
model.backbone_autocast = torch.bfloat16
opt = engine.FlatSGD(cfg, model, 1)
images, targets, rois = bench.synthetic_batch(1234, 0, 600, 2000, 21, dev)
fe = model.roi_heads.feature_extractor; head = model.roi_heads
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0) + (time.perf_counter() - t0) * 1e3; return time.perf_counter()
for it in range(8):
    if it == 3: T.clear()
    rand = DeviceRand(1234, first_stream=(1 << 20) + (it << 12)); head.set_rand(rand)
    torch.cuda.synchronize(); t = time.perf_counter()
    opt.begin_step(); t = tick("begin_step", t)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        feats = [f.float() for f in model.hip_body()(images.tensors)]
    t = tick("backbone_fwd", t)
    pooled = fe.forward_pooler(feats, rois); t = tick("roipool_fwd", t)
    cf, af = fe.forward_clean_and_aug(pooled); t = tick("fc6fc7_stacked_fwd", t)
    sim = head.model_sim(cf); t = tick("simnet_fwd", t)
    cls, det, refs, boxes = head.predictor(af, rois); t = tick("predictor_fwd", t)
    losses, accs = head.loss_evaluator([cls], [det], refs, boxes, sim, pooled, fe, head.model_sim, rois, targets); t = tick("loss_fwd(incl K-row passes)", t)
    loss = sum(losses.values()); loss.backward(); t = tick("backward_all", t)
    opt.step(); t = tick("sgd+shadow", t)
n = 5
for k, v in T.items(): print("%-32s %8.3f ms" % (k, v / n))
print("%-32s %8.3f ms" % ("sum", sum(T.values()) / n))
