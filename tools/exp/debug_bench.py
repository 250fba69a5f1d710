import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from od_wscl_amd import engine
from od_wscl_amd.utils.device_rand import DeviceRand
from od_wscl_amd.modeling.detector import build_detection_model
P = int(sys.argv[1]); size = int(sys.argv[2])
cfg = bench.build_cfg(21)
dev = torch.device("cuda", 0)
model = build_detection_model(cfg).to(dev); engine.load_formula_weights(model, 1); model.train()
images, targets, rois = bench.synthetic_batch(1234, 0, size, P, 21, dev)
def sync(tag):
    torch.cuda.synchronize(); print("ok", tag, flush=True)
rand = DeviceRand(1234)
model.roi_heads.set_rand(rand)
feats = model.hip_body()(images.tensors); sync("backbone fwd %s" % (tuple(feats[0].shape),))
fe = model.roi_heads.feature_extractor
cf, cp = fe.forward(feats, rois); sync("fe fwd")
sim = model.roi_heads.model_sim(cf); sync("sim")
ap = fe.forward_dropblock(cp); sync("dropblock")
af = fe.forward_neck(ap); sync("neck")
cls, det, refs, boxes = model.roi_heads.predictor(af, rois); sync("pred")
tr = {}
model.roi_heads.loss_evaluator.trace = tr
losses, accs = model.roi_heads.loss_evaluator([cls],[det],refs,boxes,sim,cp,fe,model.roi_heads.model_sim,rois,targets); sync("loss")
print({k: float(v) for k,v in losses.items()}, "supcon N", tr["supcon_n"])
for k,v in tr.items():
    if k.startswith(("iou_samples","pgt_instance")): print(k, v.numel())
loss = sum(losses.values())
loss.backward(); sync("backward")
print("---- loop", flush=True)
opt = engine.make_optimizer(cfg, model)
model.roi_heads.loss_evaluator.trace = None
for it in range(3):
    rand = DeviceRand(1234, first_stream=(1<<20)+(it<<12))
    losses, accs = model(images, targets, rois, rand=rand); sync("it%d fwd" % it)
    loss = sum(losses.values()); opt.zero_grad(set_to_none=True)
    loss.backward(); sync("it%d bwd" % it)
    opt.step(); sync("it%d step" % it)
