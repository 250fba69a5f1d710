"""Single-scale inference (eval forward + device post-processing) at the bench's C2 shape: images/s, proposals/s."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from od_wscl_amd import engine
from od_wscl_amd import precision as ll
from od_wscl_amd.modeling.backbone.vgg16_hip import VGGBackboneHip
from od_wscl_amd.modeling.detector import build_detection_model
dev = torch.device("cuda", 0)
cfg = bench.build_cfg(21)
cfg.merge_from_list(["MODEL.ROI_HEADS.SCORE_THRESH", 0.0, "MODEL.ROI_HEADS.NMS", 0.4])
ll.set_precision("bf16")
model = build_detection_model(cfg).to(dev)
engine.load_formula_weights(model, 1)
model.eval()
model.backbone_hip = VGGBackboneHip(model.backbone.body)
images, targets, rois = bench.synthetic_batch(cfg.SEED, 0, 600, 2000, 21, dev)
with torch.no_grad():
    for _ in range(5): res = model(images, rois=rois)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30
    for _ in range(n): res = model(images, rois=rois)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("eval forward + post-processing: %.2f ms/image = %.1f images/s = %.0f proposals/s; %d detections" % (dt * 1e3, 1 / dt, 2000 / dt, len(res[0])))
