"""Debug helper: run one e2e golden case on the GPU and print every comparison."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import e2e_inputs, load_e2e
from test_e2e_gpu import build_model
from od_wscl_amd import synthetic
from od_wscl_amd.structures import BoxList, to_image_list
from od_wscl_amd.utils.device_rand import DeviceRand
from oracle import hotpath_ref as H

name = sys.argv[1]
w = synthetic.init_state_dict(H.param_shapes(21), 1, overrides={"predictor": 0.002, "model_sim.mlp.2": 0.05})
g = load_e2e(name)
seed, batch, boxes, labels, cfg = e2e_inputs(g)
model = build_model(cfg["pooler"], w)
rois, targets = [], []
for k, (h, ww, p) in enumerate(g["spec_images"]):
    rois.append(BoxList(boxes[k].cuda(), (int(ww), int(h)), "xyxy"))
    t = BoxList(torch.zeros((len(labels[k]), 4)).cuda(), (int(ww), int(h)), "xyxy"); t.add_field("labels", labels[k].cuda()); targets.append(t)
trace = {}
model.roi_heads.loss_evaluator.trace = trace
losses, accs = model(to_image_list(batch.cuda()), targets, rois, rand=DeviceRand(seed))
for k, v in losses.items():
    ref = float(g["loss/" + k]); print("%-16s got %.8g ref %.8g rel %.2e" % (k, float(v), ref, abs(float(v)-ref)/max(abs(ref),1e-12)))
for k in sorted(g.files):
    if k.startswith(("pseudo_", "pgt_instance_")):
        a = trace[k].cpu().numpy()
        if not np.array_equal(a, g[k]):
            print("MISMATCH", k, "got", a[:20] if a.size<40 else np.nonzero(a!=g[k])[0], "ref", g[k][:20] if a.size<40 else g[k][a!=g[k]])
    if k.startswith("weights_"):
        a = trace[k].cpu().numpy(); d = np.abs(a-g[k]).max()
        if d > 1e-5: print("weights diff", k, d, np.nonzero(np.abs(a-g[k])>1e-5)[0][:10])
# oracle intermediate comparison
sd = {k: torch.from_numpy(v) for k, v in w.items()}
tr = {}
H.forward(batch, boxes, labels, sd, H.Rand(seed), cfg, tr)
feat = model.hip_body()(batch.cuda())[0]
print("feat max abs diff", (feat.cpu()-tr["feat"]).abs().max().item(), "scale", tr["feat"].abs().max().item())
