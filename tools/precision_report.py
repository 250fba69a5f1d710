"""Deviation of every precision mode from the reference's goldens (tests/golden/e2e_*.npz, produced by the imported
reference in fp32): per loss the relative deviation, the number of selection index sets that differ, the worst
gradient-norm deviation.  Writes gpurun_out/precision_deviation.json (copied to profiles/ by hand).

    python tools/precision_report.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(name, mode):
    from conftest import e2e_arch, e2e_classes, e2e_inputs, load_e2e, weights_for
    from test_e2e_gpu import build_model
    from od_wscl_amd import precision
    from od_wscl_amd.structures import BoxList, to_image_list
    from od_wscl_amd.utils.device_rand import DeviceRand
    precision.set_precision(mode)
    g = load_e2e(name)
    seed, batch, boxes, labels, cfg = e2e_inputs(g)
    model = build_model(cfg["pooler"], weights_for(e2e_arch(g), e2e_classes(g)), "fused", e2e_arch(g), e2e_classes(g))
    rois, targets = [], []
    for k, (h, w, p) in enumerate(g["spec_images"]):
        rois.append(BoxList(boxes[k].cuda(), (int(w), int(h)), "xyxy"))
        t = BoxList(torch.zeros((len(labels[k]), 4)).cuda(), (int(w), int(h)), "xyxy")
        t.add_field("labels", labels[k].cuda())
        targets.append(t)
    trace = {}
    model.roi_heads.loss_evaluator.trace = trace
    losses, accs = model(to_image_list(batch.cuda()), targets, rois, rand=DeviceRand(seed))
    sum(losses.values()).backward()
    out = {"loss_rel": {}, "selection_sets": 0, "selection_mismatch": 0}
    for k, v in losses.items():
        ref = float(g["loss/" + k])
        out["loss_rel"][k] = abs(float(v) - ref) / max(abs(ref), 1e-6)
    for k in g.files:
        if k.startswith(("pseudo_", "pgt_instance_")):
            out["selection_sets"] += 1
            a, b = trace[k].cpu().numpy(), g[k]
            if a.shape != b.shape or not np.array_equal(a, b):
                out["selection_mismatch"] += 1
    worst = (0.0, "")
    for n, p in model.named_parameters():
        key = "gradnorm/" + n
        if key in g.files and float(g[key]) > 1e-5:
            d = abs(p.grad.double().norm().item() - float(g[key])) / float(g[key])
            worst = max(worst, (d, n))
    out["gradnorm_worst_rel"], out["gradnorm_worst_param"] = worst
    out["loss_worst_rel"] = max(out["loss_rel"].values())
    return out


def main():
    names = ["e2e_voc_2img", "e2e_voc_1img", "e2e_align_1img", "e2e_r50_2img", "e2e_coco_2img"]
    report = {}
    for mode in ("bf16x3", "bf16x2", "bf16"):
        for name in names:
            r = run(name, mode)
            report["%s/%s" % (mode, name)] = r
            print("%-7s %-16s loss worst %.2e  selection %d/%d differ  gradnorm worst %.2e (%s)" % (
                mode, name, r["loss_worst_rel"], r["selection_mismatch"], r["selection_sets"], r["gradnorm_worst_rel"],
                r["gradnorm_worst_param"]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "precision_deviation.json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
