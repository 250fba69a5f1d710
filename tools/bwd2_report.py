"""VERDICT r04 next #7 (second half): what a TWO-plane backward of the small-K Linears buys and costs in "bf16x2f".

    python tools/bwd2_report.py > profiles/r05/bwd2_report.txt

Runs the bench's step function at the C2 workload (VGG16, P = 2000 @ 600 px; the 3-label synthetic image) once with the
default single-plane backward and once with ODW_BWD2=1 (fc7, Sim_Net and the predictor: input AND weight gradients on two
bf16 planes per operand = three plane products; fc6 and the convolutions unchanged), each in its own process, and prints
side by side: the step time (median of 20 steps) and the relative L2 error of every trainable tensor's gradient against
the CPU oracle's (oracle/hotpath_ref.py, fp32 autograd) on the first step."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def child():
    import numpy as np
    import torch
    import bench
    import fullsize_seed_scan as S
    from conftest import weights_for
    from oracle import hotpath_ref as H
    from od_wscl_amd import engine
    from od_wscl_amd.structures import BoxList, to_image_list
    from od_wscl_amd.utils.device_rand import DeviceRand
    dev = torch.device("cuda", 0)
    os.environ["ODW_NO_TIMER"] = "1"
    seed = 302
    size, p, classes = S.CASES["c2"][:3]
    batch, boxes, lab, _ = S.inputs("c2", seed)
    w_np = weights_for("vgg16", classes)
    cfg = bench.build_cfg(classes)
    step, info = engine.build_training_step(cfg, dev, dtype="bf16x2f", world=1, seed=cfg.SEED)
    model, opt = step.model, step.optimizer
    with torch.no_grad():
        for n, q in list(model.named_parameters()) + list(model.named_buffers()):
            q.copy_(torch.from_numpy(w_np[n]))
    opt.sync_from_params(model)
    images = to_image_list(batch.to(dev))
    rois = [BoxList(b.to(dev), (size, size), "xyxy") for b in boxes]
    targets = []
    for l in lab:
        t = BoxList(torch.zeros((len(l), 4), device=dev), (size, size), "xyxy")
        t.add_field("labels", l.to(dev))
        t.add_field("labels_host", l.tolist())
        targets.append(t)
    trainable = [n for n, q in model.named_parameters() if q.requires_grad]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    sd = {}
    for k, v in w_np.items():
        t = torch.from_numpy(v.copy())
        if k in trainable:
            t.requires_grad_(True)
        sd[k] = t
    ocfg = dict(nms=0.1, lmda=0.03, thres=0.5, temp=0.2, pooler="ROIPool", sampling_ratio=0, arch="vgg16", scale=0.125)
    stream0 = 1 << 20
    ref_losses, _ = H.forward(batch, boxes, lab, sd, H.Rand(seed, first_stream=stream0), ocfg, {})
    sum(ref_losses.values()).backward()
    losses, _ = step(images, targets, rois, DeviceRand(seed, first_stream=stream0, device=dev))
    torch.cuda.synchronize()
    errs = {}
    for n in trainable:
        o, k = opt.slices[n]
        g = opt.flat_g[o:o + k].cpu().double()
        r = sd[n].grad.reshape(-1).double()
        if float(r.norm()) < 1e-9 or n.endswith("det_score.bias"):
            continue
        errs[n] = float((g - r).norm() / r.norm())
    loss_dev = max(abs(float(losses[k].detach()) - float(v)) / max(abs(float(v)), 1e-5) for k, v in ref_losses.items())
    # timing: 5 warm-up + 20 timed steps, HIP events per step
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    for it in range(5):
        step(images, targets, rois, DeviceRand(seed, first_stream=stream0 + ((it + 1) << 12), device=dev))
    torch.cuda.synchronize()
    for it in range(20):
        ev[it].record()
        step(images, targets, rois, DeviceRand(seed, first_stream=stream0 + ((it + 6) << 12), device=dev))
    ev[20].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))
    print("BWD2JSON " + json.dumps({"median_ms": ms[10], "mean_ms": sum(ms) / 20, "loss_dev": loss_dev, "errs": errs}))


def main():
    out = {}
    for flag in ("0", "1"):
        env = dict(os.environ, ODW_BWD2=flag, ODW_BWD2_CHILD="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("BWD2JSON ")]
        if not line:
            print(r.stdout[-2000:], r.stderr[-4000:])
            raise SystemExit("child failed (ODW_BWD2=%s)" % flag)
        out[flag] = json.loads(line[0][9:])
    a, b = out["0"], out["1"]
    print("# two-plane backward of the small-K Linears (fc7, Sim_Net, predictor) in bf16x2f: ODW_BWD2=1 against the default")
    print("# C2 workload (VGG16, P = 2000 @ 600 px, 1 image), bench step function, seed 302; gradient error = relative L2 of the")
    print("# whole tensor against the CPU oracle's fp32 autograd gradient, first step")
    print("step time, median of 20 (ms):   single-plane backward %.3f    two-plane small-K backward %.3f   (+%.3f ms, +%.1f %%)"
          % (a["median_ms"], b["median_ms"], b["median_ms"] - a["median_ms"], 100 * (b["median_ms"] / a["median_ms"] - 1)))
    print("worst loss deviation vs oracle:  %.2e    %.2e" % (a["loss_dev"], b["loss_dev"]))
    print("%-58s %12s %12s %8s" % ("gradient tensor", "single-plane", "two-plane", "ratio"))
    for n in a["errs"]:
        print("%-58s %12.3e %12.3e %8.2f" % (n, a["errs"][n], b["errs"].get(n, float("nan")), a["errs"][n] / max(b["errs"].get(n, 1e-30), 1e-30)))
    wa, wb = max(a["errs"].values()), max(b["errs"].values())
    print("%-58s %12.3e %12.3e %8.2f" % ("WORST", wa, wb, wa / wb))


if __name__ == "__main__":
    if os.environ.get("ODW_BWD2_CHILD") == "1":
        child()
    else:
        main()
