"""Pick the seed of tests/test_timed_step_gpu.py: three consecutive oracle steps (CPU, oracle/hotpath_ref.py + torch.optim.SGD
over the reference's parameter groups) at the C2 workload, and per step the margins the margin-gated replay needs -- the
smallest relative arg-max gap and the largest number of proposals within TOL of a decision threshold.  A seed is usable
when every step keeps the arg-max gap above TOL and the uncertain count under MAX_UNCERTAIN.

    python tools/timed_step_seed_scan.py 301 306 [lr]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

TOL, MAX_UNCERTAIN = 1e-4, 12


def main():
    import fullsize_seed_scan as S
    from conftest import weights_for
    from oracle import hotpath_ref as H
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    lr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-5
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    w_np = weights_for("vgg16", 21)
    frozen = H.FROZEN
    names = [n for n, _ in H.param_shapes(21, "vgg16")]
    cfg = dict(nms=0.1, lmda=0.03, thres=0.5, temp=0.2, pooler="ROIPool", sampling_ratio=0, arch="vgg16", scale=0.125)
    for seed in range(lo, hi):
        batch, boxes, lab, _ = S.inputs("c2", seed)
        sd = {}
        for k, v in w_np.items():
            t = torch.from_numpy(v.copy())
            if k in names and not k.startswith(frozen):
                t.requires_grad_(True)
            sd[k] = t
        groups = [{"params": [sd[n]], "lr": lr * (2.0 if "bias" in n else 1.0), "weight_decay": 0.0 if "bias" in n else 1e-4}
                  for n in names if sd[n].requires_grad]
        opt = torch.optim.SGD(groups, lr, momentum=0.9)
        t0, rows, ok = time.time(), [], True
        for it in range(3):
            tr = {"_decisions": True}
            losses, _ = H.forward(batch, boxes, lab, sd, H.Rand(seed, first_stream=(1 << 20) + (it << 12)), cfg, tr)
            opt.zero_grad(set_to_none=True)
            sum(losses.values()).backward()
            opt.step()
            gap, unsure = 1.0, 0
            for k, d in tr.items():
                if not k.startswith("dec/"):
                    continue
                gap = min(gap, d["top_gap"])
                sm = d["sim_margin"]
                u = sm.abs() <= TOL
                below = sm < 0
                for s_neg in d["neg"]:
                    u |= torch.where(below, s_neg.abs() <= TOL, (1.0 - s_neg).abs() <= TOL)
                unsure = max(unsure, int(u.sum()))
            ok = ok and gap > 2 * TOL and unsure <= MAX_UNCERTAIN
            rows.append("step %d: top_gap %.2e uncertain %d loss %.4f" % (it + 1, gap, unsure, float(sum(losses.values()))))
        print("seed %d lr %g %s  %s  (%.0f s)" % (seed, lr, "OK " if ok else "-- ", " | ".join(rows), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
