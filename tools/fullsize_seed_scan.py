"""Pick seeds for the full-size parity tests (tests/test_fullsize_gpu.py): run the CPU oracle's forward at the C1 / C2 /
C4 shapes of BASELINE.json on formula-generated inputs and print the decision margins of each seed -- how far every
data-dependent selection (argmax, similarity threshold, NMS order, quirk Q3) was from flipping.  A seed is usable
when all margins are far above fp32 re-association noise, so that "bit-exact selection" tests the algorithm and not
the summation order of a K = 25088 dot product.

    python tools/fullsize_seed_scan.py c2 300 310
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = {  # name: (size, proposals, classes, labels[, arch]); labels = one list, or a list of lists = several images
    "c1": (300, 500, 21, [4, 11]),
    "c2": (600, 2000, 21, [3, 9]),
    "c4": (800, 4000, 81, [17]),
    "c5": (600, 2000, 21, [7, 12], "r50"),       # R-50-C5 body (configs/voc/voc07_r50_c5_contra_db_b8_lr0.02_ss.yaml)
    "c4s": (688, 4000, 81, [42]),                # another scale of the COCO config's multi-scale training (480-800)
    # the reference's own single-GPU setup: IMS_PER_BATCH 8 on one device (README.md:99-100), VOC shape
    "b8": (600, 2000, 21, [[3, 9], [15], [5, 12], [8], [1, 19], [14], [7], [2, 11]]),
    # the COCO shape with a NON-ZERO contrastive loss: two images x 4000 proposals x 81 classes, one label each (two
    # classes in the SupCon set -> loss_sim > 0 and a non-zero SupCon gradient at P = 4000), at 576 px -- a third scale
    # of the COCO config's multi-scale training (configs/coco/coco14_contra_db_b8_lr0.01_mcg.yaml: 480-800)
    "c4b": (576, 4000, 81, [[17], [42]]),
    # the smallest scale of the COCO config (480 px), three images (one label each: with 80 classes a two-label image makes the same proposal top both classes -- quirk Q3's rounding-dependent comparison, excluded by construction)
    "c4m": (480, 4000, 81, [[5], [60], [23]]),
}


def arch_of(name):
    return CASES[name][4] if len(CASES[name]) > 4 else "vgg16"


def labels_of(name):
    """list (one entry per image) of label lists"""
    lab = CASES[name][3]
    return lab if isinstance(lab[0], (list, tuple)) else [lab]
FLOORS = {"argmax_rel": 1e-3, "sim_thresh_abs": 1e-4, "q3_abs": 1e-4, "nms_order_rel": 1e-3}


def inputs(name, seed):
    from od_wscl_amd import synthetic
    size, p, classes = CASES[name][:3]
    labels = labels_of(name)
    pad = synthetic.pad_to(size)
    batch = torch.zeros(len(labels), 3, pad, pad)
    boxes = []
    for k in range(len(labels)):
        batch[k, :, :size, :size] = torch.from_numpy(synthetic.make_image(seed, k, size, size)[:, :size, :size].copy())
        boxes.append(torch.from_numpy(synthetic.make_proposals(seed, k, p, size, size, min_size=32 if arch_of(name) != "vgg16" else 20)))
    return batch, boxes, [torch.tensor(l, dtype=torch.int64) for l in labels], classes


def main():
    from conftest import weights_for
    from oracle import hotpath_ref as H
    name, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    torch.set_num_threads(8)
    classes = CASES[name][2]
    arch = arch_of(name)
    sd = {k: torch.from_numpy(v) for k, v in weights_for(arch, classes).items()}
    cfg = dict(nms=0.1, lmda=0.03, thres=0.5, temp=0.2, pooler="ROIPool", sampling_ratio=0, arch=arch,
               scale=0.125 if arch == "vgg16" else 0.0625)
    for seed in range(lo, hi):
        batch, boxes, labels, _ = inputs(name, seed)
        tr = {}
        t0 = time.time()
        with torch.no_grad():
            losses, _ = H.forward(batch, boxes, labels, sd, H.Rand(seed), cfg, tr)
        m = {k[7:]: float(v) for k, v in tr.items() if k.startswith("margin/")}
        ok = all(m[k] > FLOORS[k] for k in m)
        print("%s seed %d  %s  %s  supcon_n %d  (%.1f s)" % (name, seed, "OK " if ok else "-- ",
              " ".join("%s=%.2e" % kv for kv in sorted(m.items())), int(tr["supcon_n"]), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
