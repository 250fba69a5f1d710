import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _parity_backend_by_default():
    """Every test starts in the fp32-grade precision mode ("bf16x3": the mode the reference goldens are asserted in);
    engine.build_training_step switches the process-wide setting to its dtype argument and nothing switches it back."""
    from od_wscl_amd import precision as ll
    ll.set_precision("bf16x3")
    yield
    ll.set_precision("bf16x3")


@pytest.fixture(scope="session")
def ops_golden():
    return np.load(os.path.join(GOLDEN, "ops_ref.npz"))


_WEIGHTS = {}


def weights_for(arch, classes=21):
    """The one formula-generated weight set every e2e golden of (`arch`, `classes`) was produced with (WEIGHT_SEED=1);
    for the ResNets the frozen batch-norm buffers come with it."""
    key = arch if classes == 21 else (arch, classes)
    if key not in _WEIGHTS:
        from od_wscl_amd import synthetic
        from oracle import hotpath_ref as H
        sd = synthetic.init_state_dict(H.param_shapes(classes, arch), 1,
                                       overrides={"predictor": 0.002, "model_sim.mlp.2": 0.05})
        if arch != "vgg16":
            sd.update(synthetic.init_buffers(H.resnet_buffer_shapes(arch), 1))
        _WEIGHTS.clear()                       # one set resident at a time (0.6-1 GB each)
        _WEIGHTS[key] = sd
    return _WEIGHTS[key]


@pytest.fixture(scope="session")
def weights_np():
    return weights_for("vgg16")


def e2e_classes(g):
    return int(g["spec_classes"]) if "spec_classes" in g.files else 21


def e2e_arch(g):
    return str(g["spec_arch"]) if "spec_arch" in g.files else "vgg16"


def load_e2e(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def e2e_inputs(g):
    """Rebuild the formula-generated inputs of an e2e golden from its recipe."""
    import torch
    from od_wscl_amd import synthetic
    seed = int(g["spec_seed"])
    specs = g["spec_images"]
    cnt, flat = g["spec_labels_count"], g["spec_labels_flat"]
    hm = max(synthetic.pad_to(int(h)) for h, w, p in specs)
    wm = max(synthetic.pad_to(int(w)) for h, w, p in specs)
    batch = torch.zeros(len(specs), 3, hm, wm)
    boxes, labels, o = [], [], 0
    min_size = int(g["spec_min_size"]) if "spec_min_size" in g.files else 12
    arch = e2e_arch(g)
    for k, (h, w, p) in enumerate(specs):
        h, w, p = int(h), int(w), int(p)
        batch[k, :, :h, :w] = torch.from_numpy(synthetic.make_image(seed, k, h, w)[:, :h, :w].copy())
        boxes.append(torch.from_numpy(synthetic.make_proposals(seed, k, p, h, w, min_size=min_size)))
        labels.append(torch.tensor(flat[o:o + cnt[k]], dtype=torch.int64))
        o += cnt[k]
    cfg = dict(nms=0.1, lmda=0.03, thres=0.5, temp=0.2, pooler=str(g["spec_pooler"]), sampling_ratio=0,
               arch=arch, scale=0.125 if arch == "vgg16" else 0.0625)
    return seed, batch, boxes, labels, cfg
