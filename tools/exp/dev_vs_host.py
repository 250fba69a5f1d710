"""Per-parameter gradient difference between the device-resident lists path and the host-list path on an e2e golden."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import test_e2e_gpu as T
name = sys.argv[1] if len(sys.argv) > 1 else "e2e_voc_2img"
os.environ.pop("ODW_HOST_LISTS", None)
from od_wscl_amd.modeling.roi_heads.weak_head import loss_device
loss_device._EXACT = os.environ.get("EXACT") == "1"
l_dev, t_dev, m_dev, g = T._run_golden(name, "bf16x2f")
gd = {n: p.grad.detach().clone() for n, p in m_dev.named_parameters() if p.grad is not None}
del m_dev
os.environ["ODW_HOST_LISTS"] = "1"
l_host, t_host, m_host, _ = T._run_golden(name, "bf16x2f")
gh = {n: p.grad.detach().clone() for n, p in m_host.named_parameters() if p.grad is not None}
del m_host
l_h2, t_h2, m_h2, _ = T._run_golden(name, "bf16x2f")
print("losses dev ", {k: float(v) for k, v in l_dev.items()})
print("losses host", {k: float(v) for k, v in l_host.items()})
for n, p in m_h2.named_parameters():
    if p.grad is None:
        continue
    ref = gh[n].double()
    e1 = (gd[n].double() - ref).norm().item() / max(ref.norm().item(), 1e-12)
    e2 = (p.grad.double() - ref).norm().item() / max(ref.norm().item(), 1e-12)
    key = "gradnorm/" + n
    gn = float(g[key]) if key in g.files else float("nan")
    print("%-50s dev-host %.2e  host-host %.2e   |g| dev %.5e host %.5e oracle %.5e" % (n, e1, e2, gd[n].double().norm().item(), ref.norm().item(), gn))
