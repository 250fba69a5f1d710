import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from od_wscl_amd import gemm, precision
precision.set_precision("bf16x2f")
import test_pair_gpu as T
P, N, C, S = 100, 256, 64, 49
x, w, b = T._operands(P + N, P, N, C, S)
K = C * S
g = torch.Generator(device="cuda").manual_seed(P)
keep = (torch.rand(P, S, device="cuda", generator=g) > 0.45).float()
keep[0] = 1.0; keep[1] = 0.0; keep[2] = 0.0; keep[2, S - 1] = 1.0
ksum = keep.sum()
xd = (x.view(P, C, S) * keep[:, None, :] * keep.numel() / ksum).reshape(P, K)
pa, pb = precision.patterns("gemm")
segs = [(0, 11, 12), (P, 13, 14)]
ref = torch.empty(2 * P, N, device="cuda")
gemm.gemm_nt(precision.split_rows(torch.cat([x, xd]), pa, K), precision.split_rows(w, pb, K), 2 * P, N, 3 * K, ref, bias=b, relu=True, drop_p=0.5, segs=segs)
out = torch.full((2 * P, N), float("nan"), device="cuda")
gemm.gemm_nt_cm(gemm.split_rows_cm(x, C, S), gemm.split_rows_cm(w, C, S), P, N, C, S, out, bias=b, relu=True, drop_p=0.5, segs=segs, keep=keep, keep_sum=ksum, drop_row0=P)
differ = (out == 0) != (ref == 0)
print("rows with differences:", differ.any(1).nonzero().flatten().tolist(), differ.sum(1)[differ.any(1)].tolist())
# which is right? the generator
import numpy as np
from od_wscl_amd.utils import rng
def pattern(k0, k1, rows):
    return None
y64 = torch.relu(torch.cat([x, xd]).double() @ w.double().T + b.double())
for name, t in (("ref", ref), ("out", out)):
    wrong = ((t == 0) & (y64 > 1e-3))
    print(name, "zeros:", (t == 0).float().mean().item())
r = differ.any(1).nonzero().flatten().tolist()
if r:
    m = r[0]
    print("row", m, "ref", ref[m, :8].tolist(), "out", out[m, :8].tolist(), "y64*2", (2 * y64[m, :8]).tolist())
