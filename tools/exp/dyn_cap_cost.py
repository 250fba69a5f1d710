"""What a capacity-sized launch costs: the same live extent (M rows) under growing capacities, per dynamic kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from od_wscl_amd import dyn, gemm, precision
precision.set_precision("bf16x2f")
def t(fn, n=20):
    for _ in range(3): fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
M = int(os.environ.get("M", 450))
md = torch.tensor([M], dtype=torch.int32, device="cuda")
for cap in (512, 1024, 2048, 4096, 8192, 12032):
    if cap < M: continue
    m = dyn.Dyn(md, cap, M)
    a = torch.randn(cap, 4096, device="cuda").bfloat16(); b = torch.randn(4096, 4096, device="cuda").bfloat16()
    out = torch.empty(cap, 4096, device="cuda")
    us_g = t(lambda: dyn.gemm_nt(a, b, cap, 4096, 4096, out, m=m))
    b2 = torch.randn(128, 4096, device="cuda").bfloat16(); out2 = torch.empty(cap, 128, device="cuda")
    us_s = t(lambda: dyn.gemm_nt(a, b2, cap, 128, 4096, out2, m=m))
    w6 = torch.randn(25088, 4096, device="cuda").bfloat16(); out6 = torch.empty(cap, 25088, device="cuda")
    us_6 = t(lambda: dyn.gemm_nt(a, w6, cap, 25088, 4096, out6, m=m))
    x = torch.randn(cap, 4096, device="cuda")
    pa = precision.patterns("gemm")[0]
    xs = torch.empty(cap, 3 * 4096, dtype=torch.bfloat16, device="cuda")
    us_sp = t(lambda: dyn.split_rows(x, pa, 4096, m, out=xs))
    xt = torch.empty(4096, dyn.r64(cap), dtype=torch.bfloat16, device="cuda")
    us_tr = t(lambda: dyn.transpose(x, 4096, xt, m))
    dz = torch.empty(cap, 4096, dtype=torch.bfloat16, device="cuda")
    us_pr = t(lambda: dyn.bwd_prep(x, x, 4096, 2.0, dz, xt, None, m))
    print("M %d cap %5d: gemm 4096x4096 %6.1f us  128x4096 %5.1f  25088x4096 %6.1f   split_rows %5.1f  transpose %5.1f  prep %5.1f"
          % (M, cap, us_g, us_s, us_6, us_sp, us_tr, us_pr), flush=True)
