"""pairwise_sim (E E^T, 128-d unit rows -> P x P fp32): time per launch and fraction of the HBM roofline on the
algorithmic bytes 4 P^2 + 512 P (the '>= 60 % of HBM3E' contrastive-kernel target), split-bf16 form
against the exact-fp32 MFMA chain (ODW_PAIRWISE_FP32=1) and rocBLAS (torch.mm)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import _lib as L

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

lib = L.lib()
out = []
for P in (2000, 4000, 8000):
    E = torch.nn.functional.normalize(torch.randn(P, 128, device="cuda"), dim=1).contiguous()
    S = torch.empty(P, P, device="cuda")
    wsb = lib.odw_pairwise_sim_workspace(P, 128)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    us = timeit(lambda: L.check(lib.odw_pairwise_sim_ws(L.ptr(E), P, 128, L.ptr(S), L.ptr(ws), wsb, L.stream()), "ps"))
    ref = E.double() @ E.double().t()
    err = (S.double() - ref).abs().max().item()
    us_mm = timeit(lambda: torch.mm(E, E.t(), out=S))
    us_fill = timeit(lambda: S.fill_(1.0))
    nbytes = 4.0 * P * P + 512.0 * P
    r = dict(P=P, us=round(us, 2), hbm_frac=round(nbytes / (us * 1e-6) / 8e12, 3), GBps=round(nbytes / us / 1e3, 1), max_err=err,
             rocblas_us=round(us_mm, 2), fill_us=round(us_fill, 2), mode="fp32-chain" if os.environ.get("ODW_PAIRWISE_FP32") else "split-bf16")
    out.append(r); print(json.dumps(r), flush=True)
