"""pairwise_sim (default one-launch form) against P, with a dense S and with the rows of S padded to 32 floats:
time per launch (graph-replayed) and fraction of the HBM roofline."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from od_wscl_amd import _lib as L
lib = L.lib()
def graph_time(f, iters=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): f()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best
for P in [int(a) for a in sys.argv[1:]] or [2000, 3000, 4000, 4004, 4008, 4016, 4500, 4992, 5000, 5024, 5500, 6000, 7000, 8000]:
    E = torch.nn.functional.normalize(torch.randn(P, 128, device="cuda"), dim=1).contiguous()
    row = "P=%5d" % P
    for ld in (P, -(-P // 32) * 32):
        S = torch.empty(P, ld, device="cuda")
        us = graph_time(lambda: L.check(lib.odw_pairwise_sim_ld(L.ptr(E), P, 128, L.ptr(S), ld, L.stream()), "p"))
        row += "   pitch %5d floats: %6.1f us frac %.3f" % (ld, us, (4.0 * P * P + 512 * P) / (us * 1e-6) / 8e12)
    print(row, flush=True)
