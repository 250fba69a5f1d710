import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from od_wscl_amd import _lib as L
lib = L.lib()
def graph_time(f, iters=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for P in (5000, 6000, 8000):
    E = torch.nn.functional.normalize(torch.randn(P, 128, device="cuda"), dim=1).contiguous()
    S = torch.empty(P, P, device="cuda")
    wsb = lib.odw_pairwise_sim_workspace(P, 128)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    panel = lambda: L.check(lib.odw_pairwise_sim(L.ptr(E), P, 128, L.ptr(S), L.stream()), "p")
    def dma():
        L.check(lib.odw_pairwise_split_planes(L.ptr(E), P, L.ptr(ws), L.stream()), "s")
        L.check(lib.odw_pairwise_sim_planes(L.ptr(ws), P, L.ptr(S), L.stream()), "d")
    res = {"panel": [], "split+dma": []}
    for r in range(5):
        res["panel"].append(round(graph_time(panel), 1))
        res["split+dma"].append(round(graph_time(dma), 1))
    print(P, res, flush=True)
