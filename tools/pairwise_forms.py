"""pairwise_sim, the three forms side by side (graph-replayed back-to-back launches, HIP events, like bench.py):
  panel  = one launch, fp32 E split in registers (odw_pairwise_sim, no workspace)
  ws     = split kernel + DMA kernel (odw_pairwise_sim_ws with its workspace: what _C.pairwise_sim runs)
  planes = DMA kernel alone on caller-provided planes (odw_pairwise_sim_planes)
and a plain fill of S for scale.  Prints one JSON line per P.   python tools/pairwise_forms.py [P ...]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import _lib as L

lib = L.lib()


def graph_time(launch, iters=30, reps=3):
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            launch()
    g.replay()
    best = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / iters * 1e3)
    return sorted(best)[len(best) // 2]


for P in [int(a) for a in sys.argv[1:]] or [2000, 4000, 8000]:
    E = torch.nn.functional.normalize(torch.randn(P, 128, device="cuda"), dim=1).contiguous()
    S = torch.empty(P, P, device="cuda")
    S2 = torch.empty(P, P, device="cuda")
    wsb = lib.odw_pairwise_sim_workspace(P, 128)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    st = L.stream
    forms = {
        "panel": lambda: L.check(lib.odw_pairwise_sim(L.ptr(E), P, 128, L.ptr(S), st()), "panel"),
        "ws": lambda: L.check(lib.odw_pairwise_sim_ws(L.ptr(E), P, 128, L.ptr(S2), L.ptr(ws), wsb, st()), "ws"),
        "planes": lambda: L.check(lib.odw_pairwise_sim_planes(L.ptr(ws), P, L.ptr(S2), st()), "planes"),
        "split": lambda: L.check(lib.odw_pairwise_split_planes(L.ptr(E), P, L.ptr(ws), st()), "split"),
        "fill": lambda: S.fill_(1.0),
    }
    forms["panel"](); forms["ws"]()
    torch.cuda.synchronize()
    same = bool(torch.equal(S, S2))
    err = float((S2.double() - E.double() @ E.double().t()).abs().max())
    nbytes = 4.0 * P * P + 512.0 * P
    r = {"P": P, "bit_identical_to_panel": same, "max_err_vs_fp64": err}
    for k, f in forms.items():
        us = graph_time(f)
        r[k + "_us"] = round(us, 2)
        if k in ("panel", "ws", "planes"):
            r[k + "_frac"] = round(nbytes / (us * 1e-6) / 8e12, 3)
    print(json.dumps(r), flush=True)
