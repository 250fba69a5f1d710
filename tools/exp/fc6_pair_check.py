"""The shared clean + DropBlock fc6 forward (odw_gemm_nt_cm, pair form) against the stacked pass it replaces: same inputs,
same dropout keys; values compared, both timed."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from od_wscl_amd import gemm, precision, _lib as L
precision.set_precision("bf16x2f")
lib = L.lib()
P, N, C, S = int(os.environ.get("P", 2000)), 4096, 512, 49
K = C * S
torch.manual_seed(0)
x = torch.relu(torch.randn(P, K, device="cuda")) * 0.7
w = torch.randn(N, K, device="cuda") * 0.01
bias = torch.randn(N, device="cuda") * 0.1
keep = (torch.rand(P, S, device="cuda") > 0.45).float()
ksum = keep.sum()
pa, pb = precision.patterns("gemm")
# the stacked pass: rows P.. = x * keep * numel / sum (channel-major k = c * S + s)
xd = (x.view(P, C, S) * keep[:, None, :] * keep.numel() / ksum).reshape(P, K)
xs = precision.split_rows(torch.cat([x, xd]), pa, K)
ws = precision.split_rows(w, pb, K)
segs = [(0, 11, 12), (P, 13, 14)]
ref = torch.empty(2 * P, N, device="cuda")
def run_ref():
    gemm.gemm_nt(xs, ws, 2 * P, N, 3 * K, ref, bias=bias, relu=True, drop_p=0.5, segs=segs)
# cell-major planes
xc = torch.empty(P, 2 * K, dtype=torch.bfloat16, device="cuda")
wc = torch.empty(N, 2 * K, dtype=torch.bfloat16, device="cuda")
L.check(lib.odw_split_rows_cm(L.ptr(x), K, P, C, S, L.ptr(xc), 2 * K, K, L.stream()), "split x")
L.check(lib.odw_split_rows_cm(L.ptr(w), K, N, C, S, L.ptr(wc), 2 * K, K, L.stream()), "split w")
# the planes are a permutation of split_rows' planes
hi_nat = xs[:P, :K].view(P, C, S).permute(0, 2, 1).reshape(P, K)
mid_nat = xs[:P, 2 * K:].view(P, C, S).permute(0, 2, 1).reshape(P, K)
print("planes: hi equal", torch.equal(hi_nat, xc[:, :K]), " mid equal", torch.equal(mid_nat, xc[:, K:]))
out = torch.full((2 * P, N), float("nan"), device="cuda")
rows = (ctypes.c_int * 4)(0, P, 0, 0)
keys = (ctypes.c_uint32 * 8)(11, 12, 13, 14, 0, 0, 0, 0)
def run_new():
    L.check(lib.odw_gemm_nt_cm(L.ptr(xc), 2 * K, K, L.ptr(wc), 2 * K, K, P, N, C, S, L.ptr(keep), L.ptr(ksum), P, L.ptr(out), N,
                               L.ptr(bias), 1, 0.5, 2, ctypes.cast(rows, ctypes.c_void_p), ctypes.cast(keys, ctypes.c_void_p), None,
                               None, 0, L.stream()), "cm")
run_ref(); run_new(); torch.cuda.synchronize()
x64 = torch.cat([x, xd]).double()
for name, lo in (("clean", 0), ("drop", P)):
    a, b = out[lo:lo + P], ref[lo:lo + P]
    same_mask = ((a == 0) == (b == 0)).float().mean().item()
    print("%-5s max|new-ref| %.3e  (max|ref| %.2f)  zero pattern agreement %.6f  nan %d" %
          (name, (a - b).abs().max().item(), b.abs().max().item(), same_mask, int(torch.isnan(a).sum())))
# against float64 on a row sample (pre-dropout values recovered where kept)
idx = torch.arange(0, 2 * P, 97, device="cuda")
y64 = torch.relu(x64[idx] @ w.double().T + bias.double()) * 2.0
for name, t in (("new", out), ("ref", ref)):
    got = t[idx].double()
    m = got != 0
    print("%s vs fp64 on kept entries: max abs %.3e" % (name, ((got - y64).abs() * m).max().item()))
def timeit(f, iters=8):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
print("stacked pass %.3f ms   shared pass %.3f ms" % (timeit(run_ref), timeit(run_new)))
