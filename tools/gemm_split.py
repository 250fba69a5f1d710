"""Split-K planner check on the small-C / long-K products of the step: planned (split) vs unsplit, TFLOP/s."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import gemm, _lib as L
from gemm_bench import timeit  # noqa
shapes = [("conv5 wgrad", 512, 4608, 5776), ("conv4_1 wgrad", 512, 2304, 5776), ("conv3 wgrad", 256, 2304, 23104),
          ("conv3_1 wgrad", 256, 1152, 23104), ("fc6 fwd K-rows", 446, 4096, 25088), ("fc7 wgrad", 4096, 4096, 4032),
          ("sim0 wgrad K-rows", 4096, 4096, 448), ("fc6 dgrad K-rows", 446, 25088, 4096), ("fc7 fwd", 4000, 4096, 4096)]
for name, M, N, K in shapes:
    k64 = (K + 63) // 64 * 64
    a = (torch.randn(M, k64, device="cuda") * 0.1).bfloat16(); b = (torch.randn(N, k64, device="cuda") * 0.1).bfloat16()
    out = torch.empty(M, N, device="cuda")
    var = ctypes.c_int(0)
    ws = L.lib().odw_gemm_nt_bf16_workspace(M, N, K, k64, k64, L.ptr(out), N, 0, ctypes.byref(var))
    res = {"shape": name, "MNK": [M, N, K], "variant": var.value, "splits": ws // (M * N * 4)}
    fl = 2.0 * M * N * K
    os.environ.pop("ODW_GEMM_SPLITK", None)
    ms = timeit(lambda: gemm.gemm_nt(a, b, M, N, K, out), iters=20); res["planned_us"] = round(ms * 1e3, 1); res["planned_TF"] = round(fl / ms / 1e9, 1)
    os.environ["ODW_GEMM_SPLITK"] = "1"
    ms = timeit(lambda: gemm.gemm_nt(a, b, M, N, K, out), iters=20); res["unsplit_us"] = round(ms * 1e3, 1); res["unsplit_TF"] = round(fl / ms / 1e9, 1)
    os.environ.pop("ODW_GEMM_SPLITK", None)
    ms = timeit(lambda: torch.matmul(a[:, :K], b[:, :K].T), iters=20); res["hipblaslt_us"] = round(ms * 1e3, 1)
    print(json.dumps(res), flush=True)
