"""The dominant launch of the step in isolation (for tools/pmc_gemm.sh): the stacked fc6 forward,
gemm_nt_bf16_big_kernel<true, 0>, M=4000 N=4096 K=25088, bf16 output with bias + ReLU + dropout epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import gemm
M, N, K = 4000, 4096, 25088 * int(os.environ.get("ODW_ONE_PLANES", "1"))     # 3: the bf16x2f stacked fc6 forward (K' = 3 K)
if os.environ.get("ODW_ONE_SHAPE"):          # "M,N,K": a plain product of that shape, fp32 out (e.g. fc6's input gradient 2000,25088,4096)
    M, N, K = (int(v) for v in os.environ["ODW_ONE_SHAPE"].split(","))
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
    out = torch.empty(M, N, device="cuda")
    for _ in range(5):
        gemm.gemm_nt(a, b, M, N, K, out)
    torch.cuda.synchronize()
    sys.exit(0)
a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16 if os.environ.get("ODW_ONE_F32") != "1" else torch.float32)
for _ in range(5):
    gemm.gemm_nt(a, b, M, N, K, out, bias=bias, relu=True, drop_p=0.5, segs=[(0, 1, 2), (2000, 3, 4)])
torch.cuda.synchronize()
