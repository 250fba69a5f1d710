import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import gemm
M, N, K = 4000, 4096, 25088
a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.5).bfloat16()
out = torch.empty(M, N, device="cuda")
for _ in range(5): gemm.gemm_nt(a, b, M, N, K, out)
torch.cuda.synchronize()
