"""The dominant launch of the bf16x2f step in isolation (for tools/pmc_pair.sh): the shared clean + DropBlock fc6 forward,
gemm_nt_cm_kernel<true, true>, P=2000 ROIs, N=4096, C=512 x 49 cells, cell-major planes [hi | mid], fp32 out (2P rows),
bias + ReLU + dropout epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import gemm, precision
precision.set_precision("bf16x2f")
P, N, C, S = int(os.environ.get("ODW_ONE_P", "2000")), 4096, 512, 49
x = torch.relu(torch.randn(P, C * S, device="cuda")) * 0.7
w = torch.randn(N, C * S, device="cuda") * 0.01
bias = torch.randn(N, device="cuda") * 0.1
keep = (torch.rand(P, S, device="cuda") > 0.45).float()
ksum = keep.sum()
xc, wc = gemm.split_rows_cm(x, C, S), gemm.split_rows_cm(w, C, S)
del x, w
out = torch.empty(2 * P, N, device="cuda")
for _ in range(5):
    gemm.gemm_nt_cm(xc, wc, P, N, C, S, out, bias=bias, relu=True, drop_p=0.5, segs=[(0, 1, 2), (P, 3, 4)], keep=keep,
                    keep_sum=ksum, drop_row0=P)
torch.cuda.synchronize()
