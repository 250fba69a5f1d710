"""cProfile of the host side of one fused Linear forward + backward (bf16x2f, 256 rows): where its ~200 us go."""
import cProfile, pstats, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from od_wscl_amd import precision
from od_wscl_amd.layers.linear import Linear
precision.set_precision(os.environ.get("MODE", "bf16x2f"))
torch.autograd.set_multithreading_enabled(False)
x = torch.randn(256, 4096, device="cuda").requires_grad_(True)
lin = Linear(4096, 4096).cuda()
def fwd_bwd():
    y = lin.fused(x, relu=True, drop_p=0.5, key=(1, 2))
    y.backward(y)
for _ in range(50): fwd_bwd()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(300): fwd_bwd()
print("host %.1f us per fwd+bwd" % ((time.perf_counter() - t) / 300 * 1e6)); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): fwd_bwd()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
