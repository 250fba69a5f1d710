"""Fused SGD kernel over the VOC model's 153 M parameters: time and achieved HBM rate (22 B / parameter)."""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "all":
    for mode in ("0", "1", "2", "3"):
        for grid in ("2048", "8192", "32768"):
            env = dict(os.environ, ODW_SGD_MODE=mode, ODW_SGD_GRID=grid)
            subprocess.call([sys.executable, __file__], env=env)
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from od_wscl_amd import _lib as L
lib = L.lib()
n = 152_768_744
p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda"); b = torch.zeros(n, device="cuda")
sh = torch.empty(n, dtype=torch.bfloat16, device="cuda")
def run(first=0):
    L.check(lib.odw_sgd_momentum(L.ptr(p), L.ptr(g), L.ptr(b), L.ptr(sh), n, 1e-5, 1e-4, 0.9, 1.0, first, L.stream()), "sgd")
run(1)
for _ in range(3): run()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): run()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 100
print("mode %s grid %s: %.1f us = %.2f TB/s" % (os.environ.get("ODW_SGD_MODE", "0"), os.environ.get("ODW_SGD_GRID", "8192"), us, n * 22 / us / 1e6), flush=True)
