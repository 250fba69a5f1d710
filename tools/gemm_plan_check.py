"""Planner check on the step's mid-size products: time of the planner's choice against the forced ring / big variants
(split-K allowed in all three), to calibrate the per-CU rates of pick_plan."""
import os, sys, subprocess, json
shapes = [("fc7_fwd_stacked", 4000, 4096, 4096, "bf16"), ("fc7_fwd_krows", 448, 4096, 4096, "bf16"), ("fc7_dgrad", 2000, 4096, 4096, "bf16"),
          ("fc7_wgrad", 4096, 4096, 2752, "f32"), ("sim0_fwd", 2000, 4096, 4096, "bf16"), ("sim0_wgrad", 4096, 4096, 704, "f32"),
          ("pred_dgrad", 2000, 4096, 360, "bf16"), ("pred_wgrad", 357, 4096, 2000, "f32"),
          ("conv5_wgrad", 512, 4608, 5776, "f32"), ("conv3_wgrad", 256, 2304, 23104, "f32"), ("fc6_krows_fwd", 448, 4096, 25088, "bf16"),
          ("fc6_krows_dgrad", 448, 25088, 4096, "bf16")]
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from od_wscl_amd import gemm
    out = {}
    for name, M, N, K, dt in shapes:
        k64 = (K + 63) // 64 * 64
        a = (torch.randn(M, k64, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N, k64, device="cuda") * 0.5).bfloat16()
        o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16 if dt == "bf16" else torch.float32)
        for _ in range(3): gemm.gemm_nt(a, b, M, N, K, o)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): gemm.gemm_nt(a, b, M, N, K, o)
        e.record(); torch.cuda.synchronize()
        out[name] = round(s.elapsed_time(e) / 20 * 1e3, 1)
    print(json.dumps(out))
    sys.exit(0)
res = {}
for var in ("auto", "ring", "big"):
    env = dict(os.environ)
    if var != "auto":
        env["ODW_GEMM_VARIANT"] = var
    r = subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True)
    res[var] = json.loads(r.stdout.strip().splitlines()[-1])
print("%-18s %9s %9s %9s" % ("shape (us)", "auto", "ring", "big"))
for name, *_ in shapes:
    print("%-18s %9.1f %9.1f %9.1f" % (name, res["auto"][name], res["ring"][name], res["big"][name]))
