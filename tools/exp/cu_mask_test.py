"""Does reserving a few compute units keep a chain of tiny dependent launches from starving beside a chip-filling GEMM?
A: the GEMM loop on an ordinary stream; B: on a stream whose CU mask leaves `spare` CUs out.  The chain (70 launches of a
5 us kernel) runs on a high-priority stream in both.  python tools/exp/cu_mask_test.py"""
import ctypes
import sys
import torch
from od_wscl_amd import _lib as L, gemm, precision

precision.set_precision("bf16")
dev = torch.device("cuda:0")
M, N, K = 2000, 25088, 4096
a = (torch.randn(M, K, device=dev) * 0.1).bfloat16()
b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
out = torch.empty(M, N, device=dev)
small = torch.zeros(4096, device=dev)
props = torch.cuda.get_device_properties(0)
ncu = props.multi_processor_count
print("CUs:", ncu)


def masked_stream(spare, stride):
    words = (ncu + 31) // 32
    bits = [1] * ncu
    for i in range(spare):
        bits[(i * stride) % ncu] = 0
    arr = (ctypes.c_uint32 * words)()
    for i, v in enumerate(bits):
        if v:
            arr[i // 32] |= (1 << (i % 32))
    h = ctypes.c_void_p()
    L.check(L.lib().odw_stream_create_cu_mask(words, ctypes.cast(arr, ctypes.c_void_p), ctypes.byref(h)), "cu mask stream")
    return torch.cuda.ExternalStream(h.value, device=dev)


def run(big_stream, chain_stream, n_gemm=3, n_chain=70):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    fork = torch.cuda.Event()
    main = torch.cuda.current_stream()
    e0.record(main)
    fork.record(main)
    if big_stream is not None:
        big_stream.wait_event(fork)
        with torch.cuda.stream(big_stream):
            for _ in range(n_gemm):
                gemm.gemm_nt(a, b, M, N, K, out)
            e1.record(big_stream)
    chain_stream.wait_event(fork)
    with torch.cuda.stream(chain_stream):
        for _ in range(n_chain):
            small.add_(1.0)
        e2.record(chain_stream)
    torch.cuda.synchronize()
    return (e0.elapsed_time(e1) if big_stream is not None else 0.0), e0.elapsed_time(e2)


hp = torch.cuda.Stream(device=dev, priority=-1)
plain = torch.cuda.Stream(device=dev)
for _ in range(2):
    run(plain, hp)
print("chain alone: %.3f ms" % min(run(None, hp)[1] for _ in range(5)))
r = [run(plain, hp) for _ in range(5)]
print("GEMMs on an ordinary stream:  GEMMs end %.3f ms, chain ends %.3f ms" % (min(x[0] for x in r), min(x[1] for x in r)))
for spare, stride in ((8, 32), (16, 16), (32, 8), (16, 1), (64, 4)):
    ms = masked_stream(spare, stride)
    for _ in range(2):
        run(ms, hp)
    r = [run(ms, hp) for _ in range(5)]
    print("GEMMs on a stream without %2d CUs (every %2d-th): GEMMs end %.3f ms, chain ends %.3f ms"
          % (spare, stride, min(x[0] for x in r), min(x[1] for x in r)))
