"""The bf16 MFMA GEMM and the fused Linear op against PyTorch fp32 references of the same
product on bf16-rounded operands (so the only difference is accumulation order).  -m gpu."""
import numpy as np
import pytest
import torch

from od_wscl_amd.utils import rng

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _bf16_precision():
    """These tests pin the bf16 kernels against fp32 references on bf16-rounded operands (the fp32-grade split mode
    has its own file, test_split_gpu.py)."""
    from od_wscl_amd import precision
    precision.set_precision("bf16")
    yield


@pytest.fixture(scope="module")
def G():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from od_wscl_amd import gemm
    return gemm


def rnd(seed, shape, scale=1.0):
    n = int(np.prod(shape))
    return torch.from_numpy((rng.normal(seed, 1, n) * scale).reshape(shape)).cuda()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1, 1, 8), (257, 357, 4096), (300, 128, 72), (2000, 512, 2000),
                                   (64, 4096, 1024), (130, 70, 25088)])
def test_gemm_nt_matches_fp32_reference(G, M, N, K):
    k8 = (K + 7) // 8 * 8
    a = torch.zeros(M, k8, device="cuda").bfloat16()
    b = torch.zeros(N, k8, device="cuda").bfloat16()
    a[:, :K] = rnd(1, (M, K)).bfloat16()
    b[:, :K] = rnd(2, (N, K)).bfloat16()
    out = torch.empty(M, N, device="cuda")
    G.gemm_nt(a, b, M, N, K, out)
    ref = a.float() @ b.float().T
    # exact products, fp32 accumulation in a different order: error ~ sqrt(K) * eps * |a||b|
    tol = 1e-5 * np.sqrt(K) * 4
    assert (out - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item()), (out - ref).abs().max().item()
    ob = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    G.gemm_nt(a, b, M, N, K, ob)
    assert (ob.float() - ref).abs().max().item() <= 8e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 357, 4096), (2000, 512, 1984), (130, 70, 25088), (5, 3, 64),
                                   (1000, 700, 128), (513, 257, 192)])
def test_gemm_lds_dma_variant(G, M, N, K):
    """Operands padded to a multiple of 64 in K take the global_load_lds path; it must agree with the
    register-staged kernel (selected by ODW_GEMM_VARIANT=reg) to fp32 re-association."""
    import os
    k64 = (K + 63) // 64 * 64
    a = torch.zeros(M, k64, device="cuda").bfloat16()
    b = torch.zeros(N, k64, device="cuda").bfloat16()
    a[:, :K] = rnd(21, (M, K)).bfloat16()
    b[:, :K] = rnd(22, (N, K)).bfloat16()
    o1 = torch.empty(M, N, device="cuda")
    o2 = torch.empty(M, N, device="cuda")
    G.gemm_nt(a, b, M, N, K, o1)
    o3 = torch.empty(M, N, device="cuda")
    os.environ["ODW_GEMM_VARIANT"] = "reg"
    try:
        G.gemm_nt(a, b, M, N, K, o2)
        os.environ["ODW_GEMM_VARIANT"] = "ring"          # 256x128 tile, 3-stage ring, counted vmcnt
        G.gemm_nt(a, b, M, N, K, o3)


    finally:
        del os.environ["ODW_GEMM_VARIANT"]
    ref = a.float() @ b.float().T
    tol = 1e-5 * np.sqrt(K) * 4 * max(1.0, ref.abs().max().item())
    assert (o1 - ref).abs().max().item() <= tol and (o2 - ref).abs().max().item() <= tol
    assert (o3 - ref).abs().max().item() <= tol




def test_gemm_asymmetric_identity(G):
    # A = I, asymmetric B: catches transposed / permuted C writes
    n = 128
    a = torch.eye(n, device="cuda").bfloat16()
    b = (torch.arange(n * n, device="cuda").reshape(n, n) % 251).float().bfloat16()
    out = torch.empty(n, n, device="cuda")
    G.gemm_nt(a, b, n, n, n, out)
    assert torch.equal(out, b.float().T)


def test_gemm_epilogue_bias_relu_dropout_accumulate(G):
    M, N, K = 300, 200, 256
    a, b = rnd(3, (M, K)).bfloat16(), rnd(4, (N, K)).bfloat16()
    bias = rnd(5, (N,))
    ref = torch.relu(a.float() @ b.float().T * 0.5 + bias)
    k1, k2 = rng.stream_key(7, 11), rng.stream_key(7, 12)
    out = torch.empty(M, N, device="cuda")
    G.gemm_nt(a, b, M, N, K, out, bias=bias, relu=True, alpha=0.5, drop_p=0.5, segs=[(0,) + k1, (100,) + k2])
    keep = np.concatenate([rng.uniform(7, 11, 100 * N).reshape(100, N), rng.uniform(7, 12, 200 * N).reshape(200, N)]) >= 0.5
    exp = ref * torch.from_numpy(keep).cuda() * 2.0
    assert (out - exp).abs().max().item() <= 1e-3
    acc = torch.ones(M, N, device="cuda")
    G.gemm_nt(a, b, M, N, K, acc, accumulate=True)
    assert (acc - (1 + a.float() @ b.float().T)).abs().max().item() <= 1e-3


@pytest.mark.parametrize("variant", ["", "big", "ring"])
def test_gemm_carries_eight_dropout_segments(G, variant, monkeypatch):
    """One launch carries the keys of up to gemm.MAX_SEGS = 8 stacked passes (4 until round 5: an image with three positive
    classes stacks six sampled-row views, loss.py:292-305, and every Linear ran twice): every segment draws from its OWN
    stream, numbered from its first row -- against the generator itself, in every kernel variant (direct and split-K)."""
    if variant:
        monkeypatch.setenv("ODW_GEMM_VARIANT", variant)
    assert G.MAX_SEGS == 8
    M, N, K = 1100, 264, 512
    a, b = rnd(51, (M, K)).bfloat16(), rnd(52, (N, K)).bfloat16()
    bias = rnd(53, (N,))
    starts = [0, 37, 300, 301, 555, 800, 801, 1000]
    segs = [(r,) + rng.stream_key(13, 40 + i) for i, r in enumerate(starts)]
    out = torch.empty(M, N, device="cuda")
    G.gemm_nt(a, b, M, N, K, out, bias=bias, relu=True, drop_p=0.5, segs=segs)
    keep = np.concatenate([rng.uniform(13, 40 + i, (e - r) * N).reshape(e - r, N)
                           for i, (r, e) in enumerate(zip(starts, starts[1:] + [M]))]) >= 0.5
    exp = torch.relu(a.float() @ b.float().T + bias) * torch.from_numpy(keep).cuda() * 2.0
    assert ((out == 0) == (exp == 0)).all()                     # the zero pattern IS the draw
    assert (out - exp).abs().max().item() <= 1e-3 * max(1.0, exp.abs().max().item())
    with pytest.raises(AssertionError):
        G.gemm_nt(a, b, M, N, K, out, drop_p=0.5, segs=segs + [(1050,) + rng.stream_key(13, 99)])


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (700, 520, 320), (4000, 1032, 512), (130, 72, 25088), (257, 8, 64)])
def test_gemm_256x256_variant(G, M, N, K, monkeypatch):
    """The 256x256-tile kernel (asm-scheduled slices, skewed barrier, LDS-staged coalesced epilogue) forced on
    ragged shapes: plain product in both output types, then every fused epilogue feature at once."""
    monkeypatch.setenv("ODW_GEMM_VARIANT", "big")
    k64 = (K + 63) // 64 * 64
    a = torch.zeros(M, k64, device="cuda").bfloat16()
    b = torch.zeros(N, k64, device="cuda").bfloat16()
    a[:, :K] = rnd(31, (M, K)).bfloat16()
    b[:, :K] = rnd(32, (N, K)).bfloat16()
    ref = a.float() @ b.float().T
    tol = 1e-5 * np.sqrt(K) * 4 * max(1.0, ref.abs().max().item())
    out = torch.full((M + 1, N + 8), 7.0, device="cuda")
    G.gemm_nt(a, b, M, N, K, out[:M, :N])
    assert (out[:M, :N] - ref).abs().max().item() <= tol
    assert (out[M] == 7).all() and (out[:, N:] == 7).all()                  # nothing written outside C
    ob = torch.full((M + 1, N + 8), 7.0, device="cuda", dtype=torch.bfloat16)
    G.gemm_nt(a, b, M, N, K, ob[:M, :N])
    assert (ob[:M, :N].float() - ref).abs().max().item() <= 8e-3 * max(1.0, ref.abs().max().item())
    assert (ob[M] == 7).all() and (ob[:, N:] == 7).all()
    bias = rnd(33, (N,))
    s1 = min(M - 1, 100)
    k1, k2 = rng.stream_key(9, 21), rng.stream_key(9, 22)
    for dt, atol in ((torch.float32, 2e-3), (torch.bfloat16, 8e-3)):
        o = torch.empty(M, N, device="cuda", dtype=dt)
        G.gemm_nt(a, b, M, N, K, o, bias=bias, relu=True, alpha=0.5, drop_p=0.5, segs=[(0,) + k1, (s1,) + k2])
        keep = np.concatenate([rng.uniform(9, 21, s1 * N).reshape(s1, N),
                               rng.uniform(9, 22, (M - s1) * N).reshape(M - s1, N)]) >= 0.5
        exp = torch.relu(ref * 0.5 + bias) * torch.from_numpy(keep).cuda() * 2.0
        assert (o.float() - exp).abs().max().item() <= atol * max(1.0, exp.abs().max().item())
    acc = torch.ones(M, N, device="cuda")
    G.gemm_nt(a, b, M, N, K, acc, accumulate=True)
    assert (acc - (1 + ref)).abs().max().item() <= tol


@pytest.mark.parametrize("M,N,K,forced", [(512, 4608, 5776, None), (256, 2304, 23104, None), (446, 4096, 25088, None),
                                          (300, 200, 4096, 4), (130, 72, 1024, 3), (64, 576, 36864, None),
                                          (2000, 357, 4096, None), (97, 61, 2048, 4)])
def test_gemm_split_k(G, M, N, K, forced, monkeypatch):
    """Products with a small C and a long K are split along K (fp32 partials + one reduction pass that applies the
    fused epilogue): conv weight-gradient shapes, the fc6 pass over the sampled rows, and forced splits on ragged
    shapes; against the fp32 reference and against the unsplit kernel."""
    import ctypes
    from od_wscl_amd import _lib as L
    if forced:
        monkeypatch.setenv("ODW_GEMM_SPLITK", str(forced))
    k64 = (K + 63) // 64 * 64
    a = torch.zeros(M, k64, device="cuda").bfloat16()
    b = torch.zeros(N, k64, device="cuda").bfloat16()
    a[:, :K] = (rnd(41, (M, K)) * 0.1).bfloat16()
    b[:, :K] = (rnd(42, (N, K)) * 0.1).bfloat16()
    out = torch.empty(M, N, device="cuda")
    var = ctypes.c_int(0)
    ws = L.lib().odw_gemm_nt_bf16_workspace(M, N, K, k64, k64, L.ptr(out), N, 0, ctypes.byref(var))
    ldw = (N + 3) // 4 * 4          # partial rows are padded to a multiple of 4 floats (N = 357: the fused predictor)
    assert ws > 0 and ws % (M * ldw * 4) == 0, "the planner should split this product"
    ref = a.float() @ b.float().T
    tol = 1e-5 * np.sqrt(K) * 4 * max(1.0, ref.abs().max().item())
    G.gemm_nt(a, b, M, N, K, out)
    assert (out - ref).abs().max().item() <= tol
    bias = rnd(43, (N,))
    s1 = M // 3
    k1, k2 = rng.stream_key(9, 31), rng.stream_key(9, 32)
    keep = np.concatenate([rng.uniform(9, 31, s1 * N).reshape(s1, N), rng.uniform(9, 32, (M - s1) * N).reshape(M - s1, N)]) >= 0.5
    exp = torch.relu(ref * 0.5 + bias) * torch.from_numpy(keep).cuda() * 2.0
    for dt, atol in ((torch.float32, 2e-3), (torch.bfloat16, 8e-3)):
        o = torch.empty(M, N, device="cuda", dtype=dt)
        G.gemm_nt(a, b, M, N, K, o, bias=bias, relu=True, alpha=0.5, drop_p=0.5, segs=[(0,) + k1, (s1,) + k2])
        assert (o.float() - exp).abs().max().item() <= atol * max(1.0, exp.abs().max().item())
    acc = torch.ones(M, N, device="cuda")
    G.gemm_nt(a, b, M, N, K, acc, accumulate=True)
    assert (acc - (1 + ref)).abs().max().item() <= tol
    monkeypatch.setenv("ODW_GEMM_SPLITK", "1")                 # the same product unsplit
    o1 = torch.empty(M, N, device="cuda")
    G.gemm_nt(a, b, M, N, K, o1)
    assert (o1 - out).abs().max().item() <= tol


def test_transpose_and_convert(G):
    x = rnd(6, (70, 45))
    t = G.transpose_bf16(x, 70, 45)
    assert t.shape == (45, 128)
    assert torch.equal(t[:, :70], x.bfloat16().T) and (t[:, 70:] == 0).all()
    tb = G.transpose_bf16(x.bfloat16().contiguous(), 70, 45)
    assert torch.equal(tb, t)
    assert torch.equal(G.to_bf16(x), x.bfloat16())


@pytest.mark.parametrize("M,K,N,relu,drop,out_f32", [(300, 512, 256, True, 0.5, False), (190, 4096, 357, False, 0.0, True),
                                                      (64, 1024, 128, True, 0.0, False)])
def test_fused_linear_forward_backward(G, M, K, N, relu, drop, out_f32):
    x = rnd(8, (M, K), 0.5).requires_grad_(True)
    w = torch.nn.Parameter(rnd(9, (N, K), 0.05))
    b = torch.nn.Parameter(rnd(10, (N,), 0.1))
    key = rng.stream_key(3, 4)
    y = G.fused_linear(x, w, b, G.Shadow(w), relu=relu, drop_p=drop, segs=[(0,) + key] if drop else None, out_f32=out_f32)
    g = rnd(11, (M, N))
    y.backward(g.to(y.dtype))
    # fp32 reference on bf16-rounded operands
    xr = x.detach().bfloat16().float().requires_grad_(True)
    wr = w.detach().bfloat16().float().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True)
    z = xr @ wr.T + br
    if relu:
        z = torch.relu(z)
    if drop:
        keep = torch.from_numpy(rng.uniform(3, 4, M * N).reshape(M, N) >= drop).cuda()
        z = z * keep * (1.0 / (1 - drop))
    z.backward(g.to(y.dtype).float())
    s = lambda t: max(t.abs().max().item(), 1e-6)
    assert (y.float() - z).abs().max().item() <= 1e-2 * s(z)
    assert (x.grad - xr.grad).abs().max().item() <= 2e-2 * s(xr.grad)
    assert (w.grad - wr.grad).abs().max().item() <= 2e-2 * s(wr.grad)
    assert (b.grad - br.grad).abs().max().item() <= 2e-2 * s(br.grad)


def test_sgd_kernel(G):
    from od_wscl_amd import _lib as L
    n = 100003
    p, g = rnd(12, (n,)), rnd(13, (n,))
    p0 = p.clone()
    buf = torch.empty(n, device="cuda")
    sh = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([ref_p], lr=0.01, momentum=0.9, weight_decay=1e-4)
    for it in range(3):
        ref_p.grad = g.clone()
        opt.step()
        L.check(L.lib().odw_sgd_momentum(L.ptr(p), L.ptr(g), L.ptr(buf), L.ptr(sh), n, 0.01, 1e-4, 0.9, 1.0,
                                         1 if it == 0 else 0, L.stream()), "sgd")
    assert (p - ref_p.detach()).abs().max().item() <= 1e-6
    assert torch.equal(sh, p.bfloat16())
