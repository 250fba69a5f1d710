"""The C-ABI library loads and exports every symbol include/odwscl.h declares (no compute)."""
import ctypes
import os
import re

from od_wscl_amd import _build, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "odwscl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(odw_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    path = _build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), "missing export " + n
        assert n in _lib.SIGNATURES, "no ctypes signature for " + n
    assert sorted(_lib.SIGNATURES) == names


def test_error_reporting_without_gpu():
    lib = _lib.lib()
    assert lib.odw_version() >= 1
    # argument validation happens before any launch: a bad call returns ODW_EINVAL and a message
    rc = lib.odw_pairwise_sim(None, 4, 3, None, None)
    assert rc == -1 and b"pairwise_sim" in lib.odw_last_error()
    rc = lib.odw_nms(None, None, 100000, 0.5, 0, None, None, None, 0, None)
    assert rc == -1 and b"nms" in lib.odw_last_error()
    assert lib.odw_roi_pool_workspace(2000, 7, 7) >= 2000 * 29 * 4


def test_cell_split_of_the_cell_major_product_fills_whole_rounds():
    """odw_gemm_nt_cm_workspace (host logic, no launch): one workgroup per CU, so T tiles x s splits take ceil(T s / 256)
    rounds of ceil(49 / s) cells -- the split count is the one with the fewest cells on the critical path; a product that
    fills the chip by itself is not split; the pair form's partials hold both halves."""
    lib = _lib.lib()
    N, S, ldw = 4096, 49, 4096
    splits = lambda M: lib.odw_gemm_nt_cm_workspace(M, N, S) // (M * ldw * 4)
    assert splits(2000) == 0 and splits(4000) == 0                   # 256 / 512 tiles of 256 x 128
    for M in (100, 300, 446, 892, 1400):
        s = splits(M)
        tiles = -(-M // 256) * (N // 128)
        cost = lambda k: -(-tiles * k // 256) * -(-S // k)
        assert s >= 2 and cost(s) <= min(cost(k) for k in range(1, 17)) + 1, (M, s)
        assert lib.odw_gemm_nt_cm_pair_workspace(M, N, S) == 2 * lib.odw_gemm_nt_cm_workspace(M, N, S)
    assert splits(446) == 4                                           # the sampled-row views of the bench step: one round


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "od_wscl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)
                assert "liboracle" not in src
