"""Compile the reference's own CPU operator sources into oracle/_ref/.

TEST INFRASTRUCTURE ONLY.  Sources are compiled WHERE THEY LIE under
/root/reference/wetectron/csrc (vision.cpp + cpu/ROIAlign_cpu.cpp +
cpu/nms_cpu.cpp, no WITH_CUDA); nothing is copied into this repo and only the
resulting shared object lands in oracle/_ref/ (git-ignored, shipped to the GPU
box).  The .cu files need THC headers that no longer exist in torch and are
not built (and not hipified, by policy).

Gives: roi_align_forward (CPU), nms (CPU, `>=` rule).  roi_pool_* and
roi_align_backward raise "Not implemented on the CPU" in the reference
(csrc/ROIPool.h:23,44, csrc/ROIAlign.h:44).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_CSRC = "/root/reference/wetectron/csrc"
OUT = os.path.join(HERE, "_ref")
NAME = "wetectron_ref_C"


def built_path():
    p = os.path.join(OUT, NAME + ".so")
    return p if os.path.exists(p) else None


def build(verbose=False):
    """Build (if the reference tree is present) and return the .so path or None."""
    if built_path():
        return built_path()
    if not os.path.isdir(REF_CSRC):
        return None
    from torch.utils.cpp_extension import load
    os.makedirs(OUT, exist_ok=True)
    load(
        name=NAME,
        sources=[os.path.join(REF_CSRC, "vision.cpp"),
                 os.path.join(REF_CSRC, "cpu", "ROIAlign_cpu.cpp"),
                 os.path.join(REF_CSRC, "cpu", "nms_cpu.cpp")],
        extra_include_paths=[REF_CSRC],
        extra_cflags=["-O2"],
        build_directory=OUT,
        verbose=verbose,
        is_python_module=True,
    )
    return built_path()


def load_ref():
    """Import the built module (or None when it was never built)."""
    p = built_path()
    if p is None:
        return None
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location(NAME, p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    p = build(verbose="-v" in sys.argv)
    print("oracle/_ref:", p)
