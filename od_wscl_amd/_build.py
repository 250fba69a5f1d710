"""Builds libodwscl.so (every HIP kernel + the C-ABI) for gfx950, in-tree.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build
container; the resulting .so is git-ignored but ships with gpurun snapshots.
"""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libodwscl.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -ffp-contract=off: ROI bin / sample coordinates must round exactly like the
# reference's separate fp32 mul and add (SURVEY.md s7 hard part vii); kernels
# that want FMAs ask for them explicitly (fmaf / MFMA).
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


# experiment builds (tools/exp, timing studies that knowingly produce wrong results): ODW_EXTRA_FLAGS="-DODW_EXPERIMENTS"
FLAGS += [f for f in os.environ.get("ODW_EXTRA_FLAGS", "").split() if f]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=False):
    """Compile csrc/*.hip -> libodwscl.so (objects cached next to the sources)."""
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    for src in sources():
        obj = os.path.splitext(src)[0] + ".o"
        objs.append(obj)
        newest_dep = max([os.path.getmtime(src)] + [os.path.getmtime(h) for h in
                         glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))])
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > newest_dep:
            continue
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
        if verbose and out:
            print(out.decode())
    cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="-f" in sys.argv, verbose=True))
