"""Build recipe for libofq_hip.so (hand-written HIP for gfx950, no torch dependency, plain C ABI).

    python -m ofq_amd.build            # or ofq_amd.build.build()

hipcc cross-compiles without a GPU; the .so is kept in-tree (ofq_amd/lib/) so that it travels with the
repository snapshot to the GPU box.  -ffp-contract=off keeps the fp32 rounding sequence of the
quantiser formulas (no fma contraction), which is what makes the integer levels bit-exact.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "libofq.hip")
OUT_DIR = os.path.join(HERE, "lib")
OUT = os.path.join(OUT_DIR, "libofq_hip.so")
ARCH = "gfx950"


def _sources():
    d = os.path.join(HERE, "csrc")
    inc = os.path.join(os.path.dirname(HERE), "include", "ofq_hip.h")
    return [os.path.join(d, f) for f in sorted(os.listdir(d))] + [inc]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(s) > t for s in _sources())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           SRC, "-o", OUT + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
