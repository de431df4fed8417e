"""Build recipe for libofq_hip.so (hand-written HIP for gfx950, no torch dependency, plain C ABI).

    python -m ofq_amd.build            # or ofq_amd.build.build()

hipcc cross-compiles without a GPU; the .so is kept in-tree (ofq_amd/lib/) so that it travels with the
repository snapshot to the GPU box.  -ffp-contract=off keeps the fp32 rounding sequence of the
quantiser formulas (no fma contraction), which is what makes the integer levels bit-exact.
"""
import fcntl
import hashlib
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
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".h"))] + [inc]


def source_hash():
    """sha256 over the kernel sources and the C header, first 16 hex digits: compiled into the library
    (ofq_source_hash()) so that a stale .so is detected by content, not by file times (which a snapshot copy changes)."""
    h = hashlib.sha256()
    for s in _sources():
        h.update(os.path.basename(s).encode())
        with open(s, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def built_hash(path=OUT):
    """The source hash embedded in an existing library, or None (missing / unreadable / pre-hash build)."""
    if not os.path.exists(path):
        return None
    marker = b"OFQ_SOURCE_HASH="
    with open(path, "rb") as fh:
        blob = fh.read()
    i = blob.find(marker)
    return None if i < 0 else blob[i + len(marker):i + len(marker) + 16].decode("ascii", "replace")


def needs_build():
    return built_hash() != source_hash()


def build(force=False, verbose=False):
    """Compile under an exclusive file lock with a per-process temporary name: ranks started together by torchrun /
    mp.spawn serialise here, the first one builds, the others find the finished library when they get the lock."""
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return OUT
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            tmp = "%s.%d.tmp" % (OUT, os.getpid())
            cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                   '-DOFQ_SOURCE_HASH="%s"' % source_hash(), SRC, "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, OUT)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
            return OUT
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
