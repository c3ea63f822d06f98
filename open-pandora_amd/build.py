"""Build the gfx950 C-ABI library (libpandora_mi355x.so) in-tree with hipcc.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpandora_mi355x.so")
STAMP = LIB + ".stamp"
SOURCES = ["gemm.hip", "gemm256.hip", "lngemm.hip", "attn.hip", "norm.hip", "misc.hip", "peer.hip"]
HEADERS = ["common.hpp", "gemm_common.hpp", os.path.join("..", "..", "include", "pandora_mi355x.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-Wno-unused-result"]


def _digest():
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile every HIP source into one shared library; no-op when sources are unchanged."""
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == dig:
                return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
