"""Build the gfx950 C-ABI library (libpandora_mi355x.so) in-tree with hipcc.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
Every translation unit is compiled to its own object (in parallel, re-done only when that source, a header or the
flags changed) and the objects are linked into the one shared library: a kernel edit costs one file's compile time.
`-DPM_DIAG` (build(diag=True) -> libpandora_mi355x_diag.so) adds the diagnostic entry points and probe instantiations
that the shipped library does not carry (include/pandora_mi355x.h, "diagnostics").
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpandora_mi355x.so")
LIB_DIAG = os.path.join(HERE, "libpandora_mi355x_diag.so")
STAMP = LIB + ".stamp"
SOURCES = ["gemm.hip", "gemm_wide.hip", "gemm_wide_stream.hip", "gemm256.hip", "lngemm.hip", "attn.hip", "attn16.hip", "norm.hip", "misc.hip", "peer.hip"]
HEADERS = ["common.hpp", "gemm_common.hpp", "gemm_wide_loop.inc", "attn_common.hpp", os.path.join("..", "..", "include", "pandora_mi355x.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"]
LDFLAGS = ["--offload-arch=gfx950", "-fPIC", "-shared"]
# attn16.hip: no SLP vectoriser (it packs the softmax row sums of the two query blocks into v_pk_add_f32: see the file)
PER_FILE_FLAGS = {"attn16.hip": ["-fno-slp-vectorize"]}
FLAGS = CFLAGS + LDFLAGS  # (kept: tools that quote the build line)


def _hipcc():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return hipcc if os.path.exists(hipcc) else "hipcc"


def _includes(path, seen):
    """quoted #include closure of one source (headers of this repository only)"""
    import re
    with open(path, "rb") as f:
        text = f.read()
    seen[path] = text
    for m in re.finditer(rb'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
        inc = os.path.normpath(os.path.join(os.path.dirname(path), m.group(1).decode()))
        if inc not in seen and os.path.exists(inc):
            _includes(inc, seen)
    return seen


def _unit_digest(src, cflags):
    """source + the headers IT includes + flags: editing attn_common.hpp does not recompile the GEMM files"""
    h = hashlib.sha256()
    seen = _includes(os.path.join(CSRC, src), {})
    for path in sorted(seen):
        h.update(seen[path])
    h.update(" ".join(cflags + PER_FILE_FLAGS.get(src, [])).encode())
    return h.hexdigest()


def _digest(cflags=CFLAGS):
    h = hashlib.sha256()
    for s in SOURCES:
        h.update(_unit_digest(s, cflags).encode())
    h.update(" ".join(LDFLAGS).encode())
    return h.hexdigest()


def sources_digest():
    """Digest of everything the shipped library is built from (profiles quote it next to their numbers)."""
    return _digest()


def build(force=False, verbose=False, diag=False, jobs=None):
    """Compile every HIP source and link the shared library; no-op when nothing changed."""
    lib = LIB_DIAG if diag else LIB
    stamp = lib + ".stamp"
    cflags = CFLAGS + (["-DPM_DIAG=1"] if diag else [])
    dig = _digest(cflags)
    if not force and os.path.exists(lib) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == dig:
                return lib
    objdir = os.path.join(CSRC, "_obj_diag" if diag else "_obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()

    def unit(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        ud = _unit_digest(src, cflags)
        if not force and os.path.exists(obj) and os.path.exists(obj + ".stamp"):
            with open(obj + ".stamp") as f:
                if f.read().strip() == ud:
                    return obj
        cmd = [hipcc] + cflags + PER_FILE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        with open(obj + ".stamp", "w") as f:
            f.write(ud)
        return obj

    jobs = jobs or int(os.environ.get("PANDORA_BUILD_JOBS", "4"))
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(unit, SOURCES))
    tmp = lib + ".tmp"
    cmd = [hipcc] + LDFLAGS + objs + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(tmp, lib)  # (atomic: a concurrent reader / snapshot never sees a half-written library)
    with open(stamp, "w") as f:
        f.write(dig)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))
