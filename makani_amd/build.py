"""Builds libmakani_amd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmakani_amd.so")
STAMP = LIB + ".stamp"
SOURCES = ["host.cpp", "fft.hip", "gemm.hip", "gemm_x3.hip", "layout.hip", "diag.hip", "pointwise.hip", "conv_gemm.hip", "pce.hip", "pce_mlp.hip", "adam.hip"]
HEADERS = ["common.h", "fft_split.h", "lds_dma.h", "pce_common.h", os.path.join("..", "..", "include", "makani_amd.h")]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; libmakani_amd.so cannot be built")


def have_hipcc():
    return any(c and os.path.exists(c) for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"))


def _source_digest():
    """sha256 over every source and header the library is built from (content, not mtime: a snapshot copied to
    another machine keeps its digest whatever happens to the file times)."""
    import hashlib
    h = hashlib.sha256()
    h.update(os.environ.get("MK_EXTRA_HIPCC_FLAGS", "").encode() + b"\0")
    for rel in SOURCES + HEADERS:
        with open(os.path.join(CSRC, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def stale():
    """True when libmakani_amd.so is missing or was built from other sources than the ones on disk."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _source_digest()


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the shared library."""
    if not force and not stale():
        return LIB
    hipcc = _hipcc()
    objs = []
    bdir = os.path.join(CSRC, "build")
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(bdir, src + ".o")
        cmd = [hipcc, "-O3", "-std=c++20", "--offload-arch=gfx950", "-fPIC", "-x", "hip",
               "-c", os.path.join(CSRC, src), "-o", obj] + os.environ.get("MK_EXTRA_HIPCC_FLAGS", "").split()
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout}")
    # a host stub that silently failed to compile shows up as an undefined symbol at load time: check here, not on the GPU box
    import ctypes
    try:
        ctypes.CDLL(LIB)
    except OSError as e:
        raise RuntimeError(f"{LIB} was linked but does not load: {e}") from e
    with open(STAMP, "w") as f:
        f.write(_source_digest() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
