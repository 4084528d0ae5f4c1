"""Builds libmakani_amd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmakani_amd.so")
SOURCES = ["host.cpp", "fft.hip", "gemm.hip", "gemm_x3.hip", "layout.hip", "diag.hip", "pointwise.hip", "conv_gemm.hip"]
HEADERS = ["common.h", "fft_split.h", os.path.join("..", "..", "include", "makani_amd.h")]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; libmakani_amd.so cannot be built")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the shared library."""
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objs = []
    bdir = os.path.join(CSRC, "build")
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(bdir, src + ".o")
        cmd = [hipcc, "-O3", "-std=c++20", "--offload-arch=gfx950", "-fPIC", "-x", "hip",
               "-c", os.path.join(CSRC, src), "-o", obj]
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
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
