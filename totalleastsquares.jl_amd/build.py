"""Build recipe for libtlsqhip.so (gfx950 only): hipcc -> one shared object, in-tree.

    python totalleastsquares.jl_amd/build.py [--force]

The .so is git-ignored but travels to the GPU box with the repo snapshot.  hipcc cross-compiles
without a GPU.  No torch, no cmake: four HIP translation units + the host API.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libtlsqhip.so")
SOURCES = ["sweeps.hip", "fused.hip", "gemm.hip", "opgram32.hip", "opgram16.hip", "gram16.hip", "jacobi.hip", "hankel.hip", "hankelop.hip", "lanczos.hip", "subspace.hip", "matfun.hip", "cholesky.hip", "sliced.hip", "tsqr.hip", "batched.hip", "complex.hip", "grassmann.hip", "runtime.hip", "staging.hip", "svdstep.hip", "solver.hip", "solver_complex.hip", "entry.hip", "api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(ROOT, "include")]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


LAST_COMPILE_COUNT = 0   # hipcc -c commands of the last build() call


def build(force=False, verbose=True):
    global LAST_COMPILE_COUNT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "internal.hpp"), os.path.join(CSRC, "svdstep.hpp"),
               os.path.join(ROOT, "include", "tlsq.h")]
    extra = os.environ.get("TLSQ_EXTRA_FLAGS", "").split()   # development builds (tools/kbench.py ablations)
    force = force or bool(extra)
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if not force and _newer(obj, [src] + headers):
            continue
        cmd = [hipcc] + FLAGS + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    LAST_COMPILE_COUNT = len(procs)
    if force or procs or not _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
