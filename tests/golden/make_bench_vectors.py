"""Writes tests/golden/bench_vectors.json - the CPU oracle's run of the two FULL-SIZE bench problems, frozen as data:
BASELINE config 2 (rpca 20000x512 fp64, the headline) and config 4 (200000x512).  Per problem: iterations, `sv`, the rank
trajectory `svp_hist` (src/robustPCA.jl:198), the cost history (:225), the singular values of the last Z (the returned
`s.S`, :194/:238), ||A||_F^2 and ||E||_F^2, and strided samples of A and E.  bench.py's validation and
tests/test_gpu_configs.py hold the HIP path to THESE values (round 3 compared the bench with a HIP-generated file).

The oracle (oracle/rpca_oracle.py: LAPACK gesdd for both decompositions of an iteration, expression order of the
reference) is itself pinned to the reference's known-answer vectors by tests/test_oracle_golden.py.  Build container,
8 cores: config 2 takes about a minute, config 4 about a quarter of an hour and 7 GB.

    python tests/golden/make_bench_vectors.py [c2] [c4] [c5]
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import rpca_oracle as O  # noqa: E402
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("tlsq_workloads", os.path.join(ROOT, "totalleastsquares.jl_amd", "workloads.py"))
W = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(W)

OUT = os.path.join(HERE, "bench_vectors.json")
STRIDES = {"c2": 4999, "c4": 49999, "c5": 131071}      # primes: the samples walk through all rows and columns (~2048 values each)


def svp_hash(hist):
    return hashlib.sha256(",".join(str(int(v)) for v in hist).encode()).hexdigest()[:16]


def problem(name):
    if name == "c2":
        D, _, _ = W.synth_lowrank_sparse(20000, 512, 16, seed=0)
        return D
    if name == "c5":
        # BASELINE config 5 (65536 x 4096 fp32, rank 64 + 5 % sparse) in the reference's default (exact) mode: fp32 LAPACK
        # gesdd twice per iteration, about 100 TFLOP - tens of minutes and ~12 GB on the 8-core build box, once.
        D, _, _ = W.synth_lowrank_sparse(65536, 4096, 64, seed=0, dtype=np.float32)
        return D
    M4 = W.C4_SHAPE[0]
    return np.asfortranarray(W.c4_rows(0, M4))


def main():
    names = [a for a in sys.argv[1:] if a in STRIDES] or ["c2", "c4", "c5"]
    out = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            out = json.load(f)
    for name in names:
        D = problem(name)
        t0 = time.time()
        A, E, s, sv, info = O.rpca(D)
        dt = time.time() - t0
        st = STRIDES[name]
        out[name] = {
            "M": int(D.shape[0]), "N": int(D.shape[1]),
            "dtype": str(D.dtype), "D_sha256_of_float64_column_major": hashlib.sha256(D.tobytes(order="F")).hexdigest(),
            "iters": info.iters_done, "sv": int(sv), "converged": bool(info.converged),
            "svp_hist": [int(v) for v in info.svp_hist], "svp_hash": svp_hash(info.svp_hist),
            "cost_hist": [float(v) for v in info.cost_hist],
            "S": [float(v) for v in s[1]],
            "normA2": float(np.sum(A.astype(np.float64) ** 2)), "normE2": float(np.sum(E.astype(np.float64) ** 2)),
            "nnzE": int(np.count_nonzero(E)),
            "sample_stride": st,
            "A_sample": [float(v) for v in A.ravel(order="F")[::st]],
            "E_sample": [float(v) for v in E.ravel(order="F")[::st]],
            "oracle_seconds": round(dt, 1),
        }
        print(name, info.iters_done, sv, info.converged, f"{dt:.0f} s", flush=True)
        with open(OUT, "w") as f:
            json.dump(out, f)
            f.write("\n")


if __name__ == "__main__":
    main()
