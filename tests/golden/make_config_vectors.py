#!/usr/bin/env python3
"""Frozen CPU-oracle outputs for the BASELINE configs whose oracle run is too slow for the GPU suite
(tests/test_gpu_configs.py holds the HIP path to them):

  c5_small : rpca on a 6000 x 2304 fp32 matrix (rank 20 + 5 % sparse, reference defaults) — the dtype / size class of
             BASELINE config 5 (fp32, min(M,N) > 2048 -> the library's large mode); oracle = fp32 LAPACK gesdd twice
             per iteration (src/robustPCA.jl:194,225 under /root/reference), ~2 min on 8 cores.

  c5_h3    : rpca on a 32768 x 2304 fp32 matrix, rank 40 — a shape the round-5 fp16-split kernels take (M % 128 == 0,
             M >= 4096, N % 128 == 0: gram16.hip / opgram16.hip) and a rank above 32 on at least 2^26 elements (k_zsweep_wide,
             k_zx_h in the exact rebuild), so that those kernels are held to the fp32 LAPACK oracle inside a solve.

Stored: iteration count, sv, svp / cost history, and a strided sample of A and E (row_stride, col_stride).
Inputs are regenerated from the seed by the test (oracle.rpca_oracle.synth_lowrank_sparse).

    python tests/golden/make_config_vectors.py [c5_small] [c5_h3]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import rpca_oracle as O

CASES = {"c5_small": (6000, 2304, 20, 5, 41, 29), "c5_h3": (32768, 2304, 40, 6, 163, 29)}
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_vectors.json")
out = {}
if os.path.exists(OUT):
    with open(OUT) as f:
        out = json.load(f)
for name in ([a for a in sys.argv[1:] if a in CASES] or list(CASES)):
    t0 = time.time()
    M, N, r, seed, rs, cs = CASES[name]
    D, A0, _ = O.synth_lowrank_sparse(M, N, r, seed=seed, dtype=np.float32)
    A, E, s, sv, info = O.rpca(D)
    out[name] = {
        "M": M, "N": N, "rank": r, "seed": seed, "dtype": "float32",
        "iters_done": info.iters_done, "converged": bool(info.converged), "sv": int(sv),
        "svp_hist": [int(v) for v in info.svp_hist], "cost_hist": [float(v) for v in info.cost_hist],
        "row_stride": rs, "col_stride": cs,
        "A_sample": np.asarray(A[::rs, ::cs], dtype=np.float64).round(7).tolist(),
        "E_sample": np.asarray(E[::rs, ::cs], dtype=np.float64).round(7).tolist(),
        "normA": float(np.linalg.norm(A.astype(np.float64))), "normE": float(np.linalg.norm(E.astype(np.float64))),
        "S_head": [float(v) for v in s[1][:64]],
        "oracle_seconds": round(time.time() - t0, 1),
    }
    print(name, info.iters_done, sv, info.svp_hist, f"{time.time() - t0:.0f}s", flush=True)
    with open(OUT, "w") as f:
        json.dump(out, f)
print("written", os.path.getsize(OUT), "bytes")
