#!/usr/bin/env python3
"""Frozen CPU-oracle outputs for the BASELINE configs whose oracle run is too slow for the GPU suite
(tests/test_gpu_configs.py holds the HIP path to them):

  c5_small : rpca on a 6000 x 2304 fp32 matrix (rank 20 + 5 % sparse, reference defaults) — the dtype / size class of
             BASELINE config 5 (fp32, min(M,N) > 2048 -> the library's large mode); oracle = fp32 LAPACK gesdd twice
             per iteration (src/robustPCA.jl:194,225 under /root/reference), ~2 min on 8 cores.

Stored: iteration count, sv, svp / cost history, and a strided sample of A and E (every 41st row, every 29th column).
Inputs are regenerated from the seed by the test (oracle.rpca_oracle.synth_lowrank_sparse).

    python tests/golden/make_config_vectors.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import rpca_oracle as O

out = {}
t0 = time.time()
M, N, r, seed = 6000, 2304, 20, 5
D, A0, _ = O.synth_lowrank_sparse(M, N, r, seed=seed, dtype=np.float32)
A, E, s, sv, info = O.rpca(D)
out["c5_small"] = {
    "M": M, "N": N, "rank": r, "seed": seed, "dtype": "float32",
    "iters_done": info.iters_done, "converged": bool(info.converged), "sv": int(sv),
    "svp_hist": [int(v) for v in info.svp_hist], "cost_hist": [float(v) for v in info.cost_hist],
    "row_stride": 41, "col_stride": 29,
    "A_sample": np.asarray(A[::41, ::29], dtype=np.float64).round(7).tolist(),
    "E_sample": np.asarray(E[::41, ::29], dtype=np.float64).round(7).tolist(),
    "normA": float(np.linalg.norm(A.astype(np.float64))), "normE": float(np.linalg.norm(E.astype(np.float64))),
    "S_head": [float(v) for v in s[1][:32]],
    "oracle_seconds": round(time.time() - t0, 1),
}
print("c5_small", info.iters_done, sv, info.svp_hist, f"{time.time() - t0:.0f}s")
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_vectors.json"), "w") as f:
    json.dump(out, f)
print("written", os.path.getsize(os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_vectors.json")), "bytes")
