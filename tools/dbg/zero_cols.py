import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, tlsq_amd
tlsq_amd.dev_from_env()
eng = tlsq_amd.Engine(0)
for dt_, M, N in ((np.float64, 3000, 96), (np.float64, 64, 64), (np.float32, 3000, 96)):
    rng = np.random.default_rng(M + N)
    Dz = (rng.standard_normal((M, 5)) @ rng.standard_normal((5, N)) + 10 * rng.standard_normal((M, N)) * (rng.random((M, N)) < 0.05)).astype(dt_)
    Dz[:, ::3] = 0
    try:
        A, E, s, sv, rep = eng.rpca(Dz, return_report=True)
        U = np.asarray(s.U, dtype=np.float64)
        print(dt_.__name__, M, N, "ok iters", rep.iters_done, "sv", sv, "tsqr", rep.tsqr_iterations, "orth", np.abs(U.T @ U - np.eye(min(M, N))).max(), flush=True)
    except Exception as e:
        print(dt_.__name__, M, N, "FAILED", e, flush=True)
