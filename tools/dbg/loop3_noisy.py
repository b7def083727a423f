import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, tlsq_amd
tlsq_amd.dev_from_env()
def noisy(seed, M, N, r, noise):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((M, r)) @ rng.standard_normal((r, N))
            + 10 * rng.standard_normal((M, N)) * (rng.random((M, N)) < 0.05) + noise * rng.standard_normal((M, N)))
D = noisy(5, 2401, 160, 6, 1e-3)
plain = tlsq_amd.Engine(0)
A, E, s, sv, rep = plain.rpca(D, return_report=True)
print("plain", rep.iters_done, sv, rep.svp_hist, rep.tsqr_iterations, flush=True)
loop3 = tlsq_amd.Engine(devices=[0, 0, 0])
A3, E3, s3, sv3, rep3 = loop3.rpca(D, return_report=True)
print("loop3", rep3.iters_done, sv3, rep3.svp_hist, rep3.tsqr_iterations, np.abs(A-A3).max(), flush=True)
