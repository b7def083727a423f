#!/usr/bin/env python3
"""A case the subspace path cannot serve: dense noise keeps many singular values near 1/mu, so (almost) every
ALM iteration needs the full Jacobi decomposition.  Checks parity with the oracle and reports timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tlsq_amd
from oracle import rpca_oracle as O

M, N, r = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4000, 256, 8)
noise = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-2
D, A0, _ = O.synth_lowrank_sparse(M, N, r, seed=3)
D = D + noise * np.random.default_rng(5).standard_normal(D.shape)
iters = 40
eng = tlsq_amd.Engine(0)
eng.rpca(D[:200], iters=2)   # warm up
t0 = time.perf_counter()
A, E, s, sv, rep = eng.rpca(D, iters=iters, return_report=True, want_U=False)
dt = time.perf_counter() - t0
print(f"{M}x{N} noise={noise}: iters={rep.iters_done} sv={sv} svp_hist tail={rep.svp_hist[-5:]} full={rep.eig_full} fast={rep.eig_fast} "
      f"sweeps={rep.jacobi_sweeps} loop={rep.ms['loop']:.1f} ms eig/iter={rep.ms['eig']/rep.iters_done:.2f} ms")
t0 = time.perf_counter()
Ao, Eo, so, svo, io = O.rpca(D, iters=iters)
print(f"oracle: iters={io.iters_done} sv={svo} {time.perf_counter()-t0:.1f}s  svp same={rep.svp_hist == io.svp_hist} "
      f"relA={np.linalg.norm(A-Ao)/np.linalg.norm(Ao):.2e} relE={np.linalg.norm(E-Eo)/np.linalg.norm(Eo):.2e}")
if rep.svp_hist != io.svp_hist:
    print(" gpu:", rep.svp_hist); print(" cpu:", io.svp_hist)
