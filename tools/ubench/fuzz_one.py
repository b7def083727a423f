#!/usr/bin/env python3
"""development: one case of tools/fuzz_parity.py (seed, index) with the solver's debug switches taken from the environment.
   python tools/ubench/fuzz_one.py SEED INDEX"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch  # noqa: F401
import tlsq_amd
import fuzz_parity as F
from oracle import rpca_oracle as O
seed, idx = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for _ in range(idx + 1):
    D, kw, desc = F.make_case(rng)
eng = tlsq_amd.Engine(0)
A, E, s, sv, rep = eng.rpca(D, return_report=True, **kw)
Ao, Eo, so, svo, io = O.rpca(D, **kw)
k = next((i for i, (a, b) in enumerate(zip(rep.svp_hist, io.svp_hist)) if a != b), None)
print(f"errA={np.linalg.norm(A - Ao) / np.linalg.norm(D):.2e} errE={np.linalg.norm(E - Eo) / np.linalg.norm(D):.2e}")
print(desc, kw, "iters", rep.iters_done, io.iters_done, "sv", sv, svo, "first diff", k,
      "tsqr iters", rep.tsqr_iterations, "full", rep.eig_full, "fast", rep.eig_fast)
if k is not None:
    print(" gpu ", rep.svp_hist[max(0, k - 3):k + 4])
    print(" cpu ", io.svp_hist[max(0, k - 3):k + 4])
