#!/usr/bin/env python3
"""development: cases [lo, hi) of tools/fuzz_parity.py (seed) solved one after the other on ONE engine (what the fuzz run does),
printing every mismatch.    python tools/ubench/fuzz_range.py SEED LO HI"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch  # noqa: F401
import tlsq_amd
import fuzz_parity as F
from oracle import rpca_oracle as O
warnings.simplefilter("ignore")
seed, lo, hi = (int(v) for v in sys.argv[1:4])
rng = np.random.default_rng(seed)
eng = tlsq_amd.Engine(0)
for idx in range(hi):
    D, kw, desc = F.make_case(rng)
    if idx < lo:
        continue
    print(f"--- case {idx}: {desc} {kw}", file=sys.stderr, flush=True)
    A, E, s, sv, rep = eng.rpca(D, return_report=True, **kw)
    Ao, Eo, so, svo, io = O.rpca(D, **kw)
    k = next((i for i, (a, b) in enumerate(zip(rep.svp_hist, io.svp_hist)) if a != b), None)
    errA = np.linalg.norm(A - Ao) / max(np.linalg.norm(D), 1e-300)
    print(idx, desc, kw, "iters", rep.iters_done, io.iters_done, "sv", sv, svo, "first diff", k, f"errA={errA:.1e}",
          "tsqr", rep.tsqr_iterations, "fast", rep.eig_fast, flush=True)
