#!/usr/bin/env python3
"""development: rpca_ga with the robust averages (entrywise trimmed mean / median), loop time per case.
   python tools/ubench/ga_robust.py [--lib alternative libtlsqhip.so]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--cases", default="10x100000,64x200000,512x100000")
a = ap.parse_args()
import torch
torch.zeros(1, device="cuda")
from tlsq_amd import _lib as L
if a.lib:
    L.LIB_PATH = os.path.abspath(a.lib)
import tlsq_amd
eng = tlsq_amd.Engine(0)
for case in a.cases.split(","):
    d, N = (int(v) for v in case.split("x"))
    rng = np.random.default_rng(0)
    r = 2
    u = np.linalg.qr(rng.standard_normal((d, r)))[0]
    X = (u * np.array([30.0, 20.0])) @ rng.standard_normal((r, N)) + 0.01 * rng.standard_normal((d, N))
    X += 100 * rng.standard_normal((d, N)) * (rng.random((d, N)) < 0.001)
    q0 = rng.standard_normal((d, r))
    for mu in ("entrywise_trimmed_mean", "entrywise_median"):
        eng.rpca_ga(X, r, q0=q0, mu=mu, return_report=True)
        t0 = time.perf_counter()
        Q, rep = eng.rpca_ga(X, r, q0=q0, mu=mu, return_report=True)
        dt = time.perf_counter() - t0
        print(f"d={d:5d} N={N:8d} {mu:24s} iters={rep['iters']} loop {rep['ms_loop']:9.2f} ms  wall {dt*1e3:9.1f} ms  |Q|={np.abs(Q).sum():.12f}", flush=True)
