#!/usr/bin/env python3
"""rpca on low rank + sparse + dense Gaussian noise (the case where the tail of Z's spectrum is a bulk right below 1/mu):
which solver served the SVD steps, and what the loop cost.
    python tools/noisy_case.py [M N r] [--noise 1e-6 1e-4 1e-2]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import tlsq_amd
ap = argparse.ArgumentParser()
ap.add_argument("shape", nargs="*", type=int, default=[20000, 512, 16])
ap.add_argument("--noise", nargs="*", type=float, default=[0.0, 1e-6, 1e-4, 1e-2])
ap.add_argument("--check", action="store_true", help="compare iterations / sv with the CPU oracle (slow)")
a = ap.parse_args()
M, N, r = a.shape
tlsq_amd.dev_from_env()   # TLSQ_DEBUG=1 etc. from the shell
eng = tlsq_amd.Engine(0)
for noise in a.noise:
    rng = np.random.default_rng(1)
    D = (rng.standard_normal((M, r)) @ rng.standard_normal((r, N)) + 10 * rng.standard_normal((M, N)) * (rng.random((M, N)) < 0.05)
         + noise * rng.standard_normal((M, N)))
    dD = torch.from_numpy(np.ascontiguousarray(D.T)).cuda()
    dA, dE = torch.empty_like(dD), torch.empty_like(dD)
    for rep_i in range(2):
        t0 = time.perf_counter()
        sv, rep, st = eng.rpca_device(dD.data_ptr(), M, N, dA.data_ptr(), dE.data_ptr(), want_hist=False)
        dt = time.perf_counter() - t0
    line = (f"noise={noise:g}: iters={rep.iters_done} conv={rep.converged} sv={sv} subspace={rep.eig_fast} dense/tsqr={rep.eig_full} "
            f"steps={rep.subspace_steps} loop={rep.ms['loop']:.1f} ms wall={dt*1e3:.1f} ms")
    if a.check:
        from oracle import rpca_oracle as O
        Ao, Eo, so, svo, io = O.rpca(D)
        A = dA.cpu().numpy().T
        line += f" | oracle iters={io.iters_done} sv={svo} errA={np.linalg.norm(A-Ao)/np.linalg.norm(Ao):.1e}"
    print(line, flush=True)
