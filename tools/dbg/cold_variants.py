#!/usr/bin/env python3
"""C2 device-resident solve under variants of the cold start (TLSQ_COLD_Q, TLSQ_WARM_Q0): subspace steps and wall time.
    python tools/dbg/cold_variants.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import tlsq_amd
from tlsq_amd import workloads as W

M, N, r = 20000, 512, 16
D, _, _ = W.synth_lowrank_sparse(M, N, r, seed=0)
eng = tlsq_amd.Engine(0)
dD = torch.from_numpy(np.ascontiguousarray(D.T)).cuda()
dA, dE = torch.empty_like(dD), torch.empty_like(dD)
variants = [dict(), dict(WARM_Q0="3"), dict(COLD_Q="6"), dict(COLD_Q="7"), dict(COLD_Q="6", WARM_Q0="4"), dict(WARM_Q0="4"), dict(WARM_Q0="6")]
if len(sys.argv) > 1:
    variants = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[1:]]
rounds = 6
best = {i: [] for i in range(len(variants))}
info = {}
for rd in range(rounds):          # interleaved: box drift (clocks, neighbours) hits every variant alike
    for i, v in enumerate(variants):
        with tlsq_amd.dev_switches(**v):
            ts = []
            for k in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sv, rep, st = eng.rpca_device(dD.data_ptr(), M, N, dA.data_ptr(), dE.data_ptr(), want_hist=False)
                ts.append((time.perf_counter() - t0) * 1e3)
            best[i].append(min(ts[1:]))
            info[i] = (rep.iters_done, sv, rep.subspace_steps)
for i, v in enumerate(variants):
    b = sorted(best[i])
    print(f"{str(v):40s} iters={info[i][0]} sv={info[i][1]} steps={info[i][2]} wall min {b[0]:.3f} med {b[len(b)//2]:.3f} ms", flush=True)
