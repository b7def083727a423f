#!/usr/bin/env python3
"""Repeated solves of different shapes on one handle: device memory must stay flat once the workspace has grown."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tlsq_amd
from oracle import rpca_oracle as O
torch.zeros(1, device="cuda")
eng = tlsq_amd.Engine(0)
shapes = [(2000, 128, 8), (500, 50, 5), (3000, 96, 6), (1000, 300, 10), (64, 200, 3)]
Ds = [O.synth_lowrank_sparse(M, N, r, seed=i)[0] for i, (M, N, r) in enumerate(shapes)]
free0 = None
for rep in range(40):
    for D in Ds:
        eng.rpca(D, iters=40)
    eng.rtls_batched(np.random.default_rng(rep).standard_normal((64, 50, 3)), np.random.default_rng(rep + 1).standard_normal((64, 50)))
    eng.lowrankfilter(np.sin(0.1 * np.arange(2000)) + 0.1 * np.random.default_rng(rep).standard_normal(2000), 40)
    free, total = torch.cuda.mem_get_info()
    if rep == 4:
        free0 = free
    if rep in (4, 20, 39):
        print(f"rep {rep}: free {free / 2**20:.0f} MiB")
assert free0 is not None and abs(free - free0) < 64 * 2**20, (free0, free)
print("soak ok")
