#!/usr/bin/env python3
"""Run-to-run determinism of the large fp32 paths (BASELINE config 5 at a quarter of its rows): two solves per mode on one
handle and on a fresh handle must return the same bits.   python tools/dbg/det_c5.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import tlsq_amd
from tlsq_amd import _lib as L

M, N, r = 16384, 4096, 64
g = torch.Generator(device="cuda").manual_seed(3)
A0 = torch.randn(N, r, device="cuda", generator=g) @ torch.randn(r, M, device="cuda", generator=g)
D = (A0 + 10.0 * torch.randn(N, M, device="cuda", generator=g) * (torch.rand(N, M, device="cuda", generator=g) < 0.05)).contiguous()
torch.cuda.synchronize()
for tag, kw in (("randomized", dict(svd_mode=L.SVD_RANDOMIZED)), ("exact", {})):
    outs = []
    for fresh in range(2):
        eng = tlsq_amd.Engine(0)
        for rep in range(2):
            A, E = torch.empty_like(D), torch.empty_like(D)
            sv, info, st = eng.rpca_device(D.data_ptr(), M, N, A.data_ptr(), E.data_ptr(), want_hist=False, dtype=np.float32, **kw)
            torch.cuda.synchronize()
            outs.append((A.clone(), E.clone(), info.iters_done, int(sv)))
        eng.close()
    same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) and outs[0][2:] == o[2:] for o in outs[1:])
    print(f"{tag}: iters={outs[0][2]} sv={outs[0][3]} four runs bit-identical: {same}")
    assert same
