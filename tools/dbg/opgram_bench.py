#!/usr/bin/env python3
"""Y = Z'(Z X) on an fp32 panel (tlsq_k_op_gram_f32): time per call and error against torch float64, second form
(opgram32.hip) against the first (OPGRAM_OLD=1).   python tools/dbg/opgram_bench.py [M N p]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import tlsq_amd
tlsq_amd.dev_from_env()

M, N, p = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (65536, 4096, 74)
g = torch.Generator(device="cuda").manual_seed(1)
Zt = (torch.randn(8, M, device="cuda", generator=g).T @ torch.randn(8, N, device="cuda", generator=g)).T.contiguous() \
     + 0.3 * torch.randn(N, M, device="cuda", generator=g)          # (N x M row-major = M x N column-major)
Zt = Zt.float().contiguous()
Xt = torch.randn(p, N, device="cuda", generator=g, dtype=torch.float64)
ref = (Zt.double() @ (Zt.double().T @ Xt.T)).T.contiguous()          # (p x N) = column-major N x p
scale = torch.linalg.matrix_norm(Zt.double(), 2).item() ** 2 * torch.linalg.vector_norm(Xt, dim=1)
torch.cuda.synchronize()
eng = tlsq_amd.Engine(0)
for tag, sw in (("second form", {}), ("first form", {"OPGRAM_OLD": 1})):
    with tlsq_amd.dev_switches(**sw):
        Y = torch.full((p, N), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        call = lambda: eng.lib.tlsq_k_op_gram_f32(eng.h, Zt.data_ptr(), M, N, M, Xt.data_ptr(), p, Y.data_ptr())
        assert call() == 0, eng.lib.tlsq_last_error(eng.h)
        eng.synchronize()
        err = (torch.linalg.vector_norm(Y - ref, dim=1) / scale).max().item()
        for _ in range(3):
            call()
        eng.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            call()
        eng.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{tag}: {M}x{N} p={p}: {dt * 1e3:.3f} ms per product pair ({4.0 * M * N * p / dt / 1e12:.1f} TFLOP/s), max column error {err:.2e} of ||Z||^2 ||x||")
eng.close()
