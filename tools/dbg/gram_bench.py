#!/usr/bin/env python3
"""G = Z'Z of an fp32 panel (tlsq_k_gram_f32, mfma path): time per call and entry-wise error against float64, the fp16-split
kernel (gram16.hip) against the fp32-MFMA kernel (GRAM_H3=0).   python tools/dbg/gram_bench.py [M N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import tlsq_amd
tlsq_amd.dev_from_env()

noeig = '--noeig' in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith('--')]
M, N = (int(v) for v in args[:2]) if len(args) >= 2 else (65536, 4096)
g = torch.Generator(device="cuda").manual_seed(1)
Zt = (torch.randn(N, 64, device="cuda", generator=g) @ torch.randn(64, M, device="cuda", generator=g)
      + 0.3 * torch.randn(N, M, device="cuda", generator=g)).float().contiguous()
Zd = Zt.double()
ref = Zd @ Zd.T
d = torch.sqrt(torch.diagonal(ref))
scale = d[:, None] * d[None, :]
del Zd
eng = tlsq_amd.Engine(0)
for tag, sw in (("fp16 split", {}), ("fp32 MFMA", {"GRAM_H3": 0})):
    with tlsq_amd.dev_switches(**sw):
        G = torch.empty((N, N), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        call = lambda: eng.lib.tlsq_k_gram_f32(eng.h, Zt.data_ptr(), M, N, M, G.data_ptr(), N, 1)
        assert call() == 0, eng.lib.tlsq_last_error(eng.h)
        eng.synchronize()
        err = float(((G - ref).abs() / scale).max())
        lam, lam_ref = (1.0, 1.0) if noeig else (float(torch.linalg.eigvalsh(G)[-1]), float(torch.linalg.eigvalsh(ref)[-1]))
        for _ in range(2):
            call()
        eng.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            call()
        eng.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{tag}: {M}x{N}: {dt * 1e3:.3f} ms per Gram matrix ({M * N * (N + 128) / dt / 1e12:.1f} TFLOP/s lower triangle), "
              f"max entry error {err:.2e} of sqrt(G_ii G_jj), lambda_max rel err {abs(lam - lam_ref) / lam_ref:.1e}")
eng.close()
