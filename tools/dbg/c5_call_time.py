"""Whole-call time of the C5 solve (65536 x 4096 fp32 rank 64, device-resident) on the two panels bench.py has used: the
torch-generated one (rounds 4-5) and the numpy panel of the oracle fixture (round 6).   python tools/dbg/c5_call_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import tlsq_amd
from tlsq_amd import _lib as L
tlsq_amd.dev_from_env()
import importlib.util
spec = importlib.util.spec_from_file_location("w", os.path.join(os.path.dirname(tlsq_amd.__file__) if hasattr(tlsq_amd, "__file__") else ".", "workloads.py"))
from oracle import rpca_oracle as O
M5, N5, r5 = 65536, 4096, 64
eng = tlsq_amd.Engine(0)
for panel in ("torch", "numpy"):
    if panel == "torch":
        g5 = torch.Generator(device="cuda").manual_seed(5)
        A05 = (torch.randn(N5, r5, device="cuda", generator=g5) @ torch.randn(r5, M5, device="cuda", generator=g5))
        d5 = A05 + 10.0 * torch.randn(N5, M5, device="cuda", generator=g5) * (torch.rand(N5, M5, device="cuda", generator=g5) < 0.05)
    else:
        D, A0, _ = O.synth_lowrank_sparse(M5, N5, r5, seed=0, dtype=np.float32)
        d5 = torch.from_numpy(np.ascontiguousarray(D.T)).cuda()
        del D, A0
    a5, e5 = torch.empty_like(d5), torch.empty_like(d5)
    for tag, kw in (("randomized", dict(svd_mode=L.SVD_RANDOMIZED)), ("exact", {})):
        run = lambda: eng.rpca_device(d5.data_ptr(), M5, N5, a5.data_ptr(), e5.data_ptr(), want_hist=False, dtype=np.float32, **kw)
        run()
        torch.cuda.synchronize()
        t = time.perf_counter()
        per = []
        for _ in range(3):
            t1 = time.perf_counter()
            sv, rep, st = run()
            per.append((time.perf_counter() - t1) * 1e3)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        tsync = (time.perf_counter() - t2) * 1e3
        t = (time.perf_counter() - t) / 3
        print("   per call", [round(x, 1) for x in per], "final device sync", round(tsync, 2), "ms")
        print(f"{panel:6s} {tag:10s} {t*1e3:7.1f} ms per solve, {rep.iters_done} iterations, loop {rep.ms['loop']:.1f} ms, total {rep.ms['total']:.1f}, h2d {rep.ms['h2d']:.2f}, d2h {rep.ms['d2h']:.2f}, kernels {rep.kern}", flush=True)
    del d5, a5, e5
    torch.cuda.empty_cache()
eng.close()
