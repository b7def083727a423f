#!/usr/bin/env python3
"""Time of the accurate SVD route (TSQR + one-sided Jacobi on R', tlsq_k_svd_r_f64) on a C2-like panel, and its accuracy vs LAPACK.
    python tools/bench_svd_r.py [M N]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import tlsq_amd
tlsq_amd.dev_from_env()
M, N = (int(v) for v in sys.argv[1:3]) if len(sys.argv) >= 3 else (20000, 512)
rng = np.random.default_rng(0)
Z = rng.standard_normal((M, 16)) @ rng.standard_normal((16, N)) + 10 * rng.standard_normal((M, N)) * (rng.random((M, N)) < 0.05)
eng = tlsq_amd.Engine(0)
dZ = torch.from_numpy(np.ascontiguousarray(Z.T)).cuda()
dS = torch.empty(N, dtype=torch.float64, device="cuda")
dV = torch.empty((N, N), dtype=torch.float64, device="cuda")
sw = C.c_int64(0)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = eng.lib.tlsq_k_svd_r_f64(eng.h, dZ.data_ptr(), M, N, M, dS.data_ptr(), dV.data_ptr(), N, C.byref(sw))
    eng.synchronize()
    dt = time.perf_counter() - t0
    assert st == 0, eng.lib.tlsq_last_error(eng.h)
    print(f"svd_r {M}x{N}: {dt*1e3:.2f} ms, {sw.value} sweeps", flush=True)
S = dS.cpu().numpy()
V = dV.cpu().numpy().T
ref = np.linalg.svd(Z, compute_uv=False)
print("max |S - S_lapack| / S_max =", np.abs(np.sort(S)[::-1] - ref).max() / ref[0], " max rel =", (np.abs(np.sort(S)[::-1] - ref) / ref).max(),
      " ||V'V - I|| =", np.abs(V.T @ V - np.eye(N)).max())
