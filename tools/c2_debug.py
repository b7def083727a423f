#!/usr/bin/env python3
"""One C2 solve (20000 x 512 fp64, device-resident, no cost history) with TLSQ_DEBUG=1: per-iteration solver trace on stderr.
    TLSQ_DEBUG=1 python tools/c2_debug.py [M N r]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import tlsq_amd
tlsq_amd.dev_from_env()   # TLSQ_DEBUG=1 etc. from the shell (the library itself reads no environment)
from oracle import rpca_oracle as O
M, N, r = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (20000, 512, 16)
D, A0, _ = O.synth_lowrank_sparse(M, N, r, seed=0)
eng = tlsq_amd.Engine(0)
dD = torch.from_numpy(np.ascontiguousarray(D.T)).cuda()
dA, dE = torch.empty_like(dD), torch.empty_like(dD)
torch.cuda.synchronize()
for rep_i in range(4):
    t0 = time.perf_counter()
    sv, rep, st = eng.rpca_device(dD.data_ptr(), M, N, dA.data_ptr(), dE.data_ptr(), want_hist=False,
                                  phase_timing=(rep_i == 3))
    dt = time.perf_counter() - t0
    print(f"run {rep_i}: iters={rep.iters_done} sv={sv} full={rep.eig_full} fast={rep.eig_fast} steps={rep.subspace_steps} "
          f"wall={dt*1e3:.1f} ms loop={rep.ms['loop']:.1f} ms", {k: round(v / rep.iters_done, 4) for k, v in rep.ms.items()}, flush=True)
