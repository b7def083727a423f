"""Consecutive C2 solves WITH the returned decomposition (what bench.py's value_with_s times), call by call.
    python tools/dbg/ws_consecutive.py [reps]            (TLSQ_* switches from the shell)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import tlsq_amd
tlsq_amd.dev_from_env()
from oracle import rpca_oracle as O
M, N, r = 20000, 512, 16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
D, _, _ = O.synth_lowrank_sparse(M, N, r, seed=0)
eng = tlsq_amd.Engine(0)
dD = torch.from_numpy(np.ascontiguousarray(D.T)).cuda()
dA, dE = torch.empty_like(dD), torch.empty_like(dD)
dU = torch.empty((N, M), dtype=torch.float64, device="cuda")
dS = torch.empty(N, dtype=torch.float64, device="cuda")
dVt = torch.empty((N, N), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for i in range(reps):
    t0 = time.perf_counter()
    sv, rep, st = eng.rpca_device(dD.data_ptr(), M, N, dA.data_ptr(), dE.data_ptr(), dU=dU.data_ptr(), dS=dS.data_ptr(),
                                  dVt=dVt.data_ptr(), want_hist=False)
    torch.cuda.synchronize()
    print(f"call {i}: {1e3 * (time.perf_counter() - t0):.2f} ms (library total {rep.ms['total']:.2f}, loop {rep.ms['loop']:.2f}); sweeps {rep.jacobi_sweeps}", flush=True)
eng.close()
