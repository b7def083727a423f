import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, tlsq_amd
from tlsq_amd import workloads as W
tlsq_amd.dev_from_env()
M, N = 20000, 512
D = W.synth_lowrank_sparse(M, N, 16, seed=0)[0]
eng = tlsq_amd.Engine(0)
dD = torch.from_numpy(np.ascontiguousarray(D.T)).cuda(); dA = torch.empty_like(dD); dE = torch.empty_like(dD)
dU = torch.empty((N, M), dtype=torch.float64, device="cuda"); dS = torch.empty(N, dtype=torch.float64, device="cuda"); dVt = torch.empty((N, N), dtype=torch.float64, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    sv, r, st = eng.rpca_device(dD.data_ptr(), M, N, dA.data_ptr(), dE.data_ptr(), want_hist=False, dU=dU.data_ptr(), dS=dS.data_ptr(), dVt=dVt.data_ptr())
    torch.cuda.synchronize(); print(f"with s: {(time.perf_counter()-t)*1e3:.2f} ms (loop {r.ms['loop']:.2f}, total {r.ms['total']:.2f}), jacobi sweeps {r.jacobi_sweeps}", flush=True)
