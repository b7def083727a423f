import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
import tlsq_amd
from oracle import rpca_oracle as O
eng = tlsq_amd.Engine(0)
for (M, N, r) in ((500, 50, 5), (2000, 64, 6), (1000, 40, 4), (3000, 100, 8)):
    D = O.synth_lowrank_sparse(M, N, r, seed=0)[0]
    dD = torch.from_numpy(np.ascontiguousarray(D.T)).cuda(); dA = torch.empty_like(dD); dE = torch.empty_like(dD)
    eng.rpca_device(dD.data_ptr(), M, N, dA.data_ptr(), dE.data_ptr(), want_hist=False)
    t0 = time.perf_counter()
    for _ in range(5):
        sv, rep, st = eng.rpca_device(dD.data_ptr(), M, N, dA.data_ptr(), dE.data_ptr(), want_hist=False)
    dt = (time.perf_counter() - t0) / 5
    print(f"{M}x{N}: iters={rep.iters_done} {dt*1e3:.2f} ms/solve {dt/rep.iters_done*1e6:.0f} us/iter full={rep.eig_full} fast={rep.eig_fast}")
# the same problems through host pointers with the returned s (what `rpca(D)` of the drop-in pays) and on the CPU oracle
import os
os.environ.setdefault("OPENBLAS_NUM_THREADS", "8")
for (M, N, r) in ((500, 50, 5), (2000, 64, 6), (1000, 40, 4), (3000, 100, 8), (5000, 200, 10)):
    D = O.synth_lowrank_sparse(M, N, r, seed=0)[0]
    eng.rpca(D)
    t0 = time.perf_counter()
    for _ in range(5):
        A, E, s, sv, rep = eng.rpca(D, return_report=True, cost_history=False)
    dt = (time.perf_counter() - t0) / 5
    t1 = time.perf_counter()
    O.rpca(D)
    tc = time.perf_counter() - t1
    print(f"{M}x{N}: drop-in call {dt*1e3:.2f} ms ({rep.iters_done} iterations), CPU oracle {tc*1e3:.1f} ms")
