import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import tlsq_amd
import test_gpu_rr_small as T
eng = tlsq_amd.Engine(0)
rng = np.random.default_rng(1)
for p in (3, 8, 20):
    G, Y = T.warm_block(rng, 200, p, 1e-5, scale_cols=False)
    B, Hg = Y.T @ Y, Y.T @ (G @ Y)
    C, lam, st = T.run_rr(eng, torch, B, Hg)
    I = C.T @ B @ C
    Dg = C.T @ Hg @ C
    print("p", p, "CtBC-I", np.abs(I - np.eye(p)).max(), "offdiag CtHC", np.abs(Dg - np.diag(np.diag(Dg))).max(), "lam err", np.abs(np.diag(Dg) - lam).max())
    if p == 3:
        np.set_printoptions(linewidth=200, precision=4)
        print(I); print(Dg); print(C)
