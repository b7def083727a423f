import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, warnings
import tlsq_amd
from oracle import rpca_oracle as O
warnings.simplefilter("ignore")
eng = tlsq_amd.Engine(0)
D, _, _ = O.synth_lowrank_sparse(200, 30, 3, seed=3)
re = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
for it in (1, 2, 3, 4, 5, 6, 7, 1000):
    A, E, s, sv, rep = eng.rpca(D, iters=it, return_report=True)
    Ao, Eo, so, svo, io = O.rpca(D, iters=it)
    print(it, rep.iters_done, io.iters_done, "A", re(A, Ao), "E", re(E, Eo), "S", re(s[1], so[1]), "nnzE", (E != 0).sum(), (Eo != 0).sum(), flush=True)
