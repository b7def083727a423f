import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, tlsq_amd
tlsq_amd.dev_from_env()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
M, N, r = 30, 30, 1
D = rng.standard_normal((M, r)) @ rng.standard_normal((r, N)) + 10 * rng.standard_normal((M, N)) * (rng.random((M, N)) < 0.05)
e = tlsq_amd.Engine(0)
A, E, s, sv = e.rpca(D)
U = np.asarray(s.U); print("S tail", s.S[-6:] / s.S[0]); print("UtU err", np.linalg.norm(U.T @ U - np.eye(N)), "VtV", np.linalg.norm(np.asarray(s.Vt) @ np.asarray(s.Vt).T - np.eye(N)))
print("col norms tail", np.linalg.norm(U, axis=0)[-6:])
