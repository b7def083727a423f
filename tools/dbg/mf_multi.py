import sys; sys.path.insert(0,'/root/repo')
import numpy as np, warnings
warnings.simplefilter("ignore")
import torch; torch.zeros(1,device='cuda')
import tlsq_amd
from oracle import rpca_oracle as O
rng = np.random.default_rng(5)
M, N, r = 2401, 160, 6
D = (rng.standard_normal((M, r)) @ rng.standard_normal((r, N)) + 10 * rng.standard_normal((M, N)) * (rng.random((M, N)) < 0.05) + 1e-3 * rng.standard_normal((M, N)))
Ao, Eo, so, svo, io = O.rpca(D)
plain = tlsq_amd.Engine(0)
A1, E1, s1, sv1, rep1 = plain.rpca(D, return_report=True)
print("plain", rep1.iters_done, io.iters_done, sv1, svo, rep1.svp_hist == io.svp_hist, np.linalg.norm(A1-Ao)/np.linalg.norm(Ao), rep1.tsqr_iterations)
for n in (2, 3):
    multi = tlsq_amd.Engine(devices=[0]*n)
    for rep_i in range(2):
        A2, E2, s2, sv2, rep2 = multi.rpca(D, return_report=True)
        k = next((i for i, (a, b) in enumerate(zip(rep2.svp_hist, io.svp_hist)) if a != b), None)
        print("multi", n, rep2.iters_done, sv2, "first diff", k, rep2.svp_hist[max(0,(k or 0)-2):(k or 0)+3], io.svp_hist[max(0,(k or 0)-2):(k or 0)+3], np.linalg.norm(A2-Ao)/np.linalg.norm(Ao), rep2.tsqr_iterations)
