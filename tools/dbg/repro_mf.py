"""Reproduce the loop-back trajectory fork of round 2: the problems of tests/test_gpu_multi.py in module order on ONE
pair of handles per group size, then the noisy problem that takes the matrix-function route, several times.
    python tools/dbg/repro_mf.py [poison]"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

warnings.simplefilter("ignore")
import torch

torch.zeros(1, device="cuda")
import tlsq_amd
from oracle import rpca_oracle as O

if len(sys.argv) > 1 and sys.argv[1] == "poison":
    tlsq_amd.dev_set("WS_POISON", 1)


def relerr(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def noisy(seed, M, N, r, noise):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((M, r)) @ rng.standard_normal((r, N))
            + 10 * rng.standard_normal((M, N)) * (rng.random((M, N)) < 0.05) + noise * rng.standard_normal((M, N)))


Dmf = noisy(5, 2401, 160, 6, 1e-3)
Ao, Eo, so, svo, io = O.rpca(Dmf)
warm = [O.synth_lowrank_sparse(M, N, r, seed=M)[0] for M, N, r in [(1500, 96, 6), (1237, 50, 4), (3001, 130, 30), (20000, 128, 8)]]
rng = np.random.default_rng(11)
warm.append(rng.standard_normal((1200, 3)) @ rng.standard_normal((3, 48)) + 1e-3 * rng.standard_normal((1200, 48)))
warm.append(noisy(1, 2402, 160, 6, 3e-7))
bad = 0
for n in (2, 3, 8):
    plain = tlsq_amd.Engine(0)
    multi = tlsq_amd.Engine(devices=[0] * n)
    for D in warm:
        _, _, _, sv1, r1 = plain.rpca(D, return_report=True)
        _, _, _, sv2, r2 = multi.rpca(D, return_report=True)
        if r1.svp_hist != r2.svp_hist:
            bad += 1
            print("WARM-UP MISMATCH", n, D.shape, r1.svp_hist, r2.svp_hist)
    for rep in range(6):
        A1, E1, s1, sv1, rep1 = plain.rpca(Dmf, return_report=True)
        A2, E2, s2, sv2, rep2 = multi.rpca(Dmf, return_report=True)
        ok = rep1.svp_hist == io.svp_hist and rep2.svp_hist == io.svp_hist
        k = next((i for i, (a, b) in enumerate(zip(rep2.svp_hist, io.svp_hist)) if a != b), None)
        k1 = next((i for i, (a, b) in enumerate(zip(rep1.svp_hist, io.svp_hist)) if a != b), None)
        print(f"n={n} rep={rep} ok={ok} iters {rep1.iters_done}/{rep2.iters_done}/{io.iters_done} first diff plain={k1} multi={k} "
              f"errA plain {relerr(A1, Ao):.2e} multi {relerr(A2, Ao):.2e} multi-vs-plain {relerr(A2, A1):.2e} "
              f"tsqr {rep1.tsqr_iterations}/{rep2.tsqr_iterations}", flush=True)
        if not ok:
            bad += 1
            print("   oracle", io.svp_hist, "\n   plain ", rep1.svp_hist, "\n   multi ", rep2.svp_hist)
    multi.close()
    plain.close()
print("MISMATCHES", bad)
