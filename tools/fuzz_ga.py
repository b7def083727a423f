"""Randomised rpca_ga parity sweep: shapes, ranks, noise, outlier rates and all three averages against
oracle/ga_oracle.py on equal start vectors.  Reports every case whose iteration counts or components differ.

    python tools/fuzz_ga.py [--cases 200] [--seed 0]
"""
import argparse
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(cases=200, seed=0, eng=None, verbose=True):
    import tlsq_amd
    from oracle import ga_oracle as G
    own = eng is None
    if own:
        eng = tlsq_amd.Engine(0)
    rng = np.random.default_rng(seed)
    bad = []
    for c in range(cases):
        d = int(rng.choice([2, 3, 5, 8, 10, 16, 17, 31, 40, 64, 65, 100, 129, 300, 600]))
        N = int(rng.integers(max(2, 4), 3000))
        r = int(rng.integers(1, min(d, N, 5) + 1))
        mode = str(rng.choice(["mean", "mean", "entrywise_trimmed_mean", "entrywise_median"]))
        if mode != "mean" and d * N > 150000:
            N = max(4, 150000 // d)
        if mode != "mean" and r == d:
            # the last component of a full basis lives in a one-dimensional residual: every U[j, :] is +-v_j up to
            # rounding, the order statistics are decided by that rounding noise (in the reference too) — not a parity case
            r = d - 1
            if r == 0:
                continue
        eps = float(10.0 ** rng.uniform(-8, 0))
        out = float(rng.choice([0.0, 0.01, 0.05]))
        u = np.linalg.qr(rng.standard_normal((d, r)))[0]
        X = (u * (10.0 * np.arange(r, 0, -1))) @ rng.standard_normal((r, N)) + eps * rng.standard_normal((d, N))
        if out:
            X = X + 100 * rng.standard_normal((d, N)) * (rng.random((d, N)) < out)
        q0 = rng.standard_normal((d, r))
        iters = 40 if mode != "mean" else 300
        info = G.GaInfo()
        mu = G.mu_mean if mode == "mean" else getattr(G, mode)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = G.rpca_ga(X, r, q0=q0, mu=mu, iters=iters, info=info)
            got, rep = eng.rpca_ga(X, r, q0=q0, mu=None if mode == "mean" else mode, iters=iters, return_report=True)
        err = float(np.abs(got - want).max())
        if rep["iters"] != info.iters or not (err < 1e-9):
            bad.append(dict(case=c, d=d, N=N, r=r, mode=mode, eps=eps, out=out, iters=(rep["iters"], info.iters), err=err))
            if verbose:
                print("MISMATCH", bad[-1], flush=True)
    if own:
        eng.close()
    if verbose:
        print(f"{cases} cases, {len(bad)} mismatches")
    return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    import torch
    torch.zeros(1, device="cuda")
    run(a.cases, a.seed)
