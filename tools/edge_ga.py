"""Degenerate rpca_ga inputs: must neither hang nor crash, and must agree with the oracle where that is defined."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.zeros(1, device="cuda")
import tlsq_amd  # noqa: E402
from oracle import ga_oracle as G  # noqa: E402

warnings.simplefilter("ignore")
eng = tlsq_amd.Engine(0)
rng = np.random.default_rng(0)


def both(X, r, q0, **kw):
    info = G.GaInfo()
    with np.errstate(all="ignore"):
        want = G.rpca_ga(X, r, q0=q0, info=info, iters=kw.get("iters", 50))
    got, rep = eng.rpca_ga(X, r, q0=q0, return_report=True, iters=kw.get("iters", 50))
    return got, want, rep, info


cases = {
    "1x1": (np.array([[2.0]]), 1),
    "1xN": (rng.standard_normal((1, 50)), 1),
    "dx1": (rng.standard_normal((7, 1)), 1),
    "r>d": (rng.standard_normal((3, 40)), 5),
    "zero column": (np.hstack([rng.standard_normal((5, 20)), np.zeros((5, 1))]), 2),
    "all zeros": (np.zeros((4, 10)), 1),
    "nan entry": (np.where(np.arange(60).reshape(6, 10) == 7, np.nan, rng.standard_normal((6, 10))), 1),
    "big grid r>d": (rng.standard_normal((3, 20000)), 4),
    "grid zero column": (np.hstack([rng.standard_normal((70, 400)), np.zeros((70, 1))]), 2),
}
for name, (X, r) in cases.items():
    d = X.shape[0]
    q0 = rng.standard_normal((d, r))
    got, want, rep, info = both(X, r, q0)
    same_nan = np.array_equal(np.isnan(got), np.isnan(want))
    fin = ~np.isnan(want)
    err = float(np.abs(got[fin] - want[fin]).max()) if fin.any() else 0.0
    print(f"{name:18s} iters gpu {rep['iters']} oracle {info.iters}  nan-pattern same: {same_nan}  max err {err:.2e}")
eng.close()
print("edge cases done")
