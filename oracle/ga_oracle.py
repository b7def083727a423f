"""CPU oracle for `rpca_ga` (Grassmann averages) and its spherical averages.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT — see oracle/rpca_oracle.py's header; the same rule applies.

Restates, in the reference's expression and summation order (paths relative to /root/reference):
    rpca_ga                 src/robustPCA.jl:255-281
    rpca_ga_1               src/robustPCA.jl:286-310
    μ!                      src/robustPCA.jl:312-320   (here: mu_mean)
    entrywise_trimmed_mean  src/robustPCA.jl:327-337
    entrywise_median        src/robustPCA.jl:354-362

Parity pin: the reference's own tests of this path hold no numeric tables, only identities and inequalities
(test/runtests.jl:443-520): Q'Q = I to sqrt(eps) for 2 x 200 random cases, μ! = (weighted) mean,
entrywise_trimmed_mean(P=0) = weighted mean, entrywise_trimmed_mean(P=0.1) = trimmed mean of each row, and the
"robust average beats the plain one" pass rates.  tests/test_oracle_golden.py checks this restatement against all of
them.  The start vector of rpca_ga_1 is `randn(d)` from Julia's global RNG (:289), which nothing can reproduce:
the oracle (and the library) take the start vectors as an argument so that the two can be compared on equal input.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np


def _sign(x):
    # Julia's sign(): +-1, +-0 stays, NaN stays
    return np.sign(x)


def mu_mean(s, w, U):
    """μ!(s,w,U), src/robustPCA.jl:312-320: sequential weighted sum over the columns, then `s ./= ws`."""
    ws = 0.0
    s[:] = 0.0
    for n in range(U.shape[1]):
        ws += w[n]
        s += w[n] * U[:, n]
    s /= ws
    return s


def _trim_range(N, P):
    # range = (1+floor(Int, P*N)):floor(Int, (1-P)*N)          (:329), 1-based inclusive -> 0-based slice
    lo = 1 + int(math.floor(P * N))
    hi = int(math.floor((1 - P) * N))
    return lo - 1, hi


def entrywise_trimmed_mean(s, w, U, P=0.1):
    """src/robustPCA.jl:327-337: per row, drop the P fraction of smallest and largest entries, weighted mean of the rest."""
    N = U.shape[1]
    lo, hi = _trim_range(N, P)
    s[:] = 0.0
    for j in range(U.shape[0]):
        I = np.argsort(U[j, :], kind="stable")[lo:hi]
        s[j] += np.dot(w[I], U[j, I]) / np.sum(w[I])
    return s


def entrywise_median(s, w, U):
    """src/robustPCA.jl:354-362: per row, the entry whose weighted value w.*U[j,:] is the (N÷2)-th smallest."""
    N = U.shape[1]
    if N // 2 < 1:
        raise IndexError("entrywise_median needs at least 2 columns (I[end÷2], :358)")
    s[:] = 0.0
    for j in range(U.shape[0]):
        I = np.argsort(w * U[j, :], kind="stable")
        m = I[N // 2 - 1]
        s[j] = _sign(w[m]) * U[j, m]
    return s


@dataclass
class GaInfo:
    iters: List[int] = field(default_factory=list)        # iterations used by each component
    maxiter: List[bool] = field(default_factory=list)     # the @warn of :306 fired
    dq_hist: List[List[float]] = field(default_factory=list)


def rpca_ga_1(Xnorms, U, w, q0, tol=1e-7, iters=1000, mu: Callable = mu_mean, info: Optional[GaInfo] = None):
    """src/robustPCA.jl:286-310 with the start vector given (`randn(d)` at :289 is replaced by q0)."""
    q = np.array(q0, dtype=U.dtype, copy=True)
    q /= np.sqrt(np.sum(q * q))                                   # :290
    qold = q.copy()
    hist = []
    used, warned = 0, False
    for i in range(1, iters + 1):
        w[:] = _sign(U.T @ q) * Xnorms                            # :294-296
        mui = mu(q, w, U)                                          # :297  (μ! writes into q)
        q[:] = mui / np.sqrt(np.sum(mui * mui))                    # :298
        dq = math.sqrt(float(np.sum((q - qold) ** 2)))             # :299
        hist.append(dq)
        used = i
        if dq < tol:                                               # :301
            break
        qold[:] = q
        if i == iters:
            warned = True                                          # :306
    if info is not None:
        info.iters.append(used)
        info.maxiter.append(warned)
        info.dq_hist.append(hist)
    return q


def rpca_ga(X, r=None, q0=None, tol=1e-7, iters=1000, mu: Callable = mu_mean, seed=0, info: Optional[GaInfo] = None):
    """src/robustPCA.jl:255-281.  X is d x N (columns are the observations); returns Q (d x r).
    q0: d x r start vectors (column i starts component i); None -> seeded normals."""
    X = np.array(X, dtype=np.float64, copy=True, order="F")        # :257
    d, N = X.shape
    if r is None:
        r = min(d, N)
    if q0 is None:
        q0 = np.random.default_rng(seed).standard_normal((d, r))
    Q = np.zeros((d, r))
    w = np.zeros(N)
    U = np.empty_like(X)
    for i in range(r):
        Xnorms = np.sqrt(np.sum(X * X, axis=0))                   # :264
        with np.errstate(invalid="ignore", divide="ignore"):
            U[:, :] = X / Xnorms                                   # :265
        q = rpca_ga_1(Xnorms, U, w, q0[:, i], tol=tol, iters=iters, mu=mu, info=info)
        Q[:, i] = q                                                # :268
        Xs1 = q @ X                                                # :270
        X -= np.outer(q, Xs1)                                      # :271
    return Q


def subspace_gap(Q, u):
    """The statistic of test/runtests.jl:503,519: sum of the r smallest singular values of [Q u]."""
    r = u.shape[1]
    s = np.linalg.svd(np.hstack([Q, u]), compute_uv=False)
    return float(np.sum(s[r:2 * r]))
