"""Row-sharded restatement of the rpca loop — TEST INFRASTRUCTURE (see rpca_oracle.py header).

Same structure as the multi-GPU path of libtlsqhip.so: every rank owns a contiguous row
block of D; the only exchanges are sum-all-reduces of N x N Gram matrices (SVD step and opnorm) and one
scalar max at setup; the small N x N eigenproblem is replicated.  `allreduce(array, op)` is supplied by
the caller (torch.distributed with gloo in tests/test_dist_cpu.py).  Follows
/root/reference/src/robustPCA.jl:156-239 line for line otherwise.
"""
import math

import numpy as np

from . import rpca_oracle as O


def rpca_sharded(D_local, M_global, allreduce, lam=None, iters=1000, tol=None, rho=1.5, nukeA=True):
    D = np.asfortranarray(D_local, dtype=np.float64)
    Ml, N = D.shape
    if lam is None:
        lam = 1.0 / math.sqrt(max(M_global, N))
    if tol is None:
        tol = math.sqrt(np.finfo(np.float64).eps)

    def opnorm_sharded(X):
        G = allreduce(X.T @ X, "sum")
        return math.sqrt(max(np.linalg.eigvalsh(G)[-1], 0.0))

    A = np.zeros_like(D)
    E = np.zeros_like(D)
    Y = D.copy()
    norm2 = opnorm_sharded(Y)
    norminf = float(allreduce(np.array([np.max(np.abs(Y)) if Y.size else 0.0]), "max")[0]) / lam
    dual = max(norm2, norminf)
    Y /= dual
    mu = 1.25 / norm2
    mubar = mu * 1e7
    sv = 10
    svp_hist, cost_hist = [], []
    converged = False
    for k in range(1, iters + 1):
        inv_mu = 1.0 / mu
        E = O.soft_th((D - A) + inv_mu * Y, lam / mu)
        Z = (D - E) + inv_mu * Y
        G = allreduce(Z.T @ Z, "sum")
        w, V = np.linalg.eigh(G)
        w, V = w[::-1], V[:, ::-1]
        S = np.sqrt(np.maximum(w, 0.0))
        svp = int(np.sum(S >= inv_mu))
        sv = max(svp, 1)
        g = (S[:svp] - inv_mu) / S[:svp] if nukeA else np.ones(svp)
        A = (Z @ (V[:, :svp] * g)) @ V[:, :svp].T
        R = (D - A) - E
        Y = Y + mu * R
        mu = min(mu * rho, mubar)
        cost = opnorm_sharded(R) / norm2
        svp_hist.append(svp)
        cost_hist.append(cost)
        if cost < tol:
            converged = True
            break
    return A, E, sv, dict(iters_done=k, svp_hist=svp_hist, cost_hist=cost_hist, converged=converged)


def rpca_ga_sharded(X_local, r, q0, allreduce, tol=1e-7, iters=1000):
    """Column-sharded rpca_ga with the default average μ! (src/robustPCA.jl:255-320) — the structure of the library's
    multi-GPU path (grassmann.hip: ga_iteration): every rank owns a block of COLUMNS (observations); norms,
    normalisation and the deflation are local; the only exchange is a sum-all-reduce of the d+1 numbers
    [sum_n w_n U_n ; sum_n w_n] per iteration, after which q evolves identically on every rank."""
    X = np.array(X_local, dtype=np.float64, copy=True, order="F")
    d = X.shape[0]
    Q = np.zeros((d, r))
    used = []
    for i in range(r):
        norms = np.sqrt(np.sum(X * X, axis=0))
        U = X / norms
        q = np.array(q0[:, i], dtype=np.float64)
        q /= np.sqrt(np.sum(q * q))
        qold = q.copy()
        it = 0
        for it in range(1, iters + 1):
            w = np.sign(U.T @ q) * norms
            tot = allreduce(np.concatenate([U @ w, [np.sum(w)]]), "sum")
            mu = tot[:d] / tot[d]
            q = mu / np.sqrt(np.sum(mu * mu))
            dq = math.sqrt(float(np.sum((q - qold) ** 2)))
            if dq < tol:
                break
            qold = q.copy()
        used.append(it)
        Q[:, i] = q
        X -= np.outer(q, q @ X)
    return Q, used


def lowrankfilter_sharded(y, n, rank, world, allreduce, lag=1, tol=1e-3):
    """Time-window sharded lowrankfilter (src/robustPCA.jl:119-128) — the structure of tlsq_lowrankfilter_f64 with a
    communicator (SURVEY §8e "Hankel"): every rank owns a contiguous block of the rows of H = hankel(y, n, lag), i.e. a
    window of y with an (n-1)-sample halo; rpca runs row-sharded; the anti-diagonal averaging exchanges partial sums
    and counts with one sum all-reduce, then divides."""
    y = np.asarray(y, dtype=np.float64)
    Nx = y.shape[0]
    Kg = (Nx - n) // lag + 1
    base, rem = divmod(Kg, world)
    r0 = rank * base + min(rank, rem)
    r1 = r0 + base + (1 if rank < rem else 0)
    s0, Nw = r0 * lag, (r1 - r0 - 1) * lag + n
    H = O.hankel(y[s0:s0 + Nw], n, lag)                       # this rank's rows
    A, E, sv, info = rpca_sharded(H, Kg, allreduce, tol=tol)
    tot = np.zeros(Nx)
    cnt = np.zeros(Nx)
    K = r1 - r0
    for k in range(K):
        for l in range(n):
            tot[s0 + k * lag + l] += A[k, l]
            cnt[s0 + k * lag + l] += 1.0
    both = allreduce(np.concatenate([tot, cnt]), "sum")
    tot, cnt = both[:Nx], both[Nx:]
    return np.where(cnt > 0, tot / np.maximum(cnt, 1.0), 0.0), info
