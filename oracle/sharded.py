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


def _sortable(x):
    """float64 -> uint64 with the order of Julia's isless (-0.0 < 0.0, NaN last): grassmann.hip, ga_sortable"""
    u = np.ascontiguousarray(x, dtype=np.float64).view(np.uint64).copy()
    u[np.isnan(x)] = np.uint64(0x7FF8000000000000)
    neg = (u >> np.uint64(63)).astype(bool)
    u[neg] = ~u[neg]
    u[~neg] |= np.uint64(0x8000000000000000)
    return u


def select_rows_sharded(keys_local, col_off, ranks, allreduce):
    """The elements of stable rank `ranks[t]` (0-based, within the WHOLE row) of every row of a column-sharded key matrix,
    as (sortable value, global column index) per (row, target) - the structure of grassmann.hip, ga_select_rows: a
    most-significant-digit radix select on the composite key (value, column index of the whole row); every rank
    histograms the current byte over its own columns, the histograms are summed over the ranks (one all-reduce of
    d x targets x 256 counts per pass, twelve passes), and every rank fixes the same digit.  What the stable
    `sortperm(U[j,:])` (src/robustPCA.jl:332, :357) puts at that position, without gathering the row anywhere."""
    d, n = keys_local.shape
    nt = len(ranks)
    u = _sortable(keys_local.reshape(-1)).reshape(d, n)
    idx = (np.arange(n, dtype=np.uint64) + np.uint64(col_off))
    pkey = np.zeros((d, nt), dtype=np.uint64)
    pidx = np.zeros((d, nt), dtype=np.uint64)
    k = np.tile(np.asarray(ranks, dtype=np.int64), (d, 1))
    for p in range(12):
        hist = np.zeros((d, nt, 256))
        for j in range(d):
            for t in range(nt):
                if p < 8:
                    hi = np.uint64(8 * (8 - p))
                    cand = np.ones(n, bool) if p == 0 else (u[j] >> hi) == (pkey[j, t] >> hi)
                    dig = (u[j][cand] >> np.uint64(8 * (7 - p))) & np.uint64(255)
                else:
                    q = p - 8
                    cand = u[j] == pkey[j, t]
                    if q > 0:
                        hi = np.uint64(8 * (4 - q))
                        cand &= (idx >> hi) == (pidx[j, t] >> hi)
                    dig = (idx[cand] >> np.uint64(8 * (3 - q))) & np.uint64(255)
                hist[j, t] = np.bincount(dig.astype(np.int64), minlength=256)
        hist = allreduce(hist.reshape(-1), "sum").reshape(d, nt, 256)
        for j in range(d):
            for t in range(nt):
                cum = np.cumsum(hist[j, t])
                b = int(np.searchsorted(cum, k[j, t], side="right"))
                k[j, t] -= int(cum[b] - hist[j, t, b])
                if p < 8:
                    pkey[j, t] |= np.uint64(b) << np.uint64(8 * (7 - p))
                else:
                    pidx[j, t] |= np.uint64(b) << np.uint64(8 * (11 - p))
    return pkey, pidx


def rpca_ga_sharded(X_local, r, q0, allreduce, tol=1e-7, iters=1000, average="mean", P=0.1, col_off=0, N_glob=None):
    """Column-sharded rpca_ga (src/robustPCA.jl:255-362) — the structure of the library's multi-GPU path (grassmann.hip:
    ga_iteration): every rank owns a block of COLUMNS (observations); norms, normalisation and the deflation are local.
    average = "mean" (μ!, :312-320): the only exchange is a sum all-reduce of the d+1 numbers [sum_n w_n U_n ; sum_n w_n]
    per iteration.  "trimmed_mean" (:327-337): the membership of every entry in the kept range of its row is found once
    per component by select_rows_sharded (two ranks per row), then 2d sums per iteration.  "median" (:354-362): one
    selection per iteration, the rank that owns the selected column delivers sign(w_m) U[j,m], a sum all-reduce of d
    numbers.  After the exchange q evolves identically on every rank."""
    X = np.array(X_local, dtype=np.float64, copy=True, order="F")
    d, n = X.shape
    Ng = n if N_glob is None else N_glob
    Q = np.zeros((d, r))
    used = []
    gidx = np.arange(n, dtype=np.uint64) + np.uint64(col_off)
    for i in range(r):
        norms = np.sqrt(np.sum(X * X, axis=0))
        U = X / norms
        q = np.array(q0[:, i], dtype=np.float64)
        q /= np.sqrt(np.sum(q * q))
        qold = q.copy()
        mask = None
        if average == "trimmed_mean":
            lo, hi = int(math.floor(P * Ng)), int(math.floor((1 - P) * Ng))
            mask = np.zeros((d, n), bool)
            if hi > lo:
                has_hi = hi < Ng
                pk, pi = select_rows_sharded(U, col_off, [lo, hi if has_hi else lo], allreduce)
                us = _sortable(U.reshape(-1)).reshape(d, n)
                for j in range(d):
                    ge = (us[j] > pk[j, 0]) | ((us[j] == pk[j, 0]) & (gidx >= pi[j, 0]))
                    lt = np.ones(n, bool) if not has_hi else (us[j] < pk[j, 1]) | ((us[j] == pk[j, 1]) & (gidx < pi[j, 1]))
                    mask[j] = ge & lt
        it = 0
        for it in range(1, iters + 1):
            w = np.sign(U.T @ q) * norms
            if average == "mean":
                tot = allreduce(np.concatenate([U @ w, [np.sum(w)]]), "sum")
                mu = tot[:d] / tot[d]
            elif average == "trimmed_mean":
                tot = allreduce(np.concatenate([(U * mask) @ w, mask @ w]), "sum")
                mu = tot[:d] / tot[d:]
            else:
                pk, pi = select_rows_sharded(U * w, col_off, [Ng // 2 - 1], allreduce)
                m = pi[:, 0].astype(np.int64) - col_off
                mine = (m >= 0) & (m < n)
                sj = np.zeros(d)
                jj = np.nonzero(mine)[0]
                sj[jj] = np.sign(w[m[jj]]) * U[jj, m[jj]]
                mu = allreduce(sj, "sum")
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
