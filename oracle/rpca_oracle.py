"""CPU oracle: a numpy/LAPACK restatement of the reference's rpca / lowrankfilter /
tls! / rtls path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it.  The product path
(`totalleastsquares.jl_amd`, `libtlsqhip.so`) never calls anything in `oracle/`.

Reference = baggepinnen/TotalLeastSquares.jl v1.8.0 (Julia).  Julia is not present
in this image, so the reference itself cannot be executed; the heavy arithmetic
of the reference lives in Julia's stdlib LinearAlgebra (LAPACK `gesdd` behind
`svd!` and `opnorm`), which this restatement reaches through scipy's bundled
LAPACK with the *same driver* (`lapack_driver="gesdd"`).

Parity pin: the reference's own known-answer tests, transcribed as fixtures in
`tests/golden/reference_vectors.json` (5x5 rpca table test/runtests.jl:143-165,
hankel/unhankel/ishankel identities :293-299,:361-376, tls==tls! :43) — see
`tests/test_oracle_golden.py`.

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Expression order of the elementwise sweeps is kept exactly
as written there so that the HIP kernels can be compared bit-for-bit.
"""
from __future__ import annotations

import ctypes
import math
import os
from dataclasses import dataclass, field
from typing import Callable, Optional

import numpy as np
import scipy.linalg as sla

_HERE = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------
# optional fused/threaded sweeps (oracle/sweeps.c) — same arithmetic, used so the
# cpu_baseline is not penalised by numpy temporaries (Julia's broadcast is fused)
# --------------------------------------------------------------------------
_sweeps = None


def _load_sweeps():
    global _sweeps
    if _sweeps is not None:
        return _sweeps
    p = os.path.join(_HERE, "liboracle_sweeps.so")
    if os.path.exists(p):
        try:
            lib = ctypes.CDLL(p)
            dp = ctypes.POINTER(ctypes.c_double)
            lib.oracle_k1_f64.argtypes = [dp, dp, dp, dp, dp, ctypes.c_int64, ctypes.c_double,
                                          ctypes.c_double, ctypes.c_int]
            lib.oracle_k2_f64.argtypes = [dp, dp, dp, dp, dp, ctypes.c_int64, ctypes.c_double,
                                          ctypes.c_int]
            lib.oracle_k1_f64.restype = None
            lib.oracle_k2_f64.restype = None
            _sweeps = lib
        except OSError:
            _sweeps = False
    else:
        _sweeps = False
    return _sweeps


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


# --------------------------------------------------------------------------
# soft thresholds — src/robustPCA.jl:1-7
# --------------------------------------------------------------------------
def soft_th(x, eps, l=None):
    """src/robustPCA.jl:1 (2-arg), :2 (3-arg, shrink towards l), :3-7 (complex)."""
    x = np.asarray(x)
    if l is not None:
        # max(x-ϵ,l) + min(x+ϵ,l) - l      (robustPCA.jl:2)
        return (np.maximum(x - eps, l) + np.minimum(x + eps, l)) - l
    if np.iscomplexobj(x):
        # modulus shrink, keep phase          (robustPCA.jl:3-7)
        m = np.abs(x)
        a = np.angle(x)
        m = np.maximum(m - eps, 0.0) + np.minimum(m + eps, 0.0)
        return m * np.exp(1j * a)
    # max(x-ϵ,0) + min(x+ϵ,0)               (robustPCA.jl:1)
    return np.maximum(x - eps, 0.0) + np.minimum(x + eps, 0.0)


def _antidiag_indices(K, L, k):
    """1-based k in 1..K+L-1 -> 0-based (rows, cols) in the reference's order:
    ri = min(K,k):-1:max(k-L,1), ci = max(1,k-K+1):L, zipped (robustPCA.jl:13-14)."""
    r_hi, r_lo = min(K, k), max(k - L, 1)
    c_lo = max(1, k - K + 1)
    n = min(r_hi - r_lo + 1, L - c_lo + 1)
    rows = np.arange(r_hi, r_hi - n, -1) - 1
    cols = np.arange(c_lo, c_lo + n) - 1
    return rows, cols


def _seq_mean(v):
    """Julia's mean over a generator: sequential left-to-right sum, then /count."""
    tot = v[0] / 1
    for t in v[1:]:
        tot = tot + t
    return tot / len(v)


def soft_hankel_(A, eps):
    """In-place soft_hankel! — src/robustPCA.jl:9-21."""
    K, L = A.shape
    for k in range(1, K + L):
        r, c = _antidiag_indices(K, L, k)
        v = A[r, c]
        m = _seq_mean(v) if len(v) < 64 else (np.cumsum(v)[-1] / len(v))
        A[r, c] = soft_th(v, eps, m)
    return A


def unhankel(A, lag=None, N=None, D=1):
    """src/robustPCA.jl:28-39 (1-arg) and :53-68 (general lag / channels)."""
    A = np.asarray(A)
    if lag is None or (lag == 1 and D == 1):
        K, L = A.shape
        n = L + (K - 1)
        y = np.empty(n, dtype=A.dtype)
        for k in range(1, n + 1):
            r, c = _antidiag_indices(K, L, k)
            v = A[r, c]
            y[k - 1] = _seq_mean(v) if len(v) < 64 else (np.cumsum(v)[-1] / len(v))
        return y
    K = A.shape[0]
    L = A.shape[1] // D
    y = np.zeros((N, D), dtype=A.dtype)
    counts = np.zeros((N, D), dtype=np.int64)
    # index-Hankel of the (n, d) pairs, visited in column-major order (:60-65)
    rows = np.arange(N)
    idx_n = hankel(np.repeat(rows[:, None], D, axis=1), L, lag)          # K x (L*D)
    idx_d = hankel(np.repeat(np.arange(D)[None, :], N, axis=0), L, lag)
    for j in range(A.shape[1]):           # column-major traversal
        np.add.at(y, (idx_n[:, j], idx_d[:, j]), A[:, j])
        np.add.at(counts, (idx_n[:, j], idx_d[:, j]), 1)
    y = y / np.maximum(counts, 1)
    if D == 1:
        return y[:, 0]
    return y


def hankel(x, L, lag=1):
    """src/robustPCA.jl:76-92.  x: (N,) or (N,D) -> K x (L*D), K=(N-L)÷lag+1,
    channel-interleaved columns (colinds = d:D:L*D)."""
    x = np.asarray(x)
    x2 = x.reshape(x.shape[0], -1)
    N, D = x2.shape
    assert L <= N / 2, f"L has to be less than N/2 = {N / 2}"          # :79
    assert lag <= L, "lag must be <= L"                                 # :80
    K = (N - L) // lag + 1
    X = np.empty((K, L * D), dtype=x.dtype)
    idx = np.arange(K)[:, None] * lag + np.arange(L)[None, :]        # inds .+ lag per row (:88)
    for d in range(D):
        X[:, d::D] = x2[idx, d]                                        # colinds = d:D:L*D (:85)
    return X


def ishankel(A):
    """src/robustPCA.jl:94-106 — exact (!=) test of constant anti-diagonals."""
    A = np.asarray(A)
    K, L = A.shape
    for k in range(1, K + L):
        r, c = _antidiag_indices(K, L, k)
        v = A[r, c]
        if np.any(v != v[0]):
            return False
    return True


# --------------------------------------------------------------------------
# SVD helpers
# --------------------------------------------------------------------------
def _svd_full(Z):
    """LinearAlgebra.svd!(Z) -> LAPACK gesdd('S') (thin)."""
    U, S, Vt = sla.svd(Z, full_matrices=False, lapack_driver="gesdd", overwrite_a=False,
                       check_finite=False)
    return U, S, Vt


def opnorm2(X):
    """LinearAlgebra.opnorm(X) (p=2) -> svdvals -> gesdd('N')."""
    if X.size == 0:
        return 0.0
    return float(sla.svdvals(X, check_finite=False)[0])


@dataclass
class RpcaInfo:
    iters_done: int = 0
    converged: bool = False
    final_cost: float = float("nan")
    final_mu: float = float("nan")
    cost_hist: list = field(default_factory=list)
    svp_hist: list = field(default_factory=list)
    warned: bool = False


# --------------------------------------------------------------------------
# rpca — src/robustPCA.jl:156-239
# --------------------------------------------------------------------------
def rpca(D, lam=None, maxrank=None, iters=1000, tol=None, rho=1.5, verbose=False,
         nonnegA=False, nonnegE=False, hankel=False, nukeA=True,
         svd: Optional[Callable] = None, opnorm: Optional[Callable] = None,
         fused_sweeps=True):
    """Inexact-ALM robust PCA, expression-for-expression after src/robustPCA.jl:156-239.

    svd: None -> LinearAlgebra.svd! (gesdd) every iteration; otherwise a callable
         svd(Z, sv) -> (U,S,Vt) used for k>=2 (k==1 is always full, :193).
    opnorm: None -> exact; otherwise callable X -> float (:177,:225).
    Returns A, E, (U,S,Vt), sv, info.
    """
    D = np.asarray(D)
    T = D.dtype
    cplx = np.iscomplexobj(D)
    RT = np.finfo(T).dtype.type                                   # :171
    M, N = D.shape
    if lam is None:
        lam = RT(1.0 / math.sqrt(max(M, N)))                      # :157
    if tol is None:
        tol = math.sqrt(np.finfo(RT).eps)                         # :160
    if maxrank is None:
        maxrank = np.iinfo(np.int64).max                          # :158
    rho = RT(rho)
    lam = RT(lam)
    opn = opnorm if opnorm is not None else opnorm2
    A = np.zeros((M, N), dtype=T, order="F")                      # :174
    E = np.zeros((M, N), dtype=T, order="F")
    Z = np.empty((M, N), dtype=T, order="F")                      # :175
    D = np.asfortranarray(D)
    Y = D.copy(order="F")                                         # :176
    norm2 = RT(opn(Y))                                            # :177
    norminf = RT(np.max(np.abs(Y)) / lam) if Y.size else RT(0)    # :178  vector inf-norm
    dual_norm = max(norm2, norminf)                               # :179
    d_norm = norm2                                                # :180
    Y /= dual_norm                                                # :181
    mu = RT(1.25 / norm2)                                         # :182
    mubar = RT(mu * 1.0e7)                                        # :183
    sv = svp = 10                                                 # :184
    info = RpcaInfo()
    s = None
    use_c = (fused_sweeps and not cplx and T == np.float64 and _load_sweeps())
    n_el = M * N
    for k in range(1, iters + 1):                                 # :186
        inv_mu = RT(1) / mu
        thr = lam / mu
        if use_c:
            _sweeps.oracle_k1_f64(_dptr(D), _dptr(A), _dptr(Y), _dptr(E), _dptr(Z), n_el,
                                  float(inv_mu), float(thr), int(nonnegE))
        else:
            E[...] = soft_th((D - A) + inv_mu * Y, thr)           # :188
            if nonnegE:
                np.maximum(E, 0, out=E)                           # :189-191
            Z[...] = (D - E) + inv_mu * Y                         # :192
        if svd is None or k == 1:                                 # :193
            U, S, Vt = _svd_full(Z)                               # :194
        else:
            U, S, Vt = svd(Z, sv)                                 # :196
        s = (U, S, Vt)
        svp = int(np.sum(S >= inv_mu))                            # :198
        sv = svp                                                  # :199-203 (both branches)
        sv = min(max(sv, 1), maxrank)                             # :204
        if nukeA:                                                 # :205-208
            A[...] = (U[:, :svp] * (S[:svp] - inv_mu)) @ Vt[:svp, :]
        else:                                                     # :209-213
            A[...] = (U[:, :svp] * S[:svp]) @ Vt[:svp, :]
        if hankel:
            soft_hankel_(A, thr)                                  # :214-216
        if nonnegA:
            np.maximum(A, 0, out=A)                               # :217-219
        if use_c:
            _sweeps.oracle_k2_f64(_dptr(D), _dptr(A), _dptr(E), _dptr(Y), _dptr(Z), n_el,
                                  float(mu), 0)
        else:
            Z[...] = (D - A) - E                                  # :221
            Y[...] = Y + mu * Z                                   # :222
        mu = RT(min(mu * rho, mubar))                             # :223
        cost = opn(Z) / d_norm                                    # :225
        info.cost_hist.append(float(cost))
        info.svp_hist.append(svp)
        info.iters_done = k
        if verbose:
            print(f"{k} cost: {float(f'{cost:.4g}')}")            # :226
        if cost < tol:                                            # :228
            if verbose:
                print("converged")
            info.converged = True
            break
        if k == iters:
            info.warned = True                                    # :232  (@warn)
    if hankel:
        soft_hankel_(E, lam / mu)                                 # :234-236 (mu already advanced)
    info.final_cost = info.cost_hist[-1] if info.cost_hist else float("nan")
    info.final_mu = float(mu)
    return A, E, s, sv, info                                      # :238


# --------------------------------------------------------------------------
# lowrankfilter — src/robustPCA.jl:119-128
# --------------------------------------------------------------------------
def lowrankfilter(y, n=None, sv=0, lag=1, tol=1e-3, svd=None, **kw):
    y = np.asarray(y)
    N0 = y.shape[0]
    Dch = 1 if y.ndim == 1 else y.shape[1]
    if n is None:
        n = min(N0 // 20, 2000)                                   # :119
    H = hankel(y, n, lag)                                         # :120
    if sv <= 0:
        A, E, _, _, _ = rpca(H, tol=tol, svd=svd, **kw)           # :122
    else:
        U, S, Vt = _svd_full(H)                                   # :124
        A = (U[:, :sv] * S[:sv]) @ Vt[:sv, :]                     # :125
    return unhankel(A, lag, N0, Dch)                              # :127


# --------------------------------------------------------------------------
# tls! / rtls / tls — src/TotalLeastSquares.jl:48-55, 63-69, 152-156
# --------------------------------------------------------------------------
def tls_from_V(V, n):
    """tls!(s::SVD, n): x = -V21 / V22 (right division)  — TotalLeastSquares.jl:65-69."""
    V21 = V[:n, n:]
    V22 = V[n:, n:]
    # X = -V21 / V22  <=>  X V22 = -V21  <=>  V22^T X^T = -V21^T
    X = np.linalg.solve(V22.T, -V21.T).T
    return X


def tls_inplace(Ay, n):
    """tls!(Ay, n) = tls!(svd!(Ay), n) — TotalLeastSquares.jl:63."""
    _, _, Vt = _svd_full(np.asarray(Ay))
    return tls_from_V(Vt.conj().T, n)


def tls(A, y):
    """tls(A,y) — TotalLeastSquares.jl:48-55."""
    A = np.asarray(A)
    y = np.asarray(y)
    AA = np.column_stack([A, y])
    X = tls_inplace(AA, A.shape[1])
    return X[:, 0] if y.ndim == 1 else X


def rtls(A, y, **kw):
    """rtls(A,y) — TotalLeastSquares.jl:152-156: rpca([A y]; nukeA=false) then tls!(s, n)
    on the LAST svd computed inside rpca (of Z, not of A)."""
    A = np.asarray(A)
    y = np.asarray(y)
    AA = np.column_stack([A, y])
    _, _, s, _, _ = rpca(AA, nukeA=False, **kw)
    X = tls_from_V(s[2].conj().T, A.shape[1])
    return X[:, 0] if y.ndim == 1 else X


# --------------------------------------------------------------------------
# seeded synthetic workloads shared by tests and bench
# --------------------------------------------------------------------------
def synth_lowrank_sparse(M, N, rank, seed=0, sparse_frac=0.05, sparse_scale=10.0,
                         dtype=np.float64):
    """D = G1 G2 + S,  G1 (M x r), G2 (r x N) iid N(0,1); S = scale*N(0,1)*Bernoulli(frac)."""
    rng = np.random.default_rng(seed)
    G1 = rng.standard_normal((M, rank))
    G2 = rng.standard_normal((rank, N))
    A0 = G1 @ G2
    S = sparse_scale * rng.standard_normal((M, N)) * (rng.random((M, N)) < sparse_frac)
    D = np.asfortranarray((A0 + S).astype(dtype))
    return D, np.asfortranarray(A0.astype(dtype)), np.asfortranarray(S.astype(dtype))


def synth_series(N, seed=0):
    """The reference's lowrankfilter test signal scaled up (test/runtests.jl:356-379):
    y = sin(0.1 t)/q0.9 + 20 N(0,1) Bernoulli(0.01) + 0.1 N(0,1)."""
    rng = np.random.default_rng(seed)
    t = np.arange(1, N + 1, dtype=np.float64)
    y = np.sin(0.1 * t)
    y = y / np.quantile(np.abs(y), 0.9)
    n = 20 * rng.standard_normal(N) * (rng.random(N) < 0.01) + 0.1 * rng.standard_normal(N)
    return y, n
