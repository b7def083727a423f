"""Pins the CPU oracle (oracle/rpca_oracle.py) against every known-answer vector the
reference's own tests hold for the hot path (tests/golden/reference_vectors.json, transcribed from
/root/reference/test/runtests.jl) and re-expresses its statistical tests with our own seeded RNG
and the reference's thresholds."""
import math

import numpy as np
import pytest

from oracle import rpca_oracle as O


def test_rpca_5x5_known_answer(golden):          # test/runtests.jl:141-165
    g = golden["rpca_5x5"]
    D = np.array(g["D"])
    A, E, s, sv, info = O.rpca(D, nonnegE=True, nonnegA=True)
    assert np.allclose(A, np.array(g["A"]), rtol=0, atol=g["atol"])
    assert np.allclose(E, np.array(g["E"]), rtol=0, atol=g["atol"])
    assert np.linalg.norm(D - (A + E)) / np.linalg.norm(D) < math.sqrt(np.finfo(float).eps)
    assert info.converged


def test_rpca_5x5_residual_flags_off(golden):    # test/runtests.jl:168-169
    D = np.array(golden["rpca_5x5"]["D"])
    A, E, *_ = O.rpca(D)
    assert np.linalg.norm(D - (A + E)) / np.linalg.norm(D) < math.sqrt(np.finfo(float).eps)


def test_hankel_identities(golden):              # test/runtests.jl:293-299
    for key in ("hankel_L2", "hankel_L3_lag2"):
        g = golden[key]
        X = O.hankel(np.array(g["x"]), g["L"], g["lag"])
        assert X.dtype.kind == "i"
        assert np.array_equal(X, np.array(g["X"]))
    g = golden["ishankel_true"]
    A = O.hankel(np.array(g["x"]), g["L"])
    assert O.ishankel(A)
    rng = np.random.default_rng(0)
    assert not O.ishankel(A + 0.1 * rng.standard_normal(A.shape))


def _y(T=1000):
    y = np.sin(0.1 * np.arange(1, T + 1))
    return y / np.quantile(np.abs(y), 0.9)


def test_unhankel_round_trips(golden):           # test/runtests.jl:355-376
    T = golden["unhankel"]["T"]
    y = _y(T)
    H = O.hankel(y, 2)
    assert O.ishankel(H)
    assert np.array_equal(O.unhankel(H), y)
    assert np.array_equal(O.unhankel(O.hankel(y, 2, 2), 2, T), y)
    yh = O.unhankel(O.hankel(y, 5, 2), 2, T)
    assert np.allclose(yh[:-1], y[:-1]) and yh[-1] == 0.0
    y2 = np.random.default_rng(1).standard_normal(T)
    yy = np.column_stack([y, y2])
    yh = O.unhankel(O.hankel(yy, 5, 2), 2, T, 2)
    assert np.allclose(yh[:-1], yy[:-1])


def test_hankel_asserts():                        # src/robustPCA.jl:79-80
    with pytest.raises(AssertionError):
        O.hankel(np.arange(10.0), 6)
    with pytest.raises(AssertionError):
        O.hankel(np.arange(20.0), 2, 3)


def test_tls_equals_tls_inplace():                # test/runtests.jl:43
    rng = np.random.default_rng(0)
    x = rng.standard_normal(3)
    A = rng.standard_normal((50, 3))
    An = A + rng.standard_normal(A.shape)
    yn = A @ x + 0.01 * rng.standard_normal(50)
    xa = O.tls(An, yn)
    xb = O.tls_inplace(np.column_stack([An, yn]), 3)[:, 0]
    assert np.allclose(xa, xb)
    assert np.linalg.norm(x - xa) < 1


def test_soft_th_forms():                         # src/robustPCA.jl:1-7
    x = np.array([-3.0, -0.5, 0.0, 0.5, 3.0])
    assert np.array_equal(O.soft_th(x, 1.0), [-2.0, 0.0, 0.0, 0.0, 2.0])
    m = 0.3
    r = O.soft_th(x, 1.0, m)
    assert r[1] == m and r[2] == m and r[3] == m       # exactly l inside the dead zone
    assert np.allclose(r, [-2.0, m, m, m, 2.0])
    z = np.array([3 + 4j, 0.1j])
    w = O.soft_th(z, 1.0)
    assert np.allclose(np.abs(w), [4.0, 0.0]) and np.allclose(np.angle(w[0]), np.angle(z[0]))


def test_missing_values_lowrankfilter():          # test/runtests.jl:172-185 (20 trials, < 0.025)
    rng = np.random.default_rng(0)
    res = []
    for _ in range(6):
        N = 500
        y = np.sin(0.1 * np.arange(1, N + 1)) + 0.1 * rng.standard_normal(N)
        miss = rng.random(N) < 0.1
        yn = y + miss * 1e2
        yf = O.lowrankfilter(yn, 40)
        res.append(np.mean((y - yf) ** 2) / np.mean(y ** 2))
    assert np.mean(res) < 0.025


def test_complex_rpca():                          # test/runtests.jl:187-199
    rng = np.random.default_rng(0)
    c = lambda *s: (rng.standard_normal(s) + 1j * rng.standard_normal(s)) / math.sqrt(2)
    u, v = c(100), c(20)
    E = c(100, 20) * 10 * (rng.random((100, 20)) < 0.01)
    A = np.outer(u, v.conj())
    Ah, Eh, s, sv, _ = O.rpca(A + E)
    assert np.sum(np.abs(Eh - E) ** 2) / np.sum(np.abs(E) ** 2) < 1e-5
    assert np.sum(np.abs(Ah - A) ** 2) / np.sum(np.abs(A) ** 2) < 1e-5


def test_rtls_beats_tls():                        # test/runtests.jl:221-235 (>0.9 over 1000 trials)
    rng = np.random.default_rng(0)
    wins, n = 0, 60
    for _ in range(n):
        x = rng.standard_normal(3)
        A = rng.standard_normal((50, 3))
        s = 50
        An = A + s * rng.standard_normal(A.shape) * (rng.random(A.shape) < 0.1)
        y = A @ x
        yn = y + s * rng.standard_normal(50) * (rng.random(50) < 0.1)
        xt = O.tls(An, yn)
        xr = O.rtls(An, yn)
        wins += np.linalg.norm(x - xr) < np.linalg.norm(x - xt)
    assert wins / n > 0.8


def test_soft_hankel_moves_closer_and_exact_hankel():   # test/runtests.jl:296-348
    rng = np.random.default_rng(0)
    A = O.hankel(np.arange(1.0, 9.0), 4)
    An = A + 0.1 * rng.standard_normal(A.shape)
    Anc = An.copy()
    O.soft_hankel_(An, 0.1)
    assert np.sum((An - A) ** 2) < np.sum((Anc - A) ** 2)
    wins, n = 0, 200
    for _ in range(n):
        y = rng.standard_normal(100)
        H = O.hankel(y, 5)
        A1, *_ = O.rpca(H, nukeA=False)
        A2, E2, *_ = O.rpca(H, nukeA=False, hankel=True)
        assert O.ishankel(A2) and O.ishankel(E2)          # exact
        wins += np.mean((A2 - H) ** 2) < np.mean((A1 - H) ** 2)
    assert wins / n > 0.8


def test_lowrankfilter_sine_spikes():             # test/runtests.jl:378-381, 401-405
    T = 1000
    y = _y(T)
    rng = np.random.default_rng(0)
    n = 20 * rng.standard_normal(T) * (rng.random(T) < 0.01) + 0.1 * rng.standard_normal(T)
    qn = lambda x: x / np.quantile(np.abs(x), 0.9)
    yf = qn(O.lowrankfilter(y + n))
    assert np.mean((y - yf) ** 2) / np.mean(n ** 2) < 0.001
    n = rng.standard_normal(T)
    yf = qn(O.lowrankfilter(y + n, sv=2))
    assert np.mean((y - yf) ** 2) / np.mean(n ** 2) < 0.05


# ---- rpca_ga (oracle/ga_oracle.py): the reference's own tests of this path, test/runtests.jl:443-520 ----------
from oracle import ga_oracle as G  # noqa: E402


def _mu_blas(s, w, U):     # μ! with a BLAS summation order (the statistical tests below do not depend on it)
    s[:] = (U @ w) / np.sum(w)
    return s


def _lowrank_case(rng, d, N, r, eps, sigma=None):
    u, sv, vt = np.linalg.svd(rng.standard_normal((d, N)), full_matrices=False)
    u, v = u[:, :r], vt[:r, :].T
    sig = 10.0 * np.arange(1, r + 1) if sigma is None else sv[:r]
    return u, (u * sig) @ v.T + eps * rng.standard_normal((d, N))


@pytest.mark.parametrize("shape", [(10, 40), (40, 10)])
def test_rpca_ga_orthonormal_components(shape):   # test/runtests.jl:446-464
    rng = np.random.default_rng(1)
    for r in range(1, 11):
        for eps in 10.0 ** np.linspace(-8, 0, 5):
            _, A = _lowrank_case(rng, *shape, r, eps)
            Q = G.rpca_ga(A, r, seed=int(rng.integers(1 << 30)))
            assert np.linalg.norm(Q.T @ Q - np.eye(r)) < math.sqrt(np.finfo(float).eps)


def test_ga_averages_identities():                # test/runtests.jl:469-490
    rng = np.random.default_rng(2)
    U = rng.standard_normal((10, 10))
    s = np.zeros(10)
    w = np.ones(10)
    assert np.allclose(G.mu_mean(s, w, U).copy(), U.mean(axis=1))
    assert np.allclose(G.entrywise_trimmed_mean(s, w, U, 0).copy(), U.mean(axis=1))
    w = rng.standard_normal(10)
    ref = (U * w).sum(axis=1) / w.sum()
    assert np.allclose(G.mu_mean(s, w, U).copy(), ref)
    assert np.allclose(G.entrywise_trimmed_mean(s, w, U, 0).copy(), ref)
    w = np.ones(10)
    m2 = G.entrywise_trimmed_mean(s, w, U, 0.1).copy()
    for i in range(10):                           # StatsBase.trim(x, prop=0.1): drop floor(0.1 n) from each end
        assert np.isclose(m2[i], np.mean(np.sort(U[i, :])[1:-1]))
    # the sequential and the BLAS summation of μ! agree to rounding
    assert np.allclose(G.mu_mean(np.zeros(10), ref, U), _mu_blas(np.zeros(10), ref, U), rtol=1e-12)


def test_ga_median_definition():                  # src/robustPCA.jl:354-362
    U = np.array([[3.0, -1.0, 2.0, 5.0], [0.5, 0.25, -4.0, 1.0]])
    w = np.array([1.0, -2.0, 1.0, 0.5])
    s = G.entrywise_median(np.zeros(2), w, U)
    # row 1: w.*U = [3, 2, 2, 2.5]; stable order -> columns 2,3,4,1; I[end÷2] = I[2] = column 3 -> sign(1)*2
    # row 2: w.*U = [0.5, -0.5, -4, 0.5] -> columns 3,2,1,4; I[2] = column 2 -> sign(-2)*0.25
    assert s.tolist() == [2.0, -0.25]


@pytest.mark.parametrize("robust,thr", [(G.entrywise_trimmed_mean, 0.8), (G.entrywise_median, 0.9)])
def test_ga_robust_average_beats_mean(robust, thr):   # test/runtests.jl:492-520
    rng = np.random.default_rng(1)
    wins = []
    for r in range(1, 5):
        for eps in 10.0 ** np.linspace(-8, -1, 3):
            u, A = _lowrank_case(rng, 10, 1000, r, eps, sigma="svd")
            A = A + 1000 * rng.standard_normal(A.shape) * (rng.random(A.shape) < 0.01)
            q0 = rng.standard_normal((10, r))
            Qr = G.rpca_ga(A, r, q0=q0, mu=robust, iters=120)
            Qm = G.rpca_ga(A, r, q0=q0, mu=_mu_blas, iters=120)
            wins.append(G.subspace_gap(Qr, u) < G.subspace_gap(Qm, u))
    assert np.mean(wins) > thr


# ---- frozen oracle outputs: guards the oracle itself against drift ---------------------------
@pytest.mark.parametrize("name", ["c1_500x50_r5", "c2s_2000x128_r8"])
def test_oracle_reproduces_frozen_vectors(oracle_golden, name):
    import hashlib
    g = oracle_golden[name]
    D, _, _ = O.synth_lowrank_sparse(g["M"], g["N"], g["rank"], seed=g["seed"])
    assert hashlib.sha256(np.asfortranarray(D).tobytes(order="F")).hexdigest() == g["D_sha256_of_float64_column_major"]
    A, E, s, sv, info = O.rpca(D)
    assert info.iters_done == g["iters_done"] and sv == g["sv"] and list(info.svp_hist) == g["svp_hist"]
    assert np.allclose(info.cost_hist, g["cost_hist"], rtol=1e-9, atol=1e-14)
    assert np.allclose(s[1], g["S"], rtol=1e-10, atol=1e-12 * g["S"][0])
    k = g["sample_stride"]
    assert np.abs(A.ravel(order="F")[::k] - np.array(g["A_sample"])).max() <= 1e-10 * g["normA"]
    assert np.abs(E.ravel(order="F")[::k] - np.array(g["E_sample"])).max() <= 1e-10 * g["normE"]
