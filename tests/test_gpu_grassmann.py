"""GPU parity tests of rpca_ga (Grassmann averages) and its spherical averages, through the
C ABI, against oracle/ga_oracle.py on the same inputs and start vectors, plus the reference's own tests of this
path (test/runtests.jl:443-520).

Tolerances (fp64): averages 1e-12 relative (a different, fixed summation order); rpca_ga components 1e-9 absolute
with the same iteration count per component: the iteration ends on an exactly repeated sign pattern, after which
both sides hold normalise(sum_n sign_n norm_n U_n) of the same signs.
"""
import math
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch   # torch first: see tests/test_gpu_parity.py
    assert torch.cuda.is_available()
    torch.zeros(1, device="cuda")
    import tlsq_amd
    e = tlsq_amd.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def G():
    from oracle import ga_oracle
    return ga_oracle


def _mu_blas(s, w, U):
    s[:] = (U @ w) / np.sum(w)
    return s


def _data(rng, d, N, r, eps=1e-2, outliers=0.0):
    u, sv, vt = np.linalg.svd(rng.standard_normal((d, min(N, 4 * d + 50))), full_matrices=False)
    u = u[:, :r]
    X = (u * (10.0 * np.arange(r, 0, -1))) @ rng.standard_normal((r, N)) + eps * rng.standard_normal((d, N))
    if outliers:
        X = X + 100 * rng.standard_normal((d, N)) * (rng.random((d, N)) < outliers)
    return u, X


# ---- the averages on their own ---------------------------------------------------------------------------------
@pytest.mark.parametrize("d,N", [(10, 10), (3, 257), (16, 1000), (33, 400), (64, 2049), (100, 777), (300, 1500),
                                 (1100, 300), (2048, 260), (2500, 530)])
def test_mean_matches_oracle(eng, G, d, N):
    rng = np.random.default_rng(d * 1000 + N)
    U = rng.standard_normal((d, N))
    w = rng.standard_normal(N)
    got = eng.mu_(np.zeros(d), w, U)
    want = G.mu_mean(np.zeros(d), w, U)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-13 * np.abs(want).max())


def test_reference_average_identities(eng):          # test/runtests.jl:469-490
    rng = np.random.default_rng(5)
    U = rng.standard_normal((10, 10))
    s = np.zeros(10)
    w = np.ones(10)
    assert np.allclose(eng.mu_(s, w, U).copy(), U.mean(axis=1))
    assert np.allclose(eng.entrywise_trimmed_mean(s, w, U, 0).copy(), U.mean(axis=1))
    w = rng.standard_normal(10)
    ref = (U * w).sum(axis=1) / w.sum()
    assert np.allclose(eng.mu_(s, w, U).copy(), ref)
    assert np.allclose(eng.entrywise_trimmed_mean(s, w, U, 0).copy(), ref)
    w = np.ones(10)
    m2 = eng.entrywise_trimmed_mean(s, w, U, 0.1).copy()
    for i in range(10):
        assert np.isclose(m2[i], np.mean(np.sort(U[i, :])[1:-1]))


@pytest.mark.parametrize("d,N,P", [(10, 10, 0.1), (7, 1000, 0.1), (40, 333, 0.25), (130, 600, 0.05), (2100, 200, 0.1),
                                   (12, 50, 0.0), (5, 20, 0.6)])
def test_trimmed_mean_matches_oracle(eng, G, d, N, P):
    rng = np.random.default_rng(d + N)
    U = rng.standard_normal((d, N))
    w = rng.standard_normal(N)
    got = eng.entrywise_trimmed_mean(np.zeros(d), w, U, P)
    want = G.entrywise_trimmed_mean(np.zeros(d), w, U, P)
    assert np.allclose(got, want, rtol=1e-11, atol=1e-13, equal_nan=True)


@pytest.mark.parametrize("d,N", [(2, 4), (10, 10), (7, 1001), (40, 334), (130, 600), (2100, 200)])
def test_median_matches_oracle(eng, G, d, N):
    rng = np.random.default_rng(d + N)
    U = rng.standard_normal((d, N))
    w = rng.standard_normal(N)
    got = eng.entrywise_median(np.zeros(d), w, U)
    want = G.entrywise_median(np.zeros(d), w, U)
    assert np.array_equal(got, want)                  # a selection: exact


def test_median_stable_ties(eng, G):                  # sortperm is stable: ties resolve by column index
    U = np.array([[3.0, -1.0, 2.0, 5.0], [0.5, 0.25, -4.0, 1.0]])
    w = np.array([1.0, -2.0, 1.0, 0.5])
    assert eng.entrywise_median(np.zeros(2), w, U).tolist() == [2.0, -0.25]
    assert G.entrywise_median(np.zeros(2), w, U).tolist() == [2.0, -0.25]


# ---- rpca_ga against the oracle ---------------------------------------------------------------------------------
# (N (d+1) <= 18000 and d <= 64: the single-workgroup kernel k_ga_solo; otherwise the grid path)
@pytest.mark.parametrize("d,N,r", [(10, 40, 3), (40, 10, 4), (4, 500, 2), (3, 4400, 3), (64, 270, 3), (33, 500, 4),
                                   (4, 6000, 2), (10, 3000, 2), (24, 3000, 3), (64, 5000, 3),
                                   (200, 4000, 2), (700, 1500, 2), (2048, 300, 1), (2300, 400, 2)])
def test_rpca_ga_mean_matches_oracle(eng, G, d, N, r):
    rng = np.random.default_rng(7 * d + N)
    _, X = _data(rng, d, N, r, outliers=0.01)
    q0 = rng.standard_normal((d, r))
    info = G.GaInfo()
    want = G.rpca_ga(X, r, q0=q0, info=info, mu=G.mu_mean if d * N < 200000 else _mu_blas)
    got, rep = eng.rpca_ga(X, r, q0=q0, return_report=True)
    assert rep["iters"] == info.iters
    assert rep["status"] == [0] * r
    assert np.abs(got - want).max() < 1e-9
    assert np.linalg.norm(got.T @ got - np.eye(r)) < math.sqrt(np.finfo(float).eps)
    for i in range(r):                                # the dq trace of every component (what `verbose` prints)
        assert np.allclose(rep["dq_hist"][i], info.dq_hist[i], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("mode", ["entrywise_trimmed_mean", "entrywise_median"])
@pytest.mark.parametrize("d,N,r", [(10, 1000, 3), (40, 301, 2), (130, 900, 2)])
def test_rpca_ga_robust_matches_oracle(eng, G, mode, d, N, r):
    rng = np.random.default_rng(d + N)
    _, X = _data(rng, d, N, r, outliers=0.01)
    q0 = rng.standard_normal((d, r))
    info = G.GaInfo()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = G.rpca_ga(X, r, q0=q0, info=info, mu=getattr(G, mode), iters=60)
        got, rep = eng.rpca_ga(X, r, q0=q0, mu=mode, iters=60, return_report=True)
    assert rep["iters"] == info.iters
    assert rep["status"] == [int(b) for b in info.maxiter]
    assert np.abs(got - want).max() < 1e-9


@pytest.mark.parametrize("d,N,r", [(10, 400, 3), (40, 301, 2), (130, 2000, 2)])
def test_rpca_ga_user_average_through_the_callback(eng, G, d, N, r):
    """`rpca_ga(X, r; μ = f)` with ANY function of the host language (src/robustPCA.jl:286, :297): the weights and the unit
    columns visit the host, `f(q, w, U)` runs there (tlsq_ga_avg_cb) - the reference's own averages passed as plain callables
    reproduce the device averages and the oracle (same iterations, same dq trace), a made-up robust average (weighted mean
    of the entries within two scaled MADs of the row median) reproduces the oracle run with the same function, and an
    exception inside the function comes back as that exception."""
    rng = np.random.default_rng(11 * d + N)
    _, X = _data(rng, d, N, r, outliers=0.02)
    q0 = rng.standard_normal((d, r))

    def mad_mean(s, w, U):
        out = np.empty(U.shape[0])
        for j in range(U.shape[0]):
            v = w * U[j]
            med = np.median(v)
            keep = np.abs(v - med) <= 2.0 * 1.4826 * np.median(np.abs(v - med)) + 1e-300
            out[j] = np.sum(v[keep]) / np.sum(w[keep])
        return out

    for f, dev in ((G.mu_mean, None), (G.entrywise_median, "entrywise_median"), (mad_mean, None)):
        info = G.GaInfo()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = G.rpca_ga(X, r, q0=q0, info=info, mu=f, iters=60)
            got, rep = eng.rpca_ga(X, r, q0=q0, mu=f, iters=60, return_report=True)
        assert rep["iters"] == info.iters and rep["status"] == [int(b) for b in info.maxiter], f.__name__
        assert np.abs(got - want).max() < 1e-9, f.__name__
        for i in range(r):
            assert np.allclose(rep["dq_hist"][i], info.dq_hist[i], rtol=1e-6, atol=1e-12)
        if f is not mad_mean:   # ... and equals the device average of the same name
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                gd, rd = eng.rpca_ga(X, r, q0=q0, mu=dev, iters=60, return_report=True)
            assert rd["iters"] == rep["iters"] and np.abs(gd - got).max() < 1e-9

    class Boom(Exception):
        pass

    def bad(s, w, U):
        raise Boom("average failed")
    with pytest.raises(Boom):
        eng.rpca_ga(X, r, q0=q0, mu=bad)


def test_rpca_ga_reference_property(eng):             # test/runtests.jl:446-464, library start vectors
    rng = np.random.default_rng(1)
    for shape in ((10, 40), (40, 10)):
        for r in range(1, 11):
            for eps in 10.0 ** np.linspace(-8, 0, 4):
                u, sv, vt = np.linalg.svd(rng.standard_normal(shape), full_matrices=False)
                A = (u[:, :r] * (10.0 * np.arange(1, r + 1))) @ vt[:r, :] + eps * rng.standard_normal(shape)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    Q = eng.rpca_ga(A, r, seed=int(rng.integers(1 << 30)))
                assert np.linalg.norm(Q.T @ Q - np.eye(r)) < math.sqrt(np.finfo(float).eps)


def test_rpca_ga_default_rank_and_seed(eng):
    rng = np.random.default_rng(3)
    X = rng.standard_normal((6, 50))
    Q1 = eng.rpca_ga(X, seed=4)                       # r = minimum(size(X))  (:255)
    Q2 = eng.rpca_ga(X, seed=4)
    Q3 = eng.rpca_ga(X, seed=5)
    assert Q1.shape == (6, 6)
    assert np.array_equal(Q1, Q2)                     # reproducible run to run
    assert not np.array_equal(Q1, Q3)


def test_rpca_ga_iteration_limit_warns(eng, G):       # :306
    rng = np.random.default_rng(0)
    _, X = _data(rng, 12, 300, 2)
    q0 = rng.standard_normal((12, 2))
    with pytest.warns(UserWarning, match="Reached maximum number of iterations"):
        Q, rep = eng.rpca_ga(X, 2, q0=q0, iters=1, return_report=True)
    info = G.GaInfo()
    want = G.rpca_ga(X, 2, q0=q0, iters=1, info=info)
    assert rep["iters"] == [1, 1] and rep["status"] == [1, 1] and info.maxiter == [True, True]
    assert np.abs(Q - want).max() < 1e-12


def test_rpca_ga_bad_arguments(eng):
    import tlsq_amd
    with pytest.raises(tlsq_amd.TlsqError):
        eng.rpca_ga(np.ones((3, 1)), 1, mu="entrywise_median")      # I[end÷2] out of bounds (:358)
    with pytest.raises(tlsq_amd.TlsqError):
        eng.rpca_ga(np.ones((3, 4)), 1, mu="geometric_median")


@pytest.mark.parametrize("mode,thr", [("entrywise_trimmed_mean", 0.8), ("entrywise_median", 0.9)])
def test_rpca_ga_robust_average_beats_mean(eng, G, mode, thr):   # test/runtests.jl:492-520
    rng = np.random.default_rng(1)
    wins = []
    for r in range(1, 5):
        for eps in 10.0 ** np.linspace(-8, -1, 3):
            u, sv, vt = np.linalg.svd(rng.standard_normal((10, 1000)), full_matrices=False)
            A = (u[:, :r] * sv[:r]) @ vt[:r, :] + eps * rng.standard_normal((10, 1000))
            A = A + 1000 * rng.standard_normal(A.shape) * (rng.random(A.shape) < 0.01)
            q0 = rng.standard_normal((10, r))
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                Qr = eng.rpca_ga(A, r, q0=q0, mu=mode, iters=120)
                Qm = eng.rpca_ga(A, r, q0=q0, iters=120)
            wins.append(G.subspace_gap(Qr, u[:, :r]) < G.subspace_gap(Qm, u[:, :r]))
    assert np.mean(wins) > thr


def test_rpca_ga_large_panel_properties(eng):
    """512 x 200000 (0.8 GB): orthonormal components spanning the planted subspace despite gross outliers."""
    rng = np.random.default_rng(11)
    d, N, r = 512, 200000, 3
    u = np.linalg.qr(rng.standard_normal((d, r)))[0]
    X = (u * np.array([30.0, 20.0, 10.0])) @ rng.standard_normal((r, N)) + 0.01 * rng.standard_normal((d, N))
    Q, rep = eng.rpca_ga(X, r, seed=1, return_report=True)
    assert rep["status"] == [0, 0, 0]
    assert np.linalg.norm(Q.T @ Q - np.eye(r)) < 1e-8
    s = np.linalg.svd(np.hstack([Q, u]), compute_uv=False)
    assert s[r:].max() < 0.05                         # same subspace


def test_rpca_ga_fuzz(eng):
    """Randomised shapes / ranks / noise / outlier rates / averages (tools/fuzz_ga.py): same iteration counts and
    components (1e-9) as the oracle in every case."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fuzz_ga", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_ga.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = mod.run(cases=120, seed=3, eng=eng, verbose=False)
    assert not bad, bad


@pytest.mark.parametrize("d,N,r", [(12, 300, 2), (48, 2500, 3)])     # single-workgroup path / grid path
def test_rpca_ga_device_pointers_and_leading_dimensions(eng, G, d, N, r):
    """TLSQ_MEM_DEVICE through the raw C ABI: X, q0 and Q live in HBM with leading dimensions larger than d."""
    import ctypes as C
    import torch
    from tlsq_amd import _lib as L
    rng = np.random.default_rng(d + N)
    _, X = _data(rng, d, N, r, outliers=0.01)
    q0 = rng.standard_normal((d, r))
    ldX, ldq, ldQ = d + 3, d + 1, d + 5
    def dev(a, ld):       # column-major with leading dimension ld, padded rows filled with NaN
        buf = np.full((a.shape[1], ld), np.nan)
        buf[:, : a.shape[0]] = a.T
        return torch.from_numpy(buf).cuda()
    dX, dq0 = dev(X, ldX), dev(q0, ldq)
    dQ = torch.full((r, ldQ), float("nan"), dtype=torch.float64, device="cuda")
    o = L.GaOpts()
    eng.lib.tlsq_ga_opts_default(C.byref(o))
    o.memory = L.MEM_DEVICE
    it = np.zeros(r, dtype=np.int64)
    info = L.GaInfo()
    info.iters = it.ctypes.data_as(C.POINTER(C.c_int64))
    st = eng.lib.tlsq_rpca_ga_f64(eng.h, C.c_void_p(dX.data_ptr()), d, N, ldX, r, C.byref(o),
                                  C.c_void_p(dq0.data_ptr()), ldq, C.c_void_p(dQ.data_ptr()), ldQ, C.byref(info))
    assert st == 0
    got = dQ.cpu().numpy()
    assert np.isnan(got[:, d:]).all()                                  # padding untouched
    ginfo = G.GaInfo()
    want = G.rpca_ga(X, r, q0=q0, info=ginfo)
    assert it.tolist() == ginfo.iters
    assert np.abs(got[:, :d].T - want).max() < 1e-9
    assert np.array_equal(dX.cpu().numpy()[:, :d].T, X)                # the input is not modified (:257)


# ---- Float32 observations through the fp32 entry (tlsq_rpca_ga_f32, round 6) --------------------------------------------------
@pytest.mark.parametrize("d,N,r,mode", [(10, 40, 3, "mean"), (64, 5000, 3, "mean"), (700, 1500, 2, "mean"),
                                        (40, 301, 2, "entrywise_trimmed_mean"), (130, 900, 2, "entrywise_median")])
def test_rpca_ga_float32_entry(eng, G, d, N, r, mode):
    """`rpca_ga(X::Matrix{Float32})` (the reference's method is generic, src/robustPCA.jl:255): the panel travels as float32, is
    widened once on the device and iterated in the fp64 kernels; Q comes back in float32.  Against the fp64 entry on the widened
    panel: the same iterations per component, the same components to float32 rounding."""
    rng = np.random.default_rng(3 * d + N)
    _, X = _data(rng, d, N, r, outliers=0.01)
    X32 = np.asfortranarray(X.astype(np.float32))
    q0 = rng.standard_normal((d, r)).astype(np.float32)
    kw = {} if mode == "mean" else {"mu": mode}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got, rep = eng.rpca_ga(X32, r, q0=q0, return_report=True, **kw)
        want, rep64 = eng.rpca_ga(X32.astype(np.float64), r, q0=q0.astype(np.float64), return_report=True, **kw)
    assert got.dtype == np.float32 and got.shape == (d, r)
    assert rep["iters"] == rep64["iters"] and rep["status"] == rep64["status"]
    assert np.abs(got.astype(np.float64) - want).max() < 2e-7
    # raw C ABI with device pointers and leading dimensions
    import ctypes as C
    import torch
    from tlsq_amd import _lib as L
    ldX, ldq, ldQ = d + 3, d + 1, d + 5
    def dev(a, ld):
        buf = np.full((a.shape[1], ld), np.nan, dtype=np.float32)
        buf[:, : a.shape[0]] = a.T
        return torch.from_numpy(buf).cuda()
    dX, dq0 = dev(X32, ldX), dev(q0, ldq)
    dQ = torch.full((r, ldQ), float("nan"), dtype=torch.float32, device="cuda")
    o = L.GaOpts()
    eng.lib.tlsq_ga_opts_default(C.byref(o))
    o.memory = L.MEM_DEVICE
    o.average = {"mean": L.GA_MEAN, "entrywise_trimmed_mean": L.GA_TRIMMED_MEAN, "entrywise_median": L.GA_MEDIAN}[mode]
    st = eng.lib.tlsq_rpca_ga_f32(eng.h, C.c_void_p(dX.data_ptr()), d, N, ldX, r, C.byref(o), C.c_void_p(dq0.data_ptr()), ldq,
                                  C.c_void_p(dQ.data_ptr()), ldQ, None)
    assert st == (1 if any(rep["status"]) else 0)        # (TLSQ_MAXITER = the reference's @warn, :306 - not an error)
    gd = dQ.cpu().numpy()
    assert np.array_equal(gd[:, :d].T, got) and np.all(np.isnan(gd[:, d:]))
