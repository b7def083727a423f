"""The spectrum slicer in front of the accurate route's Jacobi (csrc/sliced.hip, round 6): the returned `s` - the complete SVD of
the last Z, src/robustPCA.jl:194, :238 under /root/reference - must not depend on whether the slicer ran: same singular values to
1e-10 of each (the bar tests/test_gpu_parity.py holds the plain route to against LAPACK), orthonormal vectors, U diag(S) Vt = Z."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch  # noqa: F401
    import tlsq_amd
    e = tlsq_amd.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("variant", ["normwise", "normwise_reg", "factor"])
@pytest.mark.parametrize("shape", [(20000, 512, 16), (6000, 256, 8), (9000, 384, 24), (5000, 1024, 12)])
def test_returned_svd_with_and_without_the_slicer(eng, shape, variant):
    """variant `normwise`: the default of the returned `s` (slices of the deflated Gram matrix itself, certified to 8 N eps ||G||);
    `factor` (SLICE_NORMWISE=0): slices of the Cholesky factor's K = L'L, then Jacobi sweeps inside the slices and over all pairs."""
    import tlsq_amd
    from oracle import rpca_oracle as O
    M, N, r = shape
    D, _, _ = O.synth_lowrank_sparse(M, N, r, seed=M + N)
    # (`normwise_reg`, SLICE_MIDJ=0: the slices' eigenproblems by Cholesky + the register kernel on row windows instead of one
    #  LDS-resident workgroup per slice)
    with tlsq_amd.dev_switches(**({} if variant == "normwise" else dict(SLICE_MIDJ=0) if variant == "normwise_reg" else dict(SLICE_NORMWISE=0))):
        A, E, s, sv, rep = eng.rpca(D, return_report=True)
    with tlsq_amd.dev_switches(NO_SLICED_EIG=1):
        A1, E1, s1, sv1, rep1 = eng.rpca(D, return_report=True)
    assert sv == sv1 == r and rep.iters_done == rep1.iters_done
    assert np.array_equal(A, A1) and np.array_equal(E, E1)        # the loop itself does not see the slicer
    S, S1 = np.asarray(s.S), np.asarray(s1.S)
    assert np.all(np.diff(S) <= 0) and np.all(S > 0)
    np.testing.assert_allclose(S, S1, rtol=1e-10, atol=64 * 2.2e-16 * np.sqrt(N) * S1[0])
    U, Vt = np.asarray(s.U), np.asarray(s.Vt)
    assert np.max(np.abs(Vt @ Vt.T - np.eye(N))) < 1e-11
    assert np.max(np.abs(U.T @ U - np.eye(N))) < 1e-9
    # U diag(S) Vt of both decompositions is the same matrix (the last Z)
    Z, Z1 = (U * S) @ Vt, (np.asarray(s1.U) * S1) @ np.asarray(s1.Vt)
    assert np.linalg.norm(Z - Z1) <= 1e-11 * np.linalg.norm(Z1)
    # the slicer really ran: the plain route needs 12-14 sweeps over all pairs on these flat spectra
    assert 0 < rep.jacobi_sweeps < rep1.jacobi_sweeps


def test_rtls_through_the_slicer(eng):
    """tls!(s, n) reads the trailing right singular vectors (src/TotalLeastSquares.jl:65-69): the part of `s` the slicer changes
    the computation of.  256 columns so that the slicer's shape rule applies."""
    import tlsq_amd
    rng = np.random.default_rng(7)
    M, n, q = 8000, 250, 6
    Amat = rng.standard_normal((M, n))
    x0 = rng.standard_normal((n, q))
    y = Amat @ x0 + 1e-3 * rng.standard_normal((M, q))
    x = eng.rtls(Amat, y)
    with tlsq_amd.dev_switches(NO_SLICED_EIG=1):
        x1 = eng.rtls(Amat, y)
    assert np.linalg.norm(x - x1) <= 1e-9 * np.linalg.norm(x1)
    assert np.all(np.isfinite(x))


# ---- lowrankfilter's plain truncation branch on the series itself (csrc/hankelop.hip, round 6) -------------------------------
@pytest.mark.parametrize("Ns,n,sv,dtype", [(4000, 40, 2, np.float64), (30001, 256, 6, np.float64), (20000, 200, 32, np.float64),
                                           (9000, 1000, 3, np.float64), (12000, 64, 4, np.float32)])
def test_lowrankfilter_truncation_without_the_hankel_panel(eng, Ns, n, sv, dtype):
    """`lowrankfilter(y, n; sv = k)` (src/robustPCA.jl:123-126: `s = svd(H); A = U[:, 1:sv] S V'`, then unhankel): the Gram
    matrix of H from lagged autocorrelation sums and H V as FIR filters - H is never stored - against the panel form of the same
    call (HANKEL_STRUCT=0) to 1e-12 and against the CPU oracle to 1e-8 (fp32 series: the element type's accuracy)."""
    import tlsq_amd
    from oracle import rpca_oracle as O
    y, noise = O.synth_series(Ns, seed=Ns)
    x = (y + noise).astype(dtype)
    yf = eng.lowrankfilter(x, n, sv=sv)
    with tlsq_amd.dev_switches(HANKEL_STRUCT=0):
        yp = eng.lowrankfilter(x, n, sv=sv)
    assert yf.dtype == dtype and yf.shape == x.shape
    tol = 1e-12 if dtype == np.float64 else 2e-5
    assert np.linalg.norm(yf - yp) <= tol * np.linalg.norm(yp)
    if Ns <= 12000:
        yo = O.lowrankfilter(x.astype(np.float64), n, sv=sv)
        assert np.linalg.norm(yf - yo) <= (1e-8 if dtype == np.float64 else 1e-4) * np.linalg.norm(yo)


def test_truncation_at_config3_size_holds_no_panel(eng):
    """N = 1e7, n = 256 (BASELINE config 3's series) with sv = 4: the two 20 GB panels of the materialised form are never
    allocated - a fresh handle ends up with the series, the 1e7 x 4 factor and small matrices - and the filter does what the
    reference's test asks of it (test/runtests.jl:356-381)."""
    import torch
    import tlsq_amd
    from oracle import rpca_oracle as O
    Ns, n = 10_000_000, 256
    y, noise = O.synth_series(Ns, seed=0)
    free0, _ = torch.cuda.mem_get_info(0)
    e2 = tlsq_amd.Engine(0)
    try:
        yf = e2.lowrankfilter(y + 0.1 * noise, n, sv=2)
        free1, _ = torch.cuda.mem_get_info(0)
    finally:
        e2.close()
    assert (free0 - free1) < 2.5e9, f"{(free0 - free1) / 1e9:.2f} GB resident"       # (a K x n panel is 20.5 GB)
    qn = lambda v: v / np.quantile(np.abs(v), 0.9)
    assert np.mean((y - qn(yf)) ** 2) / np.mean((0.1 * noise) ** 2) < 0.05


# ---- the factor form on its own: eig_full's Cholesky route through the kernel-level entry -------------------------------------
def _spd(kind, N, rng):
    Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
    if kind == "flat":            # a flat bulk: what the returned `s` leaves after the deflation
        lam = rng.uniform(1.0, 2.0, N)
    elif kind == "outliers":      # a few large eigenvalues above a flat bulk (an undeflated Gram matrix)
        lam = np.concatenate([rng.uniform(1e4, 1e6, 12), rng.uniform(1.0, 1.5, N - 12)])
    elif kind == "graded":        # five decades, evenly in the logarithm
        lam = np.geomspace(1.0, 1e-5, N)
    elif kind == "clusters":      # eight tight clusters
        lam = np.repeat(np.arange(1.0, 9.0), N // 8) * (1.0 + 1e-9 * rng.standard_normal(N))
    else:                         # two eigenvalues 1e-7 apart right where the first cut goes (the mean)
        lam = np.linspace(1.0, 2.0, N)
        lam[N // 2] = lam.mean() * (1 - 5e-8)
        lam[N // 2 + 1] = lam.mean() * (1 + 5e-8)
    return (Q * lam) @ Q.T, np.sort(lam)[::-1]


@pytest.mark.parametrize("kind", ["flat", "outliers", "graded", "clusters", "knife"])
@pytest.mark.parametrize("N", [256, 384, 512, 1024])
def test_symeig_through_the_slicer_on_synthetic_spectra(eng, N, kind):
    """tlsq_k_symeig_chol_f64 = eig_full's Cholesky route: with the slicer in front (default) and without (NO_SLICED_EIG=1) - the
    same eigenvalues (N eps lambda_max, the accuracy of the Cholesky factor), orthonormal vectors, small residuals; whatever the
    slicer declines (too many equal eigenvalues for a slice) must come out of the plain route unchanged."""
    import ctypes as C
    import torch
    import tlsq_amd
    rng = np.random.default_rng(N + len(kind))
    G, ref = _spd(kind, N, rng)
    G = (G + G.T) / 2
    dG = torch.from_numpy(np.ascontiguousarray(G.T)).cuda()
    out = {}
    # (`normwise`, SLICE_NORMWISE=force: the form the returned `s` of rpca takes - slices of G itself, the slices' eigenproblems one
    #  LDS-resident workgroup each, one refinement, certified to 8 N eps ||G||_F; the entry reports a negative sweep count when
    #  that form served the call - it may decline, e.g. clusters wider than a slice, and leave the call to the other two)
    for tag, sw in (("sliced", {}), ("plain", dict(NO_SLICED_EIG=1)), ("normwise", dict(SLICE_NORMWISE="force"))):
        dl = torch.zeros(N, dtype=torch.float64, device="cuda")
        dV = torch.zeros((N, N), dtype=torch.float64, device="cuda")
        sweeps = C.c_int64()
        with tlsq_amd.dev_switches(**sw):
            st = eng.lib.tlsq_k_symeig_chol_f64(eng.h, C.c_void_p(dG.data_ptr()), N, N, C.c_void_p(dl.data_ptr()),
                                                C.c_void_p(dV.data_ptr()), N, C.byref(sweeps))
        assert st == 0, eng.lib.tlsq_last_error(eng.h)
        out[tag] = (dl.cpu().numpy(), dV.cpu().numpy().T, sweeps.value)
    for tag in ("sliced", "plain", "normwise"):
        lam, V, sw = out[tag]
        assert np.all(np.diff(lam) <= 0)
        bar = 32 * N * 2.2e-16 * (np.linalg.norm(G) if (tag == "normwise" and sw < 0) else ref[0])
        assert np.max(np.abs(lam - ref)) < bar, (tag, np.max(np.abs(lam - ref)) / ref[0])
        assert np.max(np.abs(V.T @ V - np.eye(N))) < 1e-10, tag
        assert np.linalg.norm(G @ V - V * lam[None, :]) < 1e-11 * np.linalg.norm(G) * np.sqrt(N), tag
    if kind == "flat":
        assert out["normwise"][2] < 0     # a flat spectrum the normwise form must serve itself (outliers need the caller's hints)
    assert np.max(np.abs(out["sliced"][0] - out["plain"][0])) < 32 * N * 2.2e-16 * ref[0]
