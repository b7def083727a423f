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


@pytest.mark.parametrize("variant", ["normwise", "factor"])
@pytest.mark.parametrize("shape", [(20000, 512, 16), (6000, 256, 8), (9000, 384, 24), (5000, 1024, 12)])
def test_returned_svd_with_and_without_the_slicer(eng, shape, variant):
    """variant `normwise`: the default of the returned `s` (slices of the deflated Gram matrix itself, certified to 8 N eps ||G||);
    `factor` (SLICE_NORMWISE=0): slices of the Cholesky factor's K = L'L, then Jacobi sweeps inside the slices and over all pairs."""
    import tlsq_amd
    from oracle import rpca_oracle as O
    M, N, r = shape
    D, _, _ = O.synth_lowrank_sparse(M, N, r, seed=M + N)
    with tlsq_amd.dev_switches(**({} if variant == "normwise" else dict(SLICE_NORMWISE=0))):
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
