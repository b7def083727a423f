"""GPU tests of the TSQR route (tsqr.hip + one-sided Jacobi on R'): the north star's "LDS-staged Householder panel
reduction", i.e. the accurate stand-in for LAPACK's gesdd at src/robustPCA.jl:194 (reference paths relative to
/root/reference).  Everything goes through the C ABI (tlsq_k_tsqr_f64, tlsq_k_svd_r_f64, tlsq_tls_f64, tlsq_rpca_f64)."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg as sla

pytestmark = pytest.mark.gpu

EPS = 2.220446049250313e-16


@pytest.fixture(scope="module")
def torch_mod():
    import torch   # torch first: see tests/test_gpu_parity.py
    assert torch.cuda.is_available()
    torch.zeros(1, device="cuda")
    return torch


@pytest.fixture(scope="module")
def eng(torch_mod):
    import tlsq_amd
    e = tlsq_amd.Engine(0)
    yield e
    e.close()


def to_dev(torch, a):
    a = np.asarray(a)
    if a.ndim == 1:
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return torch.from_numpy(np.ascontiguousarray(a.T)).cuda()


def to_host(t):
    a = t.cpu().numpy()
    return a if a.ndim == 1 else a.T


def dptr(t):
    return C.c_void_p(t.data_ptr())


def graded(rng, M, N, lo):
    """M x N matrix with singular values geomspace(1, lo, N) exactly known"""
    U, _ = np.linalg.qr(rng.standard_normal((M, N)))
    V, _ = np.linalg.qr(rng.standard_normal((N, N)))
    s = np.geomspace(1.0, lo, N)
    return (U * s) @ V.T, s, V


# shapes: one block / several level-0 blocks / three tree levels / ragged last block / N not a multiple of 32 / N = M
SHAPES = [(8, 3), (40, 40), (256, 32), (257, 33), (300, 64), (1500, 130), (5000, 96), (20000, 512), (33000, 100),
          (700, 513), (130, 130), (4096, 1), (65, 64)]


@pytest.mark.parametrize("M,N", SHAPES)
def test_tsqr_r_factor(eng, torch_mod, M, N):
    """R'R = Z'Z to eps, R upper triangular, |R| equal to LAPACK's R (unique up to row signs for full rank)."""
    torch = torch_mod
    rng = np.random.default_rng(M * 1000 + N)
    Z = rng.standard_normal((M, N)) * np.geomspace(1.0, 1e-3, N)[None, :]
    dZ = to_dev(torch, Z)
    dR = torch.full((N, N), np.nan, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    assert eng.lib.tlsq_k_tsqr_f64(eng.h, dptr(dZ), M, N, M, dptr(dR), N) == 0, eng.lib.tlsq_last_error(eng.h)
    R = to_host(dR)
    assert np.all(np.isfinite(R))
    assert np.all(np.tril(R, -1) == 0.0)
    Rref = np.linalg.qr(Z, mode="r")
    scale = np.linalg.norm(Z)
    assert np.max(np.abs(np.abs(R) - np.abs(Rref))) < 64 * EPS * scale * np.sqrt(N)
    # column norms are preserved exactly by orthogonal transformations: a check that is independent of LAPACK
    assert np.max(np.abs(np.linalg.norm(R, axis=0) - np.linalg.norm(Z, axis=0)) / np.linalg.norm(Z, axis=0)) < 64 * EPS * np.sqrt(M)


@pytest.mark.parametrize("M,N,lo", [(400, 40, 1e-12), (3000, 130, 1e-10), (20000, 512, 1e-9), (600, 64, 1e-14), (50, 17, 1e-8)])
def test_svd_via_r_small_singular_values(eng, torch_mod, M, N, lo):
    """Every singular value to a few eps * sigma_max (the Gram route loses everything below ~1e-8 sigma_max), right
    singular vectors orthonormal with residual ||Z v - sigma u|| at the same level."""
    torch = torch_mod
    rng = np.random.default_rng(N)
    Z, s_true, _ = graded(rng, M, N, lo)
    dZ = to_dev(torch, Z)
    dS = torch.zeros(N, dtype=torch.float64, device="cuda")
    dV = torch.zeros((N, N), dtype=torch.float64, device="cuda")
    sw = C.c_int64()
    torch.cuda.synchronize()
    assert eng.lib.tlsq_k_svd_r_f64(eng.h, dptr(dZ), M, N, M, dptr(dS), dptr(dV), N, C.byref(sw)) == 0, \
        eng.lib.tlsq_last_error(eng.h)
    S, V = to_host(dS), to_host(dV)
    ref = sla.svdvals(Z)                       # LAPACK gesdd, the driver behind the reference's svd!
    assert np.all(np.diff(S) <= 0)
    tol = 32 * EPS * np.sqrt(N) * ref[0]
    assert np.max(np.abs(S - ref)) < tol, (np.max(np.abs(S - ref)), tol)
    assert np.max(np.abs(S - s_true)) < 4 * tol
    assert np.max(np.abs(V.T @ V - np.eye(N))) < 64 * EPS * N
    # Z V = U S with orthonormal U: column norms of Z V reproduce S
    ZV = Z @ V
    assert np.max(np.abs(np.linalg.norm(ZV, axis=0) - S)) < 4 * tol
    assert sw.value <= 30


def test_tls_ill_conditioned_vs_lapack(eng):
    """tls! takes the vectors of the SMALLEST singular values (src/TotalLeastSquares.jl:66-68).  With
    nearly collinear [A y] (cond ~ 1e9) the Gram route returns a wrong x; the TSQR route matches LAPACK."""
    from oracle import rpca_oracle as O
    rng = np.random.default_rng(3)
    M, n = 2000, 6
    B = rng.standard_normal((M, 3))
    A = np.hstack([B, B @ rng.standard_normal((3, 3)) + 1e-7 * rng.standard_normal((M, 3))])   # cond(A) ~ 1e8
    x0 = rng.standard_normal(n)
    y = A @ x0 + 1e-9 * rng.standard_normal(M)
    Ay = np.hstack([A, y[:, None]])
    assert np.linalg.cond(Ay) > 1e8
    x = eng.tls(A, y)
    xo = O.tls(A, y)
    assert np.linalg.norm(x.ravel() - xo.ravel()) / np.linalg.norm(xo) < 1e-6
    # and rtls on the same data returns the oracle's answer too (the returned `s` comes from the TSQR route)
    xr = eng.rtls(A, y)
    xro = O.rtls(A, y)
    assert np.linalg.norm(xr.ravel() - xro.ravel()) / np.linalg.norm(xro) < 1e-6


def test_rpca_returned_s_all_singular_values(eng):
    """s.S of the returned SVD (src/robustPCA.jl:238) down to the smallest value, rtol 1e-10 of each
    value above the fp64 noise floor, and U S Vt = the oracle's last Z."""
    from oracle import rpca_oracle as O
    D, _, _ = O.synth_lowrank_sparse(800, 96, 6, seed=4)
    A, E, s, sv, rep = eng.rpca(D, return_report=True)
    Ao, Eo, so, svo, io = O.rpca(D)
    assert rep.iters_done == io.iters_done and rep.svp_hist == io.svp_hist and sv == svo
    Uo, So, Vto = so
    S = np.asarray(s.S)
    floor = 64 * EPS * So[0] * np.sqrt(96)
    big = So > 1e3 * floor
    assert np.max(np.abs(S[big] - So[big]) / So[big]) < 1e-10
    assert np.max(np.abs(S - So)) < floor * 16
    Zo = (Uo * So) @ Vto
    Zg = (np.asarray(s.U) * S) @ np.asarray(s.Vt)
    assert np.linalg.norm(Zg - Zo) / np.linalg.norm(Zo) < 1e-9


@pytest.mark.parametrize("M,N,p", [(4096, 512, 20), (8192, 1024, 74), (5000, 700, 96), (3001, 257, 130), (65536, 4096, 74),
                                   (32768, 2304, 40), (16384, 4096, 20), (32768, 2304, 58), (16384, 4096, 9)])
def test_operator_product_on_the_fp32_mfma(eng, torch_mod, M, N, p):
    """Y = Z'(Z X) for an fp32 panel (gemm.hip, op_gram_f32: both halves on the fp32 MFMA, fp64 fold-in) against numpy in
    float64 on the same fp32 data: per column 2e-6 of ||Z||_2^2 ||x|| (X rounded to fp32 on the way in, fp32 partial sums
    over at most 64 columns / 32 rows) - the randomized hook's sketch in large mode is made of these products."""
    torch = torch_mod
    rng = np.random.default_rng(M + N + p)
    Z = (rng.standard_normal((M, 8)) @ rng.standard_normal((8, N)) + 0.3 * rng.standard_normal((M, N))).astype(np.float32)
    X = rng.standard_normal((N, p))
    dZ = torch.from_numpy(np.ascontiguousarray(Z.T)).cuda()
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    dY = torch.full((p, N), float("nan"), dtype=torch.float64, device="cuda")
    assert eng.lib.tlsq_k_op_gram_f32(eng.h, dZ.data_ptr(), M, N, M, dX.data_ptr(), p, dY.data_ptr()) == 0, \
        eng.lib.tlsq_last_error(eng.h)
    eng.synchronize()
    Y = dY.cpu().numpy().T
    Zd = Z.astype(np.float64)
    ref = Zd.T @ (Zd @ X)
    scale = np.linalg.norm(Zd, 2) ** 2 * np.linalg.norm(X, axis=0)
    assert np.isfinite(Y).all()
    assert (np.linalg.norm(Y - ref, axis=0) / scale).max() < 2e-6
    if M * N >= 1 << 26:
        # third form (opgram16.hip): every operand split in registers into two fp16 numbers, products on the fp16 MFMA - what the
        # randomized hook runs when the sweep has left the panel's maximum; here forced, with a pass for the maximum.  22 bits
        # of every operand: a few 1e-9 of the same scale, and the same T32 = Z X panel to the fp32 rounding
        import tlsq_amd
        dY3 = torch.full((p, N), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        with tlsq_amd.dev_switches(OPGRAM_H3=2):
            assert eng.lib.tlsq_k_op_gram_f32(eng.h, dZ.data_ptr(), M, N, M, dX.data_ptr(), p, dY3.data_ptr()) == 0, \
                eng.lib.tlsq_last_error(eng.h)
        eng.synchronize()
        Y3 = dY3.cpu().numpy().T
        assert np.isfinite(Y3).all()
        assert (np.linalg.norm(Y3 - ref, axis=0) / scale).max() < 1e-7
