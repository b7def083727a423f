"""The fused Rayleigh-Ritz kernel of warm subspace steps (subspace.hip, k_rr_small; the `gesdd` work of
src/robustPCA.jl:194 under /root/reference): from B = Y'Y and Hg = Y'GY it must deliver C with C'BC = I and C'HgC = diag(lam),
lam = the eigenvalues of the pencil (Hg, B) (LAPACK: scipy.linalg.eigh(Hg, B)), or decline (status[1] != 0)."""
import numpy as np
import pytest
import scipy.linalg as sla

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    import tlsq_amd
    eng = tlsq_amd.Engine(0)
    yield eng, torch
    eng.close()


def run_rr(eng, torch, B, Hg, nt=None, tau2=0.0):
    p = B.shape[0]
    nt = p if nt is None else nt
    dB = torch.from_numpy(np.asfortranarray(B).T.copy()).cuda()      # column-major p x p
    dH = torch.from_numpy(np.asfortranarray(Hg).T.copy()).cuda()
    dC = torch.full((p, p), float("nan"), dtype=torch.float64, device="cuda")
    dl = torch.full((p,), float("nan"), dtype=torch.float64, device="cuda")
    ds = torch.full((8,), float("nan"), dtype=torch.float64, device="cuda")
    st = eng.lib.tlsq_k_rr_small_f64(eng.h, dB.data_ptr(), dH.data_ptr(), p, nt, tau2, dC.data_ptr(), dl.data_ptr(), ds.data_ptr())
    assert st == 0
    eng.synchronize()
    print("rr_small p=%d status" % p, " ".join("%.3g" % v for v in ds.cpu().numpy()[:5]))
    return dC.cpu().numpy().T.copy(), dl.cpu().numpy(), ds.cpu().numpy()


def warm_block(rng, N, p, eps, spread=3.0, scale_cols=True):
    """a symmetric PSD G, and Y = (nearly) its leading eigenvectors, perturbed by eps and with uneven column norms"""
    Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
    lam = np.concatenate([np.logspace(0, -spread, p), 1e-5 * rng.random(N - p)])
    G = (Q * lam) @ Q.T
    Y = Q[:, :p] + eps * rng.standard_normal((N, p))
    if scale_cols:
        Y = Y * np.logspace(0, 6, p)[rng.permutation(p)]
    return G, Y


@pytest.mark.parametrize("p", [3, 12, 20, 27, 32])
@pytest.mark.parametrize("eps", [1e-9, 1e-5, 1e-3, 1e-2])
def test_rotation_diagonalises_the_pencil(ctx, p, eps):
    eng, torch = ctx
    rng = np.random.default_rng(100 * p + int(-np.log10(eps)))
    G, Y = warm_block(rng, 200, p, eps)
    B, Hg = Y.T @ Y, Y.T @ (G @ Y)
    C, lam, st = run_rr(eng, torch, B, Hg)
    if eps >= 1e-2 and p >= 20 and st[1] == 2:
        # far from a warm block (||Bh - I|| ~ 0.3 - 0.4 and Ritz values closer than their coupling): the kernel may give up -
        # what it returns is still a basis of the same span
        assert np.linalg.matrix_rank(C) == p
        return
    assert st[1] == 0, st
    ref = sla.eigh(0.5 * (Hg + Hg.T), B, eigvals_only=True)
    lmax = ref.max()
    assert np.abs(np.sort(lam) - ref).max() <= 2e-13 * lmax
    I = C.T @ B @ C
    Dg = C.T @ (0.5 * (Hg + Hg.T)) @ C
    assert np.abs(I - np.eye(p)).max() <= 1e-13
    assert np.abs(Dg - np.diag(lam)).max() <= 5e-14 * lmax
    # as the rotation of the step: X = Y C is orthonormal and its residuals ||G x - theta x|| are those of Rayleigh-Ritz on span(Y)
    X = Y @ C
    assert np.abs(X.T @ X - np.eye(p)).max() <= 1e-12
    Qo, _ = np.linalg.qr(Y)
    th, S = np.linalg.eigh(Qo.T @ G @ Qo)
    Xr = Qo @ S
    res_ref = np.linalg.norm(G @ Xr - Xr * th, axis=0)
    order = np.argsort(lam)
    res = np.linalg.norm(G @ X - X * lam, axis=0)[order]
    assert np.all(res <= res_ref + 1e-13 * lmax)


def test_pad_columns_are_decoupled_not_rotated(ctx):
    """The shape of a warm step of the ALM loop: nt wanted columns (converged Ritz vectors under G^3) and pad columns
    G x_pad that lean on the dominant directions (||Bh - I|| = O(1), pivots below 0.25).  Wanted pairs come out as Ritz
    pairs; the pad block is orthonormal, decoupled from them, and left unrotated - accepted because its Gershgorin bound
    stays below tau2, refused (status 3) when tau2 is lowered into the pad block."""
    eng, torch = ctx
    rng = np.random.default_rng(11)
    N, nt, P = 300, 16, 4
    p = nt + P
    Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
    lamv = np.concatenate([np.logspace(0, -1, nt), 1e-6 * (1 + rng.random(N - nt))])     # a flat bulk behind the signal
    G = (Q * lamv) @ Q.T
    X = Q[:, :p] + 1e-7 * rng.standard_normal((N, p))
    X, _ = np.linalg.qr(X)
    Y = np.concatenate([np.linalg.matrix_power(G, 3) @ X[:, :nt], G @ X[:, nt:]], axis=1)
    B, Hg = Y.T @ Y, Y.T @ (G @ Y)
    d = 1 / np.sqrt(np.diag(B))
    assert np.linalg.norm(d[:, None] * B * d[None, :] - np.eye(p)) > 0.5                 # not close to orthonormal
    C, lam, st = run_rr(eng, torch, B, Hg, nt=nt, tau2=1e-3)
    assert st[1] == 0, st
    Xn = Y @ C
    assert np.abs(Xn.T @ Xn - np.eye(p)).max() <= 1e-12
    S = Xn.T @ G @ Xn
    assert np.abs(S[:nt, :nt] - np.diag(lam[:nt])).max() <= 1e-13                        # wanted: diagonal
    assert np.abs(S[:nt, nt:]).max() <= 1e-13                                            # decoupled from the pads
    assert np.abs(np.sort(lam[:nt]) - np.sort(lamv[:nt])).max() <= 1e-12
    assert np.allclose(np.diag(S)[nt:], lam[nt:], rtol=1e-12)                            # pads: Rayleigh quotients
    C2, lam2, st2 = run_rr(eng, torch, B, Hg, nt=nt, tau2=1.5e-6)                        # the threshold reaches the pads
    assert st2[1] == 3


def test_declines_what_it_is_not_made_for(ctx):
    eng, torch = ctx
    rng = np.random.default_rng(7)
    # (a) columns far from orthogonal
    Y = rng.standard_normal((100, 16))
    Y[:, 1] = Y[:, 0] + 1e-3 * Y[:, 1]
    G = np.diag(np.linspace(1, 2, 100))
    C, lam, st = run_rr(eng, torch, Y.T @ Y, Y.T @ G @ Y)
    assert st[1] == 1
    assert np.allclose(C, np.diag(1 / np.linalg.norm(Y, axis=0)))           # the normalised columns: same span
    # (b) a pair of nearly equal Ritz values with a coupling of the size of their gap: the refinement cannot rotate inside such
    # a pair (the Jacobi solver can) - it gives up early, and what it returns is still a B-orthonormal basis of the same span
    Q, _ = np.linalg.qr(rng.standard_normal((100, 100)))
    lamv = np.concatenate([[3.0, 2.0, 2.0 - 1e-7, 1.0], 1e-3 * rng.random(96)])
    G = (Q * lamv) @ Q.T
    th = 0.3
    R = np.eye(4)
    R[1:3, 1:3] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
    Pm = np.eye(100)
    Pm[:4, :4] = R
    Y = (Q @ Pm)[:, :4] + 1e-5 * rng.standard_normal((100, 4))             # mixes the two cluster vectors by a large angle
    C, lam, st = run_rr(eng, torch, Y.T @ Y, Y.T @ G @ Y)
    assert st[1] in (0, 2)
    X = Y @ C
    assert np.abs(X.T @ X - np.eye(4)).max() <= 1e-9
    if st[1] == 0:
        assert np.linalg.norm(G @ X - X * lam, axis=0).max() <= 1e-8
    # (c) not finite
    Bn = np.eye(5)
    Bn[2, 2] = np.nan
    C, lam, st = run_rr(eng, torch, Bn, np.eye(5))
    assert st[1] == 1 and np.isfinite(C).all()


def test_exact_multiple_eigenvalues_are_fine(ctx):
    """a true multiple eigenvalue has no internal coupling: any orthonormal basis of its eigenspace is a Ritz basis"""
    eng, torch = ctx
    rng = np.random.default_rng(3)
    Q, _ = np.linalg.qr(rng.standard_normal((80, 80)))
    lamv = np.concatenate([[5.0, 2.0, 2.0, 2.0, 1.0, 0.5], np.zeros(74)])
    G = (Q * lamv) @ Q.T
    Rm, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    Y = Q[:, :6].copy()
    Y[:, 1:4] = Y[:, 1:4] @ Rm                                            # any basis of the triple eigenspace
    Y = Y * np.array([1.0, 3.0, 0.1, 7.0, 2.0, 1.0])
    C, lam, st = run_rr(eng, torch, Y.T @ Y, Y.T @ G @ Y)
    assert st[1] == 0, st
    X = Y @ C
    assert np.linalg.norm(G @ X - X * lam, axis=0).max() <= 1e-13 * 5
    assert np.abs(X.T @ X - np.eye(6)).max() <= 1e-12
