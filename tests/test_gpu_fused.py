"""GPU tests of the fused sweep + Gram kernel (csrc/fused.hip, `tlsq_k_zsweep_gram_f64`) - the E-free sweep of
src/robustPCA.jl:188-192 / :217-223 and the Gram matrix of the Z it writes (the gesdd work of :194) in one launch.

Parity bar: Y_{k+1}, Z_{k+1}, R_k bit-identical to k_zsweep (which the oracle pins bit for bit in test_gpu_parity.py) and to
the reference's statements evaluated in numpy; the Gram matrix within 2e-13 of the Gram kernel's on what was written
(summation order); a whole lowrankfilter call the same with the kernel as without it."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    torch.zeros(1, device="cuda")
    return torch


@pytest.fixture(scope="module")
def eng(torch_mod):
    import tlsq_amd
    e = tlsq_amd.Engine(0)
    yield e
    e.close()


def dptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def soft_th(x, e):                      # src/robustPCA.jl:1
    return np.maximum(x - e, 0) + np.minimum(x + e, 0)


@pytest.mark.parametrize("M,r,hankel,nonneg,with_r,N", [
    (40_000, 16, False, 0, True, 256),       # whole stages only
    (40_006, 8, False, 0, False, 256),       # a partial last stage, no residual store
    (33_334, 3, True, 0, True, 256),         # implicit Hankel D with zero pad rows at the end (hankel_K < M)
    (40_000, 0, True, 0, False, 256),        # rank 0: A = 0
    (36_010, 5, False, 1, True, 256),        # nonnegA / nonnegE compiled in
    (36_010, 16, True, 1, True, 256),
    (30_000, 16, False, 0, True, 512),       # N = 512: two diagonal blocks fused + the off-diagonal block from the stored Z
    (20_010, 7, False, 1, False, 512),
    (16_390, 4, True, 0, True, 512),
])
def test_fused_sweep_gram_kernel(eng, torch_mod, M, r, hankel, nonneg, with_r, N):
    """One launch against (a) k_zsweep + the Gram kernel on the same buffers: bit-identical panels, G to 1e-13; (b) the
    reference's statements in numpy started from a consistent state: to the rounding of Z."""
    import tlsq_amd
    torch = torch_mod
    rng = np.random.default_rng(M + 7 * r)
    K = M - 6 if hankel else M
    if hankel:
        y = rng.standard_normal(K + N - 1)
        D = np.zeros((M, N))
        idx = np.arange(K)[:, None] + np.arange(N)[None, :]
        D[:K] = y[idx]
    else:
        y = None
        D = rng.standard_normal((M, N))
    Y = rng.standard_normal((M, N))
    Ek = rng.standard_normal((M, N)) * (rng.random((M, N)) < 0.2)
    if hankel:
        Y[K:] = 0.0
        Ek[K:] = 0.0
    Tm = rng.standard_normal((M, max(r, 1)))[:, :r]
    Vs = rng.standard_normal((N, max(r, 1)))[:, :r] / max(np.sqrt(r), 1.0)
    if hankel and r:
        Tm[K:] = 0.0
    mu, mu_n, lam = 0.27, 0.405, 0.1
    inv_mu, inv_mu_n, thr_n = 1.0 / mu, 1.0 / mu_n, lam / mu_n
    Zk = (D - Ek) + inv_mu * Y          # :192 of the previous iteration
    A = Tm @ Vs.T if r else np.zeros((M, N))
    if nonneg:
        A = np.maximum(A, 0)
    Rref = (D - A) - Ek                 # :221
    Y2ref = Y + mu * Rref               # :222
    t = inv_mu_n * Y2ref
    En = soft_th((D - A) + t, thr_n)    # :188
    if nonneg:
        En = np.maximum(En, 0)
    Znref = (D - En) + t                # :192
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
    dD, dY, dZ = dev(D), dev(Y), dev(Zk)
    dT = dev(Tm if r else np.zeros((M, 1)))
    dV = dev(Vs if r else np.zeros((N, 1)))
    dy = torch.from_numpy(y).cuda() if hankel else None
    Y1, Z1, R1 = torch.empty_like(dY), dZ.clone(), torch.empty_like(dY)
    G1 = torch.empty((N, N), dtype=torch.float64, device="cuda")
    Y2, Z2, R2 = torch.empty_like(dY), torch.empty_like(dY), torch.full_like(dY, 7.0)
    G2 = torch.empty((N, N), dtype=torch.float64, device="cuda")
    ss1 = torch.zeros(72, dtype=torch.float64, device="cuda")
    ss2 = torch.zeros(72, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    lib, h = eng.lib, eng.h
    assert lib.tlsq_k_zsweep_f64(h, dptr(dD), dptr(dT), dptr(dV), None, dptr(dY), dptr(Y1), dptr(Z1), dptr(R1), M, N, r, mu, inv_mu,
                                 nonneg, inv_mu_n, thr_n, nonneg, dptr(ss1)) == 0
    assert lib.tlsq_k_gram_f64(h, dptr(Z1), M, N, M, dptr(G1), N) == 0
    with tlsq_amd.dev_switches(FUSED_ZGRAM_MINROWS=1000):
        st = lib.tlsq_k_zsweep_gram_f64(h, None if hankel else dptr(dD), dptr(dT), dptr(dV), dptr(dY), dptr(Y2), dptr(dZ), dptr(Z2),
                                        dptr(R2) if with_r else None, M, N, r, mu, inv_mu, nonneg, inv_mu_n, thr_n, nonneg,
                                        dptr(ss2), dptr(dy), K if hankel else 0, dptr(G2), N)
    assert st == 0, eng.lib.tlsq_last_error(h)
    eng.synchronize()
    assert torch.equal(Y1, Y2) and torch.equal(Z1, Z2)
    if with_r:
        assert torch.equal(R1, R2)
    else:
        assert bool((R2 == 7.0).all())                       # nothing stored
    Gt = Z2 @ Z2.T                                           # (N x M) @ (M x N): Z' Z of the column-major panel
    scale = Gt.abs().max().item()
    assert (G2 - Gt).abs().max().item() <= 1e-12 * scale     # (torch's own product: rocBLAS summation order)
    assert (G2 - G1).abs().max().item() <= 2e-13 * scale     # (the library's Gram kernel: fixed-order split-K sums, like ours)
    assert (G2 - G2.T).abs().max().item() == 0.0
    s1, s2 = ss1[:64].sum().item(), ss2[:64].sum().item()
    assert abs(s1 - s2) <= 1e-11 * abs(s1)
    assert ss2[64:65].view(torch.int64).view(torch.float64).item() == R1.abs().max().item()
    # the reference's statements
    zs = max(1.0, np.abs(Zk).max())
    for got, ref in ((Y2, Y2ref), (Z2, Znref)) + (((R2, Rref),) if with_r else ()):
        np.testing.assert_allclose(got.cpu().numpy().T, ref, rtol=0, atol=2e-13 * zs * max(1.0, np.abs(ref).max()))


def test_fused_kernel_declines_what_it_does_not_serve(eng, torch_mod):
    """Other widths, odd M, ranks above 16, short panels: TLSQ_ERR_UNSUPPORTED (rpca then runs the two kernels), never a
    wrong answer."""
    torch = torch_mod
    lib, h = eng.lib, eng.h
    buf = torch.zeros(1 << 20, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    call = lambda M, N, r, thr=0.1: lib.tlsq_k_zsweep_gram_f64(h, dptr(buf), dptr(buf), dptr(buf), dptr(buf), dptr(buf[8:]), dptr(buf),
                                                               dptr(buf), None, M, N, r, 0.3, 1 / 0.3, 0, 2.0, thr, 0, None, None, 0,
                                                               dptr(buf), N)
    unsupported = -5
    import tlsq_amd
    from tlsq_amd import _lib as L
    unsupported = L.TLSQ_ERR_UNSUPPORTED
    assert call(1000, 256, 4) == unsupported          # short panel
    with tlsq_amd.dev_switches(FUSED_ZGRAM_MINROWS=16):
        assert call(1001, 256, 4) == unsupported      # odd M
        assert call(1000, 384, 4) == unsupported      # other width
        assert call(1000, 256, 17) == unsupported     # rank
        assert call(1000, 256, 4, thr=-1.0) == unsupported


def test_lowrankfilter_with_and_without_the_fused_kernel(torch_mod):
    """A lowrankfilter call whose panels are tall enough for the fused kernel (implicit Hankel D, n = 256): same iteration
    count and the same filtered series (1e-10) as the run that keeps sweep and Gram in separate kernels; the reference's
    own acceptance threshold (test/runtests.jl:356-381) on top."""
    import tlsq_amd
    from oracle import rpca_oracle as O
    Ns, n = 450_000, 256
    y, noise = O.synth_series(Ns, seed=5)
    qn = lambda x: x / np.quantile(np.abs(x), 0.9)
    out = {}
    for tag, sw in (("fused", {}), ("split", {"NO_FUSED_ZGRAM": 1})):
        with tlsq_amd.dev_switches(**sw):
            e = tlsq_amd.Engine(0)
            try:
                out[tag] = e.lowrankfilter(y + noise, n, return_report=True, cost_history=True)
            finally:
                e.close()
    (yf, rep), (yf0, rep0) = out["fused"], out["split"]
    assert rep.converged and rep.iters_done == rep0.iters_done and rep.svp_hist == rep0.svp_hist
    assert np.linalg.norm(yf - yf0) <= 1e-10 * np.linalg.norm(yf0)
    np.testing.assert_allclose(rep.cost_hist, rep0.cost_hist, rtol=1e-8)
    assert np.mean((y - qn(yf)) ** 2) / np.mean(noise ** 2) < 0.001


def test_tall_rpca_512_columns_with_and_without_the_fused_kernel(torch_mod):
    """rpca on a 70000 x 512 panel (BASELINE config 4's shape at a third of its height): the sweep covers the diagonal
    256-column blocks of the Gram matrix, the off-diagonal block comes from the stored panel.  Same iterations, rank trajectory,
    A and E (1e-10) as the run with separate kernels, the same bits from two runs, and the planted low-rank part recovered."""
    import tlsq_amd
    from tlsq_amd import workloads as W
    M, N, r = 70_000, 512, 16
    D, A0, _ = W.synth_lowrank_sparse(M, N, r, seed=3)
    out = {}
    for tag, sw in (("fused", {}), ("again", {}), ("split", {"NO_FUSED_ZGRAM": 1})):
        with tlsq_amd.dev_switches(**sw):
            e = tlsq_amd.Engine(0)
            try:
                out[tag] = e.rpca(D, return_report=True, want_s=False, cost_history=False)
            finally:
                e.close()
    (A, E, _, sv, rep), (A2, E2, _, _, _), (As, Es, _, svs, reps) = out["fused"], out["again"], out["split"]
    assert rep.converged and rep.iters_done == reps.iters_done and rep.svp_hist == reps.svp_hist and sv == svs == r
    assert np.array_equal(A, A2) and np.array_equal(E, E2)
    assert np.linalg.norm(A - As) <= 1e-10 * np.linalg.norm(As) and np.linalg.norm(E - Es) <= 1e-10 * np.linalg.norm(Es)
    assert np.linalg.norm(A - A0) <= 1e-6 * np.linalg.norm(A0)


@pytest.mark.parametrize("M,N,r,nonneg,with_r,inplace", [
    (32768, 2304, 40, 0, True, True),        # r <= 64: 32 k-steps of the 32 x 32 x 2 MFMA, zero columns behind the rank
    (32768, 2304, 64, 1, False, False),      # nonnegA / nonnegE, Z double-buffered
    (16384, 4096, 77, 0, False, True),       # 64 < r <= 80: 40 k-steps
])
def test_wide_sweep_kernel_fp32(eng, torch_mod, M, N, r, nonneg, with_r, inplace):
    """The E-free sweep for ranks above 32 on an fp32 panel (`tlsq_k_zsweep_wide_f32`: T = Z Vg on the fp32 MFMA, A_k = T Vs'
    formed inside the sweep) against the reference's statements (src/robustPCA.jl:205-213, :217-222, :188-192) evaluated in
    float64 on the same fp32 inputs: a few fp32 roundings of the panel's scale (A_k itself is an fp32 quantity in the
    reference), the residual norm to 1e-5, max |R| to 1e-3, the widened factor T to the fp32 rounding of Z Vg."""
    import tlsq_amd
    torch = torch_mod
    g = torch.Generator(device="cuda").manual_seed(M + N + r)
    f32 = dict(device="cuda", dtype=torch.float32, generator=g)
    L = torch.randn(N, 24, **f32) @ torch.randn(24, M, **f32) / 5.0          # (N x M row-major = M x N column-major)
    D = L + 3.0 * torch.randn(N, M, **f32) * (torch.rand(N, M, **f32) < 0.05)
    Y = torch.randn(N, M, **f32) * 0.3
    Ek = 3.0 * torch.randn(N, M, **f32) * (torch.rand(N, M, **f32) < 0.05)
    mu, mu_n, lam = 0.27, 0.405, 0.1
    inv_mu = float(np.float32(1.0) / np.float32(mu))
    inv_mu_n, thr_n = float(np.float32(1.0 / mu_n)), float(np.float32(lam / mu_n))
    Zk = ((D - Ek) + np.float32(inv_mu) * Y).contiguous()                      # :192 of the previous iteration
    V = torch.linalg.qr(torch.randn(N, r, device="cuda", dtype=torch.float64, generator=g))[0]   # N x r, orthonormal columns
    gsc = torch.rand(r, device="cuda", dtype=torch.float64, generator=g) * 0.9
    Vs_cm = V.T.contiguous()                                                    # column-major N x r
    Vg_cm = (V * gsc).T.contiguous()
    # float64 statements
    Z64, D64, Y64 = Zk.double(), D.double(), Y.double()
    A = ((Z64.T @ (V * gsc)) @ V.T).T                                           # A = (Z Vg) Vs'   (N x M view)
    if nonneg:
        A = A.clamp_min(0)
    w = Z64 - A
    res = w - inv_mu * Y64
    y1 = mu_f = float(np.float32(mu)) * w
    tt = inv_mu_n * y1
    x = (D64 - A) + tt
    ee = torch.clamp(x - thr_n, min=0) + torch.clamp(x + thr_n, max=0)
    if nonneg:
        ee = ee.clamp_min(0)
    zn = (D64 - ee) + tt
    Yo = torch.empty_like(Y)
    Zin = Zk.clone()
    Zo = Zin if inplace else torch.empty_like(Zk)
    R = torch.full_like(Y, 7.0)
    Tm = torch.empty((r, M), dtype=torch.float64, device="cuda")
    ss = torch.zeros(72, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    st = eng.lib.tlsq_k_zsweep_wide_f32(eng.h, dptr(D), dptr(Vg_cm), dptr(Vs_cm), r, dptr(Y), dptr(Yo), dptr(Zin), dptr(Zo),
                                        dptr(R) if with_r else None, dptr(Tm), M, N, mu, inv_mu, nonneg, inv_mu_n, thr_n, nonneg,
                                        dptr(ss))
    assert st == 0, eng.lib.tlsq_last_error(eng.h)
    eng.synchronize()
    scale = float(Z64.abs().max())
    eps = float(np.finfo(np.float32).eps)
    assert float((Yo.double() - y1).abs().max()) <= 32 * eps * scale
    assert float((Zo.double() - zn).abs().max()) <= 64 * eps * scale
    if with_r:
        assert float((R.double() - res).abs().max()) <= 32 * eps * scale
    else:
        assert bool((R == 7.0).all())
    T64 = (Z64.T @ (V * gsc)).T                                                 # r x M
    assert float((Tm - T64).abs().max()) <= 32 * eps * float(T64.abs().max())
    fro = float((res * res).sum())
    assert abs(float(ss[:64].sum()) - fro) <= 1e-5 * fro
    mx = ss[64:65].view(torch.int64).view(torch.float64).item()
    assert abs(mx - float(res.abs().max())) <= 1e-3 * float(res.abs().max())
    # shapes the kernel does not take
    assert eng.lib.tlsq_k_zsweep_wide_f32(eng.h, dptr(D), dptr(Vg_cm), dptr(Vs_cm), 20, dptr(Y), dptr(Yo), dptr(Zin), dptr(Zo), None,
                                          None, M, N, mu, inv_mu, 0, inv_mu_n, thr_n, 0, None) != 0


def test_large_fp32_rpca_with_and_without_the_wide_sweep(torch_mod):
    """rpca on a 32768 x 2304 fp32 panel of rank 40 (large mode: min(M, N) > 2048): the loop that forms A_k inside the sweep
    against the loop that stores it (NO_WIDE_SWEEP): same iterations and rank trajectory, A and E to 1e-4 (fp32 factors
    against fp64 ones), the planted low-rank part recovered to the fp32 bar, in the exact and the randomized mode."""
    import tlsq_amd
    from tlsq_amd import workloads as W
    M, N, r = 32768, 2304, 40
    D, A0, _ = W.synth_lowrank_sparse(M, N, r, seed=5)
    D = D.astype(np.float32)
    for kw in ({}, {"svd": "randomized"}):
        out = {}
        for tag, sw in (("wide", {}), ("stored", {"NO_WIDE_SWEEP": 1})):
            with tlsq_amd.dev_switches(**sw):
                e = tlsq_amd.Engine(0)
                try:
                    out[tag] = e.rpca(D, return_report=True, want_s=False, cost_history=False, **kw)
                finally:
                    e.close()
        (A, E, _, sv, rep), (As, Es, _, svs, reps) = out["wide"], out["stored"]
        assert rep.converged and reps.converged and sv == svs == r
        assert abs(rep.iters_done - reps.iters_done) <= (0 if not kw else 1)
        if not kw:
            assert rep.svp_hist == reps.svp_hist
        assert np.linalg.norm(A.astype(np.float64) - As) <= 1e-4 * np.linalg.norm(As)
        assert np.linalg.norm(E.astype(np.float64) - Es) <= 1e-3 * np.linalg.norm(Es)
        assert np.linalg.norm(A.astype(np.float64) - A0) <= 1e-3 * np.linalg.norm(A0)
