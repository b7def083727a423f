"""Single-process multi-GPU handle (tlsq_create_multi): the path a Julia session takes to more than
one GPU.  The GPU box of the test run has ONE MI355X, so the group is exercised with ngpus = 1: same code path (worker
orchestration, scatter / gather of row blocks through strided 2-D copies, a one-rank RCCL communicator from
ncclCommInitAll, the TSQR all-gather + stacked reduction), compared with the plain single-GPU handle and the oracle.
The rank > 1 arithmetic is covered on CPU by tests/test_dist_cpu.py (gloo, world size 2)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines():
    import torch  # noqa: F401  (its HIP runtime first, see tests/test_gpu_parity.py)
    import tlsq_amd
    plain = tlsq_amd.Engine(0)
    multi = tlsq_amd.Engine(ngpus=1)
    yield plain, multi
    multi.close()
    plain.close()


def relerr(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def test_multi_handle_reports_its_size(engines):
    plain, multi = engines
    assert plain.ngpus == 1 and multi.ngpus == 1
    import tlsq_amd
    with pytest.raises(tlsq_amd.TlsqError):
        tlsq_amd.Engine(ngpus=64)          # more GPUs than the box has: tlsq_create_multi fails loudly


@pytest.mark.parametrize("M,N,r,kw", [(1500, 96, 6, {}), (1237, 50, 4, {"nonnegE": True}), (3000, 130, 30, {"nukeA": False})])
def test_multi_rpca_vs_plain_and_oracle(engines, M, N, r, kw):
    from oracle import rpca_oracle as O
    plain, multi = engines
    D, _, _ = O.synth_lowrank_sparse(M, N, r, seed=M)
    A1, E1, s1, sv1, rep1 = plain.rpca(D, return_report=True, **kw)
    A2, E2, s2, sv2, rep2 = multi.rpca(D, return_report=True, **kw)
    Ao, Eo, so, svo, io = O.rpca(D, **kw)
    assert rep2.iters_done == rep1.iters_done == io.iters_done
    assert rep2.svp_hist == rep1.svp_hist == io.svp_hist and sv2 == sv1 == svo
    assert relerr(A2, Ao) < 1e-8 and relerr(E2, Eo) < 1e-8
    assert relerr(A2, A1) < 1e-10 and relerr(E2, E1) < 1e-10
    assert np.allclose(rep2.cost_hist, rep1.cost_hist, rtol=1e-6, atol=1e-12)
    # the returned SVD: all of S, and U S Vt = the last Z on both handles
    assert np.allclose(s2.S, so[1], rtol=1e-10, atol=64 * 2.2e-16 * np.sqrt(N) * so[1][0])
    Z1 = (np.asarray(s1.U) * s1.S) @ np.asarray(s1.Vt)
    Z2 = (np.asarray(s2.U) * s2.S) @ np.asarray(s2.Vt)
    assert relerr(Z2, Z1) < 1e-9


def test_multi_rpca_hooks_run_on_the_calling_thread(engines, capsys):
    import threading
    from oracle import rpca_oracle as O
    plain, multi = engines
    D, _, _ = O.synth_lowrank_sparse(600, 40, 3, seed=2)
    multi.rpca(D, verbose=True)                       # on_iter -> print, from the calling thread only
    out = capsys.readouterr().out
    assert "cost:" in out and "converged" in out
    assert threading.active_count() >= 1


def test_multi_lowrankfilter_vs_plain(engines):
    from oracle import rpca_oracle as O
    plain, multi = engines
    y, n = O.synth_series(6000, seed=3)
    f1 = plain.lowrankfilter(y + n, 40)
    f2 = multi.lowrankfilter(y + n, 40)
    assert relerr(f2, f1) < 1e-9
    assert relerr(f2, O.lowrankfilter(y + n, 40)) < 1e-8


def test_group_handle_wide_hankel_panel_runs_on_the_first_gpu(engines):
    """A short multichannel series gives a WIDE Hankel panel (47 x 54 here): too few rows to shard, and the single-GPU path
    goes through the general rpca entry with its own device panel - which used to mistake that for a group call with device
    pointers (TLSQ_ERR_UNSUPPORTED; found by tools/fuzz_lrf.py)."""
    plain, multi = engines
    rng = np.random.default_rng(3)
    t = np.arange(120)
    y = np.stack([np.sin(t / 7.0), np.cos(t / 11.0)], axis=1) + 0.01 * rng.standard_normal((120, 2))
    a = plain.lowrankfilter(y, 27, lag=2)
    b = multi.lowrankfilter(y, 27, lag=2)
    assert np.array_equal(a, b)


def test_multi_handle_other_entry_points_run_on_first_gpu(engines):
    from oracle import rpca_oracle as O
    plain, multi = engines
    rng = np.random.default_rng(0)
    A = rng.standard_normal((300, 5))
    x0 = rng.standard_normal(5)
    y = A @ x0 + 0.01 * rng.standard_normal(300)
    assert relerr(multi.tls(A, y), plain.tls(A, y)) < 1e-12
    assert relerr(multi.rtls(A, y), O.rtls(A, y)) < 1e-6
    W = rng.standard_normal((40, 90))                 # wide: does not shard, runs on the first GPU alone
    Aw, Ew, *_ = multi.rpca(W)
    Ap, Ep, *_ = plain.rpca(W)
    assert np.array_equal(Aw, Ap) and np.array_equal(Ew, Ep)
    assert np.array_equal(multi.hankel(np.arange(20.0), 3), plain.hankel(np.arange(20.0), 3))


def test_multi_handle_rejects_device_pointers(engines):
    import torch
    import tlsq_amd
    plain, multi = engines
    d = torch.zeros((96, 1500), dtype=torch.float64, device="cuda")
    a, e = torch.empty_like(d), torch.empty_like(d)
    with pytest.raises(tlsq_amd.TlsqError) as ei:
        multi.rpca_device(d.data_ptr(), 1500, 96, a.data_ptr(), e.data_ptr())
    assert ei.value.code == tlsq_amd._lib.TLSQ_ERR_UNSUPPORTED


# ---- rank > 1 on a one-GPU box: a loop-back group (the same device named several times) -------------------------
# RCCL refuses duplicate devices, so tlsq_create_multi gives such a group a host-staged communicator (runtime.hip,
# LocalGroup): every rank is a handle with its own stream and worker thread, all-reduce / all-gather reduce the ranks'
# buffers in rank order.  What this exercises is everything ABOVE the collective: the row-sharded solver's control flow
# with nranks > 1 (every decision has to come out the same on all ranks or the group dead-locks - the loop-back barrier
# then times out and the call fails), uneven row blocks, the TSQR gather + stacked factorisation, time-window shards.
@pytest.fixture(scope="module", params=[2, 3, 8])
def loopback(request):
    import torch  # noqa: F401
    import tlsq_amd
    plain = tlsq_amd.Engine(0)
    multi = tlsq_amd.Engine(devices=[0] * request.param)
    assert multi.ngpus == request.param
    yield plain, multi, request.param
    multi.close()
    plain.close()


@pytest.mark.parametrize("M,N,r,kw", [(1500, 96, 6, {}), (1237, 50, 4, {"nonnegE": True}), (3001, 130, 30, {"nukeA": False}),
                                      (20000, 128, 8, {})])
def test_loopback_rpca_rank_gt_1_vs_plain_and_oracle(loopback, M, N, r, kw):
    from oracle import rpca_oracle as O
    plain, multi, n = loopback
    D, _, _ = O.synth_lowrank_sparse(M, N, r, seed=M)
    A1, E1, s1, sv1, rep1 = plain.rpca(D, return_report=True, **kw)
    A2, E2, s2, sv2, rep2 = multi.rpca(D, return_report=True, **kw)
    assert rep2.iters_done == rep1.iters_done and rep2.svp_hist == rep1.svp_hist and sv2 == sv1
    assert relerr(A2, A1) < 1e-9 and relerr(E2, E1) < 1e-9
    assert np.allclose(rep2.cost_hist, rep1.cost_hist, rtol=1e-6, atol=1e-12)
    if M <= 3001:
        Ao, Eo, so, svo, io = O.rpca(D, **kw)
        assert rep2.iters_done == io.iters_done and rep2.svp_hist == io.svp_hist
        assert relerr(A2, Ao) < 1e-8 and relerr(E2, Eo) < 1e-8
        assert np.allclose(s2.S, so[1], rtol=1e-10, atol=64 * 2.2e-16 * np.sqrt(N) * so[1][0])


def test_loopback_user_hooks_on_row_shards(loopback):
    """Round 6 (missing 1): the caller's `svd(Z, sv)` / `opnorm(X)` closures (src/robustPCA.jl:168-169, :177,
    :193-197, :225) on a ROW-SHARDED group - every use gathers the shards, rank 0 calls the hook on the whole M x N panel on
    the calling thread, the results go back to the ranks.  With LAPACK behind both hooks the run reproduces the default path
    and the single-GPU run with the same hooks; the hook sees the exact panel shape, once per use, never from a worker thread."""
    import threading
    import scipy.linalg as sla
    from oracle import rpca_oracle as O
    plain, multi, n = loopback
    if n > 3:
        pytest.skip("two and three ranks cover it (every collective of the host-staged loop-back group costs a thread barrier)")
    M, N, r = 1237, 50, 4            # uneven row blocks on 2 and 3 ranks
    D, _, _ = O.synth_lowrank_sparse(M, N, r, seed=77)
    calls = {"svd": 0, "opnorm": 0, "shapes": set(), "threads": set()}

    def my_svd(Z, sv):
        calls["svd"] += 1
        calls["shapes"].add(Z.shape)
        calls["threads"].add(threading.get_ident())
        return sla.svd(Z, full_matrices=False, lapack_driver="gesdd")

    def my_opnorm(X):
        calls["opnorm"] += 1
        calls["shapes"].add(X.shape)
        calls["threads"].add(threading.get_ident())
        return sla.svdvals(X)[0]

    A0, E0, s0, sv0, rep0 = plain.rpca(D, return_report=True)
    A1, E1, s1, sv1, rep1 = plain.rpca(D, svd=my_svd, opnorm=my_opnorm, return_report=True)
    ncalls = dict(calls)
    calls.update(svd=0, opnorm=0)
    A2, E2, s2, sv2, rep2 = multi.rpca(D, svd=my_svd, opnorm=my_opnorm, return_report=True)
    assert rep2.iters_done == rep1.iters_done == rep0.iters_done and rep2.svp_hist == rep0.svp_hist and sv2 == sv0
    assert relerr(A2, A1) < 1e-9 and relerr(E2, E1) < 1e-9
    assert relerr(A2, A0) < 1e-8 and relerr(E2, E0) < 1e-8
    assert calls["svd"] == ncalls["svd"] == rep0.iters_done - 1      # iteration 1 is the library's own full SVD (:193)
    assert calls["opnorm"] == ncalls["opnorm"] == rep0.iters_done + 1
    assert calls["shapes"] == {(M, N)}
    assert calls["threads"] == {threading.get_ident()}
    # a truncating hook (rank sv, like rsvd) against the oracle with the same hook; only the svd hook this time
    trunc = lambda Z, sv: tuple(x[..., :sv] if i == 0 else (x[:sv] if i == 1 else x[:sv, :])
                                for i, x in enumerate(sla.svd(Z, full_matrices=False)))
    A3, E3, s3, sv3, rep3 = multi.rpca(D, svd=trunc, return_report=True)
    Ao, Eo, so, svo, io = O.rpca(D, svd=trunc)
    assert rep3.iters_done == io.iters_done and rep3.svp_hist == io.svp_hist and sv3 == svo
    assert relerr(A3, Ao) < 1e-8 and relerr(E3, Eo) < 1e-8
    # fp32 shards: the hook receives the float32 panel
    seen = []
    multi.rpca(D.astype(np.float32), opnorm=lambda X: (seen.append((X.dtype, X.shape)), float(sla.svdvals(X)[0]))[1], iters=3)
    assert seen and all(dt == np.float32 and sh == (M, N) for dt, sh in seen)
    # a failing hook fails the call on every rank (no hang)
    class Boom(Exception):
        pass

    def bad(Z, sv):
        raise Boom("hook failed")
    with pytest.raises(Boom):
        multi.rpca(D, svd=bad)
    # and the group is usable afterwards
    A4, E4, s4, sv4, rep4 = multi.rpca(D, return_report=True)
    assert rep4.iters_done == rep0.iters_done and relerr(A4, A0) < 1e-9


def test_loopback_noisy_problem_takes_the_tsqr_route(loopback):
    """Dense noise on top of the low-rank part: late iterations count singular values inside the noise of the Gram
    matrix and go through the TSQR route - on shards: per-rank factorisation, all-gather of the triangular factors,
    stacked factorisation, redundantly on every rank."""
    from oracle import rpca_oracle as O
    plain, multi, n = loopback
    rng = np.random.default_rng(11)
    M, N = 1200, 48
    D = rng.standard_normal((M, 3)) @ rng.standard_normal((3, N)) + 1e-3 * rng.standard_normal((M, N))
    A1, E1, s1, sv1, rep1 = plain.rpca(D, return_report=True)
    A2, E2, s2, sv2, rep2 = multi.rpca(D, return_report=True)
    Ao, Eo, so, svo, io = O.rpca(D)
    assert rep2.iters_done == rep1.iters_done == io.iters_done
    assert rep2.svp_hist == rep1.svp_hist == io.svp_hist
    assert relerr(A2, Ao) < 1e-8 and relerr(E2, Eo) < 1e-8
    assert rep2.tsqr_iterations > 0


def test_loopback_tiny_noise_deflated_certificate(loopback):
    """Noise at the level of float32 rounding: the late iterations are certified on the deflated panel (rows of Z minus the
    dominant part, per shard; its Gram all-reduced) instead of the TSQR route - every rank has to take the same decisions."""
    from oracle import rpca_oracle as O
    plain, multi, n = loopback
    rng = np.random.default_rng(1)
    M, N, r = 2402, 160, 6
    D = (rng.standard_normal((M, r)) @ rng.standard_normal((r, N)) + 10 * rng.standard_normal((M, N)) * (rng.random((M, N)) < 0.05)
         + 3e-7 * rng.standard_normal((M, N)))
    A1, E1, s1, sv1, rep1 = plain.rpca(D, return_report=True)
    A2, E2, s2, sv2, rep2 = multi.rpca(D, return_report=True)
    Ao, Eo, so, svo, io = O.rpca(D)
    assert rep2.iters_done == rep1.iters_done == io.iters_done and rep2.iters_done > 32
    assert rep2.svp_hist == rep1.svp_hist == io.svp_hist
    assert relerr(A2, Ao) < 1e-8 and relerr(E2, Eo) < 1e-8 and relerr(A2, A1) < 1e-9
    assert rep2.tsqr_iterations <= 1 and rep1.tsqr_iterations <= 1


def test_loopback_noisy_problem_matrix_function_route(loopback):
    """Noisy data on row shards: the count and A of the late iterations come from matrix functions of the all-reduced Gram
    matrix of the deflated panel (replicated N x N work, every rank has to take the same decisions), A = Z Phi per shard -
    the same route as on one GPU, the same trajectory as the oracle."""
    from oracle import rpca_oracle as O
    plain, multi, n = loopback
    rng = np.random.default_rng(5)
    M, N, r = 2401, 160, 6
    D = (rng.standard_normal((M, r)) @ rng.standard_normal((r, N)) + 10 * rng.standard_normal((M, N)) * (rng.random((M, N)) < 0.05)
         + 1e-3 * rng.standard_normal((M, N)))
    A1, E1, s1, sv1, rep1 = plain.rpca(D, return_report=True)
    A2, E2, s2, sv2, rep2 = multi.rpca(D, return_report=True)
    Ao, Eo, so, svo, io = O.rpca(D)
    assert rep2.iters_done == rep1.iters_done == io.iters_done and sv2 == sv1 == svo and sv2 > 4 * r
    assert rep2.svp_hist == rep1.svp_hist == io.svp_hist
    assert relerr(A2, Ao) < 1e-8 and relerr(E2, Eo) < 1e-8 and relerr(A2, A1) < 1e-9
    assert rep2.tsqr_iterations <= rep2.iters_done // 3 and rep1.tsqr_iterations <= rep1.iters_done // 3


def test_loopback_lowrankfilter_time_windows(loopback):
    from oracle import rpca_oracle as O
    plain, multi, n = loopback
    y, noise = O.synth_series(6000, seed=3)
    f1 = plain.lowrankfilter(y + noise, 40)
    f2 = multi.lowrankfilter(y + noise, 40)
    assert relerr(f2, f1) < 1e-9
    # fp32 series: the shards sum their anti-diagonals in fp32 (like the one-GPU kernel), fp64 across shards
    y32 = (y + noise).astype(np.float32)
    g1 = plain.lowrankfilter(y32, 40)
    g2 = multi.lowrankfilter(y32, 40)
    assert g2.dtype == np.float32 and relerr(g2.astype(np.float64), g1.astype(np.float64)) < 1e-4


@pytest.mark.parametrize("mode", [None, "entrywise_trimmed_mean", "entrywise_median"])
@pytest.mark.parametrize("d,N,r", [(10, 1000, 3), (40, 901, 2), (130, 2000, 2)])
def test_loopback_rpca_ga_on_column_shards(loopback, mode, d, N, r):
    """rpca_ga on a group handle (src/robustPCA.jl:255-310): contiguous blocks of the observations per rank, the weighted
    sums all-reduced (mean), and for the entrywise averages (:322-362) a radix select of whole rows across the shards -
    composite keys with the column index of the WHOLE row, histograms summed over the ranks in every pass - so that the
    element the stable sortperm of the full row would pick is found without gathering anything.  Held to the oracle and to
    the one-GPU solve: same iterations per component, same components."""
    import warnings
    from oracle import ga_oracle as G
    plain, multi, n = loopback
    rng = np.random.default_rng(d + N)
    Qt = np.linalg.qr(rng.standard_normal((d, r)))[0]
    X = Qt @ (rng.standard_normal((r, N)) * np.linspace(5.0, 2.0, r)[:, None]) + 0.05 * rng.standard_normal((d, N))
    out = rng.random(N) < 0.01
    X[:, out] += 20.0 * rng.standard_normal((d, int(out.sum())))
    X[:, 5] = X[:, 3]          # duplicate observations: ties in every row, resolved by the column index
    q0 = rng.standard_normal((d, r))
    kw = dict(q0=q0, iters=60, return_report=True)
    if mode:
        kw["mu"] = mode
    info = G.GaInfo()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = G.rpca_ga(X, r, q0=q0, info=info, iters=60, **({"mu": getattr(G, mode)} if mode else {}))
        got1, rep1 = plain.rpca_ga(X, r, **kw)
        got, rep = multi.rpca_ga(X, r, **kw)
    assert rep["iters"] == rep1["iters"] == info.iters
    assert rep["status"] == rep1["status"]
    assert np.abs(got - want).max() < 1e-9 and np.abs(got - got1).max() < 1e-9


def test_loopback_large_panel_shards_fused_rebuild_sweep():
    """Two ranks whose shards are large panels (>= 2^26 entries each): the fused rebuild + update + shrink sweep, the
    Frobenius shortcut with its all-reduced partial sums and the next iteration's Gram queued behind the sweep - with
    nranks = 2.  Same trajectory as the one-GPU solve of the whole matrix."""
    import torch  # noqa: F401
    import tlsq_amd
    from oracle import rpca_oracle as O
    M, N, r = 270_000, 512, 8
    D, A0, _ = O.synth_lowrank_sparse(M, N, r, seed=5)
    plain = tlsq_amd.Engine(0)
    try:
        A1, E1, s1, sv1, rep1 = plain.rpca(D, return_report=True, want_U=False, want_s=False, cost_history=False)
    finally:
        plain.close()
    multi = tlsq_amd.Engine(devices=[0, 0])
    try:
        A2, E2, s2, sv2, rep2 = multi.rpca(D, return_report=True, want_U=False, want_s=False, cost_history=False)
    finally:
        multi.close()
    assert rep1.converged and rep2.converged and sv2 == sv1 == r
    assert rep2.iters_done == rep1.iters_done and rep2.svp_hist == rep1.svp_hist
    assert relerr(A2, A1) < 1e-9 and relerr(E2, E1) < 1e-9
    assert relerr(A2, A0) < 1e-6


def test_loopback_uneven_shards_agree_on_the_collective_sequence():
    """decisions that depend on the rank-local row count and change WHICH collectives an iteration enters are agreed on
    before the loop (one min-all-reduce of the shape flags, solver.hip).  (a) 800003 x 256 fp64 on two ranks = 400002 + 400001
    rows: the fused sweep + Gram kernel (fused.hip: even row count, >= 400000 rows) can serve one shard and not the other - its
    residual-Gram mode would all-reduce R'R where the other rank all-reduces Z'Z.  (b) 16385 x 8192 fp32 = 8193 + 8192 rows: the
    fp16-split shape (M % 64 == 0) holds on one rank only, and at N = 8192 it decides between an N x N Gram all-reduce and the
    N x p all-reduces of the operator form.  Both complete (a mismatch is a hang that the loop-back barrier turns into an error) and
    reproduce the one-GPU solve."""
    import torch  # noqa: F401
    import tlsq_amd
    from oracle import rpca_oracle as O
    plain = tlsq_amd.Engine(0)
    multi = tlsq_amd.Engine(devices=[0, 0])
    try:
        D, A0, _ = O.synth_lowrank_sparse(800_003, 256, 4, seed=8)
        A1, E1, s1, sv1, rep1 = plain.rpca(D, return_report=True, want_U=False, want_s=False, cost_history=False)
        A2, E2, s2, sv2, rep2 = multi.rpca(D, return_report=True, want_U=False, want_s=False, cost_history=False)
        assert rep1.converged and rep2.converged and sv2 == sv1 == 4
        assert rep2.iters_done == rep1.iters_done and rep2.svp_hist == rep1.svp_hist
        assert relerr(A2, A1) < 1e-9 and relerr(E2, E1) < 1e-9
        del D, A0, A1, E1, A2, E2
        D, A0, _ = O.synth_lowrank_sparse(16385, 8192, 12, seed=9, dtype=np.float32)
        A1, E1, s1, sv1, rep1 = plain.rpca(D, return_report=True, want_U=False, want_s=False, cost_history=False)
        A2, E2, s2, sv2, rep2 = multi.rpca(D, return_report=True, want_U=False, want_s=False, cost_history=False)
        assert rep1.converged and rep2.converged and sv2 == sv1 == 12
        assert abs(rep2.iters_done - rep1.iters_done) <= 1
        assert relerr(A2.astype(np.float64), A1.astype(np.float64)) < 1e-3
        assert relerr(A2.astype(np.float64), A0.astype(np.float64)) < 1e-3
    finally:
        multi.close()
        plain.close()


def test_loopback_large_mode_fp32_shards():
    """BASELINE config 5's shape class on row shards: fp32, min(M, N) > 2048 (large mode: subspace solver only, the Gram
    of every shard on the fp32 MFMA with fp64 fold-in, all-reduced), two ranks; against the one-GPU solve."""
    import torch  # noqa: F401
    import tlsq_amd
    from oracle import rpca_oracle as O
    M, N, r = 6000, 2304, 12
    D, A0, _ = O.synth_lowrank_sparse(M, N, r, seed=0, dtype=np.float32)
    plain = tlsq_amd.Engine(0)
    try:
        A1, E1, s1, sv1, rep1 = plain.rpca(D, return_report=True, want_U=False)
    finally:
        plain.close()
    multi = tlsq_amd.Engine(devices=[0, 0])
    try:
        A2, E2, s2, sv2, rep2 = multi.rpca(D, return_report=True, want_U=False)
    finally:
        multi.close()
    assert A2.dtype == np.float32 and rep1.converged and rep2.converged and sv2 == sv1 == r
    assert abs(rep2.iters_done - rep1.iters_done) <= 1
    assert relerr(A2.astype(np.float64), A1.astype(np.float64)) < 1e-3
    assert relerr(A2.astype(np.float64), A0.astype(np.float64)) < 1e-3


def test_loopback_large_fp32_shards_round5_kernels():
    """The round-5 kernels of BASELINE config 5 on row shards (two ranks, 32768 rows each, 2304 columns, rank 40): the
    fp16-split Gram kernel on every shard (each with the scale of ITS rows: the all-reduce adds rescaled fp64 matrices), the
    fp32 factors and the sweep that forms A_k on the MFMA, and - with svd = "randomized" - the block power hook whose products
    are all-reduced; against the one-GPU solve of the same panel."""
    import torch  # noqa: F401
    import tlsq_amd
    from tlsq_amd import workloads as W
    M, N, r = 65536, 2304, 40
    D, A0, _ = W.synth_lowrank_sparse(M, N, r, seed=9)
    D = D.astype(np.float32)
    for kw in ({}, {"svd": "randomized"}):
        plain = tlsq_amd.Engine(0)
        try:
            A1, E1, s1, sv1, rep1 = plain.rpca(D, return_report=True, want_s=False, cost_history=False, **kw)
        finally:
            plain.close()
        multi = tlsq_amd.Engine(devices=[0, 0])
        try:
            A2, E2, s2, sv2, rep2 = multi.rpca(D, return_report=True, want_s=False, cost_history=False, **kw)
        finally:
            multi.close()
        assert rep1.converged and rep2.converged and sv2 == sv1 == r, kw
        assert abs(rep2.iters_done - rep1.iters_done) <= 1, kw
        assert relerr(A2.astype(np.float64), A1.astype(np.float64)) < 1e-4, kw
        assert relerr(E2.astype(np.float64), E1.astype(np.float64)) < 1e-3, kw
        assert relerr(A2.astype(np.float64), A0) < 1e-3, kw


def test_loopback_hankel_flag_on_row_shards(loopback):
    """rpca(H; hankel=true) and lowrankfilter(...; hankel=true) on row shards: the anti-diagonal means of soft_hankel!
    (src/robustPCA.jl:214-216, 234-236) run through several ranks' row blocks - block sums and counts at the block's
    row offset, one all-reduce, shrink towards the global means.  Same trajectory and matrices as the one-GPU solve."""
    from oracle import rpca_oracle as O
    plain, multi, n = loopback
    y, noise = O.synth_series(1500, seed=7)
    H = plain.hankel(y + noise, 30)
    for kw in ({"hankel": True}, {"hankel": True, "nukeA": False}):
        A1, E1, s1, sv1, rep1 = plain.rpca(H, return_report=True, **kw)
        A2, E2, s2, sv2, rep2 = multi.rpca(H, return_report=True, **kw)
        assert rep2.iters_done == rep1.iters_done and rep2.svp_hist == rep1.svp_hist and sv2 == sv1
        assert relerr(A2, A1) < 1e-9 and relerr(E2, E1) < 1e-9
    Ao, Eo, so, svo, io = O.rpca(H, hankel=True)
    A2, E2, s2, sv2, rep2 = multi.rpca(H, return_report=True, hankel=True)
    assert rep2.iters_done == io.iters_done and relerr(A2, Ao) < 1e-8 and relerr(E2, Eo) < 1e-8
    f1 = plain.lowrankfilter(y + noise, 30, hankel=True)
    f2 = multi.lowrankfilter(y + noise, 30, hankel=True)
    assert relerr(f2, f1) < 1e-9


def test_group_call_with_user_hooks(engines):
    """`rpca(D; svd = my_svd)` on a multi-GPU handle (round 6: row-sharded like any other call - the shards are
    gathered for every use of a hook, which runs on the calling thread): same trajectory as the plain handle with the same hooks.
    `lowrankfilter` with a hook still runs on the first GPU alone (bit-identical to the plain handle)."""
    import scipy.linalg as sla
    import tlsq_amd
    from oracle import rpca_oracle as O
    plain, multi = engines
    D, _, _ = O.synth_lowrank_sparse(900, 40, 4, seed=5)

    def my_svd(Z, sv):
        U, S, Vt = sla.svd(Z, full_matrices=False, lapack_driver="gesdd")
        return U, S, Vt

    def my_opnorm(X):
        return float(sla.svd(X, compute_uv=False, lapack_driver="gesdd")[0])

    A1, E1, s1, sv1, rep1 = plain.rpca(D, return_report=True, svd=my_svd, opnorm=my_opnorm)
    A2, E2, s2, sv2, rep2 = multi.rpca(D, return_report=True, svd=my_svd, opnorm=my_opnorm)
    assert rep2.svp_hist == rep1.svp_hist and sv2 == sv1
    assert relerr(A2, A1) < 1e-10 and relerr(E2, E1) < 1e-10
    loop2 = tlsq_amd.Engine(devices=[0, 0])
    try:
        A3, E3, s3, sv3, rep3 = loop2.rpca(D, return_report=True, svd=my_svd, opnorm=my_opnorm)
        assert relerr(A3, A1) < 1e-9 and relerr(E3, E1) < 1e-9 and rep3.svp_hist == rep1.svp_hist
        y, noise = O.synth_series(3000, seed=3)
        f1 = plain.lowrankfilter(y + noise, 30, opnorm=my_opnorm)
        f3 = loop2.lowrankfilter(y + noise, 30, opnorm=my_opnorm)
        assert np.array_equal(f1, f3)
    finally:
        loop2.close()


def test_comm_init_is_refused_on_a_group_handle(engines):
    import tlsq_amd
    plain, multi = engines
    with pytest.raises(tlsq_amd.TlsqError) as ei:
        multi.comm_init(1, 0, multi.unique_id())
    assert ei.value.code == tlsq_amd._lib.TLSQ_ERR_ARG


def test_a_failing_rank_takes_the_group_down_instead_of_hanging():
    """a rank that leaves a group call with an error (OOM, a HIP error on one GPU) used to leave the others blocked
    in the next collective.  FAIL_RANK injects such an error on one rank of a loop-back group: the call must come back with
    that rank's error within seconds, and the handle must serve the next call."""
    import time
    import tlsq_amd
    from oracle import rpca_oracle as O
    D, _, _ = O.synth_lowrank_sparse(1500, 96, 6, seed=1500)
    loop3 = tlsq_amd.Engine(devices=[0, 0, 0])
    try:
        A0, E0, *_ = loop3.rpca(D)
        for bad_rank in (2, 0):
            with tlsq_amd.dev_switches(FAIL_RANK=bad_rank):
                t0 = time.time()
                with pytest.raises(tlsq_amd.TlsqError) as ei:
                    loop3.rpca(D)
                assert time.time() - t0 < 20.0
                assert "injected failure" in str(ei.value) and ei.value.code == tlsq_amd._lib.TLSQ_ERR_HIP
            A1, E1, *_ = loop3.rpca(D)
            assert np.array_equal(A0, A1) and np.array_equal(E0, E1)
    finally:
        loop3.close()


@pytest.mark.parametrize("n", [2, 8])
def test_bench_problems_keep_their_trajectory_on_row_shards(n):
    """What `bench.py --gpus N` validates on a multi-GPU node, checked here on loop-back groups: the two bench problems (C2:
    20000 x 512, extra.c4: 200000 x 512 from 8 seeded row blocks) row-sharded over N ranks reproduce the iterations, sv and
    rank trajectory of the CPU oracle's frozen run (tests/golden/bench_vectors.json) - the all-reduced Gram matrices
    differ from the one-GPU ones in their last bits only, and no decision of the solver may depend on those."""
    import hashlib
    import json
    import os
    import torch  # noqa: F401
    import tlsq_amd
    from tlsq_amd import workloads as W
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bench_vectors.json")))
    D2 = W.synth_lowrank_sparse(20000, 512, 16, seed=0)[0]
    D4 = W.c4_rows(0, W.C4_SHAPE[0])
    e = tlsq_amd.Engine(devices=[0] * n)
    try:
        for name, D in (("c2", D2), ("c4", D4)):
            A, E, s, sv, rep = e.rpca(D, return_report=True, want_s=False)
            h = hashlib.sha256(",".join(str(int(v)) for v in rep.svp_hist).encode()).hexdigest()[:16]
            assert (rep.iters_done, int(sv), h, bool(rep.converged)) == (ref[name]["iters"], ref[name]["sv"], ref[name]["svp_hash"],
                                                                         ref[name]["converged"]), name
            assert abs(float(np.sum(A * A)) - ref[name]["normA2"]) <= 1e-8 * ref[name]["normA2"]
            assert abs(float(np.sum(E * E)) - ref[name]["normE2"]) <= 1e-8 * ref[name]["normE2"]
            st = ref[name]["sample_stride"]
            for X, key in ((A, "A_sample"), (E, "E_sample")):
                w = np.asarray(ref[name][key])
                assert np.linalg.norm(X.ravel(order="F")[::st] - w) <= 1e-8 * np.linalg.norm(w), (name, key)
    finally:
        e.close()
