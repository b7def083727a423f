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
