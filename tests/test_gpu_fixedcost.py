"""What round 6 changed around the ALM loop of a plain rpca call (src/robustPCA.jl:177, :193-204, :225 under /root/reference): the
norms behind `opnorm` by Lanczos steps that share one launch (csrc/lanczos.hip, k_lanczos_multi), the cold block of the first
`svd!` (second step on the counted columns, Ritz pairs sorted on the device, fresh pad columns projected out of the block).
None of it may change what a solve returns: every switch against the default, and the norm kernel against numpy."""
import ctypes as C
import time

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


def _panel(rng, N, kind):
    M = 3 * N + 7
    if kind == "flat":
        return rng.standard_normal((M, N))
    if kind == "lowrank":
        return rng.standard_normal((M, 9)) @ rng.standard_normal((9, N)) + 1e-3 * rng.standard_normal((M, N))
    if kind == "graded":
        return rng.standard_normal((M, N)) * np.logspace(0, -6, N)[None, :]
    if kind == "rank1":   # Lanczos breaks down after one step: beta = 0
        return np.outer(rng.standard_normal(M), rng.standard_normal(N))
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["flat", "lowrank", "graded", "rank1"])
@pytest.mark.parametrize("N", [64, 65, 100, 500, 512, 777, 1024])
def test_opnorm_steps_in_one_launch_vs_one_launch_per_step(eng, N, kind):
    """sigma_max through tlsq_k_opnorm_f64 (Gram matrix + Lanczos): both step kernels against LAPACK's 2-norm.  N covers one row
    per workgroup, uneven rows per workgroup (65, 100, 777: the last workgroups own fewer rows or none) and the LDS limit (1024)."""
    import torch
    import tlsq_amd
    rng = np.random.default_rng(N + len(kind))
    Z = _panel(rng, N, kind)
    M = Z.shape[0]
    ref = np.linalg.norm(Z, 2)
    d = torch.from_numpy(np.ascontiguousarray(Z.T)).cuda()
    out = {}
    for mode in ("1", "0"):
        with tlsq_amd.dev_switches(LZ_MULTI=mode):
            o = C.c_double(0.0)
            assert eng.lib.tlsq_k_opnorm_f64(eng.h, C.c_void_p(d.data_ptr()), M, N, M, C.byref(o)) == 0
            out[mode] = o.value
    assert abs(out["1"] / ref - 1) < 1e-12 and abs(out["0"] / ref - 1) < 1e-12
    assert abs(out["1"] / out["0"] - 1) < 1e-12


def test_a_workgroup_that_never_publishes_is_survived():
    """LZ_MULTI=drop: workgroup 1 of k_lanczos_multi leaves without publishing its rows.  The others must give up (bounded polls,
    ~20 ms), the call must still return the right norm - from one launch per step - and the handle must not try the kernel again."""
    import torch
    import tlsq_amd
    e = tlsq_amd.Engine(0)
    try:
        rng = np.random.default_rng(11)
        N, M = 512, 2000
        Z = rng.standard_normal((M, N))
        ref = np.linalg.norm(Z, 2)
        d = torch.from_numpy(np.ascontiguousarray(Z.T)).cuda()
        o = C.c_double(0.0)
        with tlsq_amd.dev_switches(LZ_MULTI="drop"):
            t0 = time.perf_counter()
            assert e.lib.tlsq_k_opnorm_f64(e.h, C.c_void_p(d.data_ptr()), M, N, M, C.byref(o)) == 0
            t_first = time.perf_counter() - t0
            assert abs(o.value / ref - 1) < 1e-12
            assert t_first > 0.015                      # the give-up really happened
            t0 = time.perf_counter()
            assert e.lib.tlsq_k_opnorm_f64(e.h, C.c_void_p(d.data_ptr()), M, N, M, C.byref(o)) == 0
            t_second = time.perf_counter() - t0
            assert abs(o.value / ref - 1) < 1e-12
            assert t_second < 0.015                     # ... and is not paid again on this handle
    finally:
        e.close()


@pytest.mark.parametrize("shape", [(3000, 256, 8), (5000, 512, 16), (2500, 200, 5)])
def test_round6_switches_do_not_change_a_solve(eng, shape):
    """Every round-6 change of the set-up and of the first `svd!` against the path it replaced, on the same panel: same
    iterations and rank trajectory, A and E to 1e-9 (the norms agree to the last bits, not bit for bit), costs to 1e-6."""
    import tlsq_amd
    from oracle import rpca_oracle as O
    M, N, r = shape
    D, _, _ = O.synth_lowrank_sparse(M, N, r, seed=M + 3 * N)
    A, E, s, sv, rep = eng.rpca(D, return_report=True, want_s=False)
    assert sv == r and rep.converged
    for sw in (dict(LZ_MULTI=0), dict(COLD_TOP=0), dict(COLD_TOL0=0), dict(RITZ_SORT=0), dict(PAD_PROJECT=0),
               dict(LZ_MULTI=0, COLD_TOP=0, COLD_TOL0=0, RITZ_SORT=0, PAD_PROJECT=0)):
        with tlsq_amd.dev_switches(**sw):
            A1, E1, s1, sv1, rep1 = eng.rpca(D, return_report=True, want_s=False)
        assert sv1 == sv and rep1.iters_done == rep.iters_done and rep1.svp_hist == rep.svp_hist, sw
        assert np.linalg.norm(A1 - A) <= 1e-9 * np.linalg.norm(A), sw
        assert np.linalg.norm(E1 - E) <= 1e-9 * np.linalg.norm(E), sw
        np.testing.assert_allclose(rep1.cost_hist, rep.cost_hist, rtol=1e-6, err_msg=str(sw))
        assert abs(rep1.d_norm / rep.d_norm - 1) < 1e-13, sw


def test_sorted_block_when_the_rank_grows(eng):
    """A panel whose count above 1/mu grows from one ALM iteration to the next (dense noise on top of the planted parts) keeps
    re-ordering the block: the device-side sort of the Ritz pairs against the host's gather, same trajectory and results."""
    import tlsq_amd
    from oracle import rpca_oracle as O
    M, N, r = 4000, 256, 12
    D, _, _ = O.synth_lowrank_sparse(M, N, r, seed=99)
    D = D + 1e-4 * np.random.default_rng(5).standard_normal(D.shape) * np.abs(D).max()
    A, E, s, sv, rep = eng.rpca(D, return_report=True, want_s=False)
    with tlsq_amd.dev_switches(RITZ_SORT=0):
        A1, E1, s1, sv1, rep1 = eng.rpca(D, return_report=True, want_s=False)
    assert sv1 == sv and rep1.svp_hist == rep.svp_hist and rep1.iters_done == rep.iters_done
    assert np.linalg.norm(A1 - A) <= 1e-9 * np.linalg.norm(A)
    assert np.linalg.norm(E1 - E) <= 1e-9 * np.linalg.norm(E)
