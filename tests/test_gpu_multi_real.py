"""Real RCCL, more than one physical GPU: skipped on a one-GPU box, ready for the day the run lands on a node
with several MI355X.  Everything in tests/test_gpu_multi.py that needs rank > 1 runs on loop-back groups (one device named
several times: a host-staged communicator, because RCCL refuses duplicate devices); these tests name DISTINCT devices, so
tlsq_create_multi builds its communicators with ncclCommInitAll and every all-reduce / all-gather of the row-sharded solver
goes over xGMI:

  * the two bench problems (C2, C4 = BASELINE config 4) against the CPU oracle's frozen run - iterations, rank trajectory, norms
    and samples of A and E (tests/golden/bench_vectors.json);
  * a failing rank (FAIL_RANK) takes the group down through ncclCommAbort instead of leaving the others in a collective;
  * `bench.py --gpus N` under torch.distributed.run (one process per GPU, the driver's launch line) prints its JSON line with
    `validation.rccl_ranks == N`.
"""
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpus():
    try:
        import torch
        return torch.cuda.device_count()      # (counting devices does not initialise the GPU)
    except Exception:
        return 0


needs2 = pytest.mark.skipif(_ngpus() < 2, reason="needs at least two physical GPUs (real RCCL over xGMI)")


def _group_sizes():
    n = _ngpus()
    return [k for k in (2, 4, 8) if k <= n] or [2]


@needs2
@pytest.mark.parametrize("n", _group_sizes())
def test_bench_problems_on_distinct_devices_match_the_oracle_fixture(n):
    import torch  # noqa: F401
    import tlsq_amd
    from tlsq_amd import workloads as W
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_vectors.json")))
    D2 = W.synth_lowrank_sparse(20000, 512, 16, seed=0)[0]
    D4 = W.c4_rows(0, W.C4_SHAPE[0])
    e = tlsq_amd.Engine(devices=list(range(n)))
    try:
        assert e.ngpus == n
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
        # same bits from two runs of the same group (RCCL's all-reduce returns identical bits on every rank, and the
        # library's own reductions are in fixed order)
        A2, E2, *_ = e.rpca(D2, want_s=False)
        A3, E3, *_ = e.rpca(D2, want_s=False)
        assert np.array_equal(A2, A3) and np.array_equal(E2, E3)
    finally:
        e.close()


@needs2
def test_lowrankfilter_time_windows_on_distinct_devices():
    """Row shards of the Hankel panel are time windows with an n - 1 sample halo: against the one-GPU run."""
    import torch  # noqa: F401
    import tlsq_amd
    from oracle import rpca_oracle as O
    y, noise = O.synth_series(200_000, seed=11)
    one = tlsq_amd.Engine(0)
    grp = tlsq_amd.Engine(devices=list(range(min(_ngpus(), 8))))
    try:
        f1 = one.lowrankfilter(y + noise, 64)
        f2 = grp.lowrankfilter(y + noise, 64)
        assert np.linalg.norm(f2 - f1) <= 1e-9 * np.linalg.norm(f1)
    finally:
        grp.close()
        one.close()


@needs2
def test_a_failing_rank_aborts_the_real_communicators():
    """FAIL_RANK on a group of distinct devices: the other ranks sit in an RCCL collective when the failure happens - the
    library aborts the communicators (ncclCommAbort under the per-communicator mutex, runtime.hip), the call returns the
    injected error within seconds, and the group serves the next call with new communicators."""
    import torch  # noqa: F401
    import tlsq_amd
    from oracle import rpca_oracle as O
    n = min(_ngpus(), 4)
    D, _, _ = O.synth_lowrank_sparse(6000, 128, 6, seed=6000)
    grp = tlsq_amd.Engine(devices=list(range(n)))
    try:
        A0, E0, *_ = grp.rpca(D)
        for bad_rank in (n - 1, 0):
            with tlsq_amd.dev_switches(FAIL_RANK=bad_rank):
                t0 = time.time()
                with pytest.raises(tlsq_amd.TlsqError) as ei:
                    grp.rpca(D)
                assert time.time() - t0 < 60.0
                assert "injected failure" in str(ei.value)
            A1, E1, *_ = grp.rpca(D)
            assert np.array_equal(A0, A1) and np.array_equal(E0, E1)
    finally:
        grp.close()


@needs2
@pytest.mark.parametrize("n", _group_sizes()[:1])
def test_bench_py_under_torch_distributed_run(n):
    """The driver's launch line for N > 1: one process per GPU, rendezvous on 127.0.0.1, launched before anything touches a
    GPU in this (child) process tree; rank 0 prints ONE JSON line with the whole-job value and rccl_ranks = N."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
           "--no-extras"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["validation"]["rccl_ranks"] == n and rec["validation"]["ok"], rec["validation"]
    assert rec["value"] > 0 and rec["scaling"] == "strong"
