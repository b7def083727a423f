"""N>1 path on CPU: two gloo ranks exercise the host-side sharding glue (row partition, RCCL-id
exchange) and check that the row-sharded algorithm (local Gram -> all-reduce -> replicated small
eigenproblem -> local rebuild) reproduces the single-process oracle."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_row_sharding(tmp_path):
    env = dict(os.environ, OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path)]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    for r in range(2):
        res = json.load(open(tmp_path / f"rank{r}.json"))
        assert res["world"] == 2 and res["env_ok"] and res["partition_ok"] and res["uid_ok"]
        assert res["iters"][0] == res["iters"][1] and res["svp_same"] and res["sv"][0] == res["sv"][1]
        assert res["relA"] < 1e-8 and res["relE"] < 1e-8, res
        assert res["ga_iters"][0] == res["ga_iters"][1] and res["ga_err"] < 1e-10, res
        for name in ("trimmed_mean", "median"):
            used, want, err = res["ga_" + name]
            assert used == want and err < 1e-10, (name, res["ga_" + name])
        assert res["lrf_err"] < 1e-8, res
        assert res["tsqr_err"] < 1e-14 and res["tsqr_same_on_all_ranks"], res


def test_row_partition_edge_cases():
    import tlsq_amd  # noqa: F401
    from tlsq_amd import dist as tdist
    for M, W in [(10, 3), (8, 8), (5, 8), (200000, 8), (20000, 1)]:
        spans = [tdist.row_partition(M, W, r) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == M
        assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1 and sum(sizes) == M
