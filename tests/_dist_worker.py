"""Worker for tests/test_dist_cpu.py: run under torch.distributed.run with the gloo backend (CPU)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir = sys.argv[1]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from tlsq_amd import dist as tdist
    from oracle import rpca_oracle as O
    from oracle.sharded import rpca_sharded

    res = {"rank": rank, "world": world}
    # --- env plumbing + partition
    r2, w2, _ = tdist.env_rank_world()
    res["env_ok"] = (r2 == rank and w2 == world)
    M, N = 1001, 40
    lo, hi = tdist.row_partition(M, world, rank)
    spans = [None] * world
    dist.all_gather_object(spans, (lo, hi))
    res["partition_ok"] = (spans[0][0] == 0 and spans[-1][1] == M and
                           all(spans[i][1] == spans[i + 1][0] for i in range(world - 1)) and
                           max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1)
    # --- unique-id exchange (what Engine.unique_id() feeds on the GPU box)
    uid = tdist.exchange_unique_id(lambda: bytes((7 * i + 3) % 256 for i in range(128)), rank, world)
    res["uid_ok"] = uid == bytes((7 * i + 3) % 256 for i in range(128))

    # --- row-sharded rpca (Gram all-reduce) == single-process LAPACK oracle
    def allreduce(a, op):
        t = torch.from_numpy(np.ascontiguousarray(a))
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
        return t.numpy()

    D, A0, _ = O.synth_lowrank_sparse(M, N, 4, seed=11)
    A, E, sv, info = rpca_sharded(D[lo:hi], M, allreduce)
    Ao, Eo, _, svo, io = O.rpca(D)
    res["iters"] = (info["iters_done"], io.iters_done)
    res["svp_same"] = info["svp_hist"] == io.svp_hist
    res["sv"] = (sv, svo)
    res["relA"] = float(np.linalg.norm(A - Ao[lo:hi]) / np.linalg.norm(Ao[lo:hi]))
    res["relE"] = float(np.linalg.norm(E - Eo[lo:hi]) / np.linalg.norm(Eo[lo:hi]))
    # --- column-sharded rpca_ga (all-reduce of d+1 sums per iteration) == single-process oracle
    from oracle import ga_oracle as G
    from oracle.sharded import rpca_ga_sharded
    rng = np.random.default_rng(5)
    d, Ng, r = 12, 999, 3
    u = np.linalg.qr(rng.standard_normal((d, r)))[0]
    X = (u * np.array([30.0, 20.0, 10.0])) @ rng.standard_normal((r, Ng)) + 0.01 * rng.standard_normal((d, Ng))
    q0 = rng.standard_normal((d, r))
    clo, chi = tdist.row_partition(Ng, world, rank)          # the same contiguous split, over columns
    Qs, used = rpca_ga_sharded(X[:, clo:chi], r, q0, allreduce)
    ginfo = G.GaInfo()
    Qo = G.rpca_ga(X, r, q0=q0, info=ginfo)
    res["ga_iters"] = (used, ginfo.iters)
    res["ga_err"] = float(np.abs(Qs - Qo).max())
    # --- the entrywise averages on column shards: a radix select of whole rows with summed histograms (:322-362)
    import warnings
    Xr = X.copy()
    Xr[:, 7] = Xr[:, 400]                                    # ties across the shards: the column index decides
    Xr[:, rng.random(Ng) < 0.02] *= 15.0
    for name, fn in (("trimmed_mean", G.entrywise_trimmed_mean), ("median", G.entrywise_median)):
        Qs, used = rpca_ga_sharded(Xr[:, clo:chi], 2, q0, allreduce, iters=40, average=name, col_off=clo, N_glob=Ng)
        ginfo = G.GaInfo()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            Qo = G.rpca_ga(Xr, 2, q0=q0, info=ginfo, mu=fn, iters=40)
        res["ga_" + name] = (used, ginfo.iters, float(np.abs(Qs - Qo).max()))
    # --- time-window sharded lowrankfilter (halo + all-reduce of anti-diagonal sums and counts) == oracle
    from oracle.sharded import lowrankfilter_sharded
    ys, nz = O.synth_series(700, seed=2)
    yf_s, linfo = lowrankfilter_sharded(ys + nz, 20, rank, world, allreduce)
    yf_o = O.lowrankfilter(ys + nz, 20)
    res["lrf_err"] = float(np.linalg.norm(yf_s - yf_o) / np.linalg.norm(yf_o))
    # --- TSQR on row shards (tsqr.hip with a communicator): local R_r, all-gather of the N x N factors, redundant
    #     reduction of the stack -> the singular values of the whole panel to eps * sigma_max, identical on all ranks
    Zt = (np.random.default_rng(9).standard_normal((M, N)) * np.geomspace(1.0, 1e-9, N)[None, :])
    Rl = np.linalg.qr(Zt[lo:hi], mode="r")
    gathered = [torch.zeros((N, N), dtype=torch.float64) for _ in range(world)]
    pad = np.zeros((N, N))
    pad[: Rl.shape[0]] = Rl
    dist.all_gather(gathered, torch.from_numpy(pad))
    Rs = np.linalg.qr(np.vstack([g.numpy() for g in gathered]), mode="r")
    sv_stack = np.linalg.svd(Rs, compute_uv=False)
    sv_full = np.linalg.svd(Zt, compute_uv=False)
    res["tsqr_err"] = float(np.max(np.abs(sv_stack - sv_full)) / sv_full[0])
    chk = torch.from_numpy(np.ascontiguousarray(np.abs(Rs)))
    ref = chk.clone()
    dist.broadcast(ref, src=0)
    res["tsqr_same_on_all_ranks"] = bool(torch.equal(chk, ref))
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
