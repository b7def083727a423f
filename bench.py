#!/usr/bin/env python3
"""bench.py — rpca ALM iterations/sec on a 20000x512 fp64 D (BASELINE.json config 2) on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one complete rpca solve (the reference's hot loop, /root/reference/src/robustPCA.jl:156-239,
run to its own convergence test) on a synthetic rank-16 + 5%-sparse D that is already resident in HBM;
outputs A, E stay in HBM.  value = ALM iterations completed in the K timed steps / wall time (max over
ranks).  N>1 row-shards the SAME 20000x512 problem (strong scaling) and exchanges the N x N Gram matrices
with RCCL inside libtlsqhip.so.

Extra objects on the JSON line:
  roofline      ALM sweeps (HBM-bound): the bytes the shipped fused form has to move (8 passes over M*N*8 per
                iteration, 7 for panels of 2^26+ elements, + 5 for the lone shrink at k=1; SURVEY.md §8d's unfused figure is 11 and is reported
                beside it) divided by the sweeps' device time measured with HIP events on the library's
                stream inside the timed solves (tlsq_rpca_info.ms_shrink + ms_update).
  cpu_baseline  the oracle (oracle/rpca_oracle.py: LAPACK gesdd + fused OpenMP sweeps) timed on this box's
                host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=20000)
    ap.add_argument("--cols", type=int, default=512)
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--cpu-iters", type=int, default=24, help="ALM iterations of the CPU-oracle sample (0 = skip)")
    ap.add_argument("--no-c4", action="store_true", help="skip the extra 200000x512 row-sharded measurement (extra.c4)")
    args = ap.parse_args()

    import numpy as np
    import torch  # first: torch brings its own HIP runtime (see tests/test_gpu_parity.py)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    torch.zeros(1, device="cuda")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import tlsq_amd
    from tlsq_amd import dist as tdist
    from oracle import rpca_oracle as O   # cpu_baseline + input generator only

    M, N, r = args.rows, args.cols, args.rank
    D, A0, S0 = O.synth_lowrank_sparse(M, N, r, seed=0)
    lo, hi = tdist.row_partition(M, world, rank)
    Dl = np.ascontiguousarray(D[lo:hi].T)                 # (N, Ml) C-order == (Ml, N) column-major
    Ml = hi - lo
    dD = torch.from_numpy(Dl).cuda()
    dA = torch.empty_like(dD)
    dE = torch.empty_like(dD)

    eng = tlsq_amd.Engine(local_rank)
    tdist.init_engine_comm(eng, rank, world)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def solve():
        # like a non-verbose reference call: no per-iteration cost history requested
        return eng.rpca_device(dD.data_ptr(), Ml, N, dA.data_ptr(), dE.data_ptr(), m_global=M, want_hist=False)

    for _ in range(args.warmup):
        solve()
    barrier()
    t0 = time.perf_counter()
    iters_total = 0
    rskip_total = 0
    hbm_sweeps_total = hbm_total = 0.0
    ms = {}
    last = None
    for _ in range(args.steps):
        sv, rep, st = solve()
        iters_total += rep.iters_done
        rskip_total += rep.residual_stores_skipped
        hbm_sweeps_total += rep.hbm_bytes_sweeps
        hbm_total += rep.hbm_bytes
        for k, v in rep.ms.items():
            ms[k] = ms.get(k, 0.0) + v
        last = (sv, rep, st)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- extra.c4: BASELINE config 4 (rpca 200000 x 512 fp64, row-sharded over the N GPUs of this run) -------------
    # The >= 6x @ 8 GPUs target of the north star is quoted on THIS shape, so every --gpus N run also times it (after
    # the headline measurement, outside its timed region).  The matrix is defined by 8 row blocks of 25000 rows with
    # their own seeds, so N = 1, 2, 4, 8 all solve the same problem and every rank only generates its own rows.
    c4 = None
    if not args.no_c4 and args.rows == 20000 and args.cols == 512:
        M4, N4, r4, nb = 200000, 512, 16, 8
        rb = M4 // nb
        lo4, hi4 = tdist.row_partition(M4, world, rank)
        G2 = np.random.default_rng([4, 999]).standard_normal((r4, N4))
        parts = []
        for b in range(lo4 // rb, (hi4 - 1) // rb + 1):
            rg = np.random.default_rng([4, b])
            blk = rg.standard_normal((rb, r4)) @ G2 + 10.0 * rg.standard_normal((rb, N4)) * (rg.random((rb, N4)) < 0.05)
            parts.append(blk[max(lo4 - b * rb, 0): min(hi4 - b * rb, rb)])
        D4 = np.ascontiguousarray(np.vstack(parts).T)
        del parts
        M4l = hi4 - lo4
        d4 = torch.from_numpy(D4).cuda()
        a4, e4 = torch.empty_like(d4), torch.empty_like(d4)
        run4 = lambda: eng.rpca_device(d4.data_ptr(), M4l, N4, a4.data_ptr(), e4.data_ptr(), m_global=M4, want_hist=False)
        run4()                                   # warm-up (workspace growth)
        barrier()
        t4 = time.perf_counter()
        n4 = 0
        for _ in range(2):
            sv4, rep4, st4 = run4()
            n4 += rep4.iters_done
        barrier()
        t4 = time.perf_counter() - t4
        if world > 1:
            tt = torch.tensor([t4], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t4 = float(tt.item())
        c4 = {"workload": "rpca 200000x512 fp64 rank-16 + 5% sparse, row-sharded, reference defaults, to convergence",
              "value": n4 / t4, "unit": "iters/s", "n_gpus": world, "rows_per_gpu": M4l, "solves": 2,
              "ms_per_solve": t4 / 2 * 1e3, "iters_per_solve": rep4.iters_done, "sv": sv4, "converged": rep4.converged,
              "phases_ms_per_iter": {k: v / rep4.iters_done for k, v in rep4.ms.items()
                                     if k in ("shrink", "update")}}
        del d4, a4, e4, D4

    # one more solve, outside the timed region, with every phase of the iteration bracketed by HIP events: the source of
    # phases_ms_per_iter and of the Gram roofline (the timed solves only bracket the sweep kernels - each recorded
    # event costs ~6 us between two kernels, tlsq_rpca_opts.phase_timing)
    _, rep_ph, _ = eng.rpca_device(dD.data_ptr(), Ml, N, dA.data_ptr(), dE.data_ptr(), m_global=M, want_hist=False,
                                   phase_timing=True)
    barrier()
    ms_ph = dict(rep_ph.ms)

    sv, rep, st = last
    # sanity of the timed work (not part of the timing): residual and recovery on this rank's shard
    A = dA.cpu().numpy().T
    E = dE.cpu().numpy().T
    resid = float(np.linalg.norm(D[lo:hi] - (A + E)) / np.linalg.norm(D[lo:hi]))
    rel_a = float(np.linalg.norm(A - A0[lo:hi]) / np.linalg.norm(A0[lo:hi]))

    if rank == 0:
        value = iters_total / dt
        sweep_ms_per_iter = (ms["shrink"] + ms["update"]) / iters_total
        # Bytes the sweeps have to move.  SURVEY.md §8d prices the two-kernel form (K1 R3/W2 + K2 R4/W2 = 11
        # passes per iteration); the shipped path fuses K2(k) with K1(k+1) (R4/W4 = 8 passes) and runs one plain
        # K1 at k = 1 - folded together with the set-up's Y = D / dual_norm: R D / W Y,E,Z = 4 passes (k_first_shrink) - so
        # the roofline is priced against the bytes of THAT form - the smaller figure.
        fused = os.environ.get("TLSQ_NO_FUSED_SWEEP", "0") != "1"
        # large panels (>= 2^26 elements): the rebuild A = T Vs' is folded in as well (A stays in registers):
        # R D,E,Y / W R,Y,E',Z' = 7 passes
        fused_rebuild = (fused and os.environ.get("TLSQ_NO_FUSED_REBUILD", "0") != "1" and Ml % 2 == 0 and
                         (Ml * N >= 1 << 26 or os.environ.get("TLSQ_FUSED_REBUILD", "0") == "1"))
        array_bytes = float(Ml) * N * 8
        sweep_passes = 7.0 if fused_rebuild else 8.0
        # the shipped default: the E-free sweep (k_zsweep) - A from its factors in registers, E never stored while the loop
        # runs: R D,Y,Z / W R,Y',Z' = 6 passes (5 without the residual store); k_first_shrink R D / W Y,Z = 3 passes.
        # (The returned E is formed once after the loop, outside the two timed phases: not counted here.)
        zsweep = fused and os.environ.get("TLSQ_NO_ZSWEEP", "0") != "1" and os.environ.get("TLSQ_NO_FIRST_SHRINK", "0") != "1"
        if zsweep:
            alg_bytes = (3.0 * args.steps + 6.0 * iters_total - rskip_total) / iters_total * array_bytes
        elif fused:
            # sweeps that were told not to store the residual panel moved one pass less
            first_passes = 5.0 if os.environ.get("TLSQ_NO_FIRST_SHRINK", "0") == "1" else 4.0
            alg_bytes = (first_passes * args.steps + sweep_passes * iters_total - rskip_total) / iters_total * array_bytes
        else:
            alg_bytes = 11.0 * array_bytes
        achieved = alg_bytes / (sweep_ms_per_iter * 1e-3) / 1e9
        out = {
            "metric": "rpca ALM iters/sec on 20000x512 fp64 D",
            "value": value, "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"rpca {M}x{N} fp64 rank-{r} + 5% sparse, reference defaults "
                                   f"(lambda=1/sqrt(M), rho=1.5, tol=sqrt(eps)), to convergence",
                       "rows_per_gpu": Ml, "iters_per_solve": rep.iters_done, "sv": sv,
                       "converged": rep.converged, "residual": resid, "rel_err_A": rel_a,
                       "parallelism": f"row-shard x{world}" if world > 1 else "single GPU",
                       "call": "plain (non-verbose) call: no per-iteration cost history is requested, so opnorm(residual) is "
                               "only resolved far enough to settle cost < tol; identical iterations and outputs "
                               "(tests/test_gpu_parity.py::test_rpca_device_mode_and_decision_only_cost)"},
            "roofline": {"kernel": "k_zsweep (E-free fused rebuild + ALM sweep) + k_first_shrink at k=1" if zsweep else
                                   ("k_rebuild_update_shrink (fused rebuild + ALM sweep) + k_first_shrink at k=1" if fused_rebuild else "k_update_shrink (fused ALM sweep) + k_first_shrink at k=1") if fused else "k_shrink + k_update (ALM sweeps)", "bound": "hbm", "achieved": achieved,
                         "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                         "ms_per_iter": sweep_ms_per_iter, "algorithmic_bytes_per_iter": alg_bytes,
                         "passes_per_iter": alg_bytes / array_bytes, "sweeps_without_residual_store": rskip_total,
                         "survey_11_pass_equivalent_GBps": 11.0 * array_bytes / (sweep_ms_per_iter * 1e-3) / 1e9},
            "phases_ms_per_iter": {k: v / rep_ph.iters_done for k, v in ms_ph.items()
                                   if k in ("shrink", "gram", "eig", "rebuild", "update", "opnorm")},
            "phases_source": "one extra solve with tlsq_rpca_opts.phase_timing = 1 (events at all nine phase boundaries: "
                             "that solve runs ~6 % slower than the timed ones)",
            "roofline_mfma": None,
            "jacobi_sweeps_per_solve": rep.jacobi_sweeps,
            "svd_step": {"tsqr_route": rep.eig_full, "subspace": rep.eig_fast, "subspace_steps": rep.subspace_steps},
            "extra": {"c4": c4},
        }
        # on-box reference point for a streaming kernel (SURVEY §8d asks for one beside the 8 TB/s vendor figure): a plain
        # torch device-to-device copy of 1 GiB (read + write = 2 GiB moved), best of 5, on torch's stream.  NOT a ceiling:
        # the fused sweep's non-temporal 16-byte loads/stores run faster than this copy
        try:
            src = torch.empty(1 << 27, dtype=torch.float64, device="cuda").fill_(1.0)
            dst = torch.empty_like(src)
            best = 0.0
            for _ in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dst.copy_(src)
                e1.record()
                torch.cuda.synchronize()
                best = max(best, 2.0 * src.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
            out["roofline"]["torch_copy_GBps"] = best
            del src, dst
        except Exception as e:  # the bench line must not depend on this extra
            out["roofline"]["torch_copy_GBps"] = None
            print(f"# torch copy rate not measured: {e}", file=sys.stderr)
        # HBM bytes of the sweep kernels.  `traffic` = the PMC measurement of these kernels (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate passes, FETCH x2 on gfx950 - collected with tools/profile_round.sh and committed under
        # profiles/; a counter run cannot happen inside this process) scaled to this run's launches; beside it the
        # library's own accounting of what it launched in the timed solves (tlsq_rpca_info.hbm_bytes_sweeps).
        out["roofline"]["library_accounted_bytes_per_iter"] = hbm_sweeps_total / iters_total
        out["hbm_bytes_per_iter_all_panel_kernels"] = hbm_total / iters_total
        for pmc_file in ("r02_pmc_sweeps.json", "r01_pmc_sweeps.json"):
            try:
                with open(os.path.join(ROOT, "profiles", pmc_file)) as f:
                    pmc = json.load(f)
                if Ml == 20000 and N == 512:
                    sk = pmc["sweep_kernels"]
                    kname = "k_zsweep" if zsweep else "k_rebuild_update_shrink" if fused_rebuild else "k_update_shrink"
                    if fused and kname in sk:   # PMC average over the launches of one solve (with and without the R store)
                        first = sk.get("k_first_shrink", sk.get("k_shrink"))
                        tr = (first["hbm_bytes_per_launch"] +
                              rep.iters_done * sk[kname]["hbm_bytes_per_launch"]) / rep.iters_done
                    elif not fused and "k_update" in sk:
                        tr = sk["k_shrink"]["hbm_bytes_per_launch"] + sk["k_update"]["hbm_bytes_per_launch"]
                    else:
                        tr = None
                    out["roofline"]["traffic"] = tr
                    out["roofline"]["traffic_source"] = f"profiles/{pmc_file} (rocprofv3 --pmc, separate passes)"
                break
            except (OSError, KeyError):
                continue
        # second roofline: the Gram kernel (fp64 MFMA, v_mfma_f64_16x16x4_f64), flops actually executed
        # ALGORITHMIC flops of G = Z'Z: the N (N + 1) / 2 distinct entries, 2 M flop each.  The kernel issues more: the
        # strictly lower 128 x 128 tiles in full, of the diagonal tiles the 36 of 64 MFMA tiles on and below the diagonal
        nt = (N + 127) // 128
        gram_flops = 1.0 * Ml * N * (N + 1)
        issued_flops = 2.0 * Ml * 128 * 128 * (nt * (nt - 1) // 2 + nt * 36.0 / 64.0)
        if ms_ph.get("gram"):
            # Gram launches of the profiled solve: one per iteration, plus the one queued behind the last sweep
            # before its convergence is known (the library hides the host round trip behind it)
            n_gram = rep_ph.iters_done + 1
            ms = dict(ms, gram=ms_ph["gram"])
            tf = gram_flops / (ms["gram"] / n_gram * 1e-3) / 1e12
            out["roofline_mfma"] = {"kernel": "k_gram_kc Gram(Z) + k_slab_reduce", "bound": "mfma",
                                    "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6,
                                    "flops_per_launch": gram_flops, "issued_flops_per_launch": issued_flops,
                                    "issued_TFLOPs": issued_flops / (ms["gram"] / n_gram * 1e-3) / 1e12,
                                    "ms_per_launch": ms["gram"] / n_gram, "launches": n_gram}
        if world == 1 and args.cpu_iters > 0:
            ncores = os.cpu_count() or 1
            # LAPACK gesdd on a 20000x512 panel does not scale to hundreds of threads: pick the best of a few
            # thread counts on one iteration, then time the sample with it (a fair best-effort host baseline)
            best_t, best_n = None, ncores
            try:
                from threadpoolctl import threadpool_limits
            except Exception:   # noqa: BLE001
                threadpool_limits = None
            cands = sorted({c for c in (8, 16, 32, 64, ncores) if c <= ncores})
            if threadpool_limits is not None:
                O.rpca(D[:2000], iters=1)                 # warm LAPACK/OpenMP
                for c in cands:
                    with threadpool_limits(limits=c):
                        tq = time.perf_counter()
                        O.rpca(D, iters=1)
                        tq = time.perf_counter() - tq
                    if best_t is None or tq < best_t:
                        best_t, best_n = tq, c
            ctx = threadpool_limits(limits=best_n) if threadpool_limits is not None else None
            if ctx is not None:
                ctx.__enter__()
            tc = time.perf_counter()
            _, _, _, _, ci = O.rpca(D, iters=args.cpu_iters)
            tc = time.perf_counter() - tc
            if ctx is not None:
                ctx.__exit__(None, None, None)
            out["cpu_baseline"] = {"value": ci.iters_done / tc, "unit": "iters/s", "cores": best_n, "kind": "port",
                                   "host_cores_available": ncores,
                                   "sample": f"first {ci.iters_done} ALM iterations of the same {M}x{N} D "
                                             f"(oracle: LAPACK gesdd x2 per iteration + fused OpenMP sweeps; "
                                             f"thread count chosen as the fastest of {cands}), {tc:.1f} s"}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
