#!/usr/bin/env python3
"""bench.py — rpca ALM iterations/sec on a 20000x512 fp64 D (BASELINE.json config 2) on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one complete rpca solve (the reference's hot loop, /root/reference/src/robustPCA.jl:156-239, run to its own
convergence test) on a synthetic rank-16 + 5%-sparse D that is already resident in HBM; outputs A, E stay in HBM.
value = ALM iterations completed in the K timed steps / wall time (max over ranks).  N>1 row-shards the SAME 20000x512
problem (strong scaling) and exchanges the N x N Gram matrices with RCCL inside libtlsqhip.so.

Beside `value` (all outside the timed region):
  value_with_s          the same solve with the returned decomposition s = (U, S, Vt) of the last Z requested as well
                        (`A, E, s, sv = rpca(D)` of the reference returns it, src/robustPCA.jl:238; `value` does not ask for it)
  value_host_pointers   the same solve through HOST pointers (what a Julia `Array` call pays: H2D of D, D2H of A, E), no s
  roofline              the fused ALM sweep against the 8 TB/s HBM peak: algorithmic bytes / device time of the sweep kernels
                        (HIP events on the library's stream inside the timed solves)
  roofline_mfma         the Gram kernel against the fp64 MFMA peak
  cpu_baseline          the oracle (LAPACK gesdd + fused OpenMP sweeps) on this box's host cores, bounded sample (N=1 only)
  extra.c4              BASELINE config 4 (200000x512, the shape of the >= 6x @ 8 GPUs target) on the GPUs of this run
  extra.c5              BASELINE config 5 (65536x4096 fp32 rank 64, svd = randomized and exact) on one GPU: ms per iteration
  validation            every run checks itself against the CPU ORACLE's run of the same two full-size problems, frozen as data in
                        tests/golden/bench_vectors.json (tests/golden/make_bench_vectors.py: LAPACK gesdd, reference expression
                        order): identical iterations, sv and rank trajectory, cost history to 1e-6, strided samples of A and E
                        (each rank checks the samples in its row block) and ||A||^2, ||E||^2 to 1e-8, the returned singular values
                        to 1e-10 (N = 1).  The ranks must agree among themselves; rccl_ranks = size of the communicator as RCCL
                        reports it.  Any disagreement, or a run that exceeds --timeout seconds (a hung collective), ends with a
                        non-zero exit.
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
REF_PATH = os.path.join(ROOT, "tests", "golden", "bench_vectors.json")


def svp_hash(hist):
    return hashlib.sha256(",".join(str(int(v)) for v in hist).encode()).hexdigest()[:16]


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
    ap.add_argument("--no-extras", action="store_true", help="skip value_with_s / value_host_pointers")
    ap.add_argument("--no-c5", action="store_true", help="skip the extra 65536x4096 fp32 measurement (extra.c5, one GPU only)")
    ap.add_argument("--timeout", type=float, default=900.0, help="seconds after which the run is declared hung (exit 3)")
    args = ap.parse_args()

    # a collective that never completes must not hang the driver: the watchdog ends the process (no re-exec, no retry here)
    def hung():
        sys.stderr.write(f"bench.py: no result after {args.timeout:.0f} s - a collective or a kernel hangs; giving up\n")
        sys.stderr.flush()
        os._exit(3)
    dog = threading.Timer(args.timeout, hung)
    dog.daemon = True
    dog.start()

    import numpy as np
    import torch  # first: torch brings its own HIP runtime (see tests/test_gpu_parity.py)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ          # under torch.distributed.run (also with one process)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    torch.zeros(1, device="cuda")
    use_dist = launched
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import tlsq_amd
    from tlsq_amd import dist as tdist
    from tlsq_amd import workloads as W   # seeded synthetic inputs (numpy only; the oracle is imported for cpu_baseline alone)

    M, N, r = args.rows, args.cols, args.rank
    headline = (M, N, r) == (20000, 512, 16)
    D, A0, S0 = W.synth_lowrank_sparse(M, N, r, seed=0)
    lo, hi = tdist.row_partition(M, world, rank)
    Dl = np.ascontiguousarray(D[lo:hi].T)                 # (N, Ml) C-order == (Ml, N) column-major
    Ml = hi - lo
    dD = torch.from_numpy(Dl).cuda()
    dA = torch.empty_like(dD)
    dE = torch.empty_like(dD)

    eng = tlsq_amd.Engine(local_rank)
    if launched and world == 1:
        # one process under the launcher: a one-rank RCCL communicator, so that a one-GPU box runs the code of an N-GPU run
        tlsq_amd.dev_set("FORCE_COMM", 1)
    tdist.init_engine_comm(eng, rank, world, force=launched)
    rccl_ranks = eng.comm_size()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def allsum(vals):
        if not use_dist:
            return list(vals)
        t = torch.tensor(list(vals), dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    def solve(**kw):
        # like a non-verbose reference call: no per-iteration cost history requested
        return eng.rpca_device(dD.data_ptr(), Ml, N, dA.data_ptr(), dE.data_ptr(), m_global=M, want_hist=False, **kw)

    # ---- the timed region: W warm-up solves, then exactly K solves between two barriers --------------------------------
    for _ in range(args.warmup):
        solve()
    barrier()
    t0 = time.perf_counter()
    iters_total = 0
    rskip_total = 0
    hbm_sweeps_total = hbm_total = hbm_timed_total = 0.0
    sweeps_timed_total = 0
    ms = {}
    last = None
    runs = [solve() for _ in range(args.steps)]   # (the reports are read after the clock has stopped)
    barrier()
    dt = allmax(time.perf_counter() - t0)
    for sv, rep, st in runs:
        iters_total += rep.iters_done
        rskip_total += rep.residual_stores_skipped
        hbm_sweeps_total += rep.hbm_bytes_sweeps
        hbm_total += rep.hbm_bytes
        hbm_timed_total += rep.hbm_bytes_sweeps_timed
        sweeps_timed_total += rep.sweeps_timed
        for k, v in rep.ms.items():
            ms[k] = ms.get(k, 0.0) + v
        last = (sv, rep, st)
    sv, rep, st = last

    # ---- validation of the timed work against the oracle's frozen run (outside the timing) -------------------------------
    ref = None
    if headline and os.path.exists(REF_PATH):
        with open(REF_PATH) as f:
            ref = json.load(f)
    problems = []

    def sample_check(dX, key, want, lo_, hi_, Mg):
        """strided samples of the column-major Mg x N oracle panel that fall into this rank's rows [lo_, hi_): squared
        difference and squared reference (summed over the ranks by the caller)"""
        st_ = want["sample_stride"]
        idx = np.arange(0, Mg * N, st_, dtype=np.int64)
        rows, cols = idx % Mg, idx // Mg
        m = (rows >= lo_) & (rows < hi_)
        wv = np.asarray(want[key], dtype=np.float64)[m]
        flat = torch.from_numpy(cols[m] * (hi_ - lo_) + (rows[m] - lo_)).cuda()
        got = dX.reshape(-1)[flat].cpu().numpy()
        return float(np.sum((got - wv) ** 2)), float(np.sum(wv ** 2))

    def check_problem(cfg, dAx, dEx, lo_, hi_, Mg, rep_plain, sv_, rep_hist):
        mine_ = {"iters": rep_plain.iters_done, "sv": int(sv_), "svp_hash": svp_hash(rep_hist.svp_hist),
                 "converged": bool(rep_plain.converged)}
        if use_dist:
            everyone = [None] * world
            dist.all_gather_object(everyone, mine_)
            if any(e != everyone[0] for e in everyone):
                problems.append(f"{cfg}: the ranks disagree: {everyone}")
        na2_, ne2_ = allsum([float((dAx * dAx).sum().item()), float((dEx * dEx).sum().item())])
        got = dict(mine_, normA2=na2_, normE2=ne2_)
        want = ref.get(cfg) if ref else None
        if want:
            for key in ("iters", "sv", "svp_hash", "converged"):
                if got[key] != want[key]:
                    problems.append(f"{cfg}.{key}: {got[key]} (oracle {want[key]})")
            for key in ("normA2", "normE2"):
                if not abs(got[key] - want[key]) <= 1e-8 * abs(want[key]):
                    problems.append(f"{cfg}.{key}: {got[key]!r} (oracle {want[key]!r})")
            if list(rep_hist.svp_hist) != list(want["svp_hist"]):
                problems.append(f"{cfg}.svp_hist: {list(rep_hist.svp_hist)} (oracle {want['svp_hist']})")
            ch, cw = np.asarray(rep_hist.cost_hist, dtype=np.float64), np.asarray(want["cost_hist"])
            if ch.shape != cw.shape or not np.allclose(ch, cw, rtol=1e-6, atol=1e-12):
                problems.append(f"{cfg}.cost_hist differs from the oracle's beyond rtol 1e-6")
            for key, dX in (("A_sample", dAx), ("E_sample", dEx)):
                num, den = allsum(sample_check(dX, key, want, lo_, hi_, Mg))
                rel = (num / max(den, 1e-300)) ** 0.5
                got[key + "_relerr"] = rel
                if not rel <= 1e-8:
                    problems.append(f"{cfg}.{key}: relative error {rel:.3e} against the oracle (bar 1e-8)")
        return got

    A = dA.cpu().numpy().T
    E = dE.cpu().numpy().T
    resid = float(np.linalg.norm(D[lo:hi] - (A + E)) / np.linalg.norm(D[lo:hi]))
    rel_a = float(np.linalg.norm(A - A0[lo:hi]) / np.linalg.norm(A0[lo:hi]))
    # (one more solve WITH the per-iteration cost: the rank trajectory and the cost history the oracle's are compared with; the
    #  plain solves above settle cost < tol only - same iterations and outputs, tests/test_gpu_parity.py)
    _, rep_h, _ = eng.rpca_device(dD.data_ptr(), Ml, N, dA.data_ptr(), dE.data_ptr(), m_global=M, want_hist=True)
    check = {"c2": check_problem("c2", dA, dE, lo, hi, M, rep, sv, rep_h)} if headline else {}

    # ---- the same solve with the returned decomposition, and through host pointers (outside the timed region; before the
    # large configurations below, which leave the GPU in another thermal state: 10 % on a 13 ms solve) -------------
    extras = {}
    if not args.no_extras and world == 1:
        d = min(M, N)
        dU = torch.empty((d, Ml), dtype=torch.float64, device="cuda")
        dS = torch.empty(d, dtype=torch.float64, device="cuda")
        dVt = torch.empty((N, d), dtype=torch.float64, device="cuda")
        with_s = lambda: solve(dU=dU.data_ptr(), dS=dS.data_ptr(), dVt=dVt.data_ptr())
        with_s()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        ns = 0
        for _ in range(3):
            _, rs, _ = with_s()
            ns += rs.iters_done
        torch.cuda.synchronize()
        ts = time.perf_counter() - ts
        extras["value_with_s"] = ns / ts
        extras["ms_per_solve_with_s"] = ts / 3 * 1e3
        if ref and "c2" in ref:   # the returned singular values against the oracle's s.S of the last Z (:194, :238)
            Sg, Sw = dS.cpu().numpy(), np.asarray(ref["c2"]["S"])
            floor = 64 * np.finfo(float).eps * np.sqrt(d) * Sw[0]
            bad = np.abs(Sg - Sw) > 1e-10 * Sw + floor
            check["c2"]["S_max_relerr"] = float(np.max(np.abs(Sg - Sw) / Sw))
            if bad.any():
                problems.append(f"c2.S: {int(bad.sum())} of {d} returned singular values differ from the oracle's beyond rtol 1e-10")
        Df = np.asfortranarray(D)
        eng.rpca(Df, want_s=False, cost_history=False)
        keep = []   # the returned arrays stay alive: releasing 164 MB of touched pages (munmap, ~7 ms) is the caller's business
        th = time.perf_counter()
        nh = 0
        for _ in range(3):
            out_h = eng.rpca(Df, want_s=False, cost_history=False, return_report=True)
            keep.append(out_h)
            nh += out_h[4].iters_done
        th = time.perf_counter() - th
        keep.clear()
        extras["value_host_pointers"] = nh / th
        extras["ms_per_solve_host_pointers"] = th / 3 * 1e3
        extras["host_pointers_note"] = ("numpy arrays of the caller (pageable memory, outputs freshly allocated per call): H2D of D, "
                                        "D2H of A and E (246 MB per solve) inside the time, through the library's pinned staging "
                                        "pipeline (csrc/staging.hip)")
        # the drop-in call: what `A, E, s, sv = rpca(D)` of the Julia shim executes (julia/TotalLeastSquaresHIP.jl: host
        # pointers AND the returned decomposition U, S, Vt) - src/robustPCA.jl:156, :238
        eng.rpca(Df, cost_history=False)
        td = time.perf_counter()
        nd = 0
        for _ in range(3):
            out_d = eng.rpca(Df, cost_history=False, return_report=True)
            keep.append(out_d)
            nd += out_d[4].iters_done
        td = time.perf_counter() - td
        keep.clear()
        extras["value_dropin"] = nd / td
        extras["ms_per_solve_dropin"] = td / 3 * 1e3
        extras["dropin_note"] = "host pointers and s = (U, S, Vt): +82 MB of U and 2 MB of Vt over PCIe, + the accurate SVD of the last Z"
        del dU, dS, dVt

    # ---- extra.c4: BASELINE config 4 (rpca 200000 x 512 fp64, row-sharded over the N GPUs of this run) -------------
    # The >= 6x @ 8 GPUs target of the north star is quoted on THIS shape, so every --gpus N run also times it (after
    # the headline measurement, outside its timed region).  The matrix is defined by 8 row blocks of 25000 rows with
    # their own seeds, so N = 1, 2, 4, 8 all solve the same problem and every rank only generates its own rows.
    c4 = None
    if not args.no_c4 and headline:
        M4, N4, r4, nb = W.C4_SHAPE
        lo4, hi4 = tdist.row_partition(M4, world, rank)
        D4 = np.ascontiguousarray(W.c4_rows(lo4, hi4).T)
        M4l = hi4 - lo4
        d4 = torch.from_numpy(D4).cuda()
        a4, e4 = torch.empty_like(d4), torch.empty_like(d4)
        run4 = lambda **kw: eng.rpca_device(d4.data_ptr(), M4l, N4, a4.data_ptr(), e4.data_ptr(), m_global=M4, **kw)
        run4(want_hist=False)                                   # warm-up (workspace growth)
        barrier()
        t4 = time.perf_counter()
        n4 = 0
        for _ in range(2):
            sv4, rep4, st4 = run4(want_hist=False)
            n4 += rep4.iters_done
        barrier()
        t4 = allmax(time.perf_counter() - t4)
        _, rep4p, _ = run4(want_hist=False, phase_timing=True)      # all six phases bracketed (the Amdahl term per N)
        _, rep4h, _ = run4(want_hist=True)
        c4 = {"workload": "rpca 200000x512 fp64 rank-16 + 5% sparse, row-sharded, reference defaults, to convergence",
              "value": n4 / t4, "unit": "iters/s", "n_gpus": world, "rows_per_gpu": M4l, "solves": 2,
              "ms_per_solve": t4 / 2 * 1e3, "iters_per_solve": rep4.iters_done, "sv": sv4, "converged": rep4.converged,
              "phases_ms_per_iter": {k: v / rep4p.iters_done for k, v in rep4p.ms.items()
                                     if k in ("shrink", "gram", "eig", "rebuild", "update", "opnorm")},
              "phases_note": "eig = the replicated N x N solve (does not shrink with the number of GPUs); everything else is sharded"}
        check["c4"] = check_problem("c4", a4, e4, lo4, hi4, M4, rep4, sv4, rep4h)
        del d4, a4, e4, D4

    # ---- extra.c5: BASELINE config 5 (rpca 65536 x 4096 fp32, rank 64 + 5 % sparse, svd = randomized) on ONE GPU --------------
    # The panel is generated on the device (same distribution as workloads.synth_lowrank_sparse; 1.07 GB per panel), solved once
    # for the workspace and twice for the clock, in the reference's hook mode (:195-197) and in the exact mode.
    c5 = None
    if not args.no_c5 and headline and world == 1:
        from tlsq_amd import _lib as L5
        M5, N5, r5 = 65536, 4096, 64
        ref5 = ref.get("c5") if ref else None
        if ref5:
            # the oracle's own panel (numpy, seed 0: tests/golden/make_bench_vectors.py c5): generated on the host, ~20 s
            D5h, A05h, _ = W.synth_lowrank_sparse(M5, N5, r5, seed=0, dtype=np.float32)
            d5 = torch.from_numpy(np.ascontiguousarray(D5h.T)).cuda()
            A05 = torch.from_numpy(np.ascontiguousarray(A05h.T)).cuda()
            del D5h, A05h
        else:
            g5 = torch.Generator(device="cuda").manual_seed(5)
            A05 = (torch.randn(N5, r5, device="cuda", generator=g5) @ torch.randn(r5, M5, device="cuda", generator=g5))   # (N x M row-major = M x N column-major)
            d5 = A05 + 10.0 * torch.randn(N5, M5, device="cuda", generator=g5) * (torch.rand(N5, M5, device="cuda", generator=g5) < 0.05)
        a5, e5 = torch.empty_like(d5), torch.empty_like(d5)
        torch.cuda.synchronize()
        c5 = {"workload": "rpca 65536x4096 fp32 rank-64 + 5% sparse, one GPU, reference defaults, to convergence", "unit": "ms per iteration",
              "operand_precision": "the Gram matrix and the operator products of this shape run on v_mfma_f32_16x16x32_f16 with every fp32 "
                                   "entry split into two fp16 numbers (22 significant bits, three of the four cross products; fp32 "
                                   "accumulation folded into fp64): narrower operands than the reference's fp32 LAPACK (24 bits) - held to the "
                                   "fp32 LAPACK oracle inside a solve by tests/test_gpu_configs.py::test_c5_h3_shape_vs_fp32_oracle "
                                   "(32768x2304, identical rank trajectory)",
              "validated_against_oracle": None}
        for tag, kw in (("randomized", dict(svd_mode=L5.SVD_RANDOMIZED)), ("exact", {})):
            run5 = lambda: eng.rpca_device(d5.data_ptr(), M5, N5, a5.data_ptr(), e5.data_ptr(), want_hist=False, dtype=np.float32, **kw)
            run5()
            torch.cuda.synchronize()
            # (every call timed on its own - blocking call + device synchronisation - and added up: what lies BETWEEN two calls
            #  of this process - the allocator returning the gigabytes of the panel generation to the system - is not the solve's)
            t5 = 0.0
            n5 = 0
            for _ in range(2):
                t5a = time.perf_counter()
                sv5, rep5, st5 = run5()
                torch.cuda.synchronize()
                t5 += time.perf_counter() - t5a
                n5 += rep5.iters_done
            err5 = float(torch.linalg.norm(a5 - A05) / torch.linalg.norm(A05))
            c5[tag] = {"ms_per_iter": t5 / n5 * 1e3, "ms_per_solve": t5 / 2 * 1e3, "iters_per_solve": rep5.iters_done, "sv": int(sv5),
                       "converged": bool(rep5.converged), "rel_err_A_vs_planted": err5}
            c5[tag]["kernels"] = dict(rep5.kern)
            if not (rep5.converged and int(sv5) == r5 and err5 < 1e-3):
                problems.append(f"c5 {tag}: converged={rep5.converged} sv={sv5} rel_err_A={err5:.2e}")
            if tag == "exact" and ref5:
                # the fp32 LAPACK oracle's run of this panel: iterations within 1, the rank trajectory, strided samples at the fp32 bar
                _, rep5h, _ = eng.rpca_device(d5.data_ptr(), M5, N5, a5.data_ptr(), e5.data_ptr(), want_hist=True, dtype=np.float32)
                st5 = int(ref5["sample_stride"])
                nI = min(rep5h.iters_done, int(ref5["iters"]))
                okv = abs(rep5h.iters_done - int(ref5["iters"])) <= 1 and list(rep5h.svp_hist)[:nI] == list(ref5["svp_hist"])[:nI]
                rels = {}
                for key, dX in (("A_sample", a5), ("E_sample", e5)):
                    got5 = dX.reshape(-1)[::st5].double().cpu().numpy()
                    want5 = np.asarray(ref5[key], dtype=np.float64)
                    rels[key] = float(np.linalg.norm(got5 - want5) / max(np.linalg.norm(want5), 1e-300))
                    okv = okv and rels[key] <= 1e-3
                c5["validated_against_oracle"] = {"ok": bool(okv), "iters": rep5h.iters_done, "oracle_iters": int(ref5["iters"]), **rels}
                if not okv:
                    problems.append(f"c5 exact against the fp32 oracle: {c5['validated_against_oracle']}")
        c5["value"] = c5["randomized"]["ms_per_iter"]
        # roofline of the two kernels that carry a hook iteration (per launch, algorithmic figures; durations from the kernel
        # statistics of profiles/): k_zsweep_wide moves 3 reads + 2 writes of the 1.07 GB panel; k_gram_h3 multiplies the lower
        # triangle of a 4096 x 4096 Gram matrix over 65536 rows three times (hh + hl + lh) on the fp16 MFMA
        c5["roofline"] = {
            "hook_iteration_hbm_floor_ms": (5 + 6) * M5 * N5 * 4 / 8.0e12 * 1e3,
            "hook_iteration_note": "5 panel passes of the sweep + 6 of the three operator products (Z X, then Z' T) against 8 TB/s; the "
                                   "measured iteration is c5.randomized.ms_per_iter",
            "gram_h3_flops_fp16": 3 * 2.0 * M5 * N5 * (N5 + 128) / 2, "gram_h3_peak_tflops_fp16": 2500.0,
            "gram_h3_note": "three fp16 products per entry of the lower triangle (128-column tiles); 4.1 ms per launch = 0.32 of the dense "
                            "fp16 MFMA peak - the kernel is bound by staging its operand planes through LDS (DESIGN 8, round 5 point 7)"}
        del d5, a5, e5, A05
        torch.cuda.empty_cache()

    validation = {"rccl_ranks": rccl_ranks, "reference": os.path.relpath(REF_PATH, ROOT) if ref else None,
                  "reference_kind": "CPU oracle (LAPACK gesdd, reference expression order), frozen by tests/golden/make_bench_vectors.py",
                  "ok": True}
    if rccl_ranks != world and launched:
        problems.append(f"the communicator has {rccl_ranks} ranks, the run {world}")
    validation["ok"] = not problems
    validation["problems"] = problems
    validation["checked"] = check

    # one more solve, outside the timed region, with every phase of the iteration bracketed by HIP events: the source of
    # phases_ms_per_iter and of the Gram roofline (the timed solves only bracket the sweep kernels - each recorded
    # event costs ~6 us between two kernels, tlsq_rpca_opts.phase_timing)
    _, rep_ph, _ = solve(phase_timing=True)
    barrier()
    ms_ph = dict(rep_ph.ms)

    if rank == 0:
        value = iters_total / dt
        # The sweep kernels of the timed solves between HIP events on the library's stream: the first shrink of every solve and
        # the fused sweep of every fourth iteration (an event record is a packet between two kernels, ~6 us: bracketing all of
        # them cost 3 % of the headline; tlsq_rpca_info.sweeps_timed launches, .hbm_bytes_sweeps_timed their algorithmic bytes).
        # Bytes a sweep has to move.  SURVEY.md §8d prices the two-kernel form (K1 R3/W2 + K2 R4/W2 = 11 passes per
        # iteration); the shipped loop is the E-free sweep (k_zsweep: A from its factors in registers, E never stored while
        # the loop runs): R D,Y,Z / W R,Y',Z' = 6 passes, 5 when the residual store is skipped, and k_first_shrink at k = 1:
        # R D / W Y,Z = 3 passes.  (The returned E is formed once after the loop, outside the two timed phases.)
        array_bytes = float(Ml) * N * 8
        sweep_ms_timed = ms["shrink"] + ms["update"]
        ms_per_sweep = sweep_ms_timed / max(sweeps_timed_total, 1)
        bytes_per_sweep = hbm_timed_total / max(sweeps_timed_total, 1)
        achieved = hbm_timed_total / (sweep_ms_timed * 1e-3) / 1e9
        passes_timed = hbm_timed_total / array_bytes
        assert abs(passes_timed - round(passes_timed)) < 1e-9 and 3 * args.steps <= passes_timed <= 6 * sweeps_timed_total
        alg_bytes = (3.0 * args.steps + 6.0 * iters_total - rskip_total) / iters_total * array_bytes   # all sweeps, per iteration
        sweep_ms_per_iter = alg_bytes / (achieved * 1e9) * 1e3
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
                       "call": "A, E, sv of a plain (non-verbose) call on device-resident panels.  `value` does NOT include the "
                               "returned decomposition s = (U, S, Vt) of the last Z (value_with_s does) nor PCIe transfers "
                               "(value_host_pointers does); no per-iteration cost history is requested, so opnorm(residual) is "
                               "only resolved far enough to settle cost < tol - identical iterations and outputs "
                               "(tests/test_gpu_parity.py::test_rpca_device_mode_and_decision_only_cost)"},
            **extras,
            "validation": validation,
            "roofline": {"kernel": "k_zsweep (E-free fused rebuild + ALM sweep) + k_first_shrink at k=1", "bound": "hbm",
                         "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "how": "algorithmic bytes of the bracketed launches / their device time between HIP events on the library's "
                                "stream, inside the timed solves (a sample: every solve's first shrink + the sweep of every 4th iteration)",
                         "launches_timed": sweeps_timed_total, "ms_per_launch": ms_per_sweep, "bytes_per_launch": bytes_per_sweep,
                         # what the library launched in the timed solves (tlsq_rpca_info.hbm_bytes_sweeps: algorithmic bytes of
                         # every sweep launch incl. the R stores it decided on), per iteration
                         "traffic": hbm_sweeps_total / iters_total,
                         "traffic_source": "tlsq_rpca_info.hbm_bytes_sweeps of the timed solves (bytes of the launches the library made); "
                                           "PMC cross-check in traffic_pmc",
                         "ms_per_iter": sweep_ms_per_iter, "algorithmic_bytes_per_iter": alg_bytes,
                         "passes_per_iter": alg_bytes / array_bytes, "sweeps_without_residual_store": rskip_total,
                         "survey_11_pass_equivalent_GBps": 11.0 * array_bytes / (sweep_ms_per_iter * 1e-3) / 1e9,
                         # the same time priced with SURVEY.md 8(d)'s own accounting (K1 + K2 = 11 panel passes per iteration):
                         # above 1 because the loop was re-derived to move 5.3 passes, not because anything runs above the
                         # HBM peak - `frac` (bytes actually moved / time / peak) is the roofline figure
                         "frac_vs_survey_11_pass": 11.0 * array_bytes / (sweep_ms_per_iter * 1e-3) / 1e9 / 8000.0},
            "phases_ms_per_iter": {k: v / rep_ph.iters_done for k, v in ms_ph.items()
                                   if k in ("shrink", "gram", "eig", "rebuild", "update", "opnorm")},
            "phases_source": "one extra solve with tlsq_rpca_opts.phase_timing = 1 (events at all nine phase boundaries: "
                             "that solve runs ~6 % slower than the timed ones)",
            "roofline_mfma": None,
            "jacobi_sweeps_per_solve": rep.jacobi_sweeps,
            "svd_step": {"tsqr_route": rep.tsqr_iterations, "subspace": rep.eig_fast, "subspace_steps": rep.subspace_steps},
            "hbm_bytes_per_iter_all_panel_kernels": hbm_total / iters_total,
            "extra": {"c4": c4, "c5": c5},
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
        # PMC cross-check of `traffic`: FETCH_SIZE (x2 on gfx950) + WRITE_SIZE of the sweep kernels from a counter run of this
        # same command (rocprofv3 --pmc in separate passes, tools/profile_round.sh; a counter pass cannot run inside this
        # process), committed under profiles/
        out["roofline"]["traffic_pmc"] = None
        for pmc_file in ("r06_pmc_sweeps.json", "r05_pmc_sweeps.json", "r04_pmc_sweeps.json", "r03_pmc_sweeps.json", "r02_pmc_sweeps.json"):
            try:
                with open(os.path.join(ROOT, "profiles", pmc_file)) as f:
                    pmc = json.load(f)
                if Ml == 20000 and N == 512:
                    sk = pmc["sweep_kernels"]
                    first = sk.get("k_first_shrink", sk.get("k_shrink"))
                    out["roofline"]["traffic_pmc"] = {
                        "bytes_per_iter": (first["hbm_bytes_per_launch"] + rep.iters_done * sk["k_zsweep"]["hbm_bytes_per_launch"]) / rep.iters_done,
                        "source": f"profiles/{pmc_file}"}
                break
            except (OSError, KeyError):
                continue
        # second roofline: the Gram kernel (fp64 MFMA, v_mfma_f64_16x16x4_f64).  ALGORITHMIC flops of G = Z'Z: the
        # N (N + 1) / 2 distinct entries, 2 M flop each.  The kernel issues more: the strictly lower 128 x 128 tiles in full, of
        # the diagonal tiles the 36 of 64 MFMA tiles on and below the diagonal
        nt = (N + 127) // 128
        gram_flops = 1.0 * Ml * N * (N + 1)
        issued_flops = 2.0 * Ml * 128 * 128 * (nt * (nt - 1) // 2 + nt * 36.0 / 64.0)
        if ms_ph.get("gram"):
            # Gram launches of the profiled solve: one per iteration, plus the one queued behind the last sweep
            # before its convergence is known (the library hides the host round trip behind it)
            n_gram = rep_ph.iters_done + 1
            tf = gram_flops / (ms_ph["gram"] / n_gram * 1e-3) / 1e12
            out["roofline_mfma"] = {"kernel": "k_gram_kc Gram(Z) + k_slab_reduce", "bound": "mfma",
                                    "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6,
                                    "flops_per_launch": gram_flops, "issued_flops_per_launch": issued_flops,
                                    "issued_TFLOPs": issued_flops / (ms_ph["gram"] / n_gram * 1e-3) / 1e12,
                                    "ms_per_launch": ms_ph["gram"] / n_gram, "launches": n_gram}
        if world == 1 and args.cpu_iters > 0:
            from oracle import rpca_oracle as O   # the checker, timed as the CPU baseline (nothing else of bench.py uses it)
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
        print(json.dumps(out), flush=True)
        for p_ in problems:
            print("bench.py: VALIDATION FAILED: " + p_, file=sys.stderr)
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    dog.cancel()
    if problems:
        sys.exit(2)


if __name__ == "__main__":
    main()
