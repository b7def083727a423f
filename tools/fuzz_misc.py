#!/usr/bin/env python3
"""Randomised sweep of the remaining entry points against the oracle: complex rpca, rpca through device pointers, rpca on a
loop-back group with uneven row blocks and shapes at the sharding limit, the returned SVD (U S Vt = last Z, S against LAPACK),
hook modes (opnorm by power iteration, randomized svd: the reference's own thresholds), tls / tls_ / rtls, hankel / unhankel /
soft_hankel_, the three spherical averages.
    python tools/fuzz_misc.py [seed] [ncases]"""
import os, sys, time, warnings
os.environ.setdefault("OPENBLAS_NUM_THREADS", "4")
os.environ.setdefault("OMP_NUM_THREADS", "4")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def lowrank(rng, M, N, r, frac=0.05, cplx=False):
    g = (lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)) if cplx else (lambda *s: rng.standard_normal(s))
    D = g(M, r) @ g(r, N)
    mask = rng.random((M, N)) < frac
    return D + 10 * g(M, N) * mask


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def main():
    import torch
    import tlsq_amd
    from oracle import rpca_oracle as O
    from oracle import ga_oracle as G
    tlsq_amd.dev_from_env()
    warnings.simplefilter("ignore")
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=4)
    except Exception:   # noqa: BLE001
        pass
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    budget = float(os.environ.get("FUZZ_BUDGET_S", "300"))
    rng = np.random.default_rng(seed)
    eng = tlsq_amd.Engine(0)
    groups = {n: tlsq_amd.Engine(devices=[0] * n) for n in (2, 3, 5)}
    kinds = ["complex", "device", "group", "returned_s", "opnorm_power", "rsvd", "tls", "rtls", "hankel", "averages",
             "returned_s_f32", "batched_f32", "ga_group", "options"]
    bad = done = 0
    worst = 0.0
    t0 = time.time()
    for it in range(ncase):
        kind = str(rng.choice(kinds))
        desc = kind
        try:
            err, tol = 0.0, 1e-8
            if kind == "complex":
                M, N = int(rng.choice([20, 60, 200, 500])), int(rng.choice([5, 12, 40, 90]))
                r = int(rng.integers(1, max(2, min(M, N) // 4 + 1)))
                D = lowrank(rng, M, N, r, cplx=True)
                kw = dict(nukeA=bool(rng.random() < 0.7))
                desc = f"{kind} {M}x{N} r={r} {kw}"
                A, E, s, sv, rep = eng.rpca(D, return_report=True, **kw)
                Ao, Eo, so, svo, io = O.rpca(D, **kw)
                err = max(rel(A, Ao), rel(E, Eo))
                if rep.iters_done != io.iters_done or rep.svp_hist != io.svp_hist or sv != svo:
                    k = next((i for i, (x, y) in enumerate(zip(rep.svp_hist, io.svp_hist)) if x != y), None)
                    desc += f" iters {rep.iters_done}/{io.iters_done} sv {sv}/{svo} first svp diff k={k} errA={err:.1e}"
                    err = 1.0
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    np.savez(os.path.join(ROOT, "gpurun_out", f"fuzz_misc_fail_{seed}_{it}.npz"), D=D, nukeA=kw["nukeA"],
                             svp_gpu=np.array(rep.svp_hist), svp_ref=np.array(io.svp_hist))
            elif kind == "device":
                M, N = int(rng.choice([64, 333, 1000, 4000])), int(rng.choice([8, 33, 64, 130]))
                r = int(rng.integers(1, max(2, min(M, N) // 4 + 1)))
                D = lowrank(rng, M, N, r)
                desc = f"{kind} {M}x{N} r={r}"
                dD = torch.from_numpy(np.ascontiguousarray(D.T)).cuda()
                dA, dE = torch.empty_like(dD), torch.empty_like(dD)
                sv, rep, st = eng.rpca_device(dD.data_ptr(), M, N, dA.data_ptr(), dE.data_ptr(),
                                              want_hist=bool(rng.random() < 0.5))
                Ao, Eo, so, svo, io = O.rpca(D)
                err = max(rel(dA.cpu().numpy().T, Ao), rel(dE.cpu().numpy().T, Eo))
                if rep.iters_done != io.iters_done or sv != svo:
                    err = 1.0
            elif kind == "group":
                n = int(rng.choice([2, 3, 5]))
                N = int(rng.choice([5, 16, 40, 100]))
                M = int(rng.choice([max(32 * n, N), 32 * n + 1, 32 * n + 17, 777, 2003]))
                M = max(M, N)
                r = int(rng.integers(1, max(2, min(M, N) // 4 + 1)))
                D = lowrank(rng, M, N, r)
                kw = dict(nukeA=bool(rng.random() < 0.7), nonnegE=bool(rng.random() < 0.1))
                if rng.random() < 0.15:
                    kw["hankel"] = True
                desc = f"{kind} n={n} {M}x{N} r={r} {kw}"
                A, E, s, sv, rep = groups[n].rpca(D, return_report=True, **kw)
                Ao, Eo, so, svo, io = O.rpca(D, **kw)
                err = max(rel(A, Ao), rel(E, Eo))
                if rep.iters_done != io.iters_done or rep.svp_hist != io.svp_hist or sv != svo:
                    err = 1.0
                Zr = (np.asarray(s.U) * s.S) @ np.asarray(s.Vt)
                Zo = (so[0] * so[1]) @ so[2]
                err = max(err, rel(Zr, Zo))
            elif kind == "returned_s":
                M, N = int(rng.choice([30, 200, 900])), int(rng.choice([7, 30, 64, 150]))
                if rng.random() < 0.3:
                    M, N = N, M   # wide: solved as the transpose
                r = int(rng.integers(1, max(2, min(M, N) // 4 + 1)))
                D = lowrank(rng, M, N, r)
                desc = f"{kind} {M}x{N} r={r}"
                A, E, s, sv, rep = eng.rpca(D, return_report=True)
                Ao, Eo, so, svo, io = O.rpca(D)
                Zr = (np.asarray(s.U) * s.S) @ np.asarray(s.Vt)
                Zo = (so[0] * so[1]) @ so[2]
                d = min(M, N)
                parts = dict(Z=rel(Zr, Zo), S=float(np.max(np.abs(s.S - so[1])) / so[1][0]) * 1e2,
                             VtV=float(np.linalg.norm(np.asarray(s.Vt) @ np.asarray(s.Vt).T - np.eye(d))) * 1e2,
                             UtU=float(np.linalg.norm(np.asarray(s.U).T @ np.asarray(s.U) - np.eye(d))) * 1e2)
                err = max(parts.values())
                desc += " " + " ".join(f"{k}={v:.1e}" for k, v in parts.items()) + f" Smin/Smax={s.S[-1] / s.S[0]:.1e}"
            elif kind == "opnorm_power":   # rnorm-style estimate (src/robustPCA.jl, test/runtests.jl:163-170): still a valid decomposition
                M, N = int(rng.choice([200, 600])), int(rng.choice([20, 60]))
                D = lowrank(rng, M, N, 3)
                desc = f"{kind} {M}x{N}"
                A, E, s, sv = eng.rpca(D, opnorm=("power", 20))
                err = rel(A + E, D)
                tol = 1e-4
            elif kind == "rsvd":
                M, N = int(rng.choice([300, 1000])), int(rng.choice([40, 100]))
                r = int(rng.integers(2, 6))
                L0 = rng.standard_normal((M, r)) @ rng.standard_normal((r, N))
                D = L0 + 10 * rng.standard_normal((M, N)) * (rng.random((M, N)) < 0.02)
                desc = f"{kind} {M}x{N} r={r}"
                # the reference's svd(Z, sv) hook: rank-sv decompositions from k = 2 on, so the rank estimate cannot grow
                # after the first iteration (src/robustPCA.jl:193-204) - compared with the oracle under the exact rank-sv hook
                A, E, s, sv, rep = eng.rpca(D, svd="randomized", return_report=True)

                def trunc(Z, k):
                    U, S, Vt = O._svd_full(Z)
                    return U[:, :k], S[:k], Vt[:k]
                Ao, Eo, so, svo, io = O.rpca(D, svd=trunc)
                err = rel(A, Ao)
                if sv != svo:
                    err = 1.0
                desc += f" iters {rep.iters_done}/{io.iters_done} sv {sv}/{svo}"
                tol = 1e-5
            elif kind == "tls":
                M, n, q = int(rng.choice([10, 50, 400])), int(rng.integers(1, 9)), int(rng.integers(1, 3))
                A = rng.standard_normal((M, n))
                y = A @ rng.standard_normal((n, q)) + 0.01 * rng.standard_normal((M, q))
                yy = y[:, 0] if q == 1 else y
                desc = f"{kind} {M}x{n} q={q}"
                err = rel(np.asarray(eng.tls(A, yy)).reshape(n, -1), np.asarray(O.tls(A, yy)).reshape(n, -1))
                tol = 1e-9
            elif kind == "rtls":
                M, n = int(rng.choice([30, 100, 500])), int(rng.integers(1, 8))
                A = rng.standard_normal((M, n))
                y = A @ rng.standard_normal(n) + 0.01 * rng.standard_normal(M)
                y[rng.random(M) < 0.05] += 5.0
                desc = f"{kind} {M}x{n}"
                err = rel(np.asarray(eng.rtls(A, y)).ravel(), np.asarray(O.rtls(A, y)).ravel())
                tol = 1e-6
            elif kind == "hankel":
                Nx, Dch = int(rng.choice([20, 97, 1000])), int(rng.choice([1, 1, 2, 3]))
                L_ = int(rng.integers(2, max(3, Nx // 3)))
                lag = int(rng.integers(1, min(L_, 4) + 1))
                x = rng.standard_normal((Nx, Dch)) if Dch > 1 else rng.standard_normal(Nx)
                desc = f"{kind} Nx={Nx} D={Dch} L={L_} lag={lag}"
                H = eng.hankel(x, L_, lag)
                Ho = O.hankel(x, L_, lag)
                err = 0.0 if np.array_equal(H, Ho) else 1.0
                Ap = Ho + 0.1 * rng.standard_normal(Ho.shape)
                u = eng.unhankel(Ap, lag, Nx, Dch)
                uo = O.unhankel(Ap, lag, Nx, Dch)
                err = max(err, rel(u, uo) * 1e4)   # (tolerance 1e-12)
                if lag == 1 and Dch == 1:
                    B1, B2 = Ap.copy(order="F"), Ap.copy(order="F")
                    eng.soft_hankel_(B1, 0.05)
                    O.soft_hankel_(B2, 0.05)
                    err = max(err, rel(B1, B2) * 1e4)
            elif kind == "returned_s_f32":
                M, N = int(rng.choice([30, 200, 900])), int(rng.choice([7, 30, 64]))
                if rng.random() < 0.3:
                    M, N = N, M
                r = int(rng.integers(1, max(2, min(M, N) // 4 + 1)))
                D = lowrank(rng, M, N, r).astype(np.float32)
                A, E, s, sv = eng.rpca(D)
                d = min(M, N)
                U, S, Vt = (np.asarray(x, dtype=np.float64) for x in (s.U, s.S, s.Vt))
                Ao, Eo, so, svo, io = O.rpca(D)
                Zo = (so[0].astype(np.float64) * so[1].astype(np.float64)) @ so[2].astype(np.float64)
                parts = dict(UtU=float(np.linalg.norm(U.T @ U - np.eye(d))), VtV=float(np.linalg.norm(Vt @ Vt.T - np.eye(d))),
                             rec=rel((U * S) @ Vt, Zo) * 1e-1)   # (fp32 trajectories agree to ~1e-3)
                err = max(parts.values())
                desc = f"{kind} {M}x{N} r={r} " + " ".join(f"{k}={v:.1e}" for k, v in parts.items())
                tol = 2e-4
            elif kind == "batched_f32":
                B, M, N = int(rng.integers(2, 30)), int(rng.choice([50, 200, 500])), int(rng.integers(2, 33))
                r = int(rng.integers(1, max(2, N // 3 + 1)))
                Ds = np.stack([lowrank(rng, M, N, r) for _ in range(B)]).astype(np.float32)
                desc = f"{kind} B={B} {M}x{N} r={r}"
                A, E, S, Vt, svb, itb, stb, costb = eng.rpca_batched(Ds)
                for b in range(min(B, 8)):
                    Ao, Eo, _, svo, io = O.rpca(Ds[b])
                    err = max(err, rel(A[b].astype(np.float64), Ao.astype(np.float64)))
                    if abs(int(itb[b]) - io.iters_done) > 1 or int(svb[b]) != svo:
                        err = max(err, 1.0)
                tol = 2e-3
            elif kind == "ga_group":
                n = int(rng.choice([2, 3, 5]))
                d, N, r = int(rng.choice([5, 20, 100])), int(rng.choice([400, 1001, 3000])), int(rng.integers(1, 4))
                X = rng.standard_normal((d, r)) @ (rng.standard_normal((r, N)) * np.linspace(4.0, 2.0, r)[:, None]) \
                    + 0.05 * rng.standard_normal((d, N))
                X[:, rng.random(N) < 0.02] *= 10.0
                q0 = rng.standard_normal((d, r))
                mode = [None, "entrywise_trimmed_mean", "entrywise_median"][int(rng.integers(0, 3))]
                desc = f"{kind} n={n} d={d} N={N} r={r} {mode}"
                kw = {"mu": mode} if mode else {}
                got, rep = groups[n].rpca_ga(X, r, q0=q0, iters=50, return_report=True, **kw)
                info = G.GaInfo()
                want = G.rpca_ga(X, r, q0=q0, info=info, iters=50, **({"mu": getattr(G, mode)} if mode else {}))
                err = float(np.abs(got - want).max())
                if rep["iters"] != info.iters:
                    err = 1.0
                tol = 1e-9
            elif kind == "options":
                M, N = int(rng.choice([100, 400, 1200])), int(rng.choice([10, 40, 100]))
                r = int(rng.integers(1, max(2, min(M, N) // 4 + 1)))
                D = lowrank(rng, M, N, r)
                kw = dict(lam=float(rng.uniform(0.02, 0.3)), rho=float(rng.choice([1.2, 1.5, 2.0])), tol=float(rng.choice([1e-5, 1e-7, 1e-9])),
                          iters=int(rng.choice([5, 30, 1000])), maxrank=int(rng.choice([1, 3, 1000])))
                desc = f"{kind} {M}x{N} r={r} {kw}"
                A, E, s, sv, rep = eng.rpca(D, return_report=True, **kw)
                Ao, Eo, so, svo, io = O.rpca(D, **kw)
                err = max(rel(A, Ao), rel(E, Eo))
                if rep.iters_done != io.iters_done or rep.svp_hist != io.svp_hist or sv != svo:
                    desc += f" iters {rep.iters_done}/{io.iters_done} sv {sv}/{svo}"
                    err = 1.0
            else:   # averages
                d, N = int(rng.choice([3, 10, 64, 300])), int(rng.choice([2, 7, 100, 5000]))
                U = rng.standard_normal((d, N))
                U[:, rng.integers(0, N)] = U[:, 0]
                w = rng.standard_normal(N)
                desc = f"{kind} d={d} N={N}"
                for fn, gn in ((eng.mu_, G.mu_mean), (eng.entrywise_trimmed_mean, G.entrywise_trimmed_mean),
                               (eng.entrywise_median, G.entrywise_median)):
                    a = fn(np.zeros(d), w, U)
                    b = gn(np.zeros(d), w, U)
                    ok = np.isfinite(b)
                    err = max(err, float(np.max(np.abs(a[ok] - b[ok]) / (1e-300 + np.maximum(1.0, np.abs(b[ok]))))) * 1e2 if ok.any() else 0.0)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print(f"case {it} {desc}: EXCEPTION {type(e).__name__}: {e}", flush=True)
            continue
        done += 1
        worst = max(worst, err / tol)
        if not (err <= tol):
            bad += 1
            print(f"case {it} {desc}: err {err:.2e} > {tol:.0e}", flush=True)
        if time.time() - t0 > budget:
            print("time budget reached at case", it, flush=True)
            break
    print(f"{done} cases, {bad} bad, worst err/tol {worst:.2e}, {time.time()-t0:.0f}s")


if __name__ == "__main__":
    main()
