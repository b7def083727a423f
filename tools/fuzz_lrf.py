#!/usr/bin/env python3
"""Randomised parity sweep of the entry points around rpca: lowrankfilter (series length, channels, window n, lag, SSA rank,
flags - the implicit Hankel panel with any lag / channel count included), the batched tiny-problem kernels (rtls / rpca
stacks up to 32 columns), the fp32 solver, the group handle (row shards through the loop-back communicator).  GPU against
the oracle; every case that differs in iteration count / rank trajectory or by more than its tolerance is printed.
    python tools/fuzz_lrf.py [seed] [ncases] [--tiny]"""
import os, sys, time, warnings
os.environ.setdefault("OPENBLAS_NUM_THREADS", "4")
os.environ.setdefault("OMP_NUM_THREADS", "4")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


TINY = "--tiny" in sys.argv   # adds series of 8 ... 61 samples (windows up to N/2, lags up to the window)
if TINY:
    sys.argv.remove("--tiny")


def series(rng, Nx, Dch):
    t = np.arange(Nx)
    y = np.zeros((Nx, Dch))
    for c in range(Dch):
        for _ in range(int(rng.integers(1, 4))):
            y[:, c] += rng.uniform(0.5, 2.0) * np.sin(2 * np.pi * t / rng.uniform(8, 90) + rng.uniform(0, 6))
        y[:, c] += rng.uniform(0, 0.02) * t / Nx
    y += float(rng.choice([0.0, 1e-3, 0.05])) * rng.standard_normal((Nx, Dch))
    out = rng.random((Nx, Dch)) < float(rng.choice([0.0, 0.01, 0.05]))
    y[out] += 5.0 * rng.standard_normal(int(out.sum()))
    return y if Dch > 1 else y[:, 0]


def main():
    import tlsq_amd
    from oracle import rpca_oracle as O
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
    grp = tlsq_amd.Engine(devices=[0, 0, 0])
    bad = done = 0
    worst = 0.0
    t0 = time.time()
    for it in range(ncase):
        kind = str(rng.choice(["lrf", "lrf", "lrf", "lrf_f32", "lrf_group", "batched_rpca", "batched_rtls", "rpca_f32"]))
        desc = kind
        try:
            if kind.startswith("lrf"):
                Nx = int(rng.choice([120, 257, 600, 1500, 4000] + ([8, 13, 30, 61] if TINY else [])))
                Dch = int(rng.choice([1, 1, 2, 3]))
                if Nx < 100:   # tiny series: windows up to N/2 (the reference's assertion), lags up to the window
                    n = int(rng.integers(2, Nx // 2 + 1))
                    lag = int(rng.integers(1, n + 1))
                else:
                    n = int(rng.integers(4, max(5, min(Nx // 4, 48))))
                    lag = int(rng.integers(1, min(n, 4) + 1))
                sv = int(rng.choice([0, 0, 0, 2]))
                kw = {}
                if sv == 0 and rng.random() < 0.2:
                    kw["hankel"] = True
                if sv == 0 and rng.random() < 0.2:
                    kw["nukeA"] = False
                if rng.random() < 0.3:
                    kw["tol"] = float(rng.choice([1e-2, 1e-4]))
                y = series(rng, Nx, Dch)
                desc = f"{kind} Nx={Nx} D={Dch} n={n} lag={lag} sv={sv} {kw}"
                want = O.lowrankfilter(y, n, sv=sv, lag=lag, **kw)
                if kind == "lrf_f32":
                    got = eng.lowrankfilter(y.astype(np.float32), n, sv=sv, lag=lag, **kw)
                    tol = 5e-3
                elif kind == "lrf_group":
                    got = grp.lowrankfilter(y, n, sv=sv, lag=lag, **kw)
                    tol = 1e-7
                else:
                    got = eng.lowrankfilter(y, n, sv=sv, lag=lag, **kw)
                    tol = 1e-7
                err = float(np.linalg.norm(np.asarray(got, dtype=np.float64) - want) / max(np.linalg.norm(want), 1e-300))
            elif kind == "batched_rpca":
                B, M, N = int(rng.integers(2, 40)), int(rng.choice([20, 50, 200, 500])), int(rng.integers(2, 33))
                N = min(N, M)   # (the batched kernel takes tall problems)
                r = int(rng.integers(1, max(2, N // 3 + 1)))
                Ds = np.stack([rng.standard_normal((M, r)) @ rng.standard_normal((r, N))
                               + 10 * rng.standard_normal((M, N)) * (rng.random((M, N)) < 0.05) for _ in range(B)])
                desc = f"{kind} B={B} {M}x{N} r={r}"
                A, E, S, Vt, svb, itb, stb, costb = eng.rpca_batched(Ds)
                err = 0.0
                for b in range(B):
                    Ao, Eo, _, svo, io = O.rpca(Ds[b])
                    if int(itb[b]) != io.iters_done or int(svb[b]) != svo:
                        err = max(err, 1.0)
                    err = max(err, float(np.linalg.norm(A[b] - Ao) / max(np.linalg.norm(Ds[b]), 1e-300)))
                tol = 1e-8
            elif kind == "batched_rtls":
                B, M, n = int(rng.integers(2, 60)), int(rng.choice([30, 50, 200, 500])), int(rng.integers(2, 12))
                As = rng.standard_normal((B, M, n))
                x0 = rng.standard_normal((B, n))
                ys = np.einsum("bmn,bn->bm", As, x0) + 0.01 * rng.standard_normal((B, M))
                ys[rng.random((B, M)) < 0.03] += 5.0
                desc = f"{kind} B={B} {M}x({n}+1)"
                X = eng.rtls_batched(As, ys)
                err = 0.0
                for b in range(min(B, 12)):
                    xo = O.rtls(As[b], ys[b])
                    err = max(err, float(np.linalg.norm(np.asarray(X[b]).ravel() - np.asarray(xo).ravel())
                                         / max(np.linalg.norm(xo), 1e-300)))
                tol = 1e-6
            else:   # rpca_f32: same tolerances as the fp32 parity tests (tol = sqrt(eps32))
                M, N = int(rng.choice([100, 300, 1500])), int(rng.choice([17, 40, 100]))
                r = int(rng.integers(1, max(2, min(M, N) // 4 + 1)))
                D = (rng.standard_normal((M, r)) @ rng.standard_normal((r, N))
                     + 10 * rng.standard_normal((M, N)) * (rng.random((M, N)) < 0.05)).astype(np.float32)
                desc = f"{kind} {M}x{N} r={r}"
                A, E, s, sv, rep = eng.rpca(D, return_report=True)
                Ao, Eo, so, svo, io = O.rpca(D)
                err = float(np.linalg.norm(A.astype(np.float64) - Ao.astype(np.float64)) / max(np.linalg.norm(D), 1e-300))
                if abs(rep.iters_done - io.iters_done) > 1 or sv != svo:
                    err = max(err, 1.0)
                tol = 2e-3
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
