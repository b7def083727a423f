#!/usr/bin/env python3
"""Randomised parity sweep: many small/medium rpca problems (shapes, ranks, noise, flags) GPU vs oracle.
Reports every case whose iteration count / svp history / sv differs or whose A,E error exceeds 1e-8.
    python tools/fuzz_parity.py [seed] [ncases]
Also imported by tests/test_gpu_parity.py::test_fuzz_parity."""
import os, sys, time, warnings
os.environ.setdefault("OPENBLAS_NUM_THREADS", "4")   # LAPACK on tiny matrices crawls with 256 threads
os.environ.setdefault("OMP_NUM_THREADS", "4")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


UNSTABLE = [0]   # mismatches on inputs whose reference trajectory LAPACK itself does not reproduce (reference_unstable)
BIG = False   # --big: panels of 200 ... 512 columns, ranks up to 40 (warm subspace blocks of up to 50 columns, the fused
              # Rayleigh-Ritz kernel on blocks of up to 32); ~20 s of LAPACK per case on the host


def make_case(rng):
    if BIG:
        M = int(rng.choice([1500, 3000, 5000]))
        N = int(rng.choice([200, 256, 384, 512]))
        r = int(rng.integers(2, 41))
    else:
        M = int(rng.choice([5, 8, 20, 50, 51, 100, 300, 600, 1500]))
        N = int(rng.choice([3, 4, 5, 8, 17, 40, 64, 100, 130]))
        r = int(rng.integers(1, max(2, min(M, N) // 3 + 1)))
    noise = float(rng.choice([0.0, 0.0, 1e-6, 1e-3, 1e-1]))
    frac = float(rng.choice([0.0, 0.02, 0.1, 0.3]))
    scale = float(rng.choice([1e-3, 1.0, 1e4]))
    D = (rng.standard_normal((M, r)) @ rng.standard_normal((r, N))
         + 10 * rng.standard_normal((M, N)) * (rng.random((M, N)) < frac)
         + noise * rng.standard_normal((M, N))) * scale
    kw = dict(nukeA=bool(rng.random() < 0.7), nonnegA=bool(rng.random() < 0.1), nonnegE=bool(rng.random() < 0.1),
              iters=int(rng.choice([120, 60])))
    if rng.random() < 0.2:
        kw["lam"] = float(rng.uniform(0.02, 0.5))
    if rng.random() < 0.12:
        kw["hankel"] = True   # soft_hankel!(A, lambda/mu) inside the loop and soft_hankel!(E, .) after it (:214-216, :234-236)
    return D, kw, f"{M}x{N} r={r} noise={noise} frac={frac} scale={scale}"


def reference_unstable(D, kw, err_gpu):
    """A mismatch is only a finding if the reference's own trajectory is stable on that input.  Replays the case with LAPACK's
    other dense driver (gesvd) and with one ulp added to Z at k = 5 (see tools/knife_edge.py): when those runs end as far from
    the gesdd run as the GPU did (or part ways in their counts), the input is 'reference-unstable' - a run that does not
    converge and amplifies rounding differences - and says nothing about the implementation.  -> (unstable, description)"""
    from oracle import rpca_oracle as O
    import scipy.linalg as sla
    A0, E0, _, _, i0 = O.rpca(D, **kw)
    dn = max(np.linalg.norm(D), 1e-300)
    out = []
    cnt = [0]

    def gesvd(Z, sv):
        return sla.svd(Z, full_matrices=False, lapack_driver="gesvd", check_finite=False)

    def ulp(Z, sv):
        cnt[0] += 1
        return O._svd_full(np.nextafter(Z, np.inf) if cnt[0] == 4 else Z)   # (the hook sees k = 2, 3, ...: the 4th call is k = 5)

    unstable = False
    for name, f in (("gesvd", gesvd), ("Z+1ulp@k=5", ulp)):
        A, E, _, _, info = O.rpca(D, svd=f, **kw)
        kd = next((i + 1 for i, (a, b) in enumerate(zip(info.svp_hist, i0.svp_hist)) if a != b), None)
        ea = np.linalg.norm(A - A0) / dn
        out.append(f"{name}: counts differ at k={kd}, errA={ea:.1e}")
        unstable = unstable or kd is not None or ea > 0.3 * err_gpu
    return unstable, "; ".join(out)


def run_cases(eng, seed, ncase, budget_s=300.0, verbose=True):
    """returns (cases_run, trajectory_mismatches, exceptions, worst_err)"""
    from oracle import rpca_oracle as O
    warnings.simplefilter("ignore")
    try:   # LAPACK on tiny matrices crawls with one thread per host core (256 on the GPU box)
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=4)
    except Exception:   # noqa: BLE001
        pass
    rng = np.random.default_rng(seed)
    bad = exc = done = 0
    worst = 0.0
    t0 = time.time()
    for it in range(ncase):
        D, kw, desc = make_case(rng)
        try:
            A, E, s, sv, rep = eng.rpca(D, return_report=True, **kw)
        except Exception as e:   # noqa: BLE001
            exc += 1
            if verbose:
                print(f"case {it} {desc} {kw}: GPU EXCEPTION {e}", flush=True)
            continue
        Ao, Eo, so, svo, io = O.rpca(D, **kw)
        done += 1
        dn = max(np.linalg.norm(D), 1e-300)
        ea, ee = np.linalg.norm(A - Ao) / dn, np.linalg.norm(E - Eo) / dn
        same = (rep.iters_done == io.iters_done) and (rep.svp_hist == io.svp_hist) and sv == svo
        worst = max(worst, ea, ee)
        if not same or ea > 1e-8 or ee > 1e-8:
            bad += 1
            k = next((i for i, (a, b) in enumerate(zip(rep.svp_hist, io.svp_hist)) if a != b), None)
            if verbose:
                print(f"case {it} {desc} {kw}: iters {rep.iters_done}/{io.iters_done} sv {sv}/{svo} "
                      f"first svp diff at k={k} errA={ea:.1e} errE={ee:.1e}", flush=True)
                try:
                    un, why = reference_unstable(D, kw, max(ea, ee))
                    UNSTABLE[0] += int(un)
                    print(f"    reference-unstable input: {un}  (LAPACK against itself - {why})", flush=True)
                except Exception as e:   # noqa: BLE001
                    print("    (stability check failed:", e, ")", flush=True)
        if time.time() - t0 > budget_s:
            if verbose:
                print("time budget reached at case", it, flush=True)
            break
    return done, bad, exc, worst


if __name__ == "__main__":
    import tlsq_amd
    tlsq_amd.dev_from_env()
    if "--big" in sys.argv:
        sys.argv.remove("--big")
        BIG = True
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    eng = tlsq_amd.Engine(0)
    t0 = time.time()
    done, bad, exc, worst = run_cases(eng, seed, ncase, float(os.environ.get("FUZZ_BUDGET_S", "300")))
    print(f"{done} cases, {bad} mismatches ({UNSTABLE[0]} of them on reference-unstable inputs), {exc} exceptions, "
          f"worst err {worst:.2e}, {time.time()-t0:.0f}s")
