#!/usr/bin/env python3
"""Development tool: the fused sweep + Gram kernel (csrc/fused.hip, tlsq_k_zsweep_gram_f64) against the two kernels it
replaces (tlsq_k_zsweep_f64 then tlsq_k_gram_f64) on the same device buffers - Y_{k+1}, Z_{k+1}, R_k must be bit-identical,
G equal to summation order - and the time of both forms.
    python tools/dbg/fused_check.py [--cases 20000x512x16,1000000x256x3] [--reps 20] [--hankel]"""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import tlsq_amd


def timeit(eng, fn, reps):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.3:
        for _ in range(5):
            fn()
        eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    eng.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="20000x512x16,65536x256x8,200000x512x16,1000000x256x3")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--hankel", action="store_true")
    ap.add_argument("--minrows", type=int, default=0)
    ap.add_argument("--ablate", default="", help="comma list of FUSED_ABLATE values to time as well")
    a = ap.parse_args()
    torch.zeros(1, device="cuda")
    eng = tlsq_amd.Engine(0)
    lib, h = eng.lib, eng.h
    lib.tlsq_dev_set(b"FUSED_ZGRAM_N512", b"1")   # (N = 512: the diagonal 256-column blocks only - timing, G mismatches)
    if a.minrows:
        lib.tlsq_dev_set(b"FUSED_ZGRAM_MINROWS", str(a.minrows).encode())
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    bad = 0
    for case in a.cases.split(","):
        M, N, r = (int(v) for v in case.split("x"))
        g = torch.Generator(device="cuda").manual_seed(M + N + r)
        rnd = lambda *shape: torch.randn(shape, dtype=torch.float64, device="cuda", generator=g)
        hy = None
        if a.hankel:
            hy = rnd(M + N)
            K = M - 6                       # (a few zero pad rows, as lowrankfilter's panels have)
            idx = torch.arange(M, device="cuda")[None, :] + torch.arange(N, device="cuda")[:, None]
            D = hy[idx] * (torch.arange(M, device="cuda")[None, :] < K)
            D = D.contiguous()
        else:
            K = 0
            D = rnd(N, M)                   # column-major M x N
        Y, Z = rnd(N, M), rnd(N, M)
        mask = (torch.rand((N, M), device="cuda", generator=g) < 0.3).double()
        Z = Z * mask + D * 0.5              # some entries shrink to zero, some do not
        Tm = rnd(max(r, 1), M)
        Vs = rnd(max(r, 1), N) / max(r, 1) ** 0.5
        mu, mu_n, lam = 0.27, 0.405, 0.1
        inv_mu, inv_mu_n, thr_n = 1.0 / mu, 1.0 / mu_n, lam / mu_n
        # reference: the two kernels
        Y1, Z1, R1 = torch.empty_like(Y), Z.clone(), torch.empty_like(Y)
        G1 = torch.empty((N, N), dtype=torch.float64, device="cuda")
        ss1 = torch.zeros(72, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()   # (the library runs on its own stream)
        assert lib.tlsq_k_zsweep_f64(h, p(D), p(Tm), p(Vs), None, p(Y), p(Y1), p(Z1), p(R1), M, N, r, mu, inv_mu, 0, inv_mu_n,
                                     thr_n, 0, p(ss1)) == 0
        assert lib.tlsq_k_gram_f64(h, p(Z1), M, N, M, p(G1), N) == 0
        eng.synchronize()
        Y2, Z2, R2 = torch.empty_like(Y), torch.empty_like(Y), torch.empty_like(Y)
        G2 = torch.empty((N, N), dtype=torch.float64, device="cuda")
        ss2 = torch.zeros(72, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()

        def fused(R=R2, ss=ss2):
            return lib.tlsq_k_zsweep_gram_f64(h, None if a.hankel else p(D), p(Tm), p(Vs), p(Y), p(Y2), p(Z), p(Z2), p(R), M, N, r, mu,
                                              inv_mu, 0, inv_mu_n, thr_n, 0, p(ss), p(hy), K, p(G2), N)
        st = fused()
        eng.synchronize()
        if st != 0:
            print(f"{case}: status {st}: {eng.last_error() if hasattr(eng, 'last_error') else lib.tlsq_last_error(h)}")
            bad += 1
            continue
        eqY, eqZ, eqR = (bool(torch.equal(x, y)) for x, y in ((Y1, Y2), (Z1, Z2), (R1, R2)))
        gerr = ((G1 - G2).abs().max() / G1.abs().max()).item()
        gsym = (G2 - G2.T).abs().max().item()
        Gref = Z1 @ Z1.T
        gerr_ref = ((Gref - G2).abs().max() / Gref.abs().max()).item()
        s1, s2 = ss1[:64].sum().item(), ss2[:64].sum().item()
        m2 = ss2[64:65].view(torch.int64).view(torch.float64).item()
        mref = R1.abs().max().item()
        ok = eqY and eqZ and eqR and gerr < 1e-13 and gsym == 0.0 and abs(s1 - s2) <= 1e-11 * abs(s1) + 1e-300 and m2 == mref
        bad += 0 if ok else 1
        print(f"{case}{' hankel' if a.hankel else ''}: Y {eqY} Z {eqZ} R {eqR}  |G - G_2k|/|G| {gerr:.1e}  vs torch {gerr_ref:.1e}  asym {gsym:.1e}  "
              f"sumsq rel {abs(s1 - s2) / max(abs(s1), 1e-300):.1e}  max {m2 == mref}  -> {'OK' if ok else 'MISMATCH'}")
        # timing: the pair against the fused kernel (with and without the residual store)
        Zt = Z.clone()
        torch.cuda.synchronize()

        def pair(R):
            lib.tlsq_k_zsweep_f64(h, p(D), p(Tm), p(Vs), None, p(Y), p(Y1), p(Zt), p(R), M, N, r, mu, inv_mu, 0, inv_mu_n, thr_n, 0,
                                  p(ss1))
            lib.tlsq_k_gram_f64(h, p(Zt), M, N, M, p(G1), N)
        t_pair = timeit(eng, lambda: pair(None), a.reps)
        t_sweep = timeit(eng, lambda: lib.tlsq_k_zsweep_f64(h, p(D), p(Tm), p(Vs), None, p(Y), p(Y1), p(Zt), None, M, N, r, mu,
                                                           inv_mu, 0, inv_mu_n, thr_n, 0, p(ss1)), a.reps)
        t_gram = timeit(eng, lambda: lib.tlsq_k_gram_f64(h, p(Zt), M, N, M, p(G1), N), a.reps)
        t_f = timeit(eng, lambda: fused(None), a.reps)
        t_fr = timeit(eng, lambda: fused(R2), a.reps)
        for ab in [v for v in a.ablate.split(",") if v]:
            lib.tlsq_dev_set(b"FUSED_ABLATE", ab.encode())
            print(f"    ablate {ab}: {timeit(eng, lambda: fused(None), a.reps):.1f} us")
        lib.tlsq_dev_set(b"FUSED_ABLATE", None)
        passes = 4 if a.hankel else 5
        gb = passes * M * N * 8
        print(f"    sweep {t_sweep:.1f} us + gram {t_gram:.1f} us (pair {t_pair:.1f}) | fused {t_f:.1f} us ({gb / t_f / 1e6:.2f} TB/s of {passes} passes, "
              f"{M * N * (N + 1) / t_f / 1e6:.1f} TF) | fused + R store {t_fr:.1f} us")
    print("FAILED" if bad else "all OK")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
