#!/usr/bin/env python3
"""Micro-benchmarks of the kernel-level entry points (device pointers) — development tool.
   python tools/kbench.py [gram|gemm|sweeps|symeig|all] [--M 20000 --N 512 --reps 20]"""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import tlsq_amd


def timeit(eng, fn, reps):
    # the GPU idles at a low core clock: run ~0.3 s of the kernel itself before timing
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.3:
        for _ in range(10):
            fn()
        eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    eng.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--M", type=int, default=20000)
    ap.add_argument("--N", type=int, default=512)
    ap.add_argument("--r", type=int, default=16)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--f32", type=int, default=None, help="gram of an fp32 panel: 0 widened to the fp64 MFMA, 1 fp32 MFMA + fp64 fold-in")
    a = ap.parse_args()
    M, N, r = a.M, a.N, a.r
    torch.zeros(1, device="cuda")
    eng = tlsq_amd.Engine(0)
    lib, h = eng.lib, eng.h
    g = torch.Generator(device="cuda").manual_seed(0)
    Z = torch.randn((N, M), dtype=torch.float64, device="cuda", generator=g)      # column-major M x N
    p = lambda t: C.c_void_p(t.data_ptr())
    if a.what == "gram" and a.f32 is not None:
        Zf = Z.float()
        G = torch.empty((N, N), dtype=torch.float64, device="cuda")
        us = timeit(eng, lambda: lib.tlsq_k_gram_f32(h, p(Zf), M, N, M, p(G), N, a.f32), a.reps)
        ref = Zf.double() @ Zf.double().T
        err = (G - ref).abs().max().item() / ref.abs().max().item()
        print(f"gram f32 (mfma32={a.f32}) {M}x{N}: {us:.1f} us  sym-flop {2*M*N*(N+128)/2/us/1e6:.1f} TF  algorithmic {M*N*(N+1)/us/1e6:.1f} TF  "
              f"max |dG| / max |G| = {err:.2e}")
    elif a.what in ("gram", "all"):
        G = torch.empty((N, N), dtype=torch.float64, device="cuda")
        us = timeit(eng, lambda: lib.tlsq_k_gram_f64(h, p(Z), M, N, M, p(G), N), a.reps)
        print(f"gram {M}x{N}: {us:.1f} us  full-flop {2*M*N*N/us/1e6:.1f} TF  sym-flop {2*M*N*(N+128)/2/us/1e6:.1f} TF")
    if a.what in ("gemm", "all"):
        W = torch.randn((r, N), dtype=torch.float64, device="cuda", generator=g)  # N x r col-major
        T = torch.empty((r, M), dtype=torch.float64, device="cuda")
        us = timeit(eng, lambda: lib.tlsq_k_gemm_nn_f64(h, p(Z), M, N, M, p(W), r, N, p(T), M), a.reps)
        print(f"gemm_nn T=Z*W ({M}x{N} * {N}x{r}): {us:.1f} us  read {M*N*8/us/1e6:.2f} TB/s")
        V = torch.randn((r, N), dtype=torch.float64, device="cuda", generator=g)
        A = torch.empty((N, M), dtype=torch.float64, device="cuda")
        us = timeit(eng, lambda: lib.tlsq_k_gemm_nt_f64(h, p(T), M, r, M, p(V), N, N, p(A), M), a.reps)
        print(f"gemm_nt A=T*V' ({M}x{r} * {r}x{N}): {us:.1f} us  write {M*N*8/us/1e6:.2f} TB/s")
        W2 = torch.randn((N, N), dtype=torch.float64, device="cuda", generator=g)
        us = timeit(eng, lambda: lib.tlsq_k_gemm_nn_f64(h, p(Z), M, N, M, p(W2), N, N, p(A), M), a.reps)
        print(f"gemm_nn A=Z*W ({M}x{N} * {N}x{N}): {us:.1f} us  {2*M*N*N/us/1e6:.1f} TF")
    if a.what in ("sweeps", "all"):
        D, A_, Y, E, Zz, R = (torch.randn((N, M), dtype=torch.float64, device="cuda", generator=g) for _ in range(6))
        n = M * N
        us1 = timeit(eng, lambda: lib.tlsq_k_shrink_f64(h, p(D), p(A_), p(Y), p(E), p(Zz), n, 3.0, 0.5, 0), a.reps)
        us2 = timeit(eng, lambda: lib.tlsq_k_update_f64(h, p(D), p(A_), p(E), p(Y), p(R), n, 0.3, 0), a.reps)
        En, Zn = (torch.empty_like(D) for _ in range(2))
        us3 = timeit(eng, lambda: lib.tlsq_k_update_shrink_f64(h, p(D), p(A_), p(E), p(Y), p(R), p(En), p(Zn), n, 0.3, 0,
                                                                 3.0, 0.5, 0), a.reps)
        print(f"fused update+shrink {us3:.1f} us: {8*n*8/us3/1e6:.2f} TB/s actual, {11*n*8/us3/1e6:.2f} TB/s algorithmic")
        print(f"shrink {us1:.1f} us {5*n*8/us1/1e6:.2f} TB/s | update {us2:.1f} us {6*n*8/us2/1e6:.2f} TB/s | "
              f"11 passes {11*n*8/(us1+us2)/1e6:.2f} TB/s")
    if a.what in ("symeig", "all"):
        X = torch.randn((N, 2 * N), dtype=torch.float64, device="cuda", generator=g)
        Gm = (X @ X.T).contiguous()
        lam = torch.empty(N, dtype=torch.float64, device="cuda")
        Vv = torch.empty((N, N), dtype=torch.float64, device="cuda")
        sw = C.c_int64()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lib.tlsq_k_symeig_f64(h, p(Gm), N, N, p(lam), p(Vv), N, C.byref(sw))
        print(f"symeig N={N}: {(time.perf_counter()-t0)*1e3:.1f} ms, {sw.value} sweeps")
    eng.close()


if __name__ == "__main__":
    main()
