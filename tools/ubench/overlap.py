#!/usr/bin/env python3
"""development: do the HBM-bound sweep and the MFMA-bound Gram kernel overlap when launched on two streams (two handles)?
   python tools/ubench/overlap.py [M N]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import tlsq_amd
M, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20000, 512)
torch.zeros(1, device="cuda")
ea, eb = tlsq_amd.Engine(0), tlsq_amd.Engine(0)
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda: torch.randn((N, M), dtype=torch.float64, device="cuda", generator=g)
D, A_, Y, E, R, En, Zn, Z2 = (mk() for _ in range(8))
G = torch.empty((N, N), dtype=torch.float64, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
n = M * N
sweep = lambda: ea.lib.tlsq_k_update_shrink_f64(ea.h, p(D), p(A_), p(E), p(Y), p(R), p(En), p(Zn), n, 0.3, 0, 3.0, 0.5, 0)
gram = lambda: eb.lib.tlsq_k_gram_f64(eb.h, p(Z2), M, N, M, p(G), N)
def both_sync():
    ea.synchronize(); eb.synchronize()
def t(fn, reps=30):
    for _ in range(100): fn()
    both_sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    both_sync(); return (time.perf_counter() - t0) / reps * 1e6
us_s = t(sweep); us_g = t(gram)
def pair():
    sweep(); gram()
us_p = t(pair)
print(f"{M}x{N}: sweep {us_s:.1f} us, gram {us_g:.1f} us, sum {us_s+us_g:.1f}; both queued on two streams: {us_p:.1f} us per pair "
      f"(perfect overlap {max(us_s, us_g):.1f})")
