#!/usr/bin/env python3
"""k_zsweep (explicit D) alone at a given shape:  python tools/dbg/zsweep_bench.py M N r"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, tlsq_amd
tlsq_amd.dev_from_env()
M, N, r = (int(v) for v in sys.argv[1:4])
eng = tlsq_amd.Engine(0); lib, h = eng.lib, eng.h
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda *s: torch.randn(s, dtype=torch.float64, device="cuda", generator=g)
D, Y0, Y1, Z = mk(N, M), mk(N, M), mk(N, M), mk(N, M)
T, V = mk(r, M), mk(r, N)
p = lambda t: C.c_void_p(t.data_ptr())
def run():
    st = lib.tlsq_k_zsweep_f64(h, p(D), p(T), p(V), None, p(Y0), p(Y1), p(Z), None, M, N, r, 0.3, 1 / 0.3, 0, 1 / 0.45, 0.1, 0, None)
    assert st == 0
for _ in range(5): run()
eng.synchronize()
t0 = time.perf_counter(); reps = 20
for _ in range(reps): run()
eng.synchronize()
us = (time.perf_counter() - t0) / reps * 1e6
print(f"zsweep {M}x{N} r={r}: {us:.1f} us  {5*M*N*8/us/1e6:.2f} TB/s (5 passes)")
