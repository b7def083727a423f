#!/usr/bin/env python3
"""The drop-in (host-pointer) call at the C2 size: where the time goes.  Times rpca through host pointers with and without the
returned decomposition, the transfers alone, and the cost of first-touching freshly allocated output arrays.
    python tools/host_path.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import tlsq_amd
from tlsq_amd import workloads as W
M, N, r = 20000, 512, 16
D = W.synth_lowrank_sparse(M, N, r, seed=0)[0]
eng = tlsq_amd.Engine(0)
t = time.perf_counter(); X = np.empty((M, N), order="F"); X[...] = 1.0; print(f"first touch of a fresh 82 MB array: {(time.perf_counter()-t)*1e3:.2f} ms")
t = time.perf_counter(); X[...] = 2.0; print(f"second write of it:                {(time.perf_counter()-t)*1e3:.2f} ms")
for kw, name in ((dict(want_s=False), "A, E"), (dict(), "A, E, s")):
    eng.rpca(D, cost_history=False, **kw)
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        A, E, s, sv, rep = eng.rpca(D, cost_history=False, return_report=True, **kw)
        ts.append((time.perf_counter() - t) * 1e3)
    print(f"host pointers, {name:8s}: {min(ts):.2f} ms best, {np.median(ts):.2f} median; h2d {rep.ms['h2d']:.2f} d2h {rep.ms['d2h']:.2f} "
          f"loop {rep.ms['loop']:.2f} total {rep.ms['total']:.2f} ms   ({rep.iters_done} iterations)")
dD = torch.from_numpy(np.ascontiguousarray(D.T)).cuda(); dA = torch.empty_like(dD); dE = torch.empty_like(dD)
sv, rep, st = eng.rpca_device(dD.data_ptr(), M, N, dA.data_ptr(), dE.data_ptr(), want_hist=False)
A2, E2 = dA.cpu().numpy().T, dE.cpu().numpy().T
A, E, s, sv, rep = eng.rpca(D, cost_history=False, return_report=True, want_s=False)
print("host path == device path:", np.array_equal(A, A2), np.array_equal(E, E2))
# where the wall time of the Python call goes beyond the library's own total
import ctypes as C
from tlsq_amd import _lib as L
Df = np.asfortranarray(D)
for trial in range(3):
    t0 = time.perf_counter()
    A = np.empty((M, N), order="F"); E = np.empty((M, N), order="F")
    t1 = time.perf_counter()
    o = eng.make_opts()
    info, cost, svp = eng._info(1000, False)
    sv = C.c_int64(0)
    st = eng.lib.tlsq_rpca_f64(eng.h, C.c_void_p(Df.ctypes.data), M, N, M, C.byref(o), C.c_void_p(A.ctypes.data), M,
                               C.c_void_p(E.ctypes.data), M, None, M, None, None, N, C.byref(sv), C.byref(info))
    t2 = time.perf_counter()
    del A, E
    t3 = time.perf_counter()
    print(f"alloc {1e3*(t1-t0):.2f} ms, C call {1e3*(t2-t1):.2f} ms (library total {info.ms_total:.2f}), free {1e3*(t3-t2):.2f} ms")
A = np.empty((M, N), order="F"); E = np.empty((M, N), order="F"); A[...] = 0; E[...] = 0
for trial in range(3):
    t1 = time.perf_counter()
    info, cost, svp = eng._info(1000, False)
    st = eng.lib.tlsq_rpca_f64(eng.h, C.c_void_p(Df.ctypes.data), M, N, M, C.byref(o), C.c_void_p(A.ctypes.data), M,
                               C.c_void_p(E.ctypes.data), M, None, M, None, None, N, C.byref(sv), C.byref(info))
    t2 = time.perf_counter()
    print(f"outputs already touched: C call {1e3*(t2-t1):.2f} ms (library total {info.ms_total:.2f}; h2d {info.ms_h2d:.2f} d2h {info.ms_d2h:.2f})")
