"""Returned `s` at the headline shape with and without the spectrum slicer (sliced.hip): time of the whole call, sweeps, and the
difference of the singular values / orthogonality of Vt.  TLSQ_DEBUG=1 prints the slicer's own trace.
    python tools/dbg/sliced_check.py [M N rank]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
import tlsq_amd
from oracle import rpca_oracle as O

M, N, r = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (20000, 512, 16)
tlsq_amd.dev_from_env()
D, _, _ = O.synth_lowrank_sparse(M, N, r, seed=0)
eng = tlsq_amd.Engine(0)
res = {}
for tag, sw in (("normwise", {}), ("factor", dict(SLICE_NORMWISE=0)), ("plain", dict(NO_SLICED_EIG=1))):
    with tlsq_amd.dev_switches(**sw):
        for rep_i in range(3):
            t0 = time.perf_counter()
            A, E, s, sv, rep = eng.rpca(D, return_report=True, cost_history=False)
            dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        A2, E2, s2, sv2, rep2 = eng.rpca(D, return_report=True, cost_history=False, want_s=False)
        dt0 = time.perf_counter() - t0
    res[tag] = s
    Vt = np.asarray(s.Vt)
    U = np.asarray(s.U)
    print(f"{tag:9s} call {dt*1e3:8.2f} ms (without s {dt0*1e3:7.2f})  total_ms {rep.ms['total']:.2f} loop_ms {rep.ms['loop']:.2f} "
          f"sweeps {rep.jacobi_sweeps}  |VtVt'-I| {np.max(np.abs(Vt @ Vt.T - np.eye(N))):.2e}  |U'U-I| {np.max(np.abs(U.T @ U - np.eye(N))):.2e}")
b = np.asarray(res["plain"].S)
for tag in ("normwise", "factor"):
    a = np.asarray(res[tag].S)
    print(tag, "max rel diff of S vs plain:", np.max(np.abs(a - b) / b), " S[0], S[r], S[-1]:", a[0], a[r], a[-1])
eng.close()
