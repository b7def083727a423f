#!/usr/bin/env python3
"""Large mode (min(M,N) > 2048): rpca served by the subspace solver alone.  Reports recovery against the planted
low-rank matrix and, with --oracle, parity with the CPU oracle (slow: two LAPACK SVDs per iteration).

    python tools/large_case.py M N r [--f32] [--oracle] [--want-s]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (its HIP runtime has to be the one in the process)
import tlsq_amd
tlsq_amd.dev_from_env()
from oracle import rpca_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("M", type=int)
ap.add_argument("N", type=int)
ap.add_argument("r", type=int)
ap.add_argument("--f32", action="store_true")
ap.add_argument("--oracle", action="store_true")
ap.add_argument("--want-s", action="store_true")
ap.add_argument("--randomized", action="store_true", help="svd = rsvd-style hook (TLSQ_SVD_RANDOMIZED)")
ap.add_argument("--no-hist", action="store_true", help="device-resident call without cost history (what bench.py times)")
ap.add_argument("--phases", action="store_true", help="bracket every phase with events (tlsq_rpca_opts.phase_timing)")
a = ap.parse_args()
D, A0, _ = O.synth_lowrank_sparse(a.M, a.N, a.r, seed=0)
if a.f32:
    D = D.astype(np.float32)
eng = tlsq_amd.Engine(0)
eng.rpca(np.asarray(D[:256, :64]), iters=2)   # warm up
t0 = time.perf_counter()
from tlsq_amd import _lib as L
hook = dict(svd_mode=L.SVD_RANDOMIZED) if a.randomized else {}
if a.phases:
    hook["phase_timing"] = True
if a.no_hist:
    dD = torch.from_numpy(np.ascontiguousarray(D.T)).cuda()
    dA, dE = torch.empty_like(dD), torch.empty_like(dD)
    eng.rpca_device(dD.data_ptr(), a.M, a.N, dA.data_ptr(), dE.data_ptr(), want_hist=False, dtype=D.dtype, **hook)   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sv, rep, st = eng.rpca_device(dD.data_ptr(), a.M, a.N, dA.data_ptr(), dE.data_ptr(), want_hist=False, dtype=D.dtype, **hook)
    A, E = dA.cpu().numpy().T, dE.cpu().numpy().T
    s = tlsq_amd.SVD(None, np.full(1, np.nan), None) if hasattr(tlsq_amd, "SVD") else type("S", (), {"S": np.full(1, np.nan)})()
else:
    A, E, s, sv, rep = eng.rpca(D, return_report=True, want_U=a.want_s)
dt = time.perf_counter() - t0
print(f"{a.M}x{a.N} r={a.r} {'f32' if a.f32 else 'f64'}: iters={rep.iters_done} converged={rep.converged} sv={sv} "
      f"full={rep.eig_full} fast={rep.eig_fast} steps={rep.subspace_steps} wall={dt:.2f}s loop={rep.ms['loop']:.0f} ms")
print("  svp_hist", rep.svp_hist)
print(f"  s.S: {int(np.isfinite(s.S).sum())} of {s.S.size} singular values returned, head {np.asarray(s.S[:4])}")
print("  phases ms/iter:", {k: round(v / rep.iters_done, 2) for k, v in rep.ms.items() if k in
                            ("shrink", "update", "gram", "eig", "rebuild", "opnorm")})
print(f"  residual {np.linalg.norm(D - (A + E)) / np.linalg.norm(D):.2e}  rel_err_A vs planted "
      f"{np.linalg.norm(A - A0) / np.linalg.norm(A0):.2e}")
if a.oracle:
    t0 = time.perf_counter()
    Ao, Eo, so, svo, io = O.rpca(D)
    print(f"  oracle: iters={io.iters_done} sv={svo} {time.perf_counter() - t0:.0f}s svp same={rep.svp_hist == io.svp_hist} "
          f"relA={np.linalg.norm(A - Ao) / np.linalg.norm(Ao):.2e} relE={np.linalg.norm(E - Eo) / np.linalg.norm(Eo):.2e}")
    if rep.svp_hist != io.svp_hist:
        print("   cpu:", io.svp_hist)
