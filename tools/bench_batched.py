#!/usr/bin/env python3
"""Throughput of the batched tiny-problem kernel: rtls problems per second on device-resident
stacks, with the CPU oracle (numpy + LAPACK, one problem at a time like the reference's own loop,
test/runtests.jl:205-235) timed beside it.

    python tools/bench_batched.py [--M 50 --n 3] [--batch 20000] [--cpu-problems 200]
"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import tlsq_amd
from oracle import rpca_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=50)
ap.add_argument("--n", type=int, default=3)
ap.add_argument("--sigma", type=float, default=50.0)
ap.add_argument("--batch", type=int, default=20000)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--cpu-problems", type=int, default=200)
ap.add_argument("--f32", action="store_true", help="Float32 stacks (tlsq_rtls_batched_f32)")
a = ap.parse_args()
M, n, B = a.M, a.n, a.batch
rng = np.random.default_rng(0)
x0 = rng.standard_normal((B, n))
A0 = rng.standard_normal((B, M, n))
An = A0 + a.sigma * rng.standard_normal(A0.shape) * (rng.random(A0.shape) < 0.1)
yn = np.einsum("bmn,bn->bm", A0, x0)
yn = yn + a.sigma * rng.standard_normal(yn.shape) * (rng.random(yn.shape) < 0.1)

torch.zeros(1, device="cuda")
eng = tlsq_amd.Engine(0)
from tlsq_amd import _lib as L
tdt = torch.float32 if a.f32 else torch.float64
dA = torch.from_numpy(np.ascontiguousarray(np.transpose(An, (0, 2, 1)))).cuda().to(tdt)     # each problem column-major
dy = torch.from_numpy(np.ascontiguousarray(yn)).cuda().to(tdt)
dx = torch.empty((B, n), dtype=tdt, device="cuda")
dit = torch.empty(B, dtype=torch.int32, device="cuda")
dst = torch.empty(B, dtype=torch.int32, device="cuda")
o = eng.make_opts(iters=1000, memory=L.MEM_DEVICE)
p = lambda t: C.c_void_p(t.data_ptr())


def run():
    fn = eng.lib.tlsq_rtls_batched_f32 if a.f32 else eng.lib.tlsq_rtls_batched_f64
    st = fn(eng.h, p(dA), p(dy), M, n, 1, B, C.byref(o), p(dx), p(dit), p(dst))
    assert st >= 0, st


run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
it = dit.cpu().numpy()
x = dx.cpu().numpy()

nc = min(a.cpu_problems, B)
t0 = time.perf_counter()
okw = {"tol": float(np.sqrt(np.finfo(np.float32).eps))} if a.f32 else {}          # the fp32 entry point's default tol
xo = np.stack([np.ravel(O.rtls(An[b], yn[b], **okw)) for b in range(nc)])
dtc = time.perf_counter() - t0
err = float(np.max(np.linalg.norm(x[:nc] - xo, axis=1) / (1e-9 + np.linalg.norm(xo, axis=1))))
out = {"metric": f"rtls problems/sec, {M}x{n}+1 {'fp32' if a.f32 else 'fp64'}, batch {B}", "value": B / dt, "unit": "problems/s",
       "ms_per_batch": dt * 1e3, "mean_iters": float(it.mean()), "max_iters": int(it.max()),
       "alm_iters_per_s": float(it.sum()) / dt, "unconverged": int(dst.cpu().numpy().sum()),
       "cpu_baseline": {"value": nc / dtc, "unit": "problems/s", "cores": int(os.environ.get("OPENBLAS_NUM_THREADS", 0)) or "default",
                        "kind": "port", "sample": f"first {nc} problems, one at a time (oracle rtls: numpy + LAPACK gesdd)"},
       "max_rel_diff_vs_oracle": err}
print(json.dumps(out))
