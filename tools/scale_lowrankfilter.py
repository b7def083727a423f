#!/usr/bin/env python3
"""BASELINE config 3: lowrankfilter on a long series (default N=1e7, n=256) — development/validation tool.
   python tools/scale_lowrankfilter.py --N 10000000 --n 256 [--check]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tlsq_amd
from oracle import rpca_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=10_000_000)
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--check", action="store_true", help="compare with the CPU oracle (small N only)")
ap.add_argument("--no-hist", action="store_true", help="do not ask for the per-iteration cost (what a plain Julia call does)")
ap.add_argument("--phases", action="store_true", help="bracket every phase with events (tlsq_rpca_opts.phase_timing)")
ap.add_argument("--repeat", type=int, default=2, help="calls on the same handle (the first one of a process includes cold-start costs)")
a = ap.parse_args()
y, noise = O.synth_series(a.N, seed=0)
yn = y + noise
import torch
free0, total = torch.cuda.mem_get_info(0)
tlsq_amd.dev_from_env()   # TLSQ_* development switches from the shell (after torch: its HIP runtime is loaded first)
eng = tlsq_amd.Engine(0)
qn = lambda x: x / np.quantile(np.abs(x), 0.9)
K = a.N - a.n + 1
# two calls on the same handle: the first one of a process also pays for the workspace (80+ GB of hipMalloc), the first launch
# of every kernel and a GPU that has been idle (its clock ramps up over the first ~0.3 s of work); both are printed
for run in range(a.repeat):
    t0 = time.perf_counter()
    yf, rep = eng.lowrankfilter(yn, a.n, return_report=True, cost_history=not a.no_hist, phase_timing=a.phases)
    dt = time.perf_counter() - t0
    if run == 0:
        free1, _ = torch.cuda.mem_get_info(0)   # the handle keeps its workspace: what is missing now is the peak of the call
    ratio = np.mean((y - qn(yf)) ** 2) / np.mean(noise ** 2)
    print(f"N={a.N} n={a.n} (call {run + 1} of {a.repeat} on this handle): H is {K}x{a.n} ({K*a.n*8/1e9:.2f} GB/array); {rep.iters_done} ALM iterations, "
          f"converged={rep.converged}, wall {dt:.2f} s, loop {rep.ms['loop']/1e3:.3f} s, "
          f"{rep.iters_done/(rep.ms['loop']/1e3):.2f} iters/s; MSE ratio {ratio:.2e} (< 1e-3 required); "
          f"phases ms/iter: " + ", ".join(f"{k}={v/rep.iters_done:.1f}" for k, v in rep.ms.items() if k in
                                         ("shrink", "gram", "eig", "rebuild", "update", "opnorm")))
Kp = (K + 15) // 16 * 16
print(f"device memory held by the handle after the call: {(free0 - free1)/1e9:.2f} GB = {(free0 - free1)/(Kp*a.n*8):.2f} panels "
      f"of {Kp*a.n*8/1e9:.2f} GB")
if a.check:
    yo = O.lowrankfilter(yn, a.n)
    print("rel diff vs oracle:", np.linalg.norm(yf - yo) / np.linalg.norm(yo))
eng.close()
