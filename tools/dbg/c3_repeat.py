import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import tlsq_amd
from oracle import rpca_oracle as O
y, noise = O.synth_series(10_000_000, seed=0)
yn = y + noise
tlsq_amd.dev_from_env()
eng = tlsq_amd.Engine(0)
for i in range(3):
    t0 = time.perf_counter()
    yf, rep = eng.lowrankfilter(yn, 256, return_report=True, cost_history=False)
    dt = time.perf_counter() - t0
    print({k: round(v, 1) for k, v in rep.ms.items()})
    print(f"run {i}: iters {rep.iters_done} wall {dt:.2f} s loop {rep.ms['loop']:.1f} ms -> {rep.iters_done / (rep.ms['loop'] / 1e3):.2f} iters/s", flush=True)
eng.close()
