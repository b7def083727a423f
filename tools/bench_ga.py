"""rpca_ga throughput: sweeps over U per second and the HBM rate they amount to (one read of the d x N panel per
iteration), with the oracle (BLAS-ordered mean) timed on the host beside it.

    python tools/bench_ga.py [--cpu]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu", action="store_true", help="also time the oracle on the host")
    ap.add_argument("--cases", default="10x1000000,64x1000000,256x400000,512x200000,1024x200000,2048x100000,4096x50000")
    a = ap.parse_args()
    import torch  # noqa: F401  (HIP runtime order, see tests)
    torch.zeros(1, device="cuda")
    import tlsq_amd
    eng = tlsq_amd.Engine(0)
    for case in a.cases.split(","):
        d, N = (int(v) for v in case.split("x"))
        rng = np.random.default_rng(0)
        r = 3
        u = np.linalg.qr(rng.standard_normal((d, r)))[0]
        X = (u * np.array([30.0, 20.0, 10.0])) @ rng.standard_normal((r, N)) + 0.01 * rng.standard_normal((d, N))
        X += 100 * rng.standard_normal((d, N)) * (rng.random((d, N)) < 0.001)
        q0 = rng.standard_normal((d, r))
        eng.rpca_ga(X, r, q0=q0, return_report=True)     # warm-up with the same workspace (the dq history included)
        Q, rep = eng.rpca_ga(X, r, q0=q0, return_report=True)
        gb = rep["passes"] * d * N * 8 / 1e9
        line = (f"d={d:5d} N={N:8d}  iters={rep['iters']}  loop {rep['ms_loop']:8.2f} ms  "
                f"{rep['passes'] / rep['ms_loop'] * 1e3:9.1f} sweeps/s  {gb / rep['ms_loop'] * 1e3:7.1f} GB/s")
        if a.cpu:
            from oracle import ga_oracle as G

            def mu(s, w, U):
                s[:] = (U @ w) / np.sum(w)
                return s
            info = G.GaInfo()
            t0 = time.perf_counter()
            G.rpca_ga(X, r, q0=q0, mu=mu, info=info)
            dt = time.perf_counter() - t0
            line += f"   oracle {sum(info.iters) / dt:7.2f} sweeps/s ({dt:.1f} s, incl. set-up)"
        print(line, flush=True)


if __name__ == "__main__":
    main()
