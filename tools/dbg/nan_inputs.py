import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, tlsq_amd, warnings
warnings.simplefilter("ignore")
e = tlsq_amd.Engine(0)
rng = np.random.default_rng(0)
A = rng.standard_normal((50, 4)); y = rng.standard_normal(50); A[3, 1] = np.nan
def run(name, f):
    t0 = time.perf_counter()
    try:
        r = f()
        r0 = r[0] if isinstance(r, tuple) else r
        print(name, "returned, finite:", bool(np.isfinite(np.asarray(r0)).all()), f"{time.perf_counter()-t0:.2f}s")
    except Exception as ex:
        print(name, "raised", type(ex).__name__, str(ex)[:90], f"{time.perf_counter()-t0:.2f}s")
run("tls", lambda: e.tls(A, y))
run("tls_", lambda: e.tls_(np.column_stack([A, y]), 4))
run("rtls", lambda: e.rtls(A, y))
As = rng.standard_normal((5, 50, 3)); ys = rng.standard_normal((5, 50)); As[2, 4, 1] = np.nan
run("rtls_batched", lambda: e.rtls_batched(As, ys))
Ds = rng.standard_normal((4, 40, 6)); Ds[1, 2, 3] = np.inf
run("rpca_batched", lambda: e.rpca_batched(Ds))
X = rng.standard_normal((10, 500)); X[2, 7] = np.nan
run("rpca_ga", lambda: e.rpca_ga(X, 2, q0=rng.standard_normal((10, 2))))
run("rpca_ga median", lambda: e.rpca_ga(X, 2, q0=rng.standard_normal((10, 2)), mu="entrywise_median", iters=20))
g = tlsq_amd.Engine(devices=[0, 0])
D = rng.standard_normal((400, 12)); D[100, 2] = np.nan
run("group rpca", lambda: g.rpca(D))
run("plain after", lambda: e.rpca(rng.standard_normal((60, 8))))
