import sys, time, warnings
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
import tlsq_amd
from oracle import rpca_oracle as O
warnings.simplefilter("ignore")
eng = tlsq_amd.Engine(0)
cases = {"zeros": np.zeros((10, 5)), "ones": np.ones((10, 5)), "1x1": np.array([[3.0]]), "1x5": np.arange(5.0)[None, :] + 1,
         "5x1": np.arange(5.0)[:, None] + 1, "nan": np.full((6, 4), np.nan), "rank1+outlier": np.outer(np.arange(1., 9), np.arange(1., 6)) + np.eye(8, 5) * 50,
         "2x2": np.array([[1., 2.], [3., 4.]])}
for name, D in cases.items():
    t0 = time.perf_counter()
    try:
        A, E, s, sv, rep = eng.rpca(D, iters=50, return_report=True)
        try:
            Ao, Eo, so, svo, io = O.rpca(D, iters=50)
            ref = f"oracle iters={io.iters_done} sv={svo} dA={np.nanmax(np.abs(A - Ao)) if np.isfinite(Ao).all() else 'nan'}"
        except Exception as e:  # noqa
            ref = f"oracle raised {type(e).__name__}"
        print(f"{name}: iters={rep.iters_done} sv={sv} finiteA={np.isfinite(A).all()} {time.perf_counter() - t0:.2f}s | {ref}")
    except Exception as e:  # noqa
        print(f"{name}: raised {type(e).__name__}: {str(e)[:120]} ({time.perf_counter() - t0:.2f}s)")
x = eng.rtls_batched(np.random.default_rng(0).standard_normal((3, 6, 2)), np.random.default_rng(1).standard_normal((3, 6)))
print("batched tiny ok", x.shape)
