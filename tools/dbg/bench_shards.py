"""The two bench problems (C2 and extra.c4 of bench.py) on loop-back groups of 2, 4 and 8 ranks on one GPU: iterations, sv and the
rank trajectory must equal the committed one-GPU reference (tests/golden/bench_vectors.json) - what bench.py --gpus N validates."""
import os, sys, json, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, tlsq_amd
from oracle import rpca_oracle as O
sys.argv = [sys.argv[0]]
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_vectors.json")))


def svp_hash(hist):   # (bench.py)
    return hashlib.sha256(",".join(str(int(v)) for v in hist).encode()).hexdigest()[:16]


D2 = O.synth_lowrank_sparse(20000, 512, 16, seed=0)[0]
M4, N4, r4, nb = 200000, 512, 16, 8
rb = M4 // nb
G2 = np.random.default_rng([4, 999]).standard_normal((r4, N4))
parts = []
for b in range(nb):
    rg = np.random.default_rng([4, b])
    parts.append(rg.standard_normal((rb, r4)) @ G2 + 10.0 * rg.standard_normal((rb, N4)) * (rg.random((rb, N4)) < 0.05))
D4 = np.vstack(parts)
del parts
for n in (1, 2, 4, 8):
    e = tlsq_amd.Engine(devices=[0] * n) if n > 1 else tlsq_amd.Engine(0)
    for name, D in (("c2", D2), ("c4", D4)):
        A, E, s, sv, rep = e.rpca(D, return_report=True, want_s=False)
        got = {"iters": rep.iters_done, "sv": int(sv), "svp_hash": svp_hash(rep.svp_hist), "converged": bool(rep.converged)}
        want = {k: ref[name][k] for k in got}
        na2, ne2 = float(np.sum(A * A)), float(np.sum(E * E))
        print(n, name, "OK" if got == want else f"MISMATCH {got} vs {want}", "normA2 rel", abs(na2 - ref[name]["normA2"]) / ref[name]["normA2"],
              "normE2 rel", abs(ne2 - ref[name]["normE2"]) / ref[name]["normE2"], flush=True)
    e.close()
