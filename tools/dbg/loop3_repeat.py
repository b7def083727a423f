import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, tlsq_amd
from tlsq_amd import workloads as W
tlsq_amd.dev_from_env()
def noisy(seed, M, N, r, noise):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((M, r)) @ rng.standard_normal((r, N))
            + 10 * rng.standard_normal((M, N)) * (rng.random((M, N)) < 0.05) + noise * rng.standard_normal((M, N)))
D1 = noisy(5, 2401, 160, 6, 1e-3)
D2 = W.synth_lowrank_sparse(3001, 130, 30, seed=3001)[0]
plain = tlsq_amd.Engine(0)
loop3 = tlsq_amd.Engine(devices=[0, 0, 0])
others = [W.synth_lowrank_sparse(M, N, r, seed=M)[0] for M, N, r in [(1500, 96, 6), (700, 300, 5), (4000, 64, 3)]]
first = {}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    for name, D in (("noisy", D1), ("clean", D2)):
        print(f"=== rep {rep} {name}", file=sys.stderr, flush=True)
        try:
            A, E, s, sv, r = loop3.rpca(D, return_report=True)
        except Exception as e:
            print("FAILED", rep, name, e, flush=True)
            sys.exit(1)
        out = (A, E, np.array(r.svp_hist))
        if name in first:
            if not all(np.array_equal(a, b) for a, b in zip(out, first[name])):
                print("DIFFERS", rep, name, flush=True)
        else:
            first[name] = out
    (plain if rep % 2 == 0 else loop3).rpca(others[rep % 3])
print("all fine")
