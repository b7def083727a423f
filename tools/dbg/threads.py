"""Two (four) host threads, each with its own handle, solving different problems at the same time: results must equal the
sequential ones bit for bit (no hidden global state in the library)."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, tlsq_amd
from oracle import rpca_oracle as O
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 4
probs = []
for t in range(NT):
    M, N, r = [(1500, 96, 6), (3000, 130, 12), (800, 40, 3), (5000, 256, 10)][t % 4]
    D, _, _ = O.synth_lowrank_sparse(M, N, r, seed=10 + t)
    probs.append(D)
engs = [tlsq_amd.Engine(0) for _ in range(NT)]
seq = [e.rpca(D, return_report=True) for e, D in zip(engs, probs)]
out = [None] * NT
errs = []
def work(t):
    try:
        for rep in range(6):
            out[t] = engs[t].rpca(probs[t], return_report=True)
            y = np.sin(np.arange(3000) / (7.0 + t)) + 0.01 * np.random.default_rng(t).standard_normal(3000)
            engs[t].lowrankfilter(y, 40)
    except Exception as e:  # noqa
        errs.append((t, repr(e)))
ths = [threading.Thread(target=work, args=(t,)) for t in range(NT)]
[t.start() for t in ths]; [t.join() for t in ths]
print("errors:", errs)
for t in range(NT):
    a, b = seq[t], out[t]
    print(t, "A identical:", np.array_equal(a[0], b[0]), "E identical:", np.array_equal(a[1], b[1]), "iters", a[4].iters_done, b[4].iters_done, "svp same", a[4].svp_hist == b[4].svp_hist)
