"""Two runs of the same solve with DEBUG_HASH=1 must print identical "[hash]" lines; prints the first one that differs.
    python tools/dbg/hash_runs.py [poison] 2> log   (then: python tools/dbg/hash_runs.py diff log)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 2 and sys.argv[1] == "diff":
    runs, cur = [], None
    for line in open(sys.argv[2], errors="replace"):
        if line.startswith("[run]"):
            cur = []
            runs.append(cur)
        elif line.startswith("[hash]") and cur is not None:
            cur.append(line.strip())
    base = runs[0]
    for i, r in enumerate(runs[1:], 1):
        j = next((j for j, (a, b) in enumerate(zip(base, r)) if a != b), None)
        if j is None and len(base) == len(r):
            print(f"run {i}: identical ({len(r)} lines)")
        else:
            j = j if j is not None else min(len(base), len(r))
            print(f"run {i}: first difference at line {j} of {len(base)}/{len(r)}")
            for a in range(max(0, j - 3), min(j + 3, len(base), len(r))):
                print("   ", base[a], "|", r[a], "<--" if base[a] != r[a] else "")
    sys.exit(0)
import warnings

import numpy as np

warnings.simplefilter("ignore")
import torch

torch.zeros(1, device="cuda")
import tlsq_amd

if "poison" in sys.argv:
    tlsq_amd.dev_set("WS_POISON", 1)
tlsq_amd.dev_set("DEBUG_HASH", 1)
rng = np.random.default_rng(5)
M, N, r = 2401, 160, 6
D = (rng.standard_normal((M, r)) @ rng.standard_normal((r, N)) + 10 * rng.standard_normal((M, N)) * (rng.random((M, N)) < 0.05)
     + 1e-3 * rng.standard_normal((M, N)))
eng = tlsq_amd.Engine(0)
for rep in range(4):
    sys.stderr.write("[run] %d\n" % rep)
    sys.stderr.flush()
    A, E, s, sv, rp = eng.rpca(D, return_report=True)
    print(rep, rp.iters_done, sv, rp.tsqr_iterations, float(np.abs(A).sum()))
