#!/usr/bin/env python3
"""Replay one case of tools/fuzz_parity.py under development switches:
    python tools/dbg/replay_fuzz.py <seed> <case> [SWITCH=VALUE,...] [SWITCH=VALUE,...] ...   (one run per argument; "" = defaults)"""
import os, sys, warnings
os.environ.setdefault("OPENBLAS_NUM_THREADS", "4")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import tlsq_amd
import fuzz_parity as F
from oracle import rpca_oracle as O
warnings.simplefilter("ignore")
big = "--big" in sys.argv
if big:
    sys.argv.remove("--big")
    F.BIG = True
seed, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for it in range(case + 1):
    D, kw, desc = F.make_case(rng)
print("case", case, desc, kw)
Ao, Eo, so, svo, io = O.rpca(D, **kw)
eng = tlsq_amd.Engine(0)
dn = np.linalg.norm(D)
for arg in (sys.argv[3:] or [""]):
    v = dict(kv.split("=") for kv in arg.split(",") if kv)
    with tlsq_amd.dev_switches(**v):
        A, E, s, sv, rep = eng.rpca(D, return_report=True, **kw)
    k = next((i + 1 for i, (a, b) in enumerate(zip(rep.svp_hist, io.svp_hist)) if a != b), None)
    print(f"{str(v):44s} iters {rep.iters_done}/{io.iters_done} sv {sv}/{svo} first diff k={k} errA={np.linalg.norm(A-Ao)/dn:.1e} tsqr={rep.tsqr_iterations}",
          flush=True)
    if k:
        print("    gpu   ", rep.svp_hist[max(0, k - 4):k + 4])
        print("    oracle", io.svp_hist[max(0, k - 4):k + 4])
