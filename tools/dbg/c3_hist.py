import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, tlsq_amd
from oracle import rpca_oracle as O
y, noise = O.synth_series(1_000_000, seed=0)
eng = tlsq_amd.Engine(0)
yf, rep = eng.lowrankfilter(y + noise, 256, return_report=True, cost_history=True)
print("iters", rep.iters_done, "cost", ["%.3e" % c for c in rep.cost_hist])
