import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, tlsq_amd
rng=np.random.default_rng(0)
d,N,r=40,3000,2
X=rng.standard_normal((d,r))@rng.standard_normal((r,N))+0.05*rng.standard_normal((d,N))
q0=rng.standard_normal((d,r))
p=tlsq_amd.Engine(0); m=tlsq_amd.Engine(devices=[0,0,0])
for mode in [None,"entrywise_trimmed_mean","entrywise_median"]:
    kw={"mu":mode} if mode else {}
    a,ra=p.rpca_ga(X,r,q0=q0,iters=60,return_report=True,**kw); b,rb=m.rpca_ga(X,r,q0=q0,iters=60,return_report=True,**kw)
    print(mode, "bit-identical:", np.array_equal(a,b), "maxdiff", np.abs(a-b).max(), ra["iters"], rb["iters"], "ms", ra.get("ms_loop"), rb.get("ms_loop"))
