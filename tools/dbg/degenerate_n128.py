import sys, warnings
sys.path.insert(0,'/root/repo')
import numpy as np, tlsq_amd
from oracle import rpca_oracle as O
warnings.simplefilter("ignore")
eng=tlsq_amd.Engine(0)
for name,D in [("zeros 300x128",np.zeros((300,128))),("ones 300x128",np.ones((300,128))),("rank1+tiny 600x256",np.outer(np.arange(600.)+1,np.ones(256))+1e-9*np.random.default_rng(0).standard_normal((600,256))),("single spike 400x128",np.eye(400,128)*5)]:
    try:
        A,E,s,sv,rep=eng.rpca(D,return_report=True)
        Ao,Eo,so,svo,io=O.rpca(D)
        print(name,"iters",rep.iters_done,io.iters_done,"sv",sv,svo,"finite",np.isfinite(A).all(),"dA",np.abs(A-Ao).max())
    except Exception as e:
        print(name,"EXC",repr(e)[:150])
        try:
            Ao,Eo,so,svo,io=O.rpca(D); print("   oracle ok iters",io.iters_done)
        except Exception as e2: print("   oracle EXC",repr(e2)[:100])
