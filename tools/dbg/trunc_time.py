"""lowrankfilter(y, 256; sv = 4) at N = 1e7 (BASELINE config 3's series): the structured form (hankelop.hip) against the panel form
(HANKEL_STRUCT=0), device-resident series.   python tools/dbg/trunc_time.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import tlsq_amd
from tlsq_amd import _lib as L
import ctypes as C
from oracle import rpca_oracle as O
Ns = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
y, noise = O.synth_series(Ns, seed=0)
eng = tlsq_amd.Engine(0)
dy = torch.from_numpy(y + noise).cuda()
dyf = torch.empty_like(dy)
o = eng.make_opts(memory=L.MEM_DEVICE)
for tag, sw in (("structured", {}), ("panels", dict(HANKEL_STRUCT=0))):
    with tlsq_amd.dev_switches(**sw):
        for i in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = eng.lib.tlsq_lowrankfilter_f64(eng.h, C.c_void_p(dy.data_ptr()), Ns, 1, Ns, 256, 1, 4, C.byref(o), C.c_void_p(dyf.data_ptr()), Ns, None)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert st == 0, st
    res = dyf.cpu().numpy()
    print(f"{tag:10s} {dt*1e3:8.2f} ms   (checksum {float(np.sum(res)):.12e})")
eng.close()
