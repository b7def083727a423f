"""Thousands of k_lanczos_multi runs on one handle (C2-size solves and small norms in turn): no run may give up (every give-up
costs 20 ms and switches the kernel off for the handle - the timing of the last block must equal the first's), results constant.
    python tools/dbg/lz_soak.py [solves]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import tlsq_amd
from oracle import rpca_oracle as O
n_solves = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
eng = tlsq_amd.Engine(0)
M, N, r = 20000, 512, 16
D, _, _ = O.synth_lowrank_sparse(M, N, r, seed=0)
dD = torch.from_numpy(np.ascontiguousarray(D.T)).cuda()
dA, dE = torch.empty_like(dD), torch.empty_like(dD)
rng = np.random.default_rng(1)
small = [torch.from_numpy(np.ascontiguousarray(rng.standard_normal((n, 3 * n)))).cuda() for n in (64, 200, 777, 1024)]
refs = [float(np.linalg.norm(s.cpu().numpy(), 2)) for s in small]
first = None
blocks = []
t_block = time.perf_counter()
for i in range(n_solves):
    sv, rep, st = eng.rpca_device(dD.data_ptr(), M, N, dA.data_ptr(), dE.data_ptr(), want_hist=False)
    key = (sv, rep.iters_done, rep.d_norm, rep.final_cost)
    first = first or key
    assert key == first, (i, key, first)
    s = small[i % 4]
    o = C.c_double(0.0)
    n = s.shape[0]
    assert eng.lib.tlsq_k_opnorm_f64(eng.h, C.c_void_p(s.data_ptr()), 3 * n, n, 3 * n, C.byref(o)) == 0
    assert abs(o.value / refs[i % 4] - 1) < 1e-12
    if (i + 1) % 250 == 0:
        blocks.append((time.perf_counter() - t_block) / 250 * 1e3)
        t_block = time.perf_counter()
        print(f"{i + 1} solves: {blocks[-1]:.3f} ms per solve + norm", flush=True)
assert max(blocks) < 1.05 * min(blocks), blocks
print("lz soak ok:", first)
eng.close()
