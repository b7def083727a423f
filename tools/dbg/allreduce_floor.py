"""Launch-side cost of the RCCL collectives of a row-shard iteration, measured on ONE GPU: the same 25000 x 512 shard (BASELINE
config 4 at 8 GPUs) solved on a plain handle and on a handle with a one-rank RCCL communicator (tlsq_comm_init(1, 0, id)): every
all-reduce of the loop is issued (ncclAllReduce on one rank = a device copy the library does not even need, but the enqueue,
the stream dependencies and RCCL's own kernel launch are all there).  The difference per iteration / collectives per iteration is
the floor of what a collective costs the loop before any wire latency - the number DESIGN.md section 6 uses.  The N > 1 wire
latency (xGMI) cannot be measured on this box.
    python tools/dbg/allreduce_floor.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import tlsq_amd
from oracle import rpca_oracle as O
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
N, r = 512, 16
D, _, _ = O.synth_lowrank_sparse(M, N, r, seed=0)
dD = torch.from_numpy(np.ascontiguousarray(D.T)).cuda()
dA, dE = torch.empty_like(dD), torch.empty_like(dD)
res = {}
for tag in ("plain", "comm1"):
    eng = tlsq_amd.Engine(0)
    if tag == "comm1":
        tlsq_amd.dev_set("FORCE_COMM", 1)      # (a single rank gets no communicator otherwise)
        eng.comm_init(1, 0, eng.unique_id())
        assert eng.comm_size() == 1
    for i in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sv, rep, st = eng.rpca_device(dD.data_ptr(), M, N, dA.data_ptr(), dE.data_ptr(), want_hist=False)
        dt = time.perf_counter() - t0
    res[tag] = (dt, rep.iters_done)
    print(f"{tag}: {dt*1e3:.3f} ms per solve, {rep.iters_done} iterations, {dt*1e3/rep.iters_done:.4f} ms per iteration", flush=True)
    eng.close()
d = (res["comm1"][0] - res["plain"][0]) * 1e3 / res["plain"][1]
print(f"one-rank communicator: +{d*1e3:.1f} us per iteration (two collectives per steady iteration: the 72 bound slots and the N x N Gram matrix -> ~{d*1e3/2:.1f} us each, launch side only)")
