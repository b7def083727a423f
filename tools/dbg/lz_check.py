"""k_lanczos_multi (a chunk of Lanczos steps per launch) against one launch per step: sigma_max of random panels through
tlsq_k_opnorm_f64 for several N, both ways, against numpy; then the C2 solve both ways (time, iterations, d_norm, cost).
    python tools/dbg/lz_check.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import tlsq_amd
from oracle import rpca_oracle as O

eng = tlsq_amd.Engine(0)
rng = np.random.default_rng(3)
worst = 0.0
for N in (64, 65, 100, 256, 500, 512, 520, 777, 1000, 1024):
    for kind in ("flat", "lowrank", "graded"):
        M = 3 * N + 7
        if kind == "flat":
            Z = rng.standard_normal((M, N))
        elif kind == "lowrank":
            Z = rng.standard_normal((M, 9)) @ rng.standard_normal((9, N)) + 1e-3 * rng.standard_normal((M, N))
        else:
            Z = rng.standard_normal((M, N)) * np.logspace(0, -6, N)[None, :]
        ref = np.linalg.norm(Z, 2)
        d = torch.from_numpy(np.ascontiguousarray(Z.T)).cuda()
        out = {}
        for mode in ("1", "0"):
            with tlsq_amd.dev_switches(LZ_MULTI=mode):
                o = C.c_double(0.0)
                st = eng.lib.tlsq_k_opnorm_f64(eng.h, C.c_void_p(d.data_ptr()), M, N, M, C.byref(o))
                assert st == 0, (N, kind, mode, st)
                out[mode] = o.value
        e1, e0 = abs(out["1"] / ref - 1), abs(out["0"] / ref - 1)
        worst = max(worst, e1)
        print(f"N {N:5d} {kind:8s} multi {e1:.2e} classic {e0:.2e} multi-vs-classic {abs(out['1'] / out['0'] - 1):.2e}", flush=True)
        assert e1 < 1e-9, (N, kind)
print("worst", worst)

M, N, r = 20000, 512, 16
D, _, _ = O.synth_lowrank_sparse(M, N, r, seed=0)
dD = torch.from_numpy(np.ascontiguousarray(D.T)).cuda()
dA, dE = torch.empty_like(dD), torch.empty_like(dD)
torch.cuda.synchronize()
for mode in ("1", "0", "1", "0"):
    with tlsq_amd.dev_switches(LZ_MULTI=mode):
        ts = []
        for i in range(12):
            t0 = time.perf_counter()
            sv, rep, st = eng.rpca_device(dD.data_ptr(), M, N, dA.data_ptr(), dE.data_ptr(), want_hist=False)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts[2:]) * 1e3
        print(f"LZ_MULTI={mode}: {ts.mean():.3f} ms per solve (min {ts.min():.3f}), lib total {rep.ms['total']:.3f}; iters {rep.iters_done}, sv {sv}, "
              f"d_norm {rep.d_norm!r}, final cost {rep.final_cost!r}", flush=True)
for sw in (dict(COLD_TOP="0"), dict(COLD_TOL0="0"), dict(RITZ_SORT="0"), dict(PAD_PROJECT="0"), dict(RSKIP_MARGIN="8"), dict(SWEEP_TIMING_STRIDE="8"), dict(SWEEP_TIMING_STRIDE="1000"), dict(),
           dict(LZ_MULTI="0", COLD_TOP="0", COLD_TOL0="0", RITZ_SORT="0", PAD_PROJECT="0", RSKIP_MARGIN="8"), dict()):
    with tlsq_amd.dev_switches(**sw):
        ts = []
        for i in range(12):
            t0 = time.perf_counter()
            sv, rep, st = eng.rpca_device(dD.data_ptr(), M, N, dA.data_ptr(), dE.data_ptr(), want_hist=False)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts[2:]) * 1e3
        print(f"{sw}: {ts.mean():.3f} ms per solve (min {ts.min():.3f}), lib total {rep.ms['total']:.3f}; iters {rep.iters_done}, sv {sv}, subspace steps {rep.subspace_steps}", flush=True)
eng.close()
