"""free-running C2 solve time under the placement variants of the asynchronous certificate (one process per variant: the
second stream's priority is fixed when it is created)"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import time, numpy as np, torch, tlsq_amd
    from tlsq_amd import workloads as W
    tlsq_amd.dev_from_env()
    D = W.synth_lowrank_sparse(20000, 512, 16, seed=0)[0]
    eng = tlsq_amd.Engine(0)
    dD = torch.from_numpy(np.ascontiguousarray(D.T)).cuda(); dA = torch.empty_like(dD); dE = torch.empty_like(dD)
    for _ in range(5):
        eng.rpca_device(dD.data_ptr(), 20000, 512, dA.data_ptr(), dE.data_ptr(), want_hist=False)
    res = []
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20):
            sv, r, st = eng.rpca_device(dD.data_ptr(), 20000, 512, dA.data_ptr(), dE.data_ptr(), want_hist=False)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t) / 20 * 1e3)
    print(json.dumps({"ms": [round(x, 3) for x in res], "iters": r.iters_done, "upd": round(r.ms["update"] / max(r.sweeps_timed, 1), 4)}))
    sys.exit(0)
variants = {"beside gram, high": {}, "beside sweep, high": {"TLSQ_CERT_EARLY": "1"}, "beside sweep, low": {"TLSQ_CERT_EARLY": "1", "TLSQ_CERT_PRIO": "low"},
            "beside sweep, normal": {"TLSQ_CERT_EARLY": "1", "TLSQ_CERT_PRIO": "normal"}, "beside gram, low": {"TLSQ_CERT_PRIO": "low"},
            "in line (no async)": {"TLSQ_NO_CERT_ASYNC": "1"}, "in line, no spec rebuild": {"TLSQ_NO_CERT_ASYNC": "1", "TLSQ_NO_SPEC_REBUILD": "1"}}
for rnd in range(2):
    for name, env in variants.items():
        out = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), capture_output=True, text=True)
        print(f"{name:28s}", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
