"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/tlsq.h declares, resolves option defaults like the reference, and fails loudly without a GPU."""
import ctypes as C
import math
import os
import re
import subprocess

import numpy as np
import pytest

import tlsq_amd
from tlsq_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "tlsq.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tlsq_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported():
    lib = L.load()
    declared = _declared()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/tlsq.h but not exported"
    assert sorted(L.EXPORTS) == declared
    out = subprocess.check_output(["nm", "-D", "--defined-only", L.LIB_PATH]).decode()
    exported = set(re.findall(r" T (tlsq_[a-z0-9_]+)", out))
    assert set(declared) <= exported


def test_version_and_default_opts():
    lib = L.load()
    assert b"gfx950" in lib.tlsq_version()
    o = L.RpcaOpts()
    lib.tlsq_rpca_opts_default(C.byref(o))
    assert math.isnan(o.lambda_) and math.isnan(o.tol) and math.isnan(o.rho)   # -> reference defaults
    assert o.nukeA == 1 and o.iters == 0 and o.maxrank == 0
    assert o.memory == L.MEM_HOST and o.svd_mode == L.SVD_FULL


def test_struct_sizes_match_header():
    # natural alignment, no packing: these are the sizes the Julia shim's struct mirrors must have
    assert C.sizeof(L.RpcaOpts) == 128
    assert C.sizeof(L.RpcaInfo) == 256
    assert C.sizeof(L.GaOpts) == 56 and C.sizeof(L.GaInfo) == 64


def test_tls_from_vt_is_host_only_math():
    """tls!(s::SVD, n) (src/TotalLeastSquares.jl:65-69): the tiny V-partition solve needs no GPU."""
    import numpy as np
    rng = np.random.default_rng(0)
    Ay = rng.standard_normal((50, 5))
    _, _, Vt = np.linalg.svd(Ay, full_matrices=False)
    n = 3
    V = Vt.T
    expect = -V[:n, n:] @ np.linalg.inv(V[n:, n:])
    Vtf = np.asfortranarray(Vt)
    x = np.empty((n, 2), order="F")
    st = L.load().tlsq_tls_from_vt_f64(C.c_void_p(Vtf.ctypes.data), 5, 5, n, C.c_void_p(x.ctypes.data), n)
    assert st == 0
    assert np.allclose(x, expect, rtol=1e-12, atol=1e-12)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(tlsq_amd.TlsqError):
        tlsq_amd.Engine(0)


def test_product_never_imports_oracle():
    """The product path must not reach into oracle/ (grep the package sources)."""
    pkg = os.path.join(ROOT, "totalleastsquares.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "rpca_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_workload_generators_match_the_checkers_copies():
    """bench.py and the GPU tests draw their inputs from the package's neutral module (numpy only); the oracle keeps its
    own copies of the two generators: same seeds, same bits."""
    from oracle import rpca_oracle as O
    from tlsq_amd import workloads as W
    for a, b in zip(W.synth_lowrank_sparse(300, 40, 4, seed=3), O.synth_lowrank_sparse(300, 40, 4, seed=3)):
        assert np.array_equal(a, b)
    for a, b in zip(W.synth_series(5000, seed=2), O.synth_series(5000, seed=2)):
        assert np.array_equal(a, b)
    blk = W.c4_rows(24990, 25010)       # straddles two seeded row blocks
    assert blk.shape == (20, 512) and np.array_equal(blk[:10], W.c4_rows(24990, 25000))


def test_bench_vectors_are_the_oracles():
    """tests/golden/bench_vectors.json (what bench.py validates against) is the ORACLE's output: the config-2 entry is
    reproduced here for the first iterations (the full run takes half a minute: tests/golden/make_bench_vectors.py)."""
    import hashlib
    import json
    from oracle import rpca_oracle as O
    from tlsq_amd import workloads as W
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_vectors.json")))["c2"]
    D = W.synth_lowrank_sparse(20000, 512, 16, seed=0)[0]
    assert hashlib.sha256(D.tobytes(order="F")).hexdigest() == ref["D_sha256_of_float64_column_major"]
    _, _, _, _, info = O.rpca(D, iters=3)
    assert info.svp_hist == ref["svp_hist"][:3]
    assert np.allclose(info.cost_hist, ref["cost_hist"][:3], rtol=1e-12)
    assert ref["iters"] == len(ref["svp_hist"]) == len(ref["cost_hist"]) and ref["converged"]


def test_dev_switches_restore_the_previous_value():
    """nested / overlapping uses of the context manager keep the outer setting; unknown TLSQ_* variables of the
    environment are skipped with a warning instead of aborting a tool"""
    import warnings
    tlsq_amd.dev_set("PAD", 7)
    try:
        with tlsq_amd.dev_switches(PAD=3):
            assert L._dev_shadow["PAD"] == "3"
            with tlsq_amd.dev_switches(PAD=5, NO_ZSWEEP=1):
                assert L._dev_shadow["PAD"] == "5" and L._dev_shadow["NO_ZSWEEP"] == "1"
            assert L._dev_shadow["PAD"] == "3" and "NO_ZSWEEP" not in L._dev_shadow
        assert L._dev_shadow["PAD"] == "7"
    finally:
        tlsq_amd.dev_set("PAD", None)
    assert "PAD" not in L._dev_shadow
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = tlsq_amd.dev_from_env({"TLSQ_NO_SUCH_SWITCH": "1", "TLSQ_PAD": "4", "TLSQ_LIB": "x", "HOME": "/"})
    try:
        assert got == {"TLSQ_PAD": "4"} and any("NO_SUCH_SWITCH" in str(x.message) for x in w)
    finally:
        tlsq_amd.dev_set("PAD", None)


def test_report_reads_the_info_struct_on_first_use():
    """RpcaReport (round 6) converts tlsq_rpca_info lazily: nothing is read at construction, everything on the first attribute,
    unknown names are AttributeErrors, and the live-switch list of the shipped library accepts the round's new names."""
    from tlsq_amd.engine import Engine, RpcaReport
    info, cost, svp = Engine._info(8, True)
    rep = RpcaReport(info, cost, svp)
    assert "iters_done" not in rep.__dict__
    info.iters_done, info.converged, info.final_cost, info.d_norm = 3, 1, 0.25, 7.5
    info.kern_gram_h3, info.ms_total = 5, 1.5
    cost[:3] = [1.0, 0.5, 0.25]
    svp[:3] = [4, 4, 5]
    assert rep.iters_done == 3 and rep.converged and rep.final_cost == 0.25 and rep.d_norm == 7.5
    assert rep.cost_hist == [1.0, 0.5, 0.25] and rep.svp_hist == [4, 4, 5]
    assert rep.kern["gram_h3"] == 5 and rep.ms["total"] == 1.5
    with pytest.raises(AttributeError):
        rep.no_such_field
    for name in ("LZ_MULTI", "COLD_TOP", "COLD_TOL0", "RITZ_SORT", "PAD_PROJECT", "HOST_TRACE", "SWEEP_TIMING_STRIDE"):
        tlsq_amd.dev_set(name, "1")
        tlsq_amd.dev_set(name, None)
    with pytest.raises(Exception):
        tlsq_amd.dev_set("NO_FUSED_SWEEP", "1")   # an ablation switch: refused by the shipped library
