import os
import sys

# before numpy/scipy load their BLAS: the oracle's LAPACK calls on small matrices are 50x slower with one thread
# per host core (256 on the GPU box) than with a handful
os.environ.setdefault("OPENBLAS_NUM_THREADS", "8")
os.environ.setdefault("OMP_NUM_THREADS", "8")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no binaries (they are git-ignored): compile libtlsqhip.so (hipcc cross-compiles gfx950
    without a GPU) and the oracle's C helper once, with the recipes __graft_entry__.build() uses."""
    lib = os.path.join(ROOT, "totalleastsquares.jl_amd", "libtlsqhip.so")
    helper = os.path.join(ROOT, "oracle", "liboracle_sweeps.so")
    if os.path.exists(lib) and os.path.exists(helper):
        return
    # (compile only: loading the library here would put /opt/rocm's HIP runtime in the process before torch's)
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("tlsq_build", os.path.join(ROOT, "totalleastsquares.jl_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build(force=False, verbose=False)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle_golden():
    """frozen oracle outputs at C1 and a shrunken C2 (tests/golden/make_oracle_vectors.py)"""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "oracle_vectors.json")) as f:
        return json.load(f)
