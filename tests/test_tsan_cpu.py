"""Race tooling for the library's HOST threading code: runtime.hip (LocalGroup barrier and loop-back
collectives, multi_run's worker threads and group shutdown, per-communicator abort lock) and staging.hip (the staged-copy
workers) are compiled for the CPU with `g++ -fsanitize=thread` against the stub device layer in tests/tsan/ ("device" memory
is host memory, streams complete at once) and driven by tests/tsan/driver.cpp: collectives in lock step on a 4-rank loop-back
group, a rank that leaves the group call with an error while the others sit in collectives (every rank in turn), staged
copies on all ranks at once and on independent handles.  Pass = exit code 0 and no ThreadSanitizer report.
CPU container only: sanitizers never run on the GPU box."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "totalleastsquares.jl_amd", "csrc")
STUB = os.path.join(ROOT, "tests", "tsan")


def _build(tmp_path, extra=()):
    exe = os.path.join(tmp_path, "tsan_driver")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I", STUB, "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           *extra, "-x", "c++", os.path.join(CSRC, "runtime.hip"), os.path.join(CSRC, "staging.hip"),
           os.path.join(STUB, "driver.cpp"), "-o", exe, "-ldl", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and ("libtsan" in r.stderr or "sanitize" in r.stderr):
        pytest.skip("this g++ has no ThreadSanitizer runtime")
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_threading_code_is_race_free_under_tsan(tmp_path):
    exe = _build(str(tmp_path))
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[:6000]
    assert r.returncode == 0 and "tsan driver: ok" in r.stdout, (r.returncode, r.stdout, r.stderr[-2000:])


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_the_tsan_harness_sees_a_planted_race(tmp_path):
    """the tool is only worth something if it reports: the same build with the group-failure flag of LocalGroup set WITHOUT its
    lock (-DTLSQ_TSAN_PLANT_RACE) must produce a ThreadSanitizer report"""
    exe = _build(str(tmp_path), extra=("-DTLSQ_TSAN_PLANT_RACE",))
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert "ThreadSanitizer: data race" in r.stderr and r.returncode == 66
