"""The optimal odd quintics of the matrix-function route (csrc/quintic.hpp, host code): compiled here with g++ and compared
with the independent numpy statement in tools/odd_quintics.py; properties of the schedule itself."""
import os, subprocess, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import odd_quintics as OQ

DRIVER = r"""
#include "quintic.hpp"
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
    double l = atof(argv[1]), u = 1.0;
    for (int s = 0; s < 24; ++s) {
        tlsq::OddQuintic p;
        if (!tlsq::odd_quintic(l, u, &p)) { printf("fail\n"); return 1; }
        printf("%.17g %.17g %.17g %.17g\n", p.a, p.b, p.c, p.E);
        l = 1.0 - p.E; u = 1.0 + p.E;
        if (p.E < 1e-3) break;
    }
    return 0;
}
"""


def cxx_schedule(l0):
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "drv.cpp")
        open(src, "w").write(DRIVER)
        exe = os.path.join(d, "drv")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "totalleastsquares.jl_amd", "csrc"), src, "-o", exe])
        out = subprocess.check_output([exe, repr(l0)], text=True)
    return [tuple(float(v) for v in line.split()) for line in out.strip().splitlines()]


def test_header_agrees_with_the_numpy_statement():
    for l0 in (1e-2, 1e-5, 3e-8, 1e-12):
        got = cxx_schedule(l0)
        ref = OQ.schedule(l0)
        assert len(got) == len(ref)
        for (a, b, c, E), (ar, br, cr, lo, hi) in zip(got, ref):
            np.testing.assert_allclose([a, b, c], [ar, br, cr], rtol=1e-5)   # (two solvers of 4 x 4 systems with rows of size l0 .. 1; each step starts from the previous one's E)
            np.testing.assert_allclose(E, 1 - lo, rtol=1e-4, atol=1e-15)


def test_schedule_drives_the_interval_to_one():
    # every x in [l0, 1] (and its mirror image, the polynomials are odd) ends within 1e-3 of 1, monotonically in the bound; a value
    # below l0 is never thrown back (it grows by the linear coefficient a > 1 per step)
    for l0 in (1e-3, 1e-6):
        sched = cxx_schedule(l0)
        x = np.concatenate([np.geomspace(l0, 1.0, 4001), np.geomspace(l0 * 1e-3, l0, 50)])
        lo_prev = l0
        for a, b, c, E in sched:
            xn = x * (a + x * x * (b + c * x * x))
            inside = x[:4001]
            assert np.all(np.abs(1 - xn[:4001]) <= E * (1 + 1e-9) + 1e-15)
            assert np.all(xn[4001:] > x[4001:])
            assert 1 - E > lo_prev
            lo_prev = 1 - E
            x = xn
        assert sched[-1][3] < 1e-3
        # products: 3 per step against Newton-Schulz's 2 per factor 1.5
        ns_products = 2 * np.log(1 / l0) / np.log(1.5)
        assert 3 * len(sched) < 0.6 * ns_products
