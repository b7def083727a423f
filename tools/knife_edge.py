#!/usr/bin/env python3
"""The one trajectory mismatch of the fuzz sweeps (seed 202, case 15: 100x100, rank 29, noise 1e-6, 120 iterations, never
converges, the rank oscillates) examined on the host alone - no GPU needed.

The count svp = #{sigma_i(Z) >= 1/mu} (src/robustPCA.jl:198) is a discontinuous function of Z, and a run that does not
converge keeps amplifying rounding differences.  This script replays the case with the oracle (LAPACK gesdd on the
column-major Z: what LinearAlgebra.svd! calls) and then lets *LAPACK disagree with itself*: the same loop with (a) the
other dense driver (gesvd), (b) gesdd handed the row-major copy of Z (it then factors Z'), (c) gesdd on Z with every entry
moved by one unit in the last place at one early iteration.  For each variant it prints the distance ||Z_k - Z_k(ref)|| over
the iterations and the first iteration whose count differs from the reference run.  If two LAPACK variants that are
equally "the reference" part ways, the trajectory past that point is not a property of the algorithm and no implementation
can be held to it; the parity claim for such a run is the distance of A and E at the end, relative to ||D||.
    python tools/knife_edge.py [seed] [case] [--big]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import scipy.linalg as sla
from oracle import rpca_oracle as O
import fuzz_parity as F


def case(seed, idx):
    rng = np.random.default_rng(seed)
    for _ in range(idx + 1):
        D, kw, desc = F.make_case(rng)
    return D, kw, desc


def run(D, kw, decomp):
    zs = []

    def hook(Z, sv):
        zs.append(Z.copy())
        return decomp(Z, len(zs) + 1)

    A, E, s, sv, info = O.rpca(D, svd=hook, **kw)
    return A, E, info, zs


def main():
    if "--big" in sys.argv:   # the case list of `fuzz_parity.py --big`
        sys.argv.remove("--big")
        F.BIG = True
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 202
    idx = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    D, kw, desc = case(seed, idx)
    print(f"seed {seed} case {idx}: {desc} {kw}")
    dn = np.linalg.norm(D)
    ref = lambda Z, k: O._svd_full(Z)                                                       # noqa: E731
    variants = {
        "gesvd driver": lambda Z, k: sla.svd(Z, full_matrices=False, lapack_driver="gesvd", check_finite=False),
        "gesdd on the row-major copy": lambda Z, k: np.linalg.svd(Z, full_matrices=False),
        "gesdd, Z + 1 ulp at k = 5": lambda Z, k: O._svd_full(np.nextafter(Z, np.inf) if k == 5 else Z),
    }
    A0, E0, i0, z0 = run(D, kw, ref)
    Ad, Ed, _, _, idf = O.rpca(D, **kw)
    assert idf.svp_hist == i0.svp_hist and np.array_equal(Ad, A0), "the observing hook changed the run"
    print(f"reference run (gesdd): {i0.iters_done} iterations, converged {i0.converged}, counts {i0.svp_hist[:6]} ... {i0.svp_hist[-6:]}")
    for name, f in variants.items():
        A, E, info, zs = run(D, kw, f)
        n = min(len(zs), len(z0))
        dist = [np.linalg.norm(zs[i] - z0[i]) / max(np.linalg.norm(z0[i]), 1e-300) for i in range(n)]
        kdiff = next((i + 1 for i, (a, b) in enumerate(zip(info.svp_hist, i0.svp_hist)) if a != b), None)
        marks = [k for k in (2, 10, 20, 40, 60, 80, 100, n) if k <= n]
        print(f"{name}: first count that differs from the reference run: k = {kdiff}; "
              f"|A - A_ref|/|D| = {np.linalg.norm(A - A0) / dn:.1e}, |E - E_ref|/|D| = {np.linalg.norm(E - E0) / dn:.1e}")
        print("    |Z_k - Z_k(ref)|/|Z_k| at k = " + ", ".join(f"{k}: {dist[k - 2]:.1e}" for k in marks))


if __name__ == "__main__":
    main()
