"""Seeded synthetic workloads of the bench and the parity tests - numpy only, no solver code.

The reference has no data sets for this path: its tests draw `randn` panels (test/runtests.jl:141-201) and one sine
series with sparse outliers (test/runtests.jl:356-379).  These generators reproduce those models with seeded
`numpy.random.Generator`s so that bench.py, the golden-vector scripts and the tests get identical bits everywhere.
(The CPU checker under the repo's checker directory keeps identical copies; `tests/test_cabi_cpu.py` holds the two to each other.)
"""
import numpy as np


def synth_lowrank_sparse(M, N, rank, seed=0, sparse_frac=0.05, sparse_scale=10.0, dtype=np.float64):
    """D = G1 G2 + S,  G1 (M x r), G2 (r x N) iid N(0,1); S = scale*N(0,1)*Bernoulli(frac).  Returns D, A0, S
    (column-major)."""
    rng = np.random.default_rng(seed)
    G1 = rng.standard_normal((M, rank))
    G2 = rng.standard_normal((rank, N))
    A0 = G1 @ G2
    S = sparse_scale * rng.standard_normal((M, N)) * (rng.random((M, N)) < sparse_frac)
    D = np.asfortranarray((A0 + S).astype(dtype))
    return D, np.asfortranarray(A0.astype(dtype)), np.asfortranarray(S.astype(dtype))


def synth_series(N, seed=0):
    """The reference's lowrankfilter test signal scaled up (test/runtests.jl:356-379):
    y = sin(0.1 t)/q0.9 + 20 N(0,1) Bernoulli(0.01) + 0.1 N(0,1).  Returns y (clean), n (noise)."""
    rng = np.random.default_rng(seed)
    t = np.arange(1, N + 1, dtype=np.float64)
    y = np.sin(0.1 * t)
    y = y / np.quantile(np.abs(y), 0.9)
    n = 20 * rng.standard_normal(N) * (rng.random(N) < 0.01) + 0.1 * rng.standard_normal(N)
    return y, n


C4_SHAPE = (200000, 512, 16, 8)          # rows, columns, rank, seeded row blocks


def c4_rows(lo, hi):
    """Rows [lo, hi) of BASELINE config 4 (rpca 200000 x 512 fp64, rank 16 + 5 % sparse).  The matrix is defined by 8
    row blocks of 25000 rows with their own seeds, so that 1, 2, 4 and 8 ranks solve the same problem and every rank
    only generates its own rows.  Returns a C-ordered (hi-lo) x 512 array."""
    M4, N4, r4, nb = C4_SHAPE
    rb = M4 // nb
    G2 = np.random.default_rng([4, 999]).standard_normal((r4, N4))
    parts = []
    for b in range(lo // rb, (hi - 1) // rb + 1):
        rg = np.random.default_rng([4, b])
        blk = rg.standard_normal((rb, r4)) @ G2 + 10.0 * rg.standard_normal((rb, N4)) * (rg.random((rb, N4)) < 0.05)
        parts.append(blk[max(lo - b * rb, 0): min(hi - b * rb, rb)])
    return np.vstack(parts)
