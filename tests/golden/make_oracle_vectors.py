"""Writes tests/golden/oracle_vectors.json — frozen outputs of the CPU oracle (oracle/rpca_oracle.py, once it had
passed the reference's own known-answer vectors) at BASELINE config C1 (500x50) and a shrunken C2 (2000x128),
SURVEY.md §8c item (6): per-iteration cost and svp, the singular values of the last Z, Frobenius norms and a
strided sample of A and E.  The inputs come from the seeded generator (a checksum of D guards against generator
drift).  These are DATA produced by this repo's oracle, not by the reference (Julia cannot run here).

    python tests/golden/make_oracle_vectors.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rpca_oracle as O  # noqa: E402

CASES = {"c1_500x50_r5": (500, 50, 5, 0), "c2s_2000x128_r8": (2000, 128, 8, 1)}
STRIDE = 97


def main():
    out = {}
    for name, (M, N, r, seed) in CASES.items():
        D, _, _ = O.synth_lowrank_sparse(M, N, r, seed=seed)
        A, E, s, sv, info = O.rpca(D)
        out[name] = {
            "M": M, "N": N, "rank": r, "seed": seed,
            "D_sha256_of_float64_column_major": hashlib.sha256(np.asfortranarray(D).tobytes(order="F")).hexdigest(),
            "iters_done": info.iters_done, "sv": int(sv), "svp_hist": [int(v) for v in info.svp_hist],
            "cost_hist": [float(v) for v in info.cost_hist],
            "S": [float(v) for v in s[1]],
            "normA": float(np.linalg.norm(A)), "normE": float(np.linalg.norm(E)),
            "sample_stride": STRIDE,
            "A_sample": [float(v) for v in A.ravel(order="F")[::STRIDE]],
            "E_sample": [float(v) for v in E.ravel(order="F")[::STRIDE]],
        }
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.json"), "w") as f:
        json.dump(out, f)
    print({k: (v["iters_done"], v["sv"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
