#!/usr/bin/env python3
"""Optimal odd quintics for the matrix sign iteration (offline; the table goes into csrc/matfun.hip).

For an interval [l, u] (0 < l < u) the odd quintic p(x) = a x + b x^3 + c x^5 with the smallest max |1 - p(x)| on
[l, u] equioscillates at l, q, r, u (q, r: the interior extrema of p): four linear equations in (a, b, c, E) for given
q, r, then q, r from p' = 0 - a Remez iteration on two points.  Composing such steps (the image of [l, u] is
[1 - E, 1 + E], the next interval) drives every eigenvalue of a symmetric X with l <= |lambda| <= u to +-1 in far fewer
products than Newton-Schulz, whose small eigenvalues grow by 1.5 per step (here: by a ~ 8 per step while l is tiny).
Once E is small the classical cubic takes over (quadratic convergence, and its fixed points are exactly +-1).

    python tools/odd_quintics.py [l0=1e-6]      prints the schedule (a, b, c, image interval) per step
"""
import sys
import numpy as np


def best_quintic(l, u, iters=200):
    q, r = l + (u - l) * 0.3, l + (u - l) * 0.8
    for _ in range(iters):
        pts = np.array([l, q, r, u])
        sg = np.array([-1.0, 1.0, -1.0, 1.0])           # p(x_i) = 1 + sg_i E
        A = np.stack([pts, pts**3, pts**5, -sg], axis=1)
        a, b, c, E = np.linalg.solve(A, np.ones(4))
        # p'(x) = a + 3 b x^2 + 5 c x^4 = 0  ->  x^2 = (-3b -+ sqrt(9 b^2 - 20 a c)) / (10 c)
        disc = 9 * b * b - 20 * a * c
        if disc <= 0:
            break
        z1 = (-3 * b - np.sqrt(disc)) / (10 * c)
        z2 = (-3 * b + np.sqrt(disc)) / (10 * c)
        zs = sorted([z for z in (z1, z2) if z > 0])
        if len(zs) < 2:
            break
        qn, rn = np.sqrt(zs[0]), np.sqrt(zs[1])
        qn, rn = min(max(qn, l), u), min(max(rn, l), u)
        if abs(qn - q) + abs(rn - r) < 1e-15 * u:
            q, r = qn, rn
            break
        q, r = qn, rn
    return a, b, c, abs(E)


def schedule(l0, stop=1e-3, max_steps=24):
    l, u = l0, 1.0
    out = []
    for _ in range(max_steps):
        a, b, c, E = best_quintic(l, u)
        out.append((a, b, c, 1 - E, 1 + E))
        l, u = 1 - E, 1 + E
        if E < stop:
            break
    return out


if __name__ == "__main__":
    l0 = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-6
    for i, (a, b, c, lo, hi) in enumerate(schedule(l0)):
        print(f"step {i}: a={a:.17g} b={b:.17g} c={c:.17g}  -> [{lo:.6g}, {hi:.6g}]")
