// Optimal odd quintics for the matrix sign / inverse square root iterations (host side, no HIP: tests/test_quintic_cpu.py
// compiles this header with g++ and compares it with tools/odd_quintics.py).
//
// For 0 < l < u the odd quintic p(x) = a x + b x^3 + c x^5 with the smallest max |1 - p(x)| on [l, u] equioscillates at
// l, q, r, u (q < r: the interior extrema of p): p(l) = 1 - E, p(q) = 1 + E, p(r) = 1 - E, p(u) = 1 + E - four linear
// equations in (a, b, c, E) once q and r are fixed, and q, r follow from p'(x) = a + 3 b x^2 + 5 c x^4 = 0: a Remez exchange on
// two points.  p maps [l, u] onto [1 - E, 1 + E], the interval of the next step; a Newton-Schulz step multiplies an eigenvalue
// near zero by 1.5 for two products, the first quintic by 8.5 and the following ones by 4.26 for three.
#pragma once
#include <algorithm>
#include <cmath>

namespace tlsq {

struct OddQuintic {
    double a = 0.0, b = 0.0, c = 0.0;
    double E = 0.0;   // max |1 - p| on the interval it was made for
};

// false: the exchange broke down (never seen for 1e-14 <= l < u <= 2.5); the caller falls back to Newton-Schulz steps
inline bool odd_quintic(double l, double u, OddQuintic* out) {
    if (!(l > 0.0) || !(u > l)) return false;
    double q = l + 0.3 * (u - l), r = l + 0.8 * (u - l);
    double sol[4] = {0.0, 0.0, 0.0, 0.0};
    for (int iter = 0; iter < 200; ++iter) {
        const double pts[4] = {l, q, r, u};
        const double sg[4] = {-1.0, 1.0, -1.0, 1.0};
        double A[4][5];
        for (int i = 0; i < 4; ++i) {
            const double x = pts[i], x2 = x * x;
            A[i][0] = x;
            A[i][1] = x * x2;
            A[i][2] = x * x2 * x2;
            A[i][3] = -sg[i];
            A[i][4] = 1.0;
        }
        for (int k = 0; k < 4; ++k) {   // Gaussian elimination, partial pivoting
            int piv = k;
            for (int i = k + 1; i < 4; ++i)
                if (std::fabs(A[i][k]) > std::fabs(A[piv][k])) piv = i;
            if (!(std::fabs(A[piv][k]) > 0.0)) return false;
            if (piv != k)
                for (int j = 0; j < 5; ++j) std::swap(A[k][j], A[piv][j]);
            for (int i = k + 1; i < 4; ++i) {
                const double f = A[i][k] / A[k][k];
                for (int j = k; j < 5; ++j) A[i][j] -= f * A[k][j];
            }
        }
        for (int k = 3; k >= 0; --k) {
            double s = A[k][4];
            for (int j = k + 1; j < 4; ++j) s -= A[k][j] * sol[j];
            sol[k] = s / A[k][k];
        }
        const double a = sol[0], b = sol[1], c = sol[2];
        if (!std::isfinite(a) || !std::isfinite(b) || !std::isfinite(c) || !(c != 0.0)) return false;
        const double disc = 9.0 * b * b - 20.0 * a * c;
        if (!(disc > 0.0)) return false;
        const double z1 = (-3.0 * b - std::sqrt(disc)) / (10.0 * c), z2 = (-3.0 * b + std::sqrt(disc)) / (10.0 * c);
        const double zlo = std::min(z1, z2), zhi = std::max(z1, z2);
        if (!(zlo > 0.0)) return false;
        const double qn = std::min(std::max(std::sqrt(zlo), l), u), rn = std::min(std::max(std::sqrt(zhi), l), u);
        const bool done = std::fabs(qn - q) + std::fabs(rn - r) < 1e-15 * u;
        q = qn;
        r = rn;
        if (done) break;
    }
    out->a = sol[0];
    out->b = sol[1];
    out->c = sol[2];
    out->E = std::fabs(sol[3]);
    return std::isfinite(out->E) && out->E < 1.0 && out->a > 1.0;
}

}   // namespace tlsq
