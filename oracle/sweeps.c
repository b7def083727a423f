/* CPU oracle helper — TEST INFRASTRUCTURE, NOT PRODUCT (see oracle/rpca_oracle.py header).
 *
 * Fused, OpenMP-threaded versions of the four elementwise broadcasts of the
 * reference's rpca loop, with the reference's exact expression order
 * (/root/reference/src/robustPCA.jl:188-192 and :221-222).  Compiled with
 * -ffp-contract=off so that no a*b+c is fused (Julia does not contract).
 * Used only so that the cpu_baseline in bench.py is not penalised by numpy
 * temporaries: Julia's broadcast is fused (single pass per statement).
 */
#include <stdint.h>

static inline double soft_th(double x, double e) {            /* robustPCA.jl:1 */
    double a = x - e, b = x + e;
    return (a > 0.0 ? a : 0.0) + (b < 0.0 ? b : 0.0);
}

/* E = soft_th((D-A)+(1/mu)Y, lam/mu); [E=max(E,0)]; Z = (D-E)+(1/mu)Y   (:188-192) */
void oracle_k1_f64(const double *D, const double *A, const double *Y, double *E, double *Z,
                   int64_t n, double inv_mu, double thr, int nonnegE) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double t = inv_mu * Y[i];
        double e = soft_th((D[i] - A[i]) + t, thr);
        if (nonnegE) e = e > 0.0 ? e : 0.0;
        E[i] = e;
        Z[i] = (D[i] - e) + t;
    }
}

/* Z = (D-A)-E ; Y = Y + mu*Z   (:221-222) */
void oracle_k2_f64(const double *D, const double *A, const double *E, double *Y, double *Z,
                   int64_t n, double mu, int unused) {
    (void)unused;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double z = (D[i] - A[i]) - E[i];
        Z[i] = z;
        Y[i] = Y[i] + mu * z;
    }
}
