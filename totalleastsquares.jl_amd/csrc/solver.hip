// The inexact-ALM driver of rpca (src/robustPCA.jl:156-239 under /root/reference) and the small-matrix machinery
// behind its SVD step: Gram / implicit operator, warm-started subspace iteration with a Lanczos count certificate,
// dense Jacobi fall-backs, the two-level decomposition of late iterations, the low-rank rebuild, and the staging
// of caller memory (rpca_entry).  All arithmetic on M x N data runs in the HIP kernels of sweeps.hip / gemm.hip /
// jacobi.hip / subspace.hip; the host only does O(N) bookkeeping (rank count, sort, the q x q TLS partition solve).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <limits>
#include <numeric>
#include <thread>

#include "svdstep.hpp"

namespace tlsq {

// sqrt(lambda_max) of the Gram already sitting in G (N x N, ld N): Lanczos to the requested relative residual
// bound, exact Jacobi eigenvalues as the fallback.  uses WS_B, WS_LAM.
int sigma_max_of_gram(Handle* h, const double* G, int64_t N, double rel_tol, double* out, int64_t* sweeps,
                      double stop_above_sigma) {
    double lmax = 0.0;
    int steps = 0;
    // A tight answer with no yes/no shortcut (the set-up norm ||D||_2, src/robustPCA.jl:177): plain Lanczos needs hundreds
    // of dependent launches when the top of the spectrum is a cluster (a rank-16 D: 239 steps = 1.7 ms at N = 512).
    // Five squarings G -> G^32 (MFMA, each rescaled to unit Frobenius norm) concentrate any column on the dominant
    // cluster; started from the dominant column, the Krylov space is that of a ~16-dimensional problem.
    const double* v0 = nullptr;
    const bool no_pow_start = dev_is(DEV_NO_POWER_START, '1');
    if (stop_above_sigma == 0.0 && rel_tol <= 1e-9 && N >= 64 && N <= 1024 && !no_pow_start) {
        void *P1, *P2, *part, *vst;
        TLSQ_TRY(ws_get(h, WS_CP1, (size_t)N * N * 8, &P1));
        TLSQ_TRY(ws_get(h, WS_CP2, (size_t)N * N * 8, &P2));
        TLSQ_TRY(ws_get(h, WS_CPART, (size_t)std::max<int64_t>(4096, ((N + 31) / 32) * ((N + 31) / 32 + 1) / 2) * 8, &part));   // (norm partials: one per 32 x 32 block of the lower triangle)
        TLSQ_TRY(ws_get(h, WS_PW, (8 + 3 * (size_t)N + 2 * 1024) * 8, &vst));   // (same size as power_lower_bound asks for)
        const double* src = G;
        bool quick = false;
        // (N a multiple of 128: one scaling by the trace and five slab-free products, 8 us each at N = 512, instead of five
        //  split-K products + slab reductions + rescalings, 26 us each)
        // (round 5, measured: eight squarings - G^256, POWER_LEVELS=8 - change nothing at C2: the 15 Lanczos steps that remain
        //  are the dimension of the dominant cluster of a rank-16 D, not the distance of the start vector from it)
        const int plev = [] { const char* e = dev_get(DEV_POWER_LEVELS); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 10 ? v : 5; }();
        TLSQ_TRY(matfun_power_start(h, G, N, (double*)P1, (double*)P2, plev, &src, &quick));
        if (!quick) {
            double* dst = (double*)P1;
            for (int k = 0; k < 5; ++k) {
                int nb = 0;
                TLSQ_TRY(gemm_mixed(h, true, true, src, 0, N, src, 0, N, dst, 0, N, N, N, N, true, nullptr, (double*)part, &nb));
                TLSQ_TRY(launch_scale_by_norm(h, dst, N, (const double*)part, nb));
                src = dst;
                dst = (dst == (double*)P1) ? (double*)P2 : (double*)P1;
            }
        }
        TLSQ_TRY(launch_dominant_column(h, src, N, (double*)vst + 8));
        v0 = (const double*)vst + 8;
    }
    int st = lanczos_lmax_f64(h, G, N, N, rel_tol, 1000, &lmax, &steps, 0.0, stop_above_sigma * stop_above_sigma, v0);
    if (st < 0) return st;
    const bool dbg = dev_get(DEV_DEBUG) != nullptr;
    if (dbg) fprintf(stderr, "  opnorm Lanczos: %d steps (rel_tol %.0e, stop %.3e) -> sigma %.6e\n", steps, rel_tol, stop_above_sigma, std::sqrt(lmax));
    if (st == 0 || N > kFullEigMaxN) {   // large mode: no dense fallback; the Lanczos value after 1000 steps stands
        *out = std::sqrt(lmax);
        return TLSQ_OK;
    }
    void *B, *lam;
    TLSQ_TRY(ws_get(h, WS_B, (size_t)N * N * 8, &B));
    TLSQ_TRY(ws_get(h, WS_LAM, (size_t)N * 8, &lam));
    int64_t sw = 0;
    TLSQ_TRY(symeig_f64(h, G, N, N, (double*)B, nullptr, false, (double*)lam, &sw));
    if (sweeps) *sweeps += sw;
    std::vector<double> hl((size_t)N);
    TLSQ_HIP(h, hipMemcpyAsync(hl.data(), lam, (size_t)N * 8, hipMemcpyDeviceToHost, h->stream));
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    double mx = 0.0;
    for (double v : hl) mx = v > mx ? v : mx;
    *out = std::sqrt(mx);
    return TLSQ_OK;
}

// sigma_max of Z (device M x N, ld) = the default `opnorm`; uses WS_G.
template <typename T>
int opnorm_gram(Handle* h, const T* Z, int64_t M, int64_t N, int64_t ld, double* out, int64_t* sweeps,
                double rel_tol, double stop_above_sigma, int gslot) {
    void* G;
    TLSQ_TRY(ws_get(h, gslot, (size_t)N * N * 8, &G));
    TLSQ_TRY(gram_any(h, Z, Prec<T>::f32, M, N, ld, (double*)G, N));
    TLSQ_TRY(comm_allreduce(h, (double*)G, (size_t)N * N, ncclSum));
    return sigma_max_of_gram(h, (const double*)G, N, rel_tol, out, sweeps, stop_above_sigma);
}

// The `opnorm = x -> rnorm(x, mvps)` hook of the reference's tests (test/runtests.jl:384-398,
// RandomizedLinAlg.rnorm): probabilistic upper bound  alpha*sqrt(2/pi)*max_i ||Z w_i||,  w_i ~ N(0,I),
// i = 1..mvps, alpha = 0.05^(-1/mvps)  (Halko, Martinsson, Tropp 2011, Lemma 4.1).  One skinny GEMM over Z.
template <typename T>
static int opnorm_power(Handle* h, const T* Z, int64_t M, int64_t N, int64_t ld, int mvps, uint64_t seed,
                        double* out) {
    if (mvps < 1) mvps = 1;
    if (mvps > 64) mvps = 64;
    void *Om, *T1, *nn;
    TLSQ_TRY(ws_get(h, WS_VG, (size_t)N * mvps * 8, &Om));
    TLSQ_TRY(ws_get(h, WS_T, (size_t)M * mvps * 8, &T1));
    TLSQ_TRY(ws_get(h, WS_LAM, (size_t)std::max<int64_t>(N, 64) * 8, &nn));
    TLSQ_TRY(launch_fill_gauss(h, (double*)Om, N * mvps, (unsigned int)(seed * 2654435761ull + 0x1234567u)));
    TLSQ_TRY(gemm_mixed(h, true, false, Om, 0, N, Z, Prec<T>::f32, ld, T1, 0, M, mvps, M, N, false));
    TLSQ_TRY(launch_colsumsq(h, (const double*)T1, M, M, mvps, (double*)nn));
    TLSQ_TRY(comm_allreduce(h, (double*)nn, (size_t)mvps, ncclSum));
    std::vector<double> hn((size_t)mvps);
    TLSQ_HIP(h, hipMemcpyAsync(hn.data(), nn, (size_t)mvps * 8, hipMemcpyDeviceToHost, h->stream));
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    double mx = 0.0;
    for (double v : hn) mx = std::max(mx, v);
    const double alpha = std::pow(0.05, -1.0 / (double)mvps);
    *out = alpha * std::sqrt(2.0 / M_PI) * std::sqrt(mx);
    return TLSQ_OK;
}


// full eigen-decomposition of G by the block Jacobi solver: V in WS_V
// does the full solver go through the Cholesky factor (zero columns for numerically-zero eigenvalues)?
static bool chol_route(int64_t N) {
    const bool no_chol = dev_is(DEV_NO_CHOL, '1');
    return N > 64 && !no_chol;
}

int eig_full(Handle* h, const double* G, int64_t N, double** V_out, SmallSvd& s, int64_t* sweeps, bool allow_warm, int vslot,
             bool need_all_vectors, double lam_hi, int n_out, double val_out, double bulk_hi) {
    void *B, *V, *lam;
    TLSQ_TRY(ws_get(h, WS_B, (size_t)N * N * 8, &B));
    TLSQ_TRY(ws_get(h, vslot, (size_t)N * N * 8, &V));
    TLSQ_TRY(ws_get(h, WS_LAM, (size_t)N * 8, &lam));
    int64_t sw = 0;
    const bool dbg = dev_get(DEV_DEBUG) != nullptr;
    s.sigma.resize((size_t)N);
    if (chol_route(N) && !need_all_vectors) {
        // Cholesky-preconditioned route: Jacobi on L = chol(G + delta I); far fewer sweeps on graded spectra and
        // no eigenvector accumulation.  Vectors of numerically-zero eigenvalues come back as zero columns, which
        // is fine for the ALM loop (only sigma_i >= 1/mu are used).
        double delta = 0.0;
        // (round 6: the spectrum cut into slices of ~64 eigenvalues first - sliced.hip - where the shape allows; same result,
        //  a third of the sequential Jacobi rounds)
        bool sliced = false;
        if (symeig_sliced_ok(N))
            TLSQ_TRY(symeig_sliced_f64(h, G, N, N, (double*)B, (double*)V, (double*)lam, &delta, &sw, lam_hi, n_out, val_out, bulk_hi, &sliced));
        if (!sliced) TLSQ_TRY(symeig_chol_f64(h, G, N, N, (double*)B, (double*)V, (double*)lam, &delta, &sw));
        if (vslot == WS_V) h->warm_n = 0;
        if (dbg) fprintf(stderr, "  full eig (chol%s) N=%lld sweeps=%lld\n", sliced ? ", sliced" : "", (long long)N, (long long)sw);
        if (sweeps) *sweeps += sw;
        TLSQ_HIP(h, hipMemcpyAsync(s.sigma.data(), lam, (size_t)N * 8, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        for (auto& v : s.sigma) v = std::sqrt(std::max(v * v - delta, 0.0));   // sigma(L)^2 = lambda + delta
    } else {
        // consecutive ALM iterations see nearly the same eigenvectors: reuse them (cold restart every 8th time so
        // that rounding drift in the accumulated rotations cannot build up)
        const bool warm = allow_warm && vslot == WS_V && h->warm_n == N && h->warm_uses < 8 && V == h->ws[WS_V].p;
        TLSQ_TRY(symeig_f64(h, G, N, N, (double*)B, (double*)V, true, (double*)lam, &sw, false, warm));
        if (vslot == WS_V) {
            h->warm_n = N;
            h->warm_uses = warm ? h->warm_uses + 1 : 0;
        }
        if (dbg) fprintf(stderr, "  full eig N=%lld warm=%d sweeps=%lld\n", (long long)N, (int)warm, (long long)sw);
        if (sweeps) *sweeps += sw;
        TLSQ_HIP(h, hipMemcpyAsync(s.sigma.data(), lam, (size_t)N * 8, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        for (auto& v : s.sigma) v = std::sqrt(v);
    }
    s.ncols = N;
    sort_desc(s);
    *V_out = (double*)V;
    return TLSQ_OK;
}

template <typename T>
int svd_via_gram(Handle* h, const T* Z, int64_t M, int64_t N, int64_t ld, double** V_out,
                        SmallSvd& s, int64_t* sweeps, PhaseTimer* pt) {
    double* G;
    TLSQ_TRY(gram_allreduce<T>(h, Z, M, N, ld, &G));
    if (pt) pt->mark();
    return eig_full(h, G, N, V_out, s, sweeps, false, WS_V, true);   // callers (tls!, SSA truncation) want every vector
}

// The accurate route (the fp64 default of the north star): Householder TSQR of the panel (tsqr.hip),
// then one-sided Jacobi on R' (jacobi.hip).  Both steps are orthogonal transformations, so every singular value
// carries an absolute error of a few eps * sigma_max like LAPACK's gesdd (src/robustPCA.jl:194) — the Gram route only
// reaches eps * sigma_max^2 / sigma.  V in WS_V (all N right singular vectors), s = all N singular values.

template <typename T>
int svd_via_r(Handle* h, const T* Z, int64_t M, int64_t N, int64_t ld, double** V_out, SmallSvd& s, int64_t* sweeps) {
    void *B, *V, *lam;
    TLSQ_TRY(ws_get(h, WS_B, (size_t)N * N * 8, &B));
    TLSQ_TRY(ws_get(h, WS_V, (size_t)N * N * 8, &V));
    TLSQ_TRY(ws_get(h, WS_LAM, (size_t)N * 8, &lam));
    TLSQ_TRY(tsqr_lt(h, Z, Prec<T>::f32, M, N, ld, (double*)B));
    // A rank-deficient panel (zero columns of D stay zero columns of Z; exact low rank): the columns of R' that belong to the
    // zero singular values come out of the rotations as rounding noise, eps sqrt(N) ||R|| in norm, whose mutual angles never
    // settle - the sweeps would not end.  Seen on the diagonal of R (a zero or negligible pivot): such columns are then kept
    // out of the rotations (noise floor 8 sqrt(N) eps ||R||_F - below what any singular value is known to anyway), reported
    // as sigma = 0, and their right singular vectors are completed to an orthonormal basis afterwards (what LAPACK returns
    // for a null space: any orthonormal basis of it).
    const double eps = 2.220446049250313e-16;
    double floor_rel = 0.0;
    {
        std::vector<double> dg((size_t)N);
        TLSQ_HIP(h, hipMemcpy2DAsync(dg.data(), 8, B, (size_t)(N + 1) * 8, 8, (size_t)N, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        double dmax = 0.0, dmin = std::numeric_limits<double>::infinity();
        for (double v : dg) {
            dmax = std::max(dmax, std::fabs(v));
            dmin = std::min(dmin, std::fabs(v));
        }
        if (!(dmin > 8.0 * std::sqrt((double)N) * eps * dmax)) floor_rel = 8.0 * std::sqrt((double)N) * eps;
    }
    int64_t sw = 0;
    TLSQ_TRY(jacobi_factor_f64(h, (double*)B, N, (double*)V, (double*)lam, floor_rel, &sw));
    if (sweeps) *sweeps += sw;
    h->warm_n = 0;
    const bool dbg = dev_get(DEV_DEBUG) != nullptr;
    if (dbg) fprintf(stderr, "  svd via R: N=%lld sweeps=%lld floor=%.1e\n", (long long)N, (long long)sw, floor_rel);
    s.sigma.resize((size_t)N);
    TLSQ_HIP(h, hipMemcpyAsync(s.sigma.data(), lam, (size_t)N * 8, hipMemcpyDeviceToHost, h->stream));
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    s.ncols = N;
    sort_desc(s);
    *V_out = (double*)V;
    if (floor_rel > 0.0) {
        double fro2 = 0.0;
        for (double v : s.sigma) fro2 += v * v;
        const double cut = floor_rel * std::sqrt(fro2);
        int64_t ngood = 0;
        while (ngood < N && s.sigma[(size_t)s.order[(size_t)ngood]] > cut) ++ngood;
        if (ngood < N) {
            // sorted copy [kept columns | seeded random columns], block Gram-Schmidt of the second part against the first
            // (launch_orth with c_start: the kept columns are orthonormal already and stay as they are)
            void *P, *wk, *stv;
            TLSQ_TRY(ws_get(h, WS_CP1, (size_t)N * N * 8, &P));
            std::vector<int32_t> keep(s.order.begin(), s.order.begin() + ngood);
            TLSQ_TRY(gather_cols(h, (const double*)V, N, keep, (double*)P));
            TLSQ_TRY(launch_fill_hash(h, (double*)P + (size_t)N * ngood, N * (N - ngood), 0x27D4EB2Fu));
            TLSQ_TRY(ws_get(h, WS_SH, (size_t)std::max<int64_t>(N * 16, 64) * 8, &wk));
            TLSQ_TRY(ws_get(h, WS_AUX1, 64 * 8, &stv));
            bool used = false;
            TLSQ_TRY(launch_orth(h, (double*)P, nullptr, (double*)wk, N, N, (double*)stv, false, &used, false, ngood));
            TLSQ_HIP(h, hipMemcpyAsync(V, P, (size_t)N * N * 8, hipMemcpyDeviceToDevice, h->stream));
            std::vector<double> sg((size_t)N, 0.0);
            for (int64_t i = 0; i < ngood; ++i) sg[(size_t)i] = s.sigma[(size_t)s.order[(size_t)i]];
            s.sigma = sg;
            std::iota(s.order.begin(), s.order.end(), 0);
            if (dbg) fprintf(stderr, "  svd via R: %lld numerically zero singular values, null-space basis completed\n", (long long)(N - ngood));
        }
    }
    return TLSQ_OK;
}

template <typename T>
static int rebuild_factors(Handle* h, const T* Z, int64_t M, int64_t N, int64_t ldZ, const double* V,
                           const std::vector<int32_t>& sel, const std::vector<double>& g, const double** Tm_out,
                           const double** Vs_out, int slot = 0) {
    const int64_t r = (int64_t)sel.size();
    *Tm_out = nullptr;
    *Vs_out = nullptr;
    if (r == 0) return TLSQ_OK;
    void *Vg, *Vs, *T1, *aux;
    TLSQ_TRY(ws_get(h, WS_VG, (size_t)N * r * 8, &Vg));
    // (slot 1 / 2: buffers of the E-free loop, which keeps the previous iteration's factors alive through the next SVD step
    //  - WS_VS itself is scratch of the count certificate in svd_subspace)
    TLSQ_TRY(ws_get(h, slot == 1 ? WS_VS2 : slot == 2 ? WS_VS3 : WS_VS, (size_t)N * r * 8, &Vs));
    TLSQ_TRY(ws_get(h, slot == 1 ? WS_T2 : slot == 3 ? WS_T3 : WS_T, (size_t)M * r * 8, &T1));   // (3: scratch of the deflated certificate)
    TLSQ_TRY(ws_get(h, WS_AUX0, (size_t)r * 16, &aux));
    // T (M x r, fp64) = Z * Vg
    const bool no_tsmm = dev_is(DEV_NO_TSMM, '1');
    const int64_t tsmm_max = [] { const char* e = dev_get(DEV_TSMM_MAXR); return (int64_t)(e ? atoi(e) : 96); }();
    const bool no_sel = dev_is(DEV_NO_TSMM_SEL, '1');
    if (r <= 32 && r <= tsmm_max && !no_tsmm && !no_sel) {
        // short lists: selection and weights travel as kernel arguments, V[:, sel] diag(g) is gathered straight into the
        // packed operand of the factor product (no Vg panel, one launch less), Vs on the way
        SelWeights sw;
        for (int64_t i = 0; i < 32; ++i) {
            sw.sel[i] = i < r ? sel[(size_t)i] : 0;
            sw.w[i] = i < r ? g[(size_t)i] : 0.0;
        }
        TLSQ_TRY(tsmm_sel(h, Z, Prec<T>::f32, ldZ, V, sw, (double*)Vs, (double*)T1, M, M, N, r));
        *Tm_out = (const double*)T1;
        *Vs_out = (const double*)Vs;
        return TLSQ_OK;
    }
    TLSQ_TRY(gather_scale_host(h, V, N, sel, g, aux, (double*)Vg, (double*)Vs));
    if (r <= tsmm_max && r <= 96 && !no_tsmm) {
        TLSQ_TRY(tsmm_mixed(h, Z, Prec<T>::f32, ldZ, (const double*)Vg, N, (double*)T1, M, M, N, r));
    } else {
        TLSQ_TRY(gemm_mixed(h, true, false, Vg, 0, N, Z, Prec<T>::f32, ldZ, T1, 0, M, r, M, N, false));
    }
    *Tm_out = (const double*)T1;
    *Vs_out = (const double*)Vs;
    return TLSQ_OK;
}

// Aout (M x N, ldA) = Tm * Vs'   (r = 0  =>  A = 0: mul! with inner dimension 0, src/robustPCA.jl:207-208)
template <typename T>
static int rebuild_from_factors(Handle* h, const double* Tm, const double* Vs, int64_t M, int64_t N, int64_t r,
                                T* Aout, int64_t ldA) {
    if (r == 0) {
        TLSQ_HIP(h, hipMemset2DAsync(Aout, (size_t)ldA * sizeof(T), 0, (size_t)M * sizeof(T), (size_t)N, h->stream));
        return TLSQ_OK;
    }
    const bool no_store = dev_is(DEV_NO_REBUILD_STORE, '1');
    if (!no_store && rebuild_store_ok<T>(Aout, M, N, ldA, r)) return launch_rebuild_store<T>(h, Tm, Vs, Aout, M, N, r);
    TLSQ_TRY(gemm_mixed(h, false, false, Vs, 0, N, Tm, 0, M, Aout, Prec<T>::f32, ldA, N, M, r, false));
    return TLSQ_OK;
}

// Aout (M x N, ldA) = Z * V[:,sel] * diag(g) * V[:,sel]'
template <typename T>
int rebuild_lowrank(Handle* h, const T* Z, int64_t M, int64_t N, int64_t ldZ,
                           const double* V, const std::vector<int32_t>& sel,
                           const std::vector<double>& g, T* Aout, int64_t ldA) {
    const double *Tm, *Vs;
    TLSQ_TRY(rebuild_factors<T>(h, Z, M, N, ldZ, V, sel, g, &Tm, &Vs));
    return rebuild_from_factors<T>(h, Tm, Vs, M, N, (int64_t)sel.size(), Aout, ldA);
}

static double large_mode_noise(int64_t N, int64_t m_global) {
    if (N > kFullEigMaxN) return 0.0;
    const double eps = 2.220446049250313e-16;
    return 16.0 * std::max((double)N, std::sqrt((double)m_global)) * eps + 2e-12;
}

// ---- orthonormal polish of the derived singular vectors ------------------------------------------------------------------
// The accurate route delivers the right singular vectors V (one-sided Jacobi: orthonormal to rounding whatever the singular
// value) and the left ones as U = Z V diag(1/sigma).  Z v_i carries an absolute error eps ||Z||, so u_i is contaminated by the
// dominant directions at the level eps sigma_max / sigma_i: for the tail of an rpca panel (sigma_i ~ 1e-8 sigma_max) U'U = I only
// to ~1e-8, and a column with sigma_i = 0 has no direction at all - LAPACK returns orthonormal vectors in both cases
// (src/robustPCA.jl:194, :238 hand back its `s`).  Repair in Gram-Schmidt order, so that the well-determined columns stay as
// they are: B = U'U (one pass over U on the MFMA), unit diagonal by scaling, C = D (I - striu(D B D)) - the first-order
// inverse of the Cholesky factor - and U <- U C: the departure from orthonormality is squared by every pass (B' = I + O(E^2)),
// two passes in the usual case.  Columns with sigma_i <= eps max(M,N) sigma_max start from seeded normals (they span the
// left null space when the passes are done).  U S V' = Z is unchanged to rounding: column i moves by ~eps sigma_max / sigma_i,
// which S multiplies back down.  (tools/fuzz_misc.py, "returned_s", found U'U - I at 1e-6 ... 1e-4 and 1e4 for a zero
// singular value.)
template <typename T>
__global__ __launch_bounds__(256) void k_rand_cols(T* __restrict__ U, int64_t M, int64_t M_true, int64_t ld, int64_t c0,
                                                   int64_t nc, unsigned int seed, double scale) {
    const int64_t total = M * nc, stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        const int64_t r = e % M, c = e / M;
        unsigned int a = (unsigned int)e * 2654435761u + seed;
        a ^= a >> 16; a *= 2246822519u; a ^= a >> 13; a *= 3266489917u; a ^= a >> 16;
        unsigned int b = (unsigned int)e * 40503u + (seed ^ 0x68E31DA4u) + 0x9E3779B9u;
        b ^= b >> 16; b *= 2246822519u; b ^= b >> 13; b *= 3266489917u; b ^= b >> 16;
        const double u1 = ((double)a + 1.0) / 4294967297.0, u2 = ((double)b + 0.5) / 4294967296.0;
        U[r + (c0 + c) * ld] = r < M_true ? (T)(scale * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2)) : (T)0;
    }
}
// C (d x d, ld d) = D (I - striu(D B D)), D = diag(B)^-1/2; stat[0] = max |(D B D)_ij| (i != j), stat[1] = max |B_ii - 1|,
// stat[2] != 0: a diagonal entry that is not positive and finite (as bit patterns of non-negative doubles: atomicMax)
__global__ __launch_bounds__(256) void k_gs_correction(const double* __restrict__ B, int d, double* __restrict__ C,
                                                       unsigned long long* __restrict__ stat) {
    const int64_t total = (int64_t)d * d, stride = (int64_t)gridDim.x * 256;
    double e_off = 0.0, e_diag = 0.0, badv = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        const int i = (int)(e % d), j = (int)(e / d);
        const double bii = B[i + (int64_t)i * d], bjj = B[j + (int64_t)j * d];
        const bool ok = bii > 0.0 && bii < 1e300 && bjj > 0.0 && bjj < 1e300;
        const double di = ok ? 1.0 / sqrt(bii) : 0.0, dj = ok ? 1.0 / sqrt(bjj) : 0.0;
        if (!ok) badv = 1.0;
        if (i == j) {
            C[e] = di;
            e_diag = fmax(e_diag, fabs(bii - 1.0));
        } else {
            const double bh = B[e] * di * dj;
            e_off = fmax(e_off, fabs(bh));
            C[e] = i < j ? -di * bh : 0.0;   // column j is corrected by the columns before it only
        }
    }
    // (one atomic per workgroup: every thread hitting the same three words took 100 us at d = 512)
    __shared__ double sm[3][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double nanv = e_off == e_off ? 0.0 : 1.0;
    if (!(e_off == e_off)) e_off = 0.0;
    if (!(e_diag == e_diag)) e_diag = 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        e_off = fmax(e_off, __shfl_xor(e_off, off, 64));
        e_diag = fmax(e_diag, __shfl_xor(e_diag, off, 64));
        badv = fmax(fmax(badv, nanv), __shfl_xor(fmax(badv, nanv), off, 64));
    }
    if (lane == 0) {
        sm[0][w] = e_off;
        sm[1][w] = e_diag;
        sm[2][w] = badv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double eo = fmax(fmax(sm[0][0], sm[0][1]), fmax(sm[0][2], sm[0][3]));
        const double ed = fmax(fmax(sm[1][0], sm[1][1]), fmax(sm[1][2], sm[1][3]));
        const double bd = fmax(fmax(sm[2][0], sm[2][1]), fmax(sm[2][2], sm[2][3]));
        atomicMax(&stat[0], (unsigned long long)__double_as_longlong(eo));
        atomicMax(&stat[1], (unsigned long long)__double_as_longlong(ed));
        if (bd != 0.0) atomicMax(&stat[2], 1ull);
    }
}

template <typename T>
static int polish_derived_vectors(Handle* h, T* U, int64_t M, int64_t d, const std::vector<double>& sig_desc, int64_t m_global,
                                  bool replicated = false) {
    if (d <= 0 || M <= 0 || dev_is(DEV_NO_U_POLISH, '1')) return TLSQ_OK;
    // replicated: the panel is the same on every rank of a group (the N x N right vectors), not a row shard: no collectives
    struct CommOff {
        Handle* h;
        Comm* saved;
        CommOff(Handle* hh, bool off) : h(hh), saved(hh->comm) {
            if (off) h->comm = nullptr;
        }
        ~CommOff() { h->comm = saved; }
    } comm_off(h, replicated);
    const double eps_t = (double)std::numeric_limits<T>::epsilon();
    const double smax = sig_desc.empty() ? 0.0 : sig_desc[0];
    // trailing columns without a direction (nr: replaced by seeded normals) and - a superset - those whose contamination
    // eps sigma_max / sigma_i exceeds ~1e-4 (nz): outside the comfortable basin of the first-order passes, they get a plain
    // Gram-Schmidt against everything in front of them
    int64_t nr = 0, nz = 0;
    for (int64_t p = d - 1; p >= 0; --p) {
        if (sig_desc[(size_t)p] > eps_t * (double)std::max<int64_t>(m_global, d) * smax && smax > 0.0) break;
        ++nr;
    }
    // Column i of U = Z V / sigma carries ~eps sigma_max / sigma_i of the other directions.  The first-order passes below
    // square the departure from orthonormality as long as the whole of it is small in norm: contaminations of c per column
    // add up to ~c sqrt(d), kept below 0.1 - i.e. columns with sigma_i <= 10 sqrt(d) eps sigma_max (fp32 at d = 512: 2.7e-5
    // sigma_max, far below the tail of an rpca panel) are NOT left to them: they get a proper Gram-Schmidt on the device.
    // (the round-3 bound, 1e4 eps, sent every tail column of a Float32 call to a host-side Gram-Schmidt.)
    const double bar = 10.0 * std::sqrt((double)d) * eps_t * smax;
    for (int64_t p = d - 1; p >= 0; --p) {
        if (sig_desc[(size_t)p] > bar && smax > 0.0) break;
        ++nz;
    }
    nz = std::max(nz, nr);
    if (nr > 0) {
        int64_t g = (M * nr + 255) / 256;
        if (g > 2048) g = 2048;
        // (an unsharded panel may carry zero pad rows behind its m_global true ones - rpca_entry's 16-row alignment: every
        //  other column of U is zero there, and the caller cuts them off)
        const int64_t M_true = h->comm ? M : std::min<int64_t>(M, m_global);
        hipLaunchKernelGGL(k_rand_cols<T>, dim3((int)g), dim3(256), 0, h->stream, U, M, M_true, M, d - nr, nr,
                           0x51ed270bu + 977u * (unsigned int)h->rank, 1.0 / std::sqrt((double)std::max<int64_t>(m_global, 1)));
        TLSQ_HIP(h, hipGetLastError());
    }
    void *tmpv, *bv;
    TLSQ_TRY(ws_get(h, WS_UPOL, (size_t)M * d * sizeof(T), &tmpv));
    TLSQ_TRY(ws_get(h, WS_UPB, (size_t)d * d * 16 + 64, &bv));
    double* B = (double*)bv;
    double* C = B + (size_t)d * d;
    unsigned long long* stat = (unsigned long long*)(C + (size_t)d * d);
    const double tol_orth = std::max(2e-13, 64.0 * eps_t);
    // first-order passes on the leading nc columns of U (in place: the result ends up in U)
    auto first_order = [&](int64_t nc) -> int {
        if (nc <= 0) return TLSQ_OK;
        T* cur = U;
        T* oth = (T*)tmpv;
        for (int pass = 0; pass < 10; ++pass) {
            TLSQ_TRY(gram_any(h, cur, Prec<T>::f32, M, nc, M, B, nc, 0));
            if (h->comm) TLSQ_TRY(comm_allreduce(h, B, (size_t)nc * nc, ncclSum));   // row shards: the Gram adds up
            TLSQ_HIP(h, hipMemsetAsync(stat, 0, 24, h->stream));
            int64_t g = (nc * nc + 255) / 256;
            if (g > 1024) g = 1024;
            hipLaunchKernelGGL(k_gs_correction, dim3((int)g), dim3(256), 0, h->stream, (const double*)B, (int)nc, C, stat);
            TLSQ_HIP(h, hipGetLastError());
            unsigned long long hs[3];
            TLSQ_HIP(h, hipMemcpyAsync(hs, stat, 24, hipMemcpyDeviceToHost, h->stream));
            TLSQ_HIP(h, hipStreamSynchronize(h->stream));
            double e_off, e_diag;
            memcpy(&e_off, &hs[0], 8);
            memcpy(&e_diag, &hs[1], 8);
            if (dev_get(DEV_DEBUG)) fprintf(stderr, "  polish pass %d: e_off=%.3e e_diag=%.3e bad=%llu (%lld of d=%lld columns, nz=%lld, nr=%lld)\n", pass, e_off, e_diag, hs[2], (long long)nc, (long long)d, (long long)nz, (long long)nr);
            if (hs[2] != 0ull) break;                                  // not finite: leave U as it is
            if (e_off <= tol_orth && e_diag <= tol_orth) break;        // orthonormal to working precision
            if (!(e_off < 0.7)) break;                                 // outside the basin of the first-order step (never seen)
            TLSQ_TRY(gemm_mixed(h, true, false, C, 0, nc, cur, Prec<T>::f32, M, oth, Prec<T>::f32, M, nc, M, nc, false));
            std::swap(cur, oth);
        }
        if (cur != U) TLSQ_HIP(h, hipMemcpyAsync(U, cur, (size_t)M * nc * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
        return TLSQ_OK;
    };
    if (nz > 0 && !h->comm) {
        // The well-determined columns first; then the nz trailing ones by block Gram-Schmidt with re-orthogonalisation on the
        // device (subspace.hip, launch_orth with c_start: every 16-column block is projected twice against everything in front
        // of it - two multi-workgroup products per projection - and orthonormalised by the column-sequential CGS2 kernel), in
        // fp64 whatever the panel's type.  A column without any direction left (M == its index) stays zero, like sigma = 0 before.
        const int64_t nq = d - nz;
        TLSQ_TRY(first_order(nq));
        double* Uw = nullptr;
        void* wv = nullptr;
        if (Prec<T>::f32) {
            TLSQ_TRY(ws_get(h, WS_QRW, (size_t)M * d * 8, &wv));
            Uw = (double*)wv;
            TLSQ_TRY((launch_convert<T, double>(h, U, Uw, M * d)));
        } else {
            Uw = reinterpret_cast<double*>(U);
        }
        void *wk, *stv;
        TLSQ_TRY(ws_get(h, WS_SH, (size_t)std::max<int64_t>(d * 16, 64) * 8, &wk));     // c0 x 16 projections
        TLSQ_TRY(ws_get(h, WS_LAM, (size_t)std::max<int64_t>(d, 64) * 8, &stv));
        bool used = false;
        TLSQ_TRY(launch_orth(h, Uw, nullptr, (double*)wk, M, d, (double*)stv, false, &used, false, nq));
        if (Prec<T>::f32) TLSQ_TRY((launch_convert<double, T>(h, Uw + (size_t)nq * M, U + (size_t)nq * M, M * nz)));
    }
    TLSQ_TRY(first_order(d));
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    return TLSQ_OK;
}

// ------------------------------------------------------------------------------------------------
// the ALM loop on device-resident, contiguous (ld = M) panels D, A, E of element type T (fp64 or fp32).
// The small N x N work (Gram matrices, eigenvectors, singular values) is always fp64.
// Vt_host (d x N, ld ldVt, fp64) / S_host (d, fp64) / U_dev (M x d, ld M, T) optional.
// ------------------------------------------------------------------------------------------------
template <typename T>
int rpca_core(Handle* h, const T* D, int64_t M, int64_t N, const ResolvedOpts& ro,
                     const tlsq_rpca_opts* opts, T* A, T* E, T* U_dev, double* S_host,
                     double* Vt_host, int64_t ldVt, int64_t* sv_out, tlsq_rpca_info* info) {
    const int64_t n = M * N;
    const bool timing = info != nullptr;
    host_mark("core");
    void *Yv = nullptr, *Zv = nullptr, *Rv = nullptr;
    {
        // The panels first, and on row shards an agreement on the outcome: a rank that runs out of memory here (the one with
        // the remainder row, a GPU somebody else is using) must not leave the others waiting in the first collective.
        const int st_alloc = [&]() -> int {
            TLSQ_TRY(ws_get(h, WS_Y, (size_t)n * sizeof(T), &Yv));
            TLSQ_TRY(ws_get(h, WS_Z, (size_t)n * sizeof(T), &Zv));
            TLSQ_TRY(ws_get(h, WS_R, (size_t)n * sizeof(T), &Rv));
            return TLSQ_OK;
        }();
        if (h->comm) {
            double bad = st_alloc < 0 ? 1.0 : 0.0;
            TLSQ_TRY(comm_allreduce_host_scalar(h, &bad, ncclMax));
            if (bad != 0.0 && st_alloc >= 0)
                return set_err(h, TLSQ_ERR_OOM, "rpca: another rank of the group could not allocate its panels");
        }
        if (st_alloc < 0) return st_alloc;
    }
    T *Y = (T*)Yv, *R = (T*)Rv;
    const bool no_fuse = dev_is(DEV_NO_FUSED_SWEEP, '1');
    const bool no_zsweep = dev_is(DEV_NO_ZSWEEP, '1');
    const bool no_first = dev_is(DEV_NO_FIRST_SHRINK, '1');
    // The E-free loop (sweeps.hip, k_zsweep): E is not kept while the loop runs - Z is updated in place, Y is double-buffered
    // (the caller's E panel is the second buffer) and the factors of the previous A are kept, from which the returned E is
    // formed once after the loop.  Every plain call runs this way; the `hankel` flag (A is modified after the rebuild) and
    // the svd / opnorm hooks keep the classic sweeps with their E and Z double buffers.
    // (the GPU form of the randomized hook only changes where V and the singular values come from: it runs E-free as well)
    const bool zmode = !no_zsweep && !no_fuse && !no_first && !ro.hankel && ro.iters >= 1 &&
                       !(opts && ((opts->svd_mode != TLSQ_SVD_FULL && opts->svd_mode != TLSQ_SVD_RANDOMIZED) ||
                                  opts->opnorm_mode != TLSQ_OPNORM_EXACT));
    // ResolvedOpts::factors_out: the caller may pass A == nullptr; the panel is allocated the first time something needs it
    auto need_A = [&]() -> int {
        if (!A) {
            void* p;
            TLSQ_TRY(ws_get(h, WS_A, (size_t)n * sizeof(T), &p));
            A = (T*)p;
        }
        return TLSQ_OK;
    };
    h->out_factors = false;
    h->absmax_panel = nullptr;   // (no panel of this call has been written yet)
    h->kern_gram_h3 = h->kern_zx_h = h->kern_zty_h = h->kern_zsweep_wide = h->kern_fused_zgram = 0;
    if (!(zmode && ro.factors_out)) TLSQ_TRY(need_A());
    // classic loop: E and Z are double-buffered: the fused update(k)+shrink(k+1) sweep writes E_{k+1}, Z_{k+1} while E_k, Z_k
    // must survive in case iteration k is the last one
    void *E2v = nullptr, *Z2v = nullptr;
    // Speculative E-free loop (panels up to 2 GB): the count certificate of iteration k (deflation + ||S^2||_F, ~22 us of small
    // kernels) runs on the handle's second stream BESIDE the factor product, the sweep and the next Gram, and its verdict is
    // read when those are queued.  For that a sweep must not destroy anything iteration k still needs should the verdict be
    // "not certified": Z is double-buffered like Y (the sweep writes Z_{k+1} to the other buffer) and so is the Gram matrix
    // (the Gram of Z_{k+1} goes to the other slot); a failed certificate throws the queued work away and serves iteration k
    // again through the retries / the TSQR route (FAIL_CERT_AT=k injects one for the tests).
    const bool spec = zmode && !dev_is(DEV_NO_CERT_ASYNC, '1') && !dev_is(DEV_NO_CERT_OVERLAP, '1') && N <= kFullEigMaxN &&
                      (size_t)n * sizeof(T) <= ((size_t)1 << 31) && h->mailbox && h->mailbox_bytes >= 32768 &&
                      !dev_is(DEV_NO_MAILBOX, '1') &&
                      // (lowrankfilter on an implicit Hankel panel promises four resident panels: a fifth only while it is small)
                      !(ro.hankel_lazy && (size_t)n * sizeof(T) > ((size_t)1 << 28));
    if (!zmode) {
        TLSQ_TRY(ws_get(h, WS_E2, (size_t)n * sizeof(T), &E2v));
        TLSQ_TRY(ws_get(h, WS_Z2, (size_t)n * sizeof(T), &Z2v));
    } else if (spec) {
        TLSQ_TRY(ws_get(h, WS_Z2, (size_t)n * sizeof(T), &Z2v));
        TLSQ_TRY(second_stream(h));
    }
    T* Ebuf[2] = {E, (T*)E2v};
    T* Zbuf[2] = {(T*)Zv, (zmode && !spec) ? (T*)Zv : (T*)Z2v};   // (E-free loop without speculation: one Z, both entries)
    int zc = 0;                  // E-free loop: Z_k sits in Zbuf[zc], the sweep writes Z_{k+1} to Zbuf[zc ^ 1] (the same panel unless spec)
    const int Gslot[2] = {WS_G, spec ? WS_G3 : WS_G};
    int gcur = 0;                // the Gram matrix of the current Z sits in (or is computed into) workspace slot Gslot[gcur]
    // ... and the factor product of the rebuild is queued behind the Rayleigh-Ritz finish with a device-side selection list
    const bool spec_rebuild = zmode && !dev_is(DEV_NO_SPEC_REBUILD, '1') && !dev_is(DEV_NO_TSMM, '1') && !dev_is(DEV_NO_TSMM_SEL, '1') &&
                              !dev_is(DEV_NO_TSMM_SELV, '1') && !dev_is(DEV_NO_MAILBOX, '1') && (N & 3) == 0 && N <= kFullEigMaxN &&
                              ro.maxrank >= 32;
    int64_t n_spec_hits = 0;
    const int64_t fail_cert_at = [] { const char* e = dev_get(DEV_FAIL_CERT_AT); return (int64_t)(e ? atoll(e) : 0); }();
    T* Ybuf[2] = {Y, E};         // E-free loop: Y_k sits in Ybuf[ycur], the sweep writes Y_{k+1} to the other one
    int ycur = 0;
    const double *Tm_prev = nullptr, *Vs_prev = nullptr;   // factors of A_{k-1} (E-free loop)
    int64_t r_prev = 0;
    bool z_swept = false;        // the last iteration ended with an E-free sweep (Z holds Z_{k+1}, E_k is not formed yet)
    double mu_iter = 0.0;        // mu of the last iteration that ran
    int cur = 0;                 // index of the buffers holding E_k, Z_k
    bool have_next = false;      // E_k, Z_k already produced by the previous iteration's fused sweep
    const bool no_fuse_rebuild = dev_is(DEV_NO_FUSED_REBUILD, '1');
    const bool no_cert_overlap = dev_is(DEV_NO_CERT_OVERLAP, '1');
    const double *Tm_last = nullptr, *Vs_last = nullptr;   // factors of the last A (see fuse_rebuild below)
    int64_t r_last = 0;
    bool a_pending = false;                                // the last A exists only as Tm_last * Vs_last'
    // lower bound of the previous iteration's cost (Frobenius / largest entry).  Before the first one: "far above tol" - the first
    // sweep of a solve does not store a residual panel nobody is likely to read (a wrong guess recomputes it)
    double prev_lower = std::numeric_limits<double>::infinity();
    int64_t n_rskip = 0;
    bool sumsq_ready = false;   // the two accumulator sets of the Frobenius bound have been cleared
    // (round 6: 3 where the max-entry bound is in use - `prev_lower` is then within a few per cent of the cost itself, the test
    //  is only evaluated once that bound is below tol, and the cost shrinks by 0.45-0.75 per iteration there: a store is wanted
    //  once the bound is within 1 / 0.33 of tol.  8 stored R in the last five iterations of BASELINE config 2, 20 us each, to
    //  avoid one 60 us recomputation.  6 where the Frobenius bound is the larger of the two (Hankel residuals have no isolated
    //  entries): the cost is then evaluated from 2 tol on; 3 there cost config 3 a recomputation of R from the Y panels - 60 GB,
    //  11 ms)
    bool prev_lower_from_max = false;   // the previous iteration's bound was the largest entry of R, not the Frobenius one
    const double rskip_env = [] { const char* e = dev_get(DEV_RSKIP_MARGIN); return e ? atof(e) : -1.0; }();
    int64_t sweeps = 0;
    // arbitrary hooks of the host language (src/robustPCA.jl:168-169): the panel visits the host and the caller's
    // own function runs there, on the calling thread ("the CPU path with the user's closure")
    const bool cb_svd = opts && opts->svd_mode == TLSQ_SVD_CALLBACK && opts->svd_cb;
    const bool cb_opnorm = opts && opts->opnorm_mode == TLSQ_OPNORM_CALLBACK && opts->opnorm_cb;
    if (opts && ((opts->svd_mode == TLSQ_SVD_CALLBACK && !opts->svd_cb) ||
                 (opts->opnorm_mode == TLSQ_OPNORM_CALLBACK && !opts->opnorm_cb)))
        return set_err(h, TLSQ_ERR_ARG, "rpca: callback mode without a callback");
    // Callbacks on ROW SHARDS (round 6; the reference takes hooks everywhere, :168-169): a closure of the host language needs the
    // whole panel in one place, so every use gathers the shards (one all-gather of row-padded blocks), rank 0 assembles the
    // M_global x N panel on its host and calls the hook there - on the calling thread of a group handle, in rank 0's process
    // with one handle per GPU (the other ranks' callback pointers are never called) - and what comes back travels to the other
    // ranks as sums with zeros.  Correct and slow by construction (the panel crosses PCIe twice per use), like the single-GPU form.
    const bool cb_shards = (cb_svd || cb_opnorm) && h->comm != nullptr;
    std::vector<int64_t> cb_rows;       // rows of every rank's shard
    int64_t cb_maxM = M;                // largest shard
    const int64_t Mg = ro.m_global;
    if (cb_shards) {
        void *s1, *g1;
        TLSQ_TRY(ws_get(h, WS_AUX2, (size_t)std::max(h->nranks, 8) * 8, &g1));
        TLSQ_TRY(ws_get(h, WS_AUX3, 64, &s1));
        const double mine = (double)M;
        std::vector<double> all((size_t)h->nranks);
        TLSQ_HIP(h, hipMemcpyAsync(s1, &mine, 8, hipMemcpyHostToDevice, h->stream));
        TLSQ_TRY(comm_allgather(h, (const double*)s1, (double*)g1, 1));
        TLSQ_HIP(h, hipMemcpyAsync(all.data(), g1, (size_t)h->nranks * 8, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        int64_t tot = 0;
        for (int q = 0; q < h->nranks; ++q) {
            cb_rows.push_back((int64_t)all[(size_t)q]);
            cb_maxM = std::max(cb_maxM, cb_rows.back());
            tot += cb_rows.back();
        }
        if (tot != Mg)
            return set_err(h, TLSQ_ERR_ARG, "rpca: the row blocks of the ranks add up to %lld rows, m_global says %lld", (long long)tot,
                           (long long)Mg);
    }
    std::vector<T> cbZ, cbU, cbS, cbVt;
    // the whole panel on rank 0's host (cbZ: Mg x N, ld Mg); the other ranks only take part in the collective
    auto gather_panel = [&](const T* P) -> int {
        const size_t chunk_bytes = ((size_t)cb_maxM * N * sizeof(T) + 7) / 8 * 8, cnt = chunk_bytes / 8;
        void *sb, *rb;
        TLSQ_TRY(ws_get(h, WS_CBS, chunk_bytes, &sb));
        TLSQ_TRY(ws_get(h, WS_CBR, chunk_bytes * (size_t)h->nranks, &rb));
        TLSQ_HIP(h, hipMemsetAsync(sb, 0, chunk_bytes, h->stream));
        TLSQ_TRY(copy2d(h, sb, cb_maxM, P, M, M, N, sizeof(T), hipMemcpyDeviceToDevice));
        TLSQ_TRY(comm_allgather(h, (const double*)sb, (double*)rb, cnt));
        if (h->rank == 0) {
            std::vector<T> tmp((size_t)cb_maxM * N);
            cbZ.resize((size_t)Mg * N);
            int64_t r0 = 0;
            for (int q = 0; q < h->nranks; ++q) {
                TLSQ_HIP(h, hipMemcpyAsync(tmp.data(), (const char*)rb + (size_t)q * chunk_bytes, (size_t)cb_maxM * N * sizeof(T),
                                           hipMemcpyDeviceToHost, h->stream));
                TLSQ_HIP(h, hipStreamSynchronize(h->stream));
                for (int64_t j = 0; j < N; ++j)
                    memcpy(cbZ.data() + (size_t)(r0 + j * Mg), tmp.data() + (size_t)(j * cb_maxM), (size_t)cb_rows[(size_t)q] * sizeof(T));
                r0 += cb_rows[(size_t)q];
            }
        } else {
            TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        }
        return TLSQ_OK;
    };
    // `count` doubles that only rank 0 holds (host) -> every rank's host copy (a sum with zeros)
    auto bcast_from_rank0 = [&](double* v, size_t count) -> int {
        if (count == 0) return TLSQ_OK;
        void* b;
        TLSQ_TRY(ws_get(h, WS_CBS, count * 8, &b));
        if (h->rank == 0) TLSQ_HIP(h, hipMemcpyAsync(b, v, count * 8, hipMemcpyHostToDevice, h->stream));
        else TLSQ_HIP(h, hipMemsetAsync(b, 0, count * 8, h->stream));
        TLSQ_TRY(comm_allreduce(h, (double*)b, count, ncclSum));
        TLSQ_HIP(h, hipMemcpyAsync(v, b, count * 8, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        return TLSQ_OK;
    };
    auto opnorm_callback = [&](const T* P, double* out) -> int {
        if (cb_shards) {
            TLSQ_TRY(gather_panel(P));
            double v[2] = {0.0, 0.0};   // the value, "not usable"
            if (h->rank == 0) {
                v[0] = opts->opnorm_cb(cbZ.data(), Mg, N, Mg, opts->user);
                if (!std::isfinite(v[0]) || v[0] < 0.0) {
                    v[1] = 1.0;
                    v[0] = 0.0;
                }
            }
            TLSQ_TRY(bcast_from_rank0(v, 2));
            if (v[1] != 0.0) return set_err(h, TLSQ_ERR_ARG, "rpca: the opnorm callback returned a negative or non-finite value");
            *out = v[0];
            return TLSQ_OK;
        }
        cbZ.resize((size_t)n);
        TLSQ_HIP(h, hipMemcpyAsync(cbZ.data(), P, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        *out = opts->opnorm_cb(cbZ.data(), M, N, M, opts->user);
        if (!std::isfinite(*out) || *out < 0.0) return set_err(h, TLSQ_ERR_ARG, "rpca: the opnorm callback returned %g", *out);
        return TLSQ_OK;
    };
    const bool hook_svd = (opts && opts->svd_mode == TLSQ_SVD_RANDOMIZED) || cb_svd;   // `svd = rsvd`-style hook
    const bool hook_opnorm = (opts && opts->opnorm_mode == TLSQ_OPNORM_POWER) || cb_opnorm;   // `opnorm = x->rnorm(x,mvps)`
    const int mvps = opts && opts->opnorm_mvps > 0 ? opts->opnorm_mvps : 10;
    const uint64_t seed = opts ? opts->seed : 0;

    // The data panel.  Normally the caller's; for an implicit Hankel source (ro.hankel_lazy) a transient copy in Z's
    // second buffer serves the set-up and the first shrink - the loop writes that buffer for the first time in the sweep
    // that follows, which reads y itself - and the kernels without an implicit form (non-fused sweeps: ranks above 32,
    // hooks) get a real panel built on demand.
    const T* Dm = D;
    bool d_transient = false;
    const double lam_f = ro.lambda;
    auto build_hankel = [&](T* dst) -> int {
        const HankelGeom& hg = ro.hankel_geom;
        const int64_t Kh = ro.hankel_K, Lw = N / hg.Dch, Nw = (Kh - 1) * hg.lag + Lw;   // samples per channel behind these rows
        if (Kh != M) TLSQ_HIP(h, hipMemsetAsync(dst, 0, (size_t)n * sizeof(T), h->stream));
        return launch_hankel<T>(h, (const T*)ro.hankel_y, Nw, hg.Dch, hg.Dch == 1 ? Nw : hg.ldx, Lw, hg.lag, dst, M);
    };
    auto panel_D = [&](const T** out) -> int {
        if (!Dm || d_transient) {
            void* p;
            TLSQ_TRY(ws_get(h, WS_D, (size_t)n * sizeof(T), &p));
            TLSQ_TRY(build_hankel((T*)p));
            Dm = (const T*)p;
            d_transient = false;
        }
        *out = Dm;
        return TLSQ_OK;
    };
    // E-free loop: E_k = soft_th(D - A_{k-1} + Y_k / mu_k, lambda / mu_k) (:188-191) into the caller's panel, A_{k-1} from the
    // kept factors (above 32 columns through memory: the residual panel is free whenever this runs)
    auto form_final_e = [&](const T* Yk, double mu_k) -> int {
        const T* Aprev = nullptr;
        const T* hy = (const T*)ro.hankel_y;
        if (r_prev > 32) {
            TLSQ_TRY(rebuild_from_factors<T>(h, Tm_prev, Vs_prev, M, N, r_prev, R, M));
            if (ro.nonnegA) TLSQ_TRY(launch_clamp_nonneg<T>(h, R, n));
            Aprev = R;
            hy = nullptr;
        }
        const T* Dp = nullptr;
        if (!hy) {
            TLSQ_TRY(panel_D(&D));
            Dp = D;
        }
        return launch_final_e<T>(h, Dp, Tm_prev, Vs_prev, Aprev, Yk, Ebuf[0], M, N, r_prev, (T)(1.0 / mu_k), (T)(lam_f / mu_k),
                                 ro.nonnegA ? 1 : 0, ro.nonnegE ? 1 : 0, hy, ro.hankel_K, ro.hankel_geom);
    };
    if (!D) {
        if (!ro.hankel_lazy || !ro.hankel_y) return set_err(h, TLSQ_ERR_ARG, "rpca: no data panel");
        // (E-free loop: R is written for the first time by the first sweep, which reads y itself)
        T* spare = zmode ? R : Zbuf[1];
        TLSQ_TRY(build_hankel(spare));
        Dm = spare;
        d_transient = true;
    }
    D = Dm;   // (set-up below; every later use goes through panel_D or the implicit kernels)

    // ---- setup, src/robustPCA.jl:171-184 ----
    if (!zmode) {   // (the E-free loop writes both panels in full before anything reads them: A by the rebuild, E as Y's second buffer)
        TLSQ_HIP(h, hipMemsetAsync(A, 0, (size_t)n * sizeof(T), h->stream));  // :174
        TLSQ_HIP(h, hipMemsetAsync(E, 0, (size_t)n * sizeof(T), h->stream));
    }
    double norm2 = 0.0;
    // Very wide problems: G = Z'Z is never formed (see GramOp).  The Gram costs M N^2 flops (lower triangle) per
    // iteration, the ~8 products plus the Lanczos vectors of the implicit form ~130 M N p at the efficiency of the
    // skinny kernels: measured break-even near N = 128 p (65536 x 4096, p = 80: 41 ms explicit, 58 ms implicit per
    // iteration), so the switch sits at N >= 8192.  TLSQ_IMPLICIT_GRAM=0/1 overrides it (large mode only).
    const int force_implicit = [] { const char* e = dev_get(DEV_IMPLICIT_GRAM); return e ? atoi(e) : -1; }();
    // (round 5: fp32 panels the fp16-split Gram kernel takes - gram16.hip, 2x the fp32 MFMA's rate - keep the Gram matrix at
    //  N = 8192 as well: 16384 x 8192 rank 40, 176 ms per solve against 305 in operator form)
    bool h3_shape = Prec<T>::f32 && (N % 128) == 0 && (M % 64) == 0 && M >= 4096 && !dev_is(DEV_GRAM_H3, '0');
    // Row shards: decisions that change WHICH collectives an iteration enters must be the same on every rank, and these two
    // depend on the rank-local row count: h3_shape moves the switch to the operator form at N = 8192 (an N x N Gram
    // all-reduce against N x p ones), and the residual-Gram mode of the fused sweep kernel (fused_gr below) all-reduces R'R in
    // the slot where the other ranks all-reduce Z'Z.  One min-all-reduce of the flags: taken only if every rank can.
    bool group_fused_shape = true;
    if (h->comm) {
        double fl[2] = {h3_shape ? 1.0 : 0.0, 1.0};
        if constexpr (std::is_same<T, double>::value)
            fl[1] = fused_zgram_shape_ok(M, N, ro.hankel_y != nullptr) ? 1.0 : 0.0;
        TLSQ_TRY(comm_allreduce_host_vec(h, fl, 2, ncclMin));
        h3_shape = fl[0] != 0.0;
        group_fused_shape = fl[1] != 0.0;
    }
    const bool implicit_gram = N > kFullEigMaxN && (force_implicit >= 0 ? force_implicit == 1 : (h3_shape ? N > 8192 : N >= 8192));
    // The randomized hook in large mode (BASELINE config 5: `svd = rsvd`, src/robustPCA.jl:195-197, test/runtests.jl:388-398) is a
    // sketch: from iteration 2 on nothing but products with the panel - Y = Z'(Z Omega), orthonormalisation, the power passes, the
    // (sv + 10)-column Rayleigh quotient - and no N x N Gram matrix of Z (2 M N^2 flops: 12 ms of a 24 ms iteration at
    // 65536 x 4096).  opnorm(residual), in the few iterations whose cost bound does not settle the test, still goes through the
    // Gram matrix of R (Lanczos on the operator R'R needs ~100 steps of two panel passes: 46 ms).  Iteration 1 is the reference's
    // full svd: the certified subspace solver on the Gram matrix.  Default for fp32 panels, whose operator products run on the
    // fp32 MFMA (op_gram_f32); fp64 panels keep the Gram matrix below N = 8192 (their skinny products are not faster than it)
    // unless HOOK_SKETCH=1.
    const bool hook_sketch = opts && opts->svd_mode == TLSQ_SVD_RANDOMIZED && N > kFullEigMaxN &&
                             (Prec<T>::f32 ? !dev_is(DEV_HOOK_SKETCH, '0') : dev_is(DEV_HOOK_SKETCH, '1'));
    auto panel_op = [&](const T* P) {
        GramOp o;
        o.Z = P;
        o.z_f32 = Prec<T>::f32;
        o.M = M;
        o.ldZ = M;
        return o;
    };
    double maxabs = 0.0;
    // (plain calls on one GPU: the max-abs pass runs on the second stream beside the Gram matrix of the set-up norm - it is
    //  memory-bound, the Gram is not - and its result is read together with the norm's: one host round trip less, ~60 us at C2)
    // (large fp32 panels: the fp16-split Gram matrix of the set-up needs max |D| for its scale BEFORE it runs - the pass stays in
    //  line there and its result doubles as that scale, see below)
    const bool maxabs_async = !(ro.hankel_lazy && ro.hankel_y) && !cb_opnorm && !hook_opnorm && !implicit_gram && !h->comm &&
                              !(Prec<T>::f32 && N > kFullEigMaxN) && !dev_is(DEV_NO_MAXABS_ASYNC, '1');
    if (ro.hankel_lazy && ro.hankel_y) {   // every sample of the window appears in its Hankel matrix (lag <= L): max |H| = max |y|
        const HankelGeom& hg = ro.hankel_geom;
        const int64_t Nw = (ro.hankel_K - 1) * hg.lag + N / hg.Dch;
        for (int d = 0; d < hg.Dch; ++d) {
            double md = 0.0;
            TLSQ_TRY(launch_maxabs<T>(h, (const T*)ro.hankel_y + (size_t)d * hg.ldx, Nw, &md));
            maxabs = std::max(maxabs, md);
        }
    }
    else if (maxabs_async) {
        host_mark("first launch");
        TLSQ_TRY(launch_maxabs_begin<T>(h, D, n));                 // :178 norm(Y, Inf), beside the Gram matrix of :177 (read below)
    }
    else
        TLSQ_TRY(launch_maxabs<T>(h, D, n, &maxabs));              // :178 norm(Y, Inf)
    auto finish_maxabs = [&]() -> int {
        if (maxabs_async) TLSQ_TRY(launch_maxabs_end(h, &maxabs));
        TLSQ_TRY(comm_allreduce_host_scalar(h, &maxabs, ncclMax));
        // (the reference's svd! / opnorm go through LAPACK's chkfinite and throw ArgumentError("matrix contains Infs or NaNs")
        //  at :177 before anything else happens: the same input is an error here, from the max-abs pass the set-up needs anyway)
        if (!std::isfinite(maxabs)) return set_err(h, TLSQ_ERR_NONFINITE, "matrix contains Infs or NaNs");
        return TLSQ_OK;
    };
    if (!maxabs_async) TLSQ_TRY(finish_maxabs());
    if constexpr (Prec<T>::f32) {
        // max |D| is also the scale of the fp16 split of D (gram16.hip): left where a sweep would leave it (Handle::absmax_panel),
        // so that the set-up's Gram matrix does not make a max pass of its own (0.3 ms at 65536 x 4096).  An upper bound serves
        // (on row shards the all-reduced maximum).
        if (!maxabs_async && N > kFullEigMaxN && !(ro.hankel_lazy && ro.hankel_y) && maxabs > 0.0 && maxabs < 3.0e38) {
            void* sc;
            TLSQ_TRY(ws_get(h, WS_H16S, 64, &sc));
            const float mf = std::nextafter((float)maxabs, std::numeric_limits<float>::infinity());   // (rounded up: still a bound)
            unsigned int bits;
            memcpy(&bits, &mf, 4);
            TLSQ_TRY(upload_async(h, reinterpret_cast<char*>(sc) + 40, &bits, 4));
            h->absmax_panel = (const void*)D;
        }
    }
    if (cb_opnorm) TLSQ_TRY(opnorm_callback(D, &norm2));                             // :177 through the caller's hook
    else if (hook_opnorm) TLSQ_TRY(opnorm_power<T>(h, D, M, N, M, mvps, seed, &norm2));   // :177 through the hook
    else if (implicit_gram) TLSQ_TRY(sigma_max_of_op(h, panel_op(D), N, 1e-13, &norm2));
    else {
        // (Infs / NaNs in D come out of the Gram matrix as NaNs: the norm's own failure is not reported before the max-abs pass
        //  has had its say)
        const int st_n = opnorm_gram<T>(h, D, M, N, M, &norm2, &sweeps, 1e-11);                   // :177 opnorm(Y), Y = copy(D)
        if (maxabs_async) TLSQ_TRY(finish_maxabs());
        if (st_n < 0) return st_n;
    }
    host_mark("norm known");
    const double lam = ro.lambda;
    const double norminf = maxabs / lam;
    const double dual_norm = std::max(norm2, norminf);             // :179
    const double d_norm = norm2;                                   // :180
    // :181 Y = D / dual_norm is folded into the first shrink (launch_first_shrink, one pass over D instead of three)
    bool y_pending = true;
    if (no_first) {
        TLSQ_TRY(launch_div_scalar<T>(h, D, Y, n, (T)dual_norm));      // :181
        y_pending = false;
    }
    double mu = 1.25 / norm2;                                      // :182
    const double mubar = mu * 1.0e7;                               // :183
    int64_t sv = 10, svp = 10;                                     // :184
    if (info) {
        info->d_norm = d_norm;
        info->iters_done = 0;
        info->converged = 0;
    }
    SmallSvd s;
    double* V = nullptr;
    // warm-started subspace iteration (falls back to the full Jacobi solver whenever it cannot certify the
    // count).  TLSQ_FULL_EIG=1 forces the full solver every iteration.
    SubspaceState sub;
    int64_t hook_cols = 0;   // columns of the block buffer (WS_SX) holding the last decomposition's sorted Ritz vectors: the hook's warm start
    if (N > kFullEigMaxN && !dev_is(DEV_COLD_GROW, '0')) sub.cold_p = 34;   // (large mode: see the block growth below)
    // (the hook's Rayleigh-Ritz product Z Q and eigenvector matrix of this iteration, when they are still on the device: the
    //  factor of the rebuild is taken from them - SubspaceState::hook_zq)
    const float* hook_zq = nullptr;
    int64_t hook_zq_p = 0;
    const double* hook_S = nullptr;
    std::vector<int32_t> hook_order;
    const int64_t pmax = subspace_max_block(N);
    const char* force_full = dev_get(DEV_FULL_EIG);
    // Large mode (N > 2048): the full Jacobi solvers do not apply (their column blocks live in LDS); every SVD step
    // has to be served by the certified subspace iteration, whose block is enlarged on demand.  Ranks beyond
    // the largest block (subspace_max_block) are reported as TLSQ_ERR_UNSUPPORTED, and the two-level refinement
    // of very late iterations is not available.
    const bool large = N > kFullEigMaxN;
    const bool use_subspace = large || (!hook_svd && !(force_full && force_full[0] == '1') && pmax >= 11 && N >= 24);
    bool v_is_full = false;
    double sigma_top_prev = 0.0;
    int64_t n_rroute = 0;   // iterations whose SVD step was served by the TSQR route
    int64_t n_gram_dense = 0;   // ... by the dense decomposition of the Gram matrix (very tall panels)
    // the tail of the spectrum is a noise bulk whose top sits right below 1/mu (noisy data: every iteration): the
    // subspace solver cannot certify a count there, so it is not even tried until a dense result shows a gap again
    bool bulk_tail = false;
    bool power_vec_valid = false;   // the power-iteration vector of the cost evaluation has been started in this call
    const bool no_power_lb = dev_is(DEV_NO_POWER_LB, '1');
    const bool no_gram_dense = dev_is(DEV_NO_GRAM_DENSE, '1');
    // HBM traffic the panel-sized kernels of this call have to move (algorithmic bytes of what was launched: panel
    // passes x M x N x sizeof(T)); reported in tlsq_rpca_info
    const double panel_bytes = (double)n * sizeof(T);
    double hbm_sweeps = 0.0, hbm_other = 0.0;
    // Relative uncertainty of an eigenvalue of the computed Gram matrix (rounding of Z'Z accumulated over M rows, the
    // small solvers' own eps N lambda_max, Ritz residuals <= 2e-13 lambda_max per pair): see SubspaceState::noise_rel.
    // Large mode has no other solver, so no window there.
    const double noise_rel = large_mode_noise(N, ro.m_global);
    h->warm_n = 0;   // nothing from an earlier call is reused
    double cost = std::numeric_limits<double>::quiet_NaN();
    bool converged = false;
    void* meanws = nullptr;
    // soft_hankel! (:214-216, :234-236).  On row shards the anti-diagonals run through several ranks' blocks (contiguous
    // row blocks in rank order): every rank adds its block's sums and counts into full-length arrays at its row offset,
    // one all-reduce, then the shrink towards the global means.
    int64_t hk_row0 = 0;
    const int64_t hk_len = ro.m_global + N - 1;
    if (ro.hankel) TLSQ_TRY(ws_get(h, WS_AUX1, (size_t)(hk_len + 1) * sizeof(T), &meanws));
    if (ro.hankel && h->comm) {
        void* sc;
        TLSQ_TRY(ws_get(h, WS_HKSUM, (size_t)std::max<int64_t>(2 * hk_len, h->nranks) * 8, &sc));
        double mine = (double)M;
        std::vector<double> all((size_t)h->nranks);
        TLSQ_HIP(h, hipMemcpyAsync(sc, &mine, 8, hipMemcpyHostToDevice, h->stream));
        void* gat;
        TLSQ_TRY(ws_get(h, WS_AUX2, (size_t)h->nranks * 8, &gat));
        TLSQ_TRY(comm_allgather(h, (const double*)sc, (double*)gat, 1));
        TLSQ_HIP(h, hipMemcpyAsync(all.data(), gat, (size_t)h->nranks * 8, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        int64_t tot = 0;
        for (int q = 0; q < h->nranks; ++q) {
            if (q < h->rank) hk_row0 += (int64_t)all[(size_t)q];
            tot += (int64_t)all[(size_t)q];
        }
        if (tot != ro.m_global)
            return set_err(h, TLSQ_ERR_ARG, "rpca: the row blocks of the ranks add up to %lld rows, m_global says %lld",
                           (long long)tot, (long long)ro.m_global);
    }
    auto soft_hankel = [&](T* P, T eps) -> int {
        if (!h->comm) return launch_soft_hankel<T>(h, P, M, N, M, eps, (T*)meanws);
        double* sum = (double*)h->ws[WS_HKSUM].p;
        double* cnt = sum + hk_len;
        TLSQ_HIP(h, hipMemsetAsync(sum, 0, (size_t)2 * hk_len * 8, h->stream));
        TLSQ_TRY(launch_unhankel_partial<T>(h, P, M, N, 1, M, 1, M + N - 1, hk_row0, sum, cnt, hk_len));
        TLSQ_TRY(comm_allreduce(h, sum, (size_t)2 * hk_len, ncclSum));
        TLSQ_TRY(launch_unhankel_finish<T>(h, sum, cnt, hk_len, (T*)meanws));
        return launch_soft_toward<T>(h, P, M, N, M, (const T*)meanws + hk_row0, eps);
    };

    // (all phase marks only on request: tlsq_rpca_opts.phase_timing or TLSQ_PHASE_TIMING=1)
    const bool env_phases = dev_is(DEV_PHASE_TIMING, '1');
    PhaseTimer pt(h, timing, env_phases || (opts && opts->phase_timing != 0));
    // light mode: which sweeps are bracketed by events (tlsq_rpca_info::sweeps_timed)
    const int64_t tstride = [] { const char* e = dev_get(DEV_SWEEP_TIMING_STRIDE); const int v = e ? atoi(e) : 0; return v >= 1 ? v : 4; }();
    int64_t n_timed = 0;
    double hbm_timed = 0.0;
    double zero_sink = 0.0;
    // phase windows between consecutive marks: shrink | gram | eig | rebuild | sweep | read-back of the Frobenius
    // bound (booked under the cost evaluation) | next iteration's Gram queued behind the sweep (booked under gram) |
    // cost evaluation
    double* acc[8] = {info ? &info->ms_shrink : &zero_sink, info ? &info->ms_gram : &zero_sink,
                      info ? &info->ms_eig : &zero_sink,    info ? &info->ms_rebuild : &zero_sink,
                      info ? &info->ms_update : &zero_sink, info ? &info->ms_opnorm : &zero_sink,
                      info ? &info->ms_gram : &zero_sink,   info ? &info->ms_opnorm : &zero_sink};
    // ---- count certificate on the deflated panel (iterations whose threshold lies below the noise level of G = Z'Z) ----------
    // S = the Ritz values of the block that clear the threshold by G's uncertainty; Z2 = Z - (Z X_S) X_S' is formed in the
    // residual panel (free at this point of the iteration) and sigma_max(Z2) < 1/mu is certified through the Gram of Z2,
    // whose rounding is relative to ||Z2||^2 ~ (1/mu)^2 instead of sigma_max(Z)^2.  Courant-Fischer: sigma_{|S|+1}(Z) <=
    // sigma_max(Z (I - X_S X_S')) < 1/mu, and Cauchy interlacing puts |S| singular values above 1/mu - the count is |S|.
    // The Ritz values outside S (pad columns: noise of G) are zeroed so that nobody counts them.
    const bool no_defl = dev_is(DEV_NO_DEFLATED_CERT, '1');
    const bool defl_possible = !no_defl && use_subspace && !large && !hook_svd && !implicit_gram && !Prec<T>::f32;
    auto deflated_certificate = [&](const T* Zp, const double* X, SmallSvd& sv_, double inv_mu_, bool* pass) -> int {
        *pass = false;
        const double tau2 = inv_mu_ * inv_mu_;
        const double stop = sv_.ncols > 0 ? sv_.sigma[sv_.order[0]] : 0.0;
        const double dl = noise_rel * stop * stop;
        std::vector<int32_t> sel;
        for (int64_t i = 0; i < sv_.ncols; ++i) {
            const double sg = sv_.sigma[sv_.order[i]];
            if (sg * sg >= tau2 + 2.0 * dl) sel.push_back(sv_.order[i]);
        }
        const int64_t rS = (int64_t)sel.size();
        if (rS == 0 || rS > 96 || rS + 2 > sv_.ncols) return TLSQ_OK;
        // rounding of Z2 itself: ~eps sqrt(r) sigma_max(Z) per direction, to be small against 1/mu
        const double rel = 64.0 * 2.220446049250313e-16 * std::sqrt((double)rS + 1.0) * stop / inv_mu_;
        if (!(rel < 1e-3)) return TLSQ_OK;
        std::vector<double> ones((size_t)rS, 1.0);
        const double *TmS = nullptr, *VsS = nullptr;
        TLSQ_TRY(rebuild_factors<T>(h, Zp, M, N, M, X, sel, ones, &TmS, &VsS, 3));  // T_S = Z X_S
        TLSQ_TRY(rebuild_from_factors<T>(h, TmS, VsS, M, N, rS, R, M));              // R = T_S X_S'
        TLSQ_TRY(launch_diff<T>(h, Zp, R, R, n));                                    // R = Z - T_S X_S' = Z2
        hbm_other += 6.0 * panel_bytes;
        double* G2 = nullptr;
        TLSQ_TRY(gram_allreduce<T>(h, R, M, N, M, &G2));
        void* GD;
        TLSQ_TRY(ws_get(h, WS_GD, (size_t)N * N * 8, &GD));
        TLSQ_TRY(launch_deflate(h, G2, N, nullptr, nullptr, (double*)GD, N, 0, 1.0 / tau2));
        sub.cert_GD = (const double*)GD;
        sub.cert_N = N;
        sub.cert_margin = (1.0 - 2.0 * rel) * (1.0 - 1e-9);
        sub.cert_power = N <= 1024;
        if (!sub.cert_power) return TLSQ_OK;
        TLSQ_TRY(power_cert_begin(h, sub));
        TLSQ_TRY(cert_finish(h, sub, pass));
        const bool dbg = dev_get(DEV_DEBUG) != nullptr;
        if (dbg) fprintf(stderr, "  deflated certificate: |S|=%lld margin=%.3e pass=%d\n", (long long)rS, sub.cert_margin, (int)*pass);
        if (*pass) {
            std::vector<char> keep((size_t)sv_.sigma.size(), 0);
            for (int32_t i : sel) keep[(size_t)i] = 1;
            for (size_t i = 0; i < sv_.sigma.size(); ++i)
                if (!keep[i]) sv_.sigma[i] = 0.0;
        }
        return TLSQ_OK;
    };
    // ---- the matrix-function route: count and A without singular vectors (noisy data) ------------------------------------------
    // Once noise singular values cross 1/mu the rank jumps to ~N/2 and no subspace block or certificate applies: every such
    // iteration used to cost a dense decomposition (TSQR + one-sided Jacobi, 39 ms at N = 512).  The loop only needs the count
    // and A (:198, :205-213), and both are matrix functions of the Gram matrix G2 of the deflated panel Z2 = Z (I - X_S X_S')
    // (S = the dominant Ritz pairs, taken out so that G2 is accurate at the scale of the threshold):
    //     P = (I + sign(G2 mu^2 - I)) / 2,   svp = |S| + trace(P),
    //     A = Z Phi,  Phi = X_S diag(g) X_S' + F,  F = P - (P G2 mu^2 P + I - P)^(-1/2) P   (F = P without nukeA)
    // by Newton-Schulz iterations (matfun.hip: ~100 N x N x N MFMA products, 2-3 ms) and one M x N x N product.  A sign
    // iteration that does not converge (an eigenvalue within ~1e-10 of the threshold) or a trace that is not an integer to
    // 1e-6 leaves the iteration to the TSQR route.  A_k has no factor form afterwards: see last_no_factors.
    const bool no_matfun = dev_is(DEV_NO_MATFUN_ROUTE, '1');
    const bool matfun_possible = !no_matfun && use_subspace && !large && !hook_svd && !implicit_gram && !Prec<T>::f32 &&
                                 N >= 64 && N <= 1024;
    bool last_no_factors = false, prev_no_factors = false;   // A_k / A_{k-1} exist only as panels (E-free loop: how E is formed)
    int64_t mf_rS_prev = -1;   // size of the deflated set of the last matrix-function iteration (-1: none yet)
    int64_t mf_k2_last = 0;    // how many values below the dominant set it counted above the threshold
    int mf_dfl_level = 0;      // 1: the dominant set is taken at 1e5 / mu^2 from now on (see matfun_route)
    auto matfun_route = [&](const T* Zp, double inv_mu_, int64_t* svp_out, double* sigma_top_out, bool* ok) -> int {
        *ok = false;
        const bool dbg = dev_get(DEV_DEBUG) != nullptr;
        const double tau2 = inv_mu_ * inv_mu_;
        // dominant part from the Gram matrix of Z (its Ritz values are far above G's noise): no window, no certificate
        double* Gz = nullptr;
        TLSQ_TRY(gram_allreduce<T>(h, Zp, M, N, M, &Gz));
        hbm_other += panel_bytes;
        dbg_hash(h, "mf.Gz", Gz, (size_t)N * N * 8);
        GramOp gop;
        gop.G = Gz;
        SmallSvd ss;
        double* X = nullptr;
        bool got = false;
        // (only the pairs a factor 1e3 above the threshold are wanted: the solver is asked for those - it counts, and checks the
        //  residuals of, the Ritz values above sqrt(1e3) / mu - on a block cut back to its leading columns)
        // (32 columns when the dominant part seen last time leaves room: CholeskyQR2 and the single-workgroup Jacobi apply)
        const int64_t pcap = (mf_rS_prev >= 0 && mf_rS_prev + 6 <= 32) ? 32 : 48;
        if (sub.valid && sub.p > pcap) {
            sub.p = pcap;
            sub.ntop = std::min<int64_t>(sub.ntop, pcap - 8);
        }
        sub.noise_rel = 0.0;
        sub.skip_certificate = true;
        sub.defer_certificate = false;
        // How far above the threshold a pair has to be to count as dominant: 1e3 / mu^2, and 1e5 / mu^2 once that level stops
        // working (TLSQ_MATFUN_DEFL fixes it).  What stays below goes into B, whose condition number the inverse square root
        // pays for with eps * cond(B) of its accuracy - its residual is checked below.  1e3 alone handed the last iterations
        // of a noisy problem to the TSQR route as soon as the top of the noise bulk had grown past 1e3 / mu^2 - hundreds of
        // pairs that no block holds: 3 of 35 iterations and half of the run time at 20000 x 512 with noise 1e-2 (110 -> 65
        // ms).  1e7 fails the residual test; 1e5 from the start costs clean data a digit (A to 5e-11 instead of 1e-11).
        const double dfl_env = [] { const char* e = dev_get(DEV_MATFUN_DEFL); const double v = e ? atof(e) : 0.0; return v > 1.0 ? v : 0.0; }();
        const double cond_base = [] { const char* e = dev_get(DEV_MATFUN_COND); const double v = e ? atof(e) : 0.0; return v > 1.0 ? v : 1e4; }();
        std::vector<int32_t> sel;
        std::vector<double> gw;
        double stop = 0.0;
        int64_t rS = 0;
        bool have_S = false;
        for (int level = (dfl_env > 0.0 || mf_dfl_level > 0) ? 1 : 0; level < 2 && !have_S; ++level) {
            const double dfl = dfl_env > 0.0 ? dfl_env : (level == 0 ? 1e3 : 1e5);
            sub.skip_certificate = true;
            got = false;
            const int st_sub = svd_subspace(h, gop, N, inv_mu_ * std::sqrt(dfl), sub, &X, ss, &sweeps, &got);
            sub.skip_certificate = false;
            if (st_sub < 0) return st_sub;
            if (!got || !X || ss.ncols <= 0) {
                if (dbg) fprintf(stderr, "  matrix-function route: no dominant block at %.0e / mu^2 (subspace solver: %d)\n", dfl, sub.fail);
                continue;
            }
            stop = ss.sigma[ss.order[0]];
            const double dl = noise_rel * stop * stop;
            // S = the leading Ritz pairs, cut where the spectrum has a gap.  Mixing inside S is harmless (nearly equal weights g);
            // what must be small is the leak between span(S) and the rest: the solver's residual bound is 2e-13 lambda_top in
            // absolute terms, so the angle is <= 2e-13 lambda_top / gap, and the error it puts into A is sigma times that - kept
            // below 1e-10 sigma_top by asking for gap >= 2e-3 sqrt(lambda_top theta) at the cut.  Signal values pass; a noise
            // value that has grown past the level sits in a cluster of its like and is left to G2, where clusters do not matter.
            int64_t cnt = 0;
            while (cnt < ss.ncols && cnt < 32) {
                const double sg = ss.sigma[ss.order[cnt]];
                if (!(sg * sg >= std::max(dfl * tau2, tau2 + 2.0 * dl))) break;
                ++cnt;
            }
            while (cnt > 0) {
                const double sg = ss.sigma[ss.order[cnt - 1]];
                const double sg1 = cnt < ss.ncols ? ss.sigma[ss.order[cnt]] : 0.0;
                if (sg * sg - sg1 * sg1 >= 2e-3 * stop * sg) break;
                --cnt;
            }
            // What stays in G2 has to be (a) small enough for G2's own rounding to sit far below the threshold and (b) within
            // 10 x the level of the threshold: the inverse square root loses eps * cond(B) of its relative accuracy.
            const double next = cnt < ss.ncols ? ss.sigma[ss.order[cnt]] : 0.0;
            const double cond_max = cond_base * (dfl / 1e3);
            if (!(noise_rel * next * next < 1e-4 * tau2) || !(next * next <= cond_max * tau2)) {
                if (dbg) fprintf(stderr, "  matrix-function route: declined at %.0e / mu^2 (next^2 mu^2 = %.2e, noise %.2e)\n", dfl, next * next / tau2, noise_rel * next * next / tau2);
                continue;
            }
            sel.clear();
            gw.clear();
            for (int64_t i = 0; i < cnt; ++i) {
                const double sg = ss.sigma[ss.order[i]];
                sel.push_back(ss.order[i]);
                gw.push_back(ro.nukeA ? (sg - inv_mu_) / sg : 1.0);   // :205-213
            }
            rS = cnt;
            have_S = true;
            if (level == 1) mf_dfl_level = 1;   // (the bulk only grows against the threshold: later iterations start here)
        }
        if (!have_S) return TLSQ_OK;
        const T* Z2 = Zp;
        if (rS > 0) {
            std::vector<double> ones((size_t)rS, 1.0);
            const double *TmS = nullptr, *VsS = nullptr;
            TLSQ_TRY(rebuild_factors<T>(h, Zp, M, N, M, X, sel, ones, &TmS, &VsS, 3));
            TLSQ_TRY(rebuild_from_factors<T>(h, TmS, VsS, M, N, rS, R, M));
            TLSQ_TRY(launch_diff<T>(h, Zp, R, R, n));                                    // R = Z2
            Z2 = R;
            hbm_other += 5.0 * panel_bytes;
        }
        double* G2 = nullptr;
        TLSQ_TRY(gram_allreduce<T>(h, Z2, M, N, M, &G2));
        hbm_other += panel_bytes;
        dbg_hash(h, "mf.Z2", Z2, (size_t)n * sizeof(T));
        dbg_hash(h, "mf.G2", G2, (size_t)N * N * 8);
        void *G2s, *Cm, *Xs, *W1, *W2, *Wz, *Yb;
        TLSQ_TRY(ws_get(h, WS_GD, (size_t)N * N * 8, &G2s));
        TLSQ_TRY(ws_get(h, WS_MF0, (size_t)N * N * 8, &Cm));
        TLSQ_TRY(ws_get(h, WS_MF1, (size_t)N * N * 8, &Xs));
        TLSQ_TRY(ws_get(h, WS_MF2, (size_t)N * N * 8, &Wz));
        TLSQ_TRY(ws_get(h, WS_MF3, (size_t)N * N * 8, &Yb));
        TLSQ_TRY(ws_get(h, WS_CP1, (size_t)N * N * 8, &W1));
        TLSQ_TRY(ws_get(h, WS_CP2, (size_t)N * N * 8, &W2));
        TLSQ_TRY(matfun_axpbi(h, G2, (double*)G2s, N, 1.0 / tau2, 0.0));                  // G2 mu^2
        TLSQ_TRY(matfun_axpbi(h, (const double*)G2s, (double*)Cm, N, 1.0, -1.0));         // C = G2 mu^2 - I
        int it_s = 0, it_r = 0;
        bool conv = false;
        TLSQ_TRY(matfun_sign(h, (const double*)Cm, N, (double*)Xs, (double*)W1, (double*)W2, 70, &it_s, &conv));
        if (!conv) {
            if (dbg) fprintf(stderr, "  matrix-function route: sign iteration did not converge (|S|=%lld)\n", (long long)rS);
            return TLSQ_OK;
        }
        dbg_hash(h, "mf.sign", Xs, (size_t)N * N * 8);
        double* Pm = (double*)Xs;
        TLSQ_TRY(matfun_axpbi(h, (const double*)Xs, Pm, N, 0.5, 0.5));                    // P = (I + sign) / 2
        double tr = 0.0;
        TLSQ_TRY(matfun_trace_norm(h, Pm, N, &tr, nullptr));
        const double k2d = std::round(tr);
        if (!std::isfinite(tr) || std::fabs(tr - k2d) > 1e-6 || k2d < 0.0 || k2d > (double)(N - rS)) {
            if (dbg) fprintf(stderr, "  matrix-function route: trace(P) = %.9f is not a count\n", tr);
            return TLSQ_OK;
        }
        const int64_t k2 = (int64_t)k2d;
        double* Fm = (double*)Cm;   // (C is not needed any more)
        if (ro.nukeA && k2 > 0) {
            // B = P G2s P + I - P = G2s P + I - P (P is a projector that commutes with G2s)
            TLSQ_TRY(matfun_mul(h, (const double*)G2s, Pm, (double*)W1, N));
            TLSQ_TRY(matfun_lin2(h, (const double*)W1, 1.0, Pm, -1.0, 1.0, (double*)W2, N));   // B
            double bn = 0.0;
            TLSQ_TRY(matfun_trace_norm(h, (const double*)W2, N, nullptr, &bn));
            bool conv2 = false;
            // (W2 = B is only read by the first statement of the iteration; W1 and G2s serve as its scratch)
            TLSQ_TRY(matfun_invsqrt(h, (const double*)W2, N, bn, (double*)Wz, (double*)Yb, (double*)W1, (double*)G2s, 60, &it_r,
                                    &conv2));
            if (!conv2) {
                if (dbg) fprintf(stderr, "  matrix-function route: inverse square root did not converge\n");
                return TLSQ_OK;
            }
            // The coupled iteration loses accuracy quickly once cond(B) reaches a few thousand (rounding breaks the
            // commutativity it relies on): accept W only with the residual ||W B W - I||_F <= 1e-9 in hand
            {
                TLSQ_TRY(matfun_mul(h, (const double*)Wz, (const double*)W2, (double*)W1, N));      // W B
                TLSQ_TRY(matfun_mul(h, (const double*)W1, (const double*)Wz, (double*)G2s, N));     // W B W
                double rs[3];
                TLSQ_TRY(matfun_stats(h, (const double*)G2s, N, rs));
                if (!(rs[0] <= 1e-18)) {
                    if (dbg) fprintf(stderr, "  matrix-function route: inverse square root residual %.2e: rejected\n", std::sqrt(rs[0]));
                    return TLSQ_OK;
                }
            }
            dbg_hash(h, "mf.invsqrt", Wz, (size_t)N * N * 8);
            TLSQ_TRY(matfun_mul(h, (const double*)Wz, Pm, (double*)W1, N));               // B^(-1/2) P
            TLSQ_TRY(matfun_lin2(h, Pm, 1.0, (const double*)W1, -1.0, 0.0, Fm, N));        // F = P - B^(-1/2) P
        } else {
            TLSQ_TRY(matfun_axpbi(h, Pm, Fm, N, 1.0, 0.0));                               // F = P  (:211-212)
        }
        // A = Z2 F + (Z X_S) diag(g) X_S' = Z Phi,  Phi = (I - X_S X_S') F + X_S diag(g) X_S'.  (Not Z (F + ...): F only
        // annihilates X_S to ~1e-9 - rounding of G2 over the gap - and Z X_S is sigma_top large.)  The GEMM takes Phi'.
        double* Phi = (double*)Yb;
        if (rS > 0) {
            SelWeights sw;
            for (int64_t i = 0; i < 32; ++i) {
                sw.sel[i] = 0;
                sw.w[i] = i < rS ? gw[(size_t)i] : 0.0;
            }
            const double* XsS = (const double*)h->ws[WS_VS].p;   // X_S, gathered by rebuild_factors above (N x rS)
            TLSQ_TRY(launch_symm_skinny(h, Fm, N, XsS, (double*)W1, N, rS));              // Y = F X_S
            TLSQ_TRY(matfun_phi(h, Fm, XsS, (const double*)W1, sw, rS, N, Phi));
        } else {
            Phi = Fm;
        }
        // A = Z Phi through Phi' (the GEMM's first operand is indexed [column of A, k])
        TLSQ_TRY(need_A());
        dbg_hash(h, "mf.Phi", Phi, (size_t)N * N * 8);
        TLSQ_TRY(gemm_mixed(h, false, false, Phi, 0, N, Zp, Prec<T>::f32, M, A, Prec<T>::f32, M, N, M, N, false));
        hbm_other += 2.0 * panel_bytes;
        dbg_hash(h, "mf.A", A, (size_t)n * sizeof(T));
        *svp_out = rS + k2;
        *sigma_top_out = stop;
        *ok = true;
        mf_rS_prev = rS;
        mf_k2_last = k2;
        if (dbg)
            fprintf(stderr, "  matrix-function route: |S|=%lld + trace(P)=%lld, sign %d steps, inverse sqrt %d steps\n", (long long)rS,
                    (long long)k2, it_s, it_r);
        return TLSQ_OK;
    };
    bool g_ready = false;   // WS_G already holds (or will hold, in stream order) the Gram of the current Z
    if constexpr (std::is_same<T, double>::value) {
        // the fused sweep + Gram kernel (fused.hip) will most likely serve this call: its slabs (one 272 KB partial Gram per CU,
        // 134 MB) and kernel attributes are set up here, not inside the first iterations of the loop
        if (zmode && !implicit_gram && !hook_svd && !large &&
            fused_zgram_ok(M, N, 0, ro.hankel_y ? nullptr : Y, Y, Y, Y, Y, nullptr, ro.hankel_y != nullptr, lam / mu, ro.hankel_geom)) {
            GramPlan pl0;
            TLSQ_TRY(fused_zgram_plan(h, M, N, &pl0));
            TLSQ_TRY(fused_zgram_warm(h));
        }
    }
    const double t_loop0 = now_ms();
    bool prev_cost_evaluated = false;   // the previous iteration's convergence test needed opnorm(R) itself (see fused_gr below)
    int64_t k = 0;
    for (k = 1; k <= ro.iters; ++k) {                              // :186
        // (test hook, FAIL_RANK=r: rank r of a group leaves iteration 3 with an error - the others must not hang)
        if (k == 3 && h->comm)
            if (const char* fr = dev_get(DEV_FAIL_RANK))
                if (atoi(fr) == h->rank) return set_err(h, TLSQ_ERR_HIP, "rpca: injected failure on rank %d (FAIL_RANK)", h->rank);
        const double inv_mu = 1.0 / mu;
        const double thr = lam / mu;
        T* E = Ebuf[cur];
        T* Z = zmode ? Zbuf[zc] : Zbuf[cur];
        mu_iter = mu;
        z_swept = false;
        if (zmode) {   // what the previous iteration left as "last" is A_{k-1} now; this iteration's factors go to the other pair
            Tm_prev = Tm_last;
            Vs_prev = Vs_last;
            r_prev = r_last;
        }
        prev_no_factors = last_no_factors;
        last_no_factors = false;
        pt.sample = pt.full || tstride == 1 || k % tstride == 1;
        const double hbm_sweeps_at_top = hbm_sweeps;
        pt.mark(false, !have_next);
        if (!have_next)
        {
            if (!(d_transient && k == 1)) TLSQ_TRY(panel_D(&D));   // (iteration 1 may still read the transient copy)
            if (y_pending) {   // k = 1: A is zero and Y not formed yet
                bool first_fused = false;
                if constexpr (std::is_same<T, double>::value) {
                    // tall fp64 panels of 256 columns: the first shrink and the Gram of the Z_1 it writes in one kernel (fused.hip)
                    const double* hy1 = (const double*)ro.hankel_y;
                    if (zmode && !implicit_gram && !hook_svd && !large &&
                        fused_zgram_ok(M, N, 0, hy1 ? nullptr : D, Y, Y, Z, Z, nullptr, hy1 != nullptr, thr, ro.hankel_geom) &&
                        (hy1 || D)) {
                        GramPlan pl1;
                        TLSQ_TRY(fused_zgram_plan(h, M, N, &pl1));
                        TLSQ_TRY(launch_fused_zgram(h, pl1, D, nullptr, nullptr, Y, Y, Z, Z, nullptr, M, N, 0, mu, inv_mu,
                                                    0, 0.0, thr, ro.nonnegE ? 1 : 0, nullptr, nullptr, hy1, ro.hankel_K, -1, true,
                                                    dual_norm));
                        void* Gv;
                        TLSQ_TRY(ws_get(h, Gslot[gcur], (size_t)N * N * 8, &Gv));
                        TLSQ_TRY(fused_zgram_finish(h, pl1, Z, M, N, (double*)Gv));
                        TLSQ_TRY(comm_allreduce(h, (double*)Gv, (size_t)N * N, ncclSum));
                        g_ready = true;
                        first_fused = true;
                    }
                }
                if (!first_fused)
                    host_mark("first shrink");
                    TLSQ_TRY(launch_first_shrink<T>(h, D, Y, zmode ? (T*)nullptr : E, Z, n, (T)dual_norm, (T)inv_mu, (T)thr,
                                                    ro.nonnegE ? 1 : 0));
                y_pending = false;
                hbm_sweeps += (zmode ? 3.0 : 4.0) * panel_bytes - (first_fused && ro.hankel_y ? panel_bytes : 0.0);
            } else {
                TLSQ_TRY(launch_shrink<T>(h, D, A, Y, E, Z, n, (T)inv_mu, (T)thr, ro.nonnegE ? 1 : 0));  // :188-192
                hbm_sweeps += 5.0 * panel_bytes;
            }
        }
        if (!have_next) {
            dbg_hash(h, "shrink.Y", Y, (size_t)n * sizeof(T), k);
            dbg_hash(h, "shrink.Z", Z, (size_t)n * sizeof(T), k);
        }
        if (d_transient) D = nullptr;   // the copy in Zbuf[1] is not to be read any more
        pt.mark(have_next, !have_next);
        bool redo_cert_failed = false;   // second pass through the SVD step: the asynchronous certificate said no
    redo_svd_step:
        double* G = nullptr;                                                                   // :193-194
        bool fast_ok = false;
        bool cert_late = false;          // the certificate's verdict is still out (asked for once the sweep is queued)
        // Rank count, singular-value thresholding and the factors of A for the decomposition currently in (s, V).
        // Normally called once after the SVD step; with a deferred count certificate it is called right after the
        // subspace solver (the rebuild kernels queue up behind the certificate's Lanczos steps) and, should the
        // certificate fail, once more after the fall-back.
        double sigma_top = 0.0, mu_next = mu;
        bool fuse = false, fuse_rebuild = false, rebuilt = false, rebuild_marked = false;
        bool wide_sweep = false;                     // ranks 33..80 on an fp32 panel: A_k formed on the fp32 MFMA inside the sweep
        const float *wide_T32 = nullptr, *wide_Vs32 = nullptr;
        auto count_and_rebuild = [&](bool mark) -> int {
            if (mark && !rebuild_marked) {
                pt.mark();
                rebuild_marked = true;
            }
            sigma_top = s.ncols > 0 ? s.sigma[s.order[0]] : 0.0;
            sigma_top_prev = sigma_top;
            svp = 0;                                                   // :198
            for (int64_t i = 0; i < s.ncols; ++i) svp += (s.sigma[s.order[i]] >= inv_mu) ? 1 : 0;
            sv = std::min(std::max<int64_t>(svp, 1), ro.maxrank);      // :199-204
            if (s.ncols == N && svp < N) {   // a complete decomposition shows where the tail stands
                const double tr = s.sigma[s.order[svp]] / inv_mu;
                bulk_tail = tr * tr > 0.4;
            }
            std::vector<int32_t> sel((size_t)svp);
            std::vector<double> g((size_t)svp);
            for (int64_t p = 0; p < svp; ++p) {
                sel[p] = s.order[p];
                const double sg = s.sigma[sel[p]];
                g[p] = ro.nukeA ? (sg - inv_mu) / sg : 1.0;            // :205-213
            }
            mu_next = std::min(mu * ro.rho, mubar);                    // :223
            fuse = !no_fuse && k < ro.iters;
            // large panels: A = T Vs' is not written at all, the fused sweep below forms it in registers from the
            // factors (7 panel passes per iteration instead of 8 + the pass of the skinny GEMM that writes A)
            if (zmode) fuse_rebuild = fuse && svp <= 32;   // (A in registers; above 32 columns it is stored and read back)
            else
                fuse_rebuild = fuse && !no_fuse_rebuild && !ro.hankel && !hook_opnorm &&
                               rebuild_update_shrink_ok<T>(D, E, Y, R, Ebuf[cur ^ 1], Zbuf[cur ^ 1], M, N, svp);
            // fp32 panels, ranks 33..80 (BASELINE config 5: 65536 x 4096, rank 64): the factors in fp32 - T32 = Z Vg on the fp32 MFMA
            // (opgram32.hip) - and A_k formed tile by tile on the fp32 MFMA inside the sweep (sweeps.hip, k_zsweep_wide) instead of a
            // stored A (skinny GEMM + k_zsweep_lin): 1.11 + 0.78 + 1.42 ms -> 0.43 + ~1.1 ms per iteration there.  The fp64 copy of T
            // (what the final A / E are formed from) is T32 widened.
            wide_sweep = false;
            if constexpr (Prec<T>::f32) {
                if (zmode && fuse && svp > 32 && !ro.hankel_y && !dev_is(DEV_NO_WIDE_SWEEP, '1') && zsweep_wide_ok(M, N, svp) &&
                    wide_factors_ok((const float*)Z, M, M, N, svp)) {
                    void *Vgp, *Vsp, *T1, *auxp;
                    const int slot = (k & 1) ? 1 : 2;   // (the buffer pairs rebuild_factors alternates between in the E-free loop)
                    TLSQ_TRY(ws_get(h, WS_VG, (size_t)N * svp * 8, &Vgp));
                    TLSQ_TRY(ws_get(h, slot == 1 ? WS_VS2 : WS_VS3, (size_t)N * svp * 8, &Vsp));
                    TLSQ_TRY(ws_get(h, slot == 1 ? WS_T2 : WS_T, (size_t)M * svp * 8, &T1));
                    TLSQ_TRY(ws_get(h, WS_AUX0, (size_t)svp * 16, &auxp));
                    TLSQ_TRY(gather_scale_host(h, V, N, sel, g, auxp, (double*)Vgp, (double*)Vsp));
                    int lw = 0;
                    bool from_zq = false;
                    if (hook_zq && hook_zq == (const float*)h->ws[WS_OPT].p && (int64_t)hook_order.size() == hook_zq_p &&
                        !dev_is(DEV_NO_HOOK_ZQ, '1')) {
                        // Z X[:, sel] diag(g) = (Z Q) S[:, order[sel]] diag(g): the product Z Q of the hook's Rayleigh-Ritz step is
                        // still in WS_OPT (nothing has run an operator product since) - no pass over the panel
                        std::vector<int32_t> cols((size_t)svp);
                        bool okc = true;
                        for (int64_t j = 0; j < svp; ++j) {
                            okc = okc && sel[(size_t)j] >= 0 && sel[(size_t)j] < hook_zq_p;
                            cols[(size_t)j] = okc ? hook_order[(size_t)sel[(size_t)j]] : 0;
                        }
                        if (okc) {
                            void* aux2;
                            TLSQ_TRY(ws_get(h, WS_AUX1, (size_t)svp * 16 + 64, &aux2));
                            double* g_dev = (double*)aux2;
                            int32_t* c_dev = (int32_t*)((char*)aux2 + (size_t)svp * 8);
                            TLSQ_TRY(upload_async(h, g_dev, g.data(), (size_t)svp * 8));
                            TLSQ_TRY(upload_async(h, c_dev, cols.data(), (size_t)svp * 4));
                            TLSQ_TRY(wide_factors_from_zq(h, hook_zq, M, N, hook_zq_p, hook_S, c_dev, g_dev, (const double*)Vsp, svp,
                                                          (double*)T1, &wide_T32, &wide_Vs32, &lw));
                            from_zq = true;
                        }
                    }
                    hook_zq = nullptr;
                    if (!from_zq)
                        TLSQ_TRY(wide_factors_f32(h, (const float*)Z, M, M, N, (const double*)Vgp, (const double*)Vsp, svp, (double*)T1,
                                                  &wide_T32, &wide_Vs32, &lw));
                    Tm_last = (const double*)T1;
                    Vs_last = (const double*)Vsp;
                    r_last = svp;
                    if (!from_zq) hbm_other += panel_bytes;
                    wide_sweep = true;
                    fuse_rebuild = true;
                    a_pending = true;
                    rebuilt = true;
                    sub.spec.launched = false;
                    return TLSQ_OK;
                }
            }
            // (the factor product may already be queued: SubspaceState::SpecRebuild - same kernel, same list, decided on the device)
            bool spec_hit = sub.spec.launched && sub.spec.dev_ok && sub.spec.dev_r == svp && svp >= 1 &&
                            svp <= 16 * sub.spec.nct && V == (const double*)h->ws[WS_SX].p && sub.spec.Z == (const void*)Z;
            for (int64_t p = 0; p < svp && spec_hit; ++p) spec_hit = sel[(size_t)p] == (int32_t)p;
            sub.spec.launched = false;
            if (spec_hit) {
                Tm_last = sub.spec.Tout;
                Vs_last = sub.spec.Vs;
                ++n_spec_hits;
            } else {
                TLSQ_TRY(rebuild_factors<T>(h, Z, M, N, M, V, sel, g, &Tm_last, &Vs_last, zmode ? ((k & 1) ? 1 : 2) : 0));
            }
            r_last = svp;
            if (svp > 0) hbm_other += panel_bytes;                      // T = Z Vg reads Z once
            if (!fuse_rebuild) {
                TLSQ_TRY(need_A());
                TLSQ_TRY(rebuild_from_factors<T>(h, Tm_last, Vs_last, M, N, svp, A, M));
                hbm_other += panel_bytes;                               // A = T Vs' written once
            }
            a_pending = fuse_rebuild;
            rebuilt = true;
            if (svp > 0) {
                dbg_hash(h, "rebuild.Tm", Tm_last, (size_t)M * svp * 8, k);
                dbg_hash(h, "rebuild.Vs", Vs_last, (size_t)N * svp * 8, k);
            }
            return TLSQ_OK;
        };
        // How the SVD step is served.  The fast route works on the Gram matrix (warm-started subspace iteration +
        // count certificate); it is only believed when no eigenvalue lies within the Gram route's own uncertainty
        // of the threshold (1/mu)^2.  Everything else — no warm block, no spectral gap, a value inside the window,
        // a threshold at the noise level of G, tiny matrices — goes through the TSQR route (svd_via_r), whose
        // singular values are as accurate as LAPACK's.  Large mode (N > 2048) and the randomized hook keep to the
        // subspace solver.
        if (cb_svd && k >= 2) {
            // The caller's own `svd(Z, sv)` (:195-197): Z goes to the host, the hook fills U, S, Vt there (on the calling
            // thread), and A = U[:,1:svp] diag(S - 1/mu) Vt[1:svp,:] is rebuilt from the returned factors exactly as the
            // reference does (:205-213) - the hook may be approximate, so Z V V' is not a substitute.
            pt.mark(true);
            const int64_t dd = std::min(cb_shards ? Mg : M, N);
            int64_t kout = 0;
            std::vector<double> shS;   // row shards: the hook's singular values on every rank
            if (cb_shards) {
                TLSQ_TRY(gather_panel(Z));
                double hd[2] = {0.0, 0.0};   // triplets, "failed"
                if (h->rank == 0) {
                    cbU.resize((size_t)Mg * dd);
                    cbS.resize((size_t)dd);
                    cbVt.resize((size_t)dd * N);
                    const int cst = opts->svd_cb(cbZ.data(), Mg, N, Mg, sv, cbU.data(), Mg, cbS.data(), cbVt.data(), dd, &kout, opts->user);
                    if (cst != 0 || kout < 0 || kout > dd) hd[1] = 1.0;
                    else hd[0] = (double)kout;
                }
                TLSQ_TRY(bcast_from_rank0(hd, 2));
                if (hd[1] != 0.0) return set_err(h, TLSQ_ERR_ARG, "rpca: the svd callback failed on rank 0");
                kout = (int64_t)hd[0];
                shS.assign((size_t)std::max<int64_t>(kout, 1), 0.0);
                if (h->rank == 0)
                    for (int64_t i = 0; i < kout; ++i) shS[(size_t)i] = (double)cbS[(size_t)i];
                TLSQ_TRY(bcast_from_rank0(shS.data(), (size_t)kout));
            } else {
                cbZ.resize((size_t)n);
                cbU.resize((size_t)M * dd);
                cbS.resize((size_t)dd);
                cbVt.resize((size_t)dd * N);
                TLSQ_HIP(h, hipMemcpyAsync(cbZ.data(), Z, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, h->stream));
                TLSQ_HIP(h, hipStreamSynchronize(h->stream));
                const int cst = opts->svd_cb(cbZ.data(), M, N, M, sv, cbU.data(), M, cbS.data(), cbVt.data(), dd, &kout,
                                             opts->user);
                if (cst != 0 || kout < 0 || kout > dd)
                    return set_err(h, TLSQ_ERR_ARG, "rpca: the svd callback failed (status %d, %lld triplets)", cst, (long long)kout);
            }
            auto cb_sigma = [&](int64_t i) { return cb_shards ? shS[(size_t)i] : (double)cbS[(size_t)i]; };
            svp = 0;                                                   // :198
            for (int64_t i = 0; i < kout; ++i) svp += (cb_sigma(i) >= inv_mu) ? 1 : 0;
            sv = std::min(std::max<int64_t>(svp, 1), ro.maxrank);      // :199-204
            sigma_top = kout > 0 ? cb_sigma(0) : 0.0;
            sigma_top_prev = sigma_top;
            mu_next = std::min(mu * ro.rho, mubar);                    // :223
            fuse = !no_fuse && k < ro.iters;
            fuse_rebuild = false;
            Tm_last = Vs_last = nullptr;
            r_last = svp;
            if (svp > 0) {
                // the first svp triplets in the hook's order (the reference indexes 1:svp as well)
                std::vector<double> hT((size_t)M * svp), hV((size_t)N * svp);
                if (cb_shards) {
                    // rank 0 holds U (Mg x dd) and Vt: the scaled left factors go out as one buffer of row-padded blocks (every
                    // rank keeps its own), the right ones as they are
                    std::vector<double> allT((size_t)h->nranks * cb_maxM * svp, 0.0);
                    if (h->rank == 0) {
                        int64_t r0 = 0;
                        for (int q = 0; q < h->nranks; ++q) {
                            for (int64_t p = 0; p < svp; ++p) {
                                const double gp = ro.nukeA ? cb_sigma(p) - inv_mu : cb_sigma(p);     // :205-213
                                for (int64_t i = 0; i < cb_rows[(size_t)q]; ++i)
                                    allT[(size_t)(((int64_t)q * svp + p) * cb_maxM + i)] = (double)cbU[(size_t)(r0 + i + p * Mg)] * gp;
                            }
                            r0 += cb_rows[(size_t)q];
                        }
                        for (int64_t p = 0; p < svp; ++p)
                            for (int64_t j = 0; j < N; ++j) hV[(size_t)(j + p * N)] = (double)cbVt[(size_t)(p + j * dd)];
                    }
                    TLSQ_TRY(bcast_from_rank0(allT.data(), allT.size()));
                    TLSQ_TRY(bcast_from_rank0(hV.data(), hV.size()));
                    for (int64_t p = 0; p < svp; ++p)
                        for (int64_t i = 0; i < M; ++i) hT[(size_t)(i + p * M)] = allT[(size_t)(((int64_t)h->rank * svp + p) * cb_maxM + i)];
                } else
                for (int64_t p = 0; p < svp; ++p) {
                    const double sg = (double)cbS[(size_t)p];
                    const double gp = ro.nukeA ? sg - inv_mu : sg;     // :205-213
                    for (int64_t i = 0; i < M; ++i) hT[(size_t)(i + p * M)] = (double)cbU[(size_t)(i + p * M)] * gp;
                    for (int64_t j = 0; j < N; ++j) hV[(size_t)(j + p * N)] = (double)cbVt[(size_t)(p + j * dd)];
                }
                void *T1, *Vsb;
                TLSQ_TRY(ws_get(h, WS_T, (size_t)M * svp * 8, &T1));
                TLSQ_TRY(ws_get(h, WS_VS, (size_t)N * svp * 8, &Vsb));
                TLSQ_HIP(h, hipMemcpyAsync(T1, hT.data(), hT.size() * 8, hipMemcpyHostToDevice, h->stream));
                TLSQ_HIP(h, hipMemcpyAsync(Vsb, hV.data(), hV.size() * 8, hipMemcpyHostToDevice, h->stream));
                TLSQ_HIP(h, hipStreamSynchronize(h->stream));
                Tm_last = (const double*)T1;
                Vs_last = (const double*)Vsb;
            }
            pt.mark(true);
            TLSQ_TRY(rebuild_from_factors<T>(h, Tm_last, Vs_last, M, N, svp, A, M));
            a_pending = false;
            rebuilt = true;
            V = nullptr;
            v_is_full = false;
            ++sub.fast;
        } else {
        const bool hook_now = hook_svd && k >= 2;
        bool r_route = !large && !hook_now && (!use_subspace || hook_svd);
        // The threshold has reached the noise level of the Gram matrix: its small eigenvalues mean nothing any more.  While the
        // rank is stable the count can still be certified without the accurate route - on the panel itself, with the
        // dominant part taken out (deflated_certificate below); everything else goes through the TSQR route.
        bool noise_limited = false;
        if (!r_route && !large && !hook_now && sigma_top_prev > 0.0 &&
            !(inv_mu * inv_mu > 2.0 * noise_rel * sigma_top_prev * sigma_top_prev)) {
            if (defl_possible && sub.valid && !bulk_tail) noise_limited = true;
            else r_route = true;
        }
        if (!r_route) {
        GramOp op = panel_op(Z);
        const bool sketch_now = hook_sketch && hook_now;
        op.lowp_ok = sketch_now;
        const bool gram_queued_earlier = g_ready || implicit_gram || sketch_now;
        if (!implicit_gram && !sketch_now) {
            if (g_ready) G = (double*)h->ws[Gslot[gcur]].p;   // already queued behind the previous iteration's sweep (see below)
            else {
                TLSQ_TRY(gram_allreduce<T>(h, Z, M, N, M, &G, Gslot[gcur]));
                hbm_other += panel_bytes;
            }
            op = GramOp();
            op.G = G;
        }
        g_ready = false;
        pt.mark(gram_queued_earlier);
        if (G) dbg_hash(h, "G", G, (size_t)N * N * 8, k);
        if (hook_now) {
            // the reference's `svd(Z, sv)` hook (:195-197): a rank-sv randomized SVD; iteration 1 is always full
            SubspaceState rs;
            rs.hook_rank = sv;
            rs.hook_seed = seed + (uint64_t)k;
            rs.hook_carry = hook_cols;     // the previous iteration's sorted Ritz block, when it is still in the block buffer
            hook_cols = 0;
            hook_zq = nullptr;
            TLSQ_TRY(svd_subspace(h, op, N, inv_mu, rs, &V, s, &sweeps, &fast_ok));
            sub.steps += rs.steps;
            if (fast_ok) hook_cols = rs.hook_carry;
            hook_zq = fast_ok ? rs.hook_zq : nullptr;
            hook_zq_p = rs.p;
            hook_S = rs.hook_S;
            hook_order = rs.hook_order;
        } else if (bulk_tail && !large) {
            sub.fail = SubspaceState::FAIL_NONE;   // straight to the dense tier below
        } else if (noise_limited) {
            // Ritz pairs of the dominant part from G (they are far above its noise), no window and no Gram certificate: the
            // count is certified on the deflated panel.  Any failure hands the iteration to the TSQR route.
            sub.noise_rel = 0.0;
            sub.skip_certificate = true;
            sub.defer_certificate = false;
            const int st_sub = svd_subspace(h, op, N, inv_mu, sub, &V, s, &sweeps, &fast_ok);
            sub.skip_certificate = false;
            if (st_sub < 0) return st_sub;
            if (fast_ok) {
                bool pass = false;
                TLSQ_TRY(deflated_certificate(Z, V, s, inv_mu, &pass));
                if (!pass) fast_ok = false;
            }
            if (!fast_ok) sub.fail = SubspaceState::FAIL_WINDOW;   // (no block-growing retries: the accurate route decides)
        } else if (redo_cert_failed) {
            // (back here after an asynchronous certificate failed: what the synchronous form does then - larger block, retries)
            fast_ok = false;
            sub.cert_pending = false;
            sub.fail = SubspaceState::FAIL_CERT;
        } else {
            sub.noise_rel = noise_rel;
            sub.defer_certificate = !no_cert_overlap;
            sub.cert_async = spec && k < ro.iters;
            if (spec_rebuild && sub.valid && G) {
                // the buffers rebuild_factors would take for this iteration (the E-free loop keeps two pairs in turn), at full width
                void *tb, *vb;
                TLSQ_TRY(ws_get(h, (k & 1) ? WS_T2 : WS_T, (size_t)M * 32 * 8, &tb));
                TLSQ_TRY(ws_get(h, (k & 1) ? WS_VS2 : WS_VS3, (size_t)N * 32 * 8, &vb));
                sub.spec.enable = true;
                sub.spec.Z = Z;
                sub.spec.z_f32 = Prec<T>::f32;
                sub.spec.M = M;
                sub.spec.ldz = M;
                sub.spec.Tout = (double*)tb;
                sub.spec.Vs = (double*)vb;
                sub.spec.nukeA = ro.nukeA;
                sub.spec.before_launch = [&]() {
                    if (!rebuild_marked) pt.mark();
                    rebuild_marked = true;
                };
            }
            const int st_sub = svd_subspace(h, op, N, inv_mu, sub, &V, s, &sweeps, &fast_ok);
            sub.spec.enable = false;
            sub.spec.before_launch = nullptr;
            sub.defer_certificate = false;
            if (st_sub < 0) return st_sub;
            if (fast_ok && sub.cert_pending) {
                // the certificate's kernels are queued: put the rebuild right behind them, then wait for the
                // certificate's read-back only - the host round trip is hidden behind the rebuild kernels.  (Z and G
                // are not modified by the rebuild, so a failed certificate costs nothing but the redo below.)
                TLSQ_TRY(count_and_rebuild(true));
                // asynchronous form (second stream): the verdict is read after the sweep and the next Gram are queued as well
                cert_late = sub.cert_async && zmode && fuse;
                if (cert_late && sub.cert_launch && dev_is(DEV_CERT_EARLY, '1')) {   // (experiment: beside the factor product and the sweep)
                    TLSQ_TRY(sub.cert_launch());
                    sub.cert_launch = nullptr;
                }
                if (!cert_late) {
                    if (sub.cert_launch) {
                        TLSQ_TRY(sub.cert_launch());
                        sub.cert_launch = nullptr;
                    }
                    bool cert_ok = false;
                    TLSQ_TRY(svd_subspace_certify(h, sub, inv_mu, &cert_ok));
                    if (!cert_ok) {
                        fast_ok = false;
                        rebuilt = false;
                    }
                }
            }
            sub.cert_async = false;
        }
        if (!fast_ok && !hook_now && sub.fail != SubspaceState::FAIL_NONE && sub.fail != SubspaceState::FAIL_WINDOW) {
            // enlarge the block (random columns behind the current Ritz vectors) / grant more steps, and retry.
            // Large mode has nothing else; below it a few cheap attempts come before the TSQR route when the
            // block was merely too small (rank above the cold block, rank jumps) - not when convergence stalled.
            const int max_attempts = large ? 10 : 3;
            for (int attempt = 0; !fast_ok && attempt < max_attempts; ++attempt) {
                const int why = sub.fail;
                if (why == SubspaceState::FAIL_WINDOW) break;
                if (!large && why != SubspaceState::FAIL_SMALL && why != SubspaceState::FAIL_CERT) break;
                // a tail whose top sits within a factor two of the threshold is a noise bulk, not a stray value: no
                // block of this solver will ever put a 1.5x gap behind it - the dense route decides
                if (!large && why == SubspaceState::FAIL_CERT && sub.cert_tail > 0.5) {
                    bulk_tail = true;
                    break;
                }
                const bool grow = why == SubspaceState::FAIL_SMALL || why == SubspaceState::FAIL_CERT ||
                                  why == SubspaceState::FAIL_NUMERIC || attempt >= 2;
                const int64_t cap = std::min<int64_t>(pmax, N);
                if (grow) {
                    // (large mode doubles: a cold round of a wide block costs a millisecond, and ranks there are rarely below 30 -
                    //  18 -> 34 -> 51 -> 76 for BASELINE config 5's rank 64 became 34 -> 68; COLD_GROW=0: the old steps)
                    const bool fast_grow = large && !dev_is(DEV_COLD_GROW, '0');
                    const int64_t newp = std::min<int64_t>(cap, sub.p + std::max<int64_t>(16, fast_grow ? sub.p : sub.p / 2));
                    if (newp == sub.p && why == SubspaceState::FAIL_SMALL) break;   // the rank exceeds the largest block
                    if (sub.valid && newp > sub.p) {
                        double* X = (double*)h->ws[WS_SX].p;
                        TLSQ_TRY(launch_fill_hash(h, X + (size_t)N * sub.p, N * (newp - sub.p),
                                                  0xC2B2AE35u + (unsigned int)(k * 131 + attempt)));
                        sub.p = newp;
                    } else {
                        sub.cold_p = newp;
                    }
                }
                sub.extra_steps = 10;
                TLSQ_TRY(svd_subspace(h, op, N, inv_mu, sub, &V, s, &sweeps, &fast_ok));
                sub.extra_steps = 0;
            }
        }
        // Large mode has no cheap second solver.  Up to kReturnedSvdMaxN columns the TSQR route still applies (its block
        // Jacobi then keeps 2-4 columns of R' per workgroup in LDS: seconds per decomposition - the price of a rank beyond
        // the largest subspace block, but a result instead of an error); beyond that the call fails.
        const bool large_dense_ok = large && N <= kReturnedSvdMaxN && !hook_now;
        if (!fast_ok && large && !large_dense_ok)
            return set_err(h, TLSQ_ERR_UNSUPPORTED,
                           "rpca: min(M,N) = %lld > %lld and iteration %lld could not be served by the subspace solver "
                           "(block of %lld columns, reason %d): rank too large for this release",
                           (long long)N, (long long)kFullEigMaxN, (long long)k, (long long)sub.p, sub.fail);
        if (!hook_now) hook_cols = (fast_ok && sub.valid) ? sub.p : 0;   // (the sorted block the certified solver leaves in WS_SX)
        if (fast_ok) {
            ++sub.fast;
        } else if (!hook_now && G && !no_gram_dense && (double)ro.m_global > 400.0 * (double)N) {
            // Very tall panels: the TSQR route would stream the whole panel ~N/16 times, the Gram matrix is already
            // there.  Dense decomposition of G (Jacobi on its Cholesky factor), believed under the same rule as the
            // subspace result: no eigenvalue within the Gram route's uncertainty of the threshold.
            TLSQ_TRY(eig_full(h, G, N, &V, s, &sweeps));
            const double ltop = s.sigma[s.order[0]] * s.sigma[s.order[0]], dl = noise_rel * ltop, tau2 = inv_mu * inv_mu;
            bool uncertain = !(tau2 > 2.0 * dl);
            for (int64_t i = 0; i < N && !uncertain; ++i) uncertain = std::fabs(s.sigma[i] * s.sigma[i] - tau2) <= dl;
            if (uncertain) r_route = true;
            else {
                rebuilt = false;
                ++n_gram_dense;
            }
        } else {
            r_route = true;
        }
        } else {
            g_ready = false;
            pt.mark(true);
        }
        bool matfun_done = false;
        if (r_route && matfun_possible && k >= 2 && !hook_now) {
            int64_t svp_m = 0;
            double stop_m = 0.0;
            bool okm = false;
            TLSQ_TRY(matfun_route(Z, inv_mu, &svp_m, &stop_m, &okm));
            if (okm) {
                // what count_and_rebuild does, without singular vectors: A is already in memory
                if (!rebuild_marked) {
                    pt.mark();
                    rebuild_marked = true;
                }
                const bool dbg_cmp = dev_is(DEV_DEBUG, '3');
                if (dbg_cmp) {   // development: the same iteration through the TSQR route, A compared
                    std::vector<T> a_mf((size_t)n), a_rf((size_t)n);
                    TLSQ_HIP(h, hipMemcpyAsync(a_mf.data(), A, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, h->stream));
                    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
                    double* V2 = nullptr;
                    SmallSvd s2;
                    TLSQ_TRY(svd_via_r<T>(h, Z, M, N, M, &V2, s2, &sweeps));
                    std::vector<int32_t> sel2;
                    std::vector<double> g2;
                    for (int64_t i = 0; i < s2.ncols; ++i) {
                        const double sg = s2.sigma[s2.order[i]];
                        if (sg >= inv_mu) {
                            sel2.push_back(s2.order[i]);
                            g2.push_back(ro.nukeA ? (sg - inv_mu) / sg : 1.0);
                        }
                    }
                    TLSQ_TRY(rebuild_lowrank<T>(h, Z, M, N, M, V2, sel2, g2, R, M));
                    TLSQ_HIP(h, hipMemcpyAsync(a_rf.data(), R, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, h->stream));
                    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
                    double num = 0.0, den = 0.0;
                    for (size_t i = 0; i < (size_t)n; ++i) {
                        const double d = (double)a_mf[i] - (double)a_rf[i];
                        num += d * d;
                        den += (double)a_rf[i] * (double)a_rf[i];
                    }
                    fprintf(stderr, "  [cmp k=%lld] count %lld vs %lld, ||A_mf - A_svd|| / ||A_svd|| = %.3e, sigma_top %.3e, 1/mu %.3e, sigma[cnt] %.3e\n",
                            (long long)k, (long long)svp_m, (long long)sel2.size(), std::sqrt(num / std::max(den, 1e-300)),
                            s2.sigma[s2.order[0]], inv_mu, sel2.size() < (size_t)s2.ncols ? s2.sigma[s2.order[sel2.size()]] : 0.0);
                }
                r_route = false;
                matfun_done = true;
                // values of the bulk have crossed the threshold: the subspace solver has no gap to work with until a complete
                // decomposition says otherwise - the next iterations come straight here instead of growing its block first
                // (noise 1e-3 at 20000 x 512: 90 -> 48 subspace steps, 107 -> 50 ms)
                if (mf_k2_last > 0) bulk_tail = true;
                svp = svp_m;                                               // :198
                sv = std::min(std::max<int64_t>(svp, 1), ro.maxrank);      // :199-204
                sigma_top = stop_m;
                sigma_top_prev = stop_m;
                mu_next = std::min(mu * ro.rho, mubar);                    // :223
                fuse = !no_fuse && k < ro.iters;
                fuse_rebuild = false;
                Tm_last = Vs_last = nullptr;
                r_last = svp;
                a_pending = false;
                rebuilt = true;
                last_no_factors = true;
                V = nullptr;
                g_ready = false;
                ++sub.fast;
            }
        }
        if (r_route) {
            TLSQ_TRY(svd_via_r<T>(h, Z, M, N, M, &V, s, &sweeps));
            if (hook_now) s.ncols = std::min<int64_t>(s.ncols, sv);   // rank-sv truncation of the hook
            rebuilt = false;
            ++sub.full;
            ++n_rroute;
        }
        v_is_full = r_route && !hook_now;
        if (!rebuilt) TLSQ_TRY(count_and_rebuild(!rebuild_marked));
        if (use_subspace && !matfun_done) TLSQ_TRY(carry_block(h, V, N, s, svp, pmax, sub));
        }   // !(cb_svd && k >= 2)
        if (ro.hankel) TLSQ_TRY(soft_hankel(A, (T)thr));  // :214-216

        // decision-only mode: ||R||_2 >= ||R||_F / sqrt(min(M,N)).  The fused sweep accumulates ||R||_F^2 on the
        // side; while that lower bound of the cost is clearly above tol the iteration cannot be the last one and
        // the Gram + Lanczos evaluation of opnorm(R) is skipped altogether.
        const bool want_exact_cost = (info && info->cost_hist) || (opts && opts->on_iter) || k == ro.iters;
        double *sumsq_dev = nullptr, *sumsq_next = nullptr;
        // The verdict of this iteration's count certificate when it runs asynchronously (second stream: it has had the factor
        // product, the sweep and the launch of the next Gram to finish).  "Not certified": Z_k, Y_k, G_k and the factors of
        // A_{k-1} are all intact - the queued sweep wrote the other buffers - so the SVD step is served again, without the
        // subspace shortcut, and its sweep overwrites what the discarded one left.
        auto late_verdict = [&](bool* redo) -> int {
            *redo = false;
            if (!cert_late) return TLSQ_OK;
            cert_late = false;
            if (sub.cert_launch) {   // (not queued yet: the exact-cost branch has no read-back to wait for first)
                TLSQ_TRY(sub.cert_launch());
                sub.cert_launch = nullptr;
            }
            bool cert_ok = false;
            TLSQ_TRY(svd_subspace_certify(h, sub, inv_mu, &cert_ok));
            if (fail_cert_at == k && !redo_cert_failed) cert_ok = false;   // (test hook)
            if (cert_ok) return TLSQ_OK;
            if (dev_get(DEV_DEBUG)) fprintf(stderr, "  iteration %lld: asynchronous certificate failed, redoing the SVD step\n", (long long)k);
            if (sumsq_dev) TLSQ_HIP(h, hipMemsetAsync(sumsq_dev, 0, 72 * 8, h->stream));   // (the discarded sweep's partial sums)
            mu = mu_iter;
            g_ready = false;
            z_swept = false;
            redo_cert_failed = true;
            *redo = true;
            return TLSQ_OK;
        };
        if (fuse && !want_exact_cost && !hook_opnorm) {
            void* scal;
            TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &scal));
            // two sets of 64 partial sums, used alternately: each sweep clears the set of the next one (no memset
            // command between the kernels); both are cleared once on first use
            // (72 doubles per set: 64 partial sums of ||R||_F^2, then one max |R[i, j]| per rank - see max_entry below)
            double* set0 = reinterpret_cast<double*>(reinterpret_cast<char*>(scal) + 512);
            if (!sumsq_ready) {
                TLSQ_HIP(h, hipMemsetAsync(set0, 0, 2 * 72 * 8, h->stream));
                sumsq_ready = true;
            }
            sumsq_dev = set0 + 72 * (k & 1);
            sumsq_next = set0 + 72 * ((k + 1) & 1);
        }
        // Second lower bound of the cost: ||R||_2 >= max |R[i, j]|.  The residual of this iteration is a handful of isolated
        // entries on top of a much smaller dense part (the support of E still moving), so its largest entry is within a few
        // per cent of its spectral norm (measured: 0.94-0.99 of it in every iteration of C2) where the Frobenius bound is off
        // by sqrt(N / rank): the E-free sweep keeps the maximum on the side (one slot per rank: the sum all-reduce of row
        // shards then carries every rank's maximum unchanged), and "not converged" is settled without the Gram of R in all
        // iterations but the last one or two.
        const bool no_maxb = dev_is(DEV_NO_MAX_BOUND, '1');
        const int maxslot = (zmode && sumsq_dev && !no_maxb && h->nranks <= 8) ? h->rank : -1;
        // The residual panel R_k is only read by the cost evaluation.  While the Frobenius bound of the previous
        // iteration was far above tol this one's will be too (the cost shrinks by ~rho per iteration): the sweep then
        // does not store R_k at all (one panel pass less); should the bound disagree, R_k is recomputed below.
        const double rskip_margin = rskip_env >= 0.0 ? rskip_env : (prev_lower_from_max ? 3.0 : 6.0);
        const bool store_R = !(sumsq_dev && prev_lower > rskip_margin * ro.tol);
        T* Rst = store_R ? R : nullptr;
        if (!store_R) ++n_rskip;
        pt.mark(false, true);
        // Will the next iteration want the Gram of Z_{k+1}?  (Not on the TSQR route, not with the implicit operator.)  It is
        // queued before the host looks at this iteration's cost - the GPU works through that round trip, and a Gram is
        // wasted only at convergence.
        const bool r_next = !large && (!use_subspace || (!hook_svd && sigma_top > 0.0 && !(defl_possible && !bulk_tail) &&
                            !(1.0 / (mu_next * mu_next) > 2.0 * noise_rel * sigma_top * sigma_top)));
        // (not when the previous iteration's cost bound - tight since it is the largest entry of the residual - was already
        //  within 1.6 tol: this iteration is then most likely the last one, and a Gram queued now would be the wasted one;
        //  should the loop go on after all, the Gram is computed at the top of the next iteration instead)
        const double last_guess = [] { const char* e = dev_get(DEV_LAST_GUESS); return e ? atof(e) : 1.6; }();
        const bool likely_last = maxslot >= 0 && prev_lower > 0.0 && prev_lower < last_guess * ro.tol;   // (infinity before the first bound: false)
        const bool gram_next = sumsq_dev && !r_next && !implicit_gram && !likely_last && !hook_sketch;
        bool gram_queued = false;
        bool fused_gram = false;   // the sweep kernel has accumulated the Gram of Z_{k+1} as well (fused.hip): only its slabs are left to add
        bool fused_gr = false;     // ... the Gram of the residual R_k instead (and R_k was not stored)
        bool gr_ready = false;     // R_k' R_k sits in the cost evaluation's Gram slot
        GramPlan fused_pl;
        // one launch of the fused sweep over rows [r0, r1) (r1 = 0: the whole panel): E-free form or classic form
        const T* hy_sweep = (const T*)ro.hankel_y;
        if (zmode && fuse) {
            if (!fuse_rebuild) hy_sweep = nullptr;           // A from memory: the linear kernel reads a real D
            if (!hy_sweep) TLSQ_TRY(panel_D(&D));
        }
        auto sweep_rows = [&](int64_t r0, int64_t r1, size_t pad_lds) -> int {
            if (zmode)
                return launch_zsweep<T>(h, D, Tm_last, Vs_last, fuse_rebuild ? (T*)nullptr : A, Ybuf[ycur], Ybuf[ycur ^ 1],
                                        Zbuf[zc], Rst, M, N, svp, (T)mu, (T)inv_mu, ro.nonnegA ? 1 : 0, (T)(1.0 / mu_next),
                                        (T)(lam / mu_next), ro.nonnegE ? 1 : 0, sumsq_dev, sumsq_next, hy_sweep, ro.hankel_K,
                                        r0, r1, maxslot, ro.hankel_geom, Zbuf[zc ^ 1]);
            return launch_rebuild_update_shrink<T>(h, D, Tm_last, Vs_last, E, Y, Rst, Ebuf[cur ^ 1], Zbuf[cur ^ 1], M, N, svp,
                                                   (T)mu, ro.nonnegA ? 1 : 0, (T)(1.0 / mu_next), (T)(lam / mu_next),
                                                   ro.nonnegE ? 1 : 0, sumsq_dev, sumsq_next, (const T*)ro.hankel_y,
                                                   ro.hankel_K, r0, r1, pad_lds, ro.hankel_geom);
        };
        // (a max |Z| left by the previous sweep - Handle::absmax_panel - described the Z_k of this iteration's SVD step: whatever
        //  needed it has been queued, and the panel is about to be rewritten)
        h->absmax_panel = nullptr;
        if (zmode && !fuse) {
            // the last allowed iteration (no next shrink to fuse with): E_k is formed now, then the plain residual :217-221
            if (prev_no_factors) {   // A_{k-1} has no factor form: E_k = D - Z_k + Y_k / mu_k (:192)
                TLSQ_TRY(panel_D(&D));
                TLSQ_TRY(launch_e_from_z<T>(h, D, Zbuf[zc], Ybuf[ycur], Ebuf[0], n, (T)inv_mu));
            } else {
                TLSQ_TRY(form_final_e(Ybuf[ycur], mu));
            }
            if (ro.nonnegA) TLSQ_TRY(launch_clamp_nonneg<T>(h, A, n));
            if (ro.hankel_y && (!Dm || d_transient)) {
                TLSQ_TRY(launch_residual_hankel<T>(h, (const T*)ro.hankel_y, ro.hankel_K, A, Ebuf[0], R, M, N, ro.hankel_geom));
                hbm_sweeps += 6.0 * panel_bytes;
            } else {
                TLSQ_TRY(panel_D(&D));
                TLSQ_TRY(launch_residual<T>(h, D, A, Ebuf[0], R, n));
                hbm_sweeps += 7.0 * panel_bytes;
            }
        } else if (wide_sweep) {
            // A_k above 32 columns on an fp32 panel: formed on the fp32 MFMA inside the sweep (3 reads + 2 writes + R)
            if constexpr (Prec<T>::f32) {
                TLSQ_TRY(panel_D(&D));
                TLSQ_TRY(launch_zsweep_wide(h, (const float*)D, wide_T32, M, wide_Vs32, svp, (const float*)Ybuf[ycur], (float*)Ybuf[ycur ^ 1],
                                            (float*)Zbuf[zc], (float*)Zbuf[zc ^ 1], (float*)Rst, M, N, (float)mu, (float)inv_mu,
                                            ro.nonnegA ? 1 : 0, (float)(1.0 / mu_next), (float)(lam / mu_next), ro.nonnegE ? 1 : 0,
                                            sumsq_dev, sumsq_next, maxslot, true));
            }
            z_swept = true;
            hbm_sweeps += (Rst ? 6.0 : 5.0) * panel_bytes;
        } else if (zmode && !fuse_rebuild) {
            // A_k above 32 columns: stored by the rebuild, read back here (6 panel passes + R)
            TLSQ_TRY(sweep_rows(0, 0, 0));
            z_swept = true;
            hbm_sweeps += (Rst ? 7.0 : 6.0) * panel_bytes;
        } else if (fuse_rebuild) {
            // :205-213 (in registers), :217-222 and the next iteration's :188-192 in a single pass over the panels.
            // Very large panels: the sweep is HBM-bound, the Gram of what it writes MFMA-bound, and each takes many
            // milliseconds - the sweep goes row chunk by row chunk and the Gram of a finished chunk runs beside the sweep
            // of the next one, on the handle's second stream.  The two kernels do share the CUs, but each runs at about
            // half speed meanwhile (kernel trace at 1e7 x 256, 8 chunks: sweep chunk 2.6 ms alone / 3.7 beside a Gram
            // chunk, Gram chunk 1.5 / 3.6): 30.9 ms for the pair instead of 35.0, nothing at 200000 x 512 (TLSQ_OVERLAP_CHUNKS
            // forces a chunk count, 1 = off).
            const int env_chunks = [] { const char* e = dev_get(DEV_OVERLAP_CHUNKS); return e ? atoi(e) : -1; }();
            const bool f32mfma_gram = Prec<T>::f32 && N > 2048;   // (gram_any's choice: that kernel is not chunked)
            int nchunks = (gram_next && !f32mfma_gram && M * N >= ((int64_t)1 << 30)) ? 8 : 1;
            if (env_chunks >= 1 && env_chunks <= 8 && gram_next && !f32mfma_gram) nchunks = env_chunks;
            // fp64 panels of 256 columns (lowrankfilter's n = 256): sweep and Gram in ONE kernel (fused.hip) - the rows of
            // Z_{k+1} feed the MFMA from LDS on their way to memory, the Gram reads nothing from HBM and needs no second
            // stream.  Its slabs are reduced where the next Gram is queued below.
            if constexpr (std::is_same<T, double>::value) {
                if (zmode && gram_next && env_chunks < 0 &&
                    fused_zgram_ok(M, N, svp, D, Ybuf[ycur], Ybuf[ycur ^ 1], Zbuf[zc], Zbuf[zc ^ 1], Rst, hy_sweep != nullptr,
                                   lam / mu_next, ro.hankel_geom)) {
                    // The sweep was going to store R_k, i.e. the cost of this iteration will most likely be evaluated (the
                    // bound is close to tol): the kernel then accumulates R_k' R_k - what opnorm(R_k) (:225) is taken from -
                    // instead of the Gram of Z_{k+1}, and R_k is not stored at all.  Should the loop go on, the Gram of Z_{k+1}
                    // is formed at the top of the next iteration (as after any iteration that did not queue it).
                    // (only when the previous iteration's cost had to be evaluated exactly - its bounds could not settle
                    //  "not converged" - so that this one's will be too: a sweep that stores R_k is not enough of a sign, at
                    //  C3 two of four such iterations are still settled by the bounds and would pay for the Gram of Z_{k+1}
                    //  they did not accumulate)
                    //  (N = 256 only: at N = 512 the off-diagonal block would need the stored R_k)
                    fused_gr = Rst != nullptr && prev_cost_evaluated && N == 256 && group_fused_shape && !dev_is(DEV_NO_FUSED_GR, '1');
                    TLSQ_TRY(fused_zgram_plan(h, M, N, &fused_pl));
                    TLSQ_TRY(launch_fused_zgram(h, fused_pl, D, Tm_last, Vs_last, Ybuf[ycur], Ybuf[ycur ^ 1], Zbuf[zc], Zbuf[zc ^ 1],
                                                Rst, M, N, svp, mu, inv_mu, ro.nonnegA ? 1 : 0, 1.0 / mu_next, lam / mu_next,
                                                ro.nonnegE ? 1 : 0, sumsq_dev, sumsq_next, hy_sweep, ro.hankel_K, maxslot, false,
                                                1.0, fused_gr));
                    fused_gram = true;
                    nchunks = 0;
                }
            }
            if (nchunks == 0) {
                // (swept and accumulated above)
            } else if (nchunks > 1) {
                TLSQ_TRY(second_stream(h));
                // (TLSQ_OVERLAP_LDS: unused dynamic LDS per sweep workgroup, caps its residency per CU - measured: 40 KB
                // no change, 80 KB, i.e. one sweep workgroup per CU, slower)
                const size_t pad_lds = [] { const char* e = dev_get(DEV_OVERLAP_LDS); return e ? (size_t)atol(e) : (size_t)0; }();
                const int64_t rows_c = ((M + nchunks - 1) / nchunks + 511) / 512 * 512;
                GramPlan pl;
                TLSQ_TRY(gram_plan(h, Prec<T>::f32, N, rows_c, nchunks, &pl));
                void* Gv;   // (the other slot of the speculative loop's double buffer: G_k stays intact for a late verdict)
                TLSQ_TRY(ws_get(h, Gslot[gcur ^ 1], (size_t)N * N * 8, &Gv));
                for (int c = 0; c < nchunks; ++c) {
                    const int64_t r0 = std::min<int64_t>((int64_t)c * rows_c, M), r1 = std::min<int64_t>(r0 + rows_c, M);
                    if (r1 > r0) TLSQ_TRY(sweep_rows(r0, r1, pad_lds));
                    TLSQ_HIP(h, hipEventRecord(h->ev_b[c], h->stream));
                    TLSQ_HIP(h, hipStreamWaitEvent(h->stream_b, h->ev_b[c], 0));
                    TLSQ_TRY(gram_launch_chunk(h, h->stream_b, pl, (zmode ? Zbuf[zc ^ 1] : Zbuf[cur ^ 1]) + r0, M, r1 - r0, c));
                }
                TLSQ_TRY(gram_reduce(h, h->stream_b, pl, (double*)Gv, N));
                TLSQ_HIP(h, hipEventRecord(h->ev_b[8], h->stream_b));
                gram_queued = true;   // (h->stream waits for ev_b[8] below, behind the publication of the Frobenius sums)
            } else {
                TLSQ_TRY(sweep_rows(0, 0, 0));
            }
            if (zmode) z_swept = true;
            hbm_sweeps += (((Rst && !fused_gr) ? 7.0 : 6.0) - (zmode ? 1.0 : 0.0) - (ro.hankel_y ? 1.0 : 0.0)) * panel_bytes;
        } else if (fuse) {
            // :217-222 of this iteration and :188-192 of the next one in a single pass over the panels
            TLSQ_TRY(panel_D(&D));
            TLSQ_TRY(launch_update_shrink<T>(h, D, A, E, Y, Rst, Ebuf[cur ^ 1], Zbuf[cur ^ 1], n, (T)mu,
                                             ro.nonnegA ? 1 : 0, (T)(1.0 / mu_next), (T)(lam / mu_next),
                                             ro.nonnegE ? 1 : 0, sumsq_dev, sumsq_next));
            hbm_sweeps += (Rst ? 8.0 : 7.0) * panel_bytes;
        } else {
            TLSQ_TRY(panel_D(&D));
            TLSQ_TRY(launch_update<T>(h, D, A, E, Y, R, n, (T)mu, ro.nonnegA ? 1 : 0));     // :217-222
            hbm_sweeps += 6.0 * panel_bytes;
        }
        pt.mark(false, true);
        if (timing && pt.sample) {   // this iteration's sweep launches (first shrink included) sit between recorded events
            n_timed += (hbm_sweeps > hbm_sweeps_at_top) ? (have_next ? 1 : 2) : 0;
            hbm_timed += hbm_sweeps - hbm_sweeps_at_top;
        }
        if (zmode && z_swept) {
            dbg_hash(h, "sweep.Y", Ybuf[ycur ^ 1], (size_t)n * sizeof(T), k);
            dbg_hash(h, "sweep.Z", Zbuf[zc ^ 1], (size_t)n * sizeof(T), k);
            if (Rst && !fused_gr) dbg_hash(h, "sweep.R", Rst, (size_t)n * sizeof(T), k);
        }
        mu = mu_next;
        double rn = 0.0;
        bool cost_skipped = false;
        if (sumsq_dev) {
            double fro2 = 0.0, part[72];
            TLSQ_TRY(comm_allreduce(h, sumsq_dev, maxslot >= 0 ? 72 : 64, ncclSum));   // row shards: same bits on every rank afterwards
            const bool no_mailbox = dev_is(DEV_NO_MAILBOX, '1');
            const bool mail_sum = h->mailbox && h->mailbox_bytes >= 1024 && !no_mailbox;
            double mail_seq = 0.0;
            if (mail_sum) {
                mail_seq = (h->mail_seq += 1.0);
                TLSQ_TRY(launch_publish_slots(h, sumsq_dev, mail_seq));
            } else {
                TLSQ_HIP(h, hipMemcpyAsync(h->pinned, sumsq_dev, 576, hipMemcpyDeviceToHost, h->stream));
                TLSQ_HIP(h, hipEventRecord(h->ev[32], h->stream));
            }
            pt.mark();
            // The bound almost always says "not the last iteration": queue the next iteration's Gram of Z_{k+1}
            // right away so that the GPU works through the host round trip below (the opnorm evaluation, when it
            // is needed after all, goes to a Gram buffer of its own; a Gram is wasted only at convergence).
            if (gram_queued) {   // the chunks' Gram sits on the second stream: join it (+ the all-reduce of row shards)
                void* Gv;
                TLSQ_TRY(ws_get(h, Gslot[gcur ^ 1], (size_t)N * N * 8, &Gv));
                TLSQ_HIP(h, hipStreamWaitEvent(h->stream, h->ev_b[8], 0));
                TLSQ_TRY(comm_allreduce(h, (double*)Gv, (size_t)N * N, ncclSum));
                hbm_other += panel_bytes;
                g_ready = true;
            } else if (fused_gram && fused_gr) {
                // (WS_G2, never WS_G: with the speculative loop the count certificate of this iteration may still be reading G_k
                //  from WS_G on the second stream when this reduction runs; g_ready stays false)
                void* Gv;
                TLSQ_TRY(ws_get(h, WS_G2, (size_t)N * N * 8, &Gv));
                TLSQ_TRY(gram_reduce(h, h->stream, fused_pl, (double*)Gv, N));
                TLSQ_TRY(comm_allreduce(h, (double*)Gv, (size_t)N * N, ncclSum));
                gr_ready = true;
            } else if (fused_gram) {
                void* Gv;
                TLSQ_TRY(ws_get(h, Gslot[gcur ^ 1], (size_t)N * N * 8, &Gv));
                TLSQ_TRY(fused_zgram_finish(h, fused_pl, (const double*)(const void*)Zbuf[zc ^ 1], M, N, (double*)Gv));   // (fused_gram: T is double)
                TLSQ_TRY(comm_allreduce(h, (double*)Gv, (size_t)N * N, ncclSum));
                if (N != 256) hbm_other += panel_bytes;   // (the off-diagonal block reads the stored panel)
                g_ready = true;
            } else if (gram_next) {   // (the TSQR route does not use the Gram matrix)
                double* Gn = nullptr;
                TLSQ_TRY(gram_allreduce<T>(h, zmode ? Zbuf[zc ^ 1] : Zbuf[cur ^ 1], M, N, M, &Gn, Gslot[gcur ^ 1]));
                hbm_other += panel_bytes;
                g_ready = true;
            }
            pt.mark();
            pt.collect_previous(acc);                      // (host work hidden behind the sweep + Gram just queued)
            bool got_mail = false;
            if (mail_sum) {
                volatile double* mb = h->mailbox;
                const double t_poll = now_ms();
                while (mb[0] != mail_seq && now_ms() - t_poll < 2000.0) {
                }
                got_mail = mb[0] == mail_seq;
                if (got_mail) {
                    for (int i = 0; i < 72; ++i) part[i] = mb[8 + i];
                } else {
                    h->mailbox_bytes = 0;   // never seen in practice; classic read-back from now on
                    TLSQ_HIP(h, hipMemcpyAsync(h->pinned, sumsq_dev, 576, hipMemcpyDeviceToHost, h->stream));
                    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
                }
            } else {
                TLSQ_HIP(h, hipEventSynchronize(h->ev[32]));   // the copy only, not the Gram queued behind it
            }
            if (!got_mail) memcpy(part, h->pinned, 576);
            {   // (the sweep is through - its sums have arrived: now the certificate's kernels, beside the Gram just queued)
                bool redo = false;
                TLSQ_TRY(late_verdict(&redo));
                if (redo) goto redo_svd_step;
            }
            for (int i = 0; i < 64; ++i) fro2 += part[i];
            double lower = std::sqrt(fro2 / (double)std::min(ro.m_global, N)) / d_norm;   // <= cost
            bool max_settles = false;
            if (maxslot >= 0) {
                double mx = 0.0;
                for (int i = 64; i < 72; ++i)
                    if (std::isfinite(part[i]) && part[i] > mx) mx = part[i];
                const double lower_mx = mx / d_norm;               // max |R[i, j]| / ||D||_2 <= cost
                max_settles = lower_mx > ro.tol * (1.0 + 1e-6);   // (the entries of R carry ~1e-8 relative rounding)
                prev_lower_from_max = lower_mx > lower;
                if (lower_mx > lower) lower = lower_mx;
            } else {
                prev_lower_from_max = false;
            }
            if (dev_get(DEV_DEBUG))
                fprintf(stderr, "  iteration %lld: cost bound %.4e (%.2f tol), residual %s, settles %d\n", (long long)k, lower, lower / ro.tol,
                        store_R ? "stored" : "not stored", (int)(lower > 2.0 * ro.tol || max_settles));
            prev_lower = lower;
            if (lower > 2.0 * ro.tol || max_settles) {
                cost = lower;          // a lower bound of the true cost: only "not converged yet" is known
                cost_skipped = true;
            } else if (!store_R && zmode) {
                // mispredicted (E-free loop): R_k = (Y_{k+1} - Y_k) / mu_k
                TLSQ_TRY(launch_residual_from_y<T>(h, Ybuf[ycur ^ 1], Ybuf[ycur], R, n, (T)(1.0 / mu_iter)));
                hbm_sweeps += 3.0 * panel_bytes;
            } else if (!store_R) {
                // mispredicted: the residual is needed after all.  R_k = D - A_k - E_k (A_k possibly still in factors)
                if (a_pending) {
                    TLSQ_TRY(rebuild_from_factors<T>(h, Tm_last, Vs_last, M, N, r_last, A, M));
                    if (ro.nonnegA) TLSQ_TRY(launch_clamp_nonneg<T>(h, A, n));
                    a_pending = false;
                }
                if (ro.hankel_y && (!Dm || d_transient)) {
                    TLSQ_TRY(launch_residual_hankel<T>(h, (const T*)ro.hankel_y, ro.hankel_K, A, E, R, M, N, ro.hankel_geom));
                    hbm_sweeps += 3.0 * panel_bytes;
                } else {
                    TLSQ_TRY(launch_residual<T>(h, D, A, E, R, n));
                    hbm_sweeps += 4.0 * panel_bytes;
                }
            }
        } else {
            pt.mark();   // (empty read-back and "next Gram" windows)
            pt.mark();
            bool redo = false;
            TLSQ_TRY(late_verdict(&redo));
            if (redo) goto redo_svd_step;
        }
        const int cost_gslot = (g_ready || gr_ready) ? WS_G2 : WS_G;   // WS_G may already belong to the next iteration; gr_ready: R'R is in WS_G2
        if (cost_skipped) {
            // nothing to evaluate
        } else if (cb_opnorm) {
            TLSQ_TRY(opnorm_callback(R, &rn));                                                    // :225 caller's hook
            cost = rn / d_norm;
        } else if (hook_opnorm) {
            TLSQ_TRY(opnorm_power<T>(h, R, M, N, M, mvps, seed + 7919ull * (uint64_t)k, &rn));   // :225 hook
            cost = rn / d_norm;
        } else {
            // When nobody looks at the per-iteration cost (no cost_hist, no verbose hook) only the DECISION
            // cost < tol matters: the Lanczos Ritz value is a lower bound of sigma_max^2, so the test is
            // settled ("not converged") as soon as it passes (tol*d_norm)^2.  The last iteration is exact.
            const double stop_sigma = want_exact_cost ? 0.0 : ro.tol * d_norm * (1.0 + 1e-9);
            // accuracy of the value itself: 1e-8 when it is reported per iteration (cost_hist / on_iter are compared with
            // the reference's to 1e-6); when only the decision matters 1e-6 is enough - a value within 1e-5 of tol is
            // re-evaluated to full accuracy below (residual spectra are flat: 154 -> ~110 Lanczos steps at N = 4096)
            const double cost_rel = want_exact_cost ? 1e-8 : 1e-6;
            if (implicit_gram) {
                TLSQ_TRY(sigma_max_of_op(h, panel_op(R), N, cost_rel, &rn, stop_sigma));
            } else {                                                                               // :225
                void* Gc;
                TLSQ_TRY(ws_get(h, cost_gslot, (size_t)N * N * 8, &Gc));
                if (!gr_ready) {   // (gr_ready: the fused sweep has left R_k' R_k here, all-reduced)
                    TLSQ_TRY(gram_any(h, R, Prec<T>::f32, M, N, M, (double*)Gc, N));
                    TLSQ_TRY(comm_allreduce(h, (double*)Gc, (size_t)N * N, ncclSum));
                }
                bool settled = false;
                bool power_tried = false;
                // (not when the lower bounds of this iteration already sit 10 % below tol: the largest entry of the residual is
                //  within a few per cent of its norm, so "not converged" is not what three power steps are going to say - the last
                //  iteration of a solve went through them, and a host round trip, for nothing)
                const bool likely_converged = maxslot >= 0 && prev_lower_from_max && prev_lower < 0.9 * ro.tol;
                if (stop_sigma > 0.0 && !no_power_lb && !likely_converged) {
                    // "not converged" from three power steps on the vector carried over from the previous evaluation
                    // (||G v|| <= lambda_max for unit v): no Lanczos run unless the bound falls short of the mark
                    double lb = 0.0;
                    const int pst = power_lower_bound(h, (const double*)Gc, N, N, !power_vec_valid, 3, &lb);
                    if (pst < 0) return pst;
                    if (pst == 0) {
                        power_vec_valid = true;
                        power_tried = true;
                        if (lb >= stop_sigma * stop_sigma) {
                            rn = std::sqrt(lb);
                            settled = true;
                        }
                    }
                }
                if (!settled) {
                    if (power_tried || likely_converged) h->lz_first_chunk = 17;   // (16 pairs: what a flat residual spectrum takes to 1e-6)
                    TLSQ_TRY(sigma_max_of_gram(h, (const double*)Gc, N, cost_rel, &rn, &sweeps, stop_sigma));
                }
            }
            if (!gr_ready) hbm_other += panel_bytes;
            cost = rn / d_norm;
            if (std::fabs(cost - ro.tol) <= 1e-5 * ro.tol) {       // too close to call: full accuracy
                if (implicit_gram) TLSQ_TRY(sigma_max_of_op(h, panel_op(R), N, 1e-13, &rn));
                else TLSQ_TRY(sigma_max_of_gram(h, (const double*)h->ws[cost_gslot].p, N, 1e-13, &rn, &sweeps));
                cost = rn / d_norm;
            }
        }
        prev_cost_evaluated = !cost_skipped;
        pt.mark(cost_skipped);
        pt.next_iteration(acc);   // (no stream-wide synchronisation here: the next Gram may still be running)
        if (info) {
            info->iters_done = k;
            if (info->cost_hist && k <= info->hist_capacity) info->cost_hist[k - 1] = cost;
            if (info->svp_hist && k <= info->hist_capacity) info->svp_hist[k - 1] = svp;
        }
        if (opts && opts->on_iter) opts->on_iter(k, cost, svp, opts->user);  // :226
        if (cost < ro.tol) {                                       // :228
            converged = true;
            break;
        }
        if (fuse) {
            if (zmode) {
                ycur ^= 1;
                zc ^= 1;                         // (both entries are the same panel unless the loop speculates)
                if (g_ready) gcur ^= 1;          // (likewise)
            } else {
                cur ^= 1;
            }
            have_next = true;
        } else {
            have_next = false;
        }
    }
    if (k > ro.iters) k = ro.iters;
    host_mark("loop left");
    h->absmax_panel = nullptr;   // (what follows may rewrite Z: the maximum a sweep left for it does not describe it any more)
    // (no synchronisation here: the final A and E are queued first, the phase timer's events are read behind the one below)
    if (dev_get(DEV_DEBUG)) fprintf(stderr, "  speculative factor products used: %lld of %lld iterations\n", (long long)n_spec_hits, (long long)k);
    T* Z = zmode ? Zbuf[zc] : Zbuf[cur];
    // (factors_out: the caller takes A as factors - unhankel reads them directly - and does not want E)
    const bool give_factors = ro.factors_out && zmode && a_pending && !ro.nonnegA && r_last <= 32 && !(S_host || Vt_host || U_dev);
    if (a_pending && !give_factors) {   // the loop never stored A: materialise the final one (:205-213, :217-219)
        TLSQ_TRY(need_A());
        TLSQ_TRY(rebuild_from_factors<T>(h, Tm_last, Vs_last, M, N, r_last, A, M));
        if (ro.nonnegA) TLSQ_TRY(launch_clamp_nonneg<T>(h, A, n));
    }
    h->out_factors = give_factors;
    h->out_Tm = Tm_last;
    h->out_Vs = Vs_last;
    h->out_r = r_last;
    if (zmode && z_swept && ro.factors_out && !(S_host || Vt_host || U_dev)) {
        // nothing: neither E nor the last Z is wanted
    } else if (zmode && z_swept) {
        // The E-free loop stopped behind a sweep: Y_k in Ybuf[ycur], Y_{k+1} in the other buffer, Z already Z_{k+1}.
        // Z_k = A_k + Y_{k+1} / mu_k for the returned decomposition (:194, :238), then E_k over whichever Y buffer E is.
        if ((S_host || Vt_host || U_dev) && Zbuf[zc] == Zbuf[zc ^ 1]) {   // (a double-buffered Z still holds Z_k itself)
            TLSQ_TRY(launch_z_from_y<T>(h, A, Ybuf[ycur ^ 1], Z, n, (T)(1.0 / mu_iter)));
            hbm_sweeps += 3.0 * panel_bytes;
        }
        if (prev_no_factors) {   // A_{k-1} has no factor form: E_k = D - A_k - R_k, R_k = (Y_{k+1} - Y_k) / mu_k (:221-222)
            TLSQ_TRY(panel_D(&D));
            TLSQ_TRY(launch_e_from_residual<T>(h, D, A, Ybuf[ycur ^ 1], Ybuf[ycur], Ebuf[0], n, (T)(1.0 / mu_iter)));
            hbm_sweeps += 5.0 * panel_bytes;
        } else {
            TLSQ_TRY(form_final_e(Ybuf[ycur], mu_iter));
            hbm_sweeps += (ro.hankel_y && r_prev <= 32 ? 2.0 : 3.0) * panel_bytes;
        }
    }
    if (cur != 0)   // the last E_k sits in the spare buffer: move it to the caller's panel
        TLSQ_HIP(h, hipMemcpyAsync(E, Ebuf[cur], (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
    if (ro.hankel) TLSQ_TRY(soft_hankel(E, (T)(lam / mu)));  // :234-236
    host_mark("A, E queued");
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    host_mark("synchronised");
    pt.finish(acc);
    if (ro.ae_final && *ro.ae_final && (S_host || Vt_host || U_dev)) (*ro.ae_final)();
    if (info) {
        info->ms_loop = now_ms() - t_loop0;
        info->converged = converged ? 1 : 0;
        info->final_cost = cost;
        info->final_mu = mu;
        info->jacobi_sweeps = sweeps;
        info->eig_full = sub.full + n_gram_dense;
        info->eig_fast = sub.fast;
        info->subspace_steps = sub.steps;
        info->tsqr_iterations = (int32_t)n_rroute;
        info->hbm_bytes_sweeps = hbm_sweeps;
        info->hbm_bytes = hbm_sweeps + hbm_other;
        info->residual_stores_skipped = n_rskip;
        info->sweeps_timed = n_timed;
        info->hbm_bytes_sweeps_timed = hbm_timed;
        info->kern_gram_h3 = h->kern_gram_h3;
        info->kern_zx_h = h->kern_zx_h;
        info->kern_zty_h = h->kern_zty_h;
        info->kern_zsweep_wide = h->kern_zsweep_wide;
        info->kern_fused_zgram = h->kern_fused_zgram;
    }
    if (sv_out) *sv_out = sv;

    // ---- the returned `s` (SVD of the last Z), src/robustPCA.jl:194,238 ----
    const int64_t d = std::min(ro.m_global, N);
    // Large mode (N > 2048): the loop never needs a dense decomposition, but the returned `s` is the complete SVD of the
    // last Z (:194, :238).  Up to kReturnedSvdMaxN columns it is computed once after the loop by the TSQR route (the
    // block Jacobi then keeps 2-4 columns of R' per workgroup in LDS: seconds, not milliseconds - only when the caller
    // asks for S / Vt / U); beyond that only the Ritz triplets of the last block exist.
    const bool slow_full_s = large && N <= kReturnedSvdMaxN && !h->comm;
    if ((S_host || Vt_host || U_dev) && V && large && !slow_full_s) {
        // large mode: only the Ritz triplets of the last block are available; the rest of S is NaN and the
        // corresponding vectors are zero (documented in include/tlsq.h)
        const int64_t have = std::min<int64_t>(s.ncols, d);
        if (S_host)
            for (int64_t p = 0; p < d; ++p)
                S_host[p] = p < have ? s.sigma[s.order[p]] : std::numeric_limits<double>::quiet_NaN();
        if (Vt_host) {
            std::vector<double> hv((size_t)N * have);
            TLSQ_HIP(h, hipMemcpyAsync(hv.data(), V, (size_t)N * have * 8, hipMemcpyDeviceToHost, h->stream));
            TLSQ_HIP(h, hipStreamSynchronize(h->stream));
            for (int64_t p = 0; p < d; ++p)
                for (int64_t j = 0; j < N; ++j)
                    Vt_host[p + j * ldVt] = p < have ? hv[(size_t)s.order[p] * N + j] : 0.0;
        }
        if (U_dev) {
            TLSQ_HIP(h, hipMemsetAsync(U_dev, 0, (size_t)M * d * sizeof(T), h->stream));
            std::vector<int32_t> sel((size_t)have);
            std::vector<double> g((size_t)have);
            for (int64_t p = 0; p < have; ++p) {
                sel[p] = s.order[p];
                const double sg = s.sigma[sel[p]];
                g[p] = sg > 0.0 ? 1.0 / sg : 0.0;
            }
            void *Vg, *aux;
            TLSQ_TRY(ws_get(h, WS_VG, (size_t)N * have * 8, &Vg));
            TLSQ_TRY(ws_get(h, WS_AUX0, (size_t)have * 16 + 64, &aux));
            int32_t* dsel = (int32_t*)aux;
            double* dg = (double*)((char*)aux + ((have * 4 + 7) / 8) * 8);
            TLSQ_TRY(upload_async(h, dsel, sel.data(), (size_t)have * 4));
            TLSQ_TRY(upload_async(h, dg, g.data(), (size_t)have * 8));
            TLSQ_TRY(launch_gather_scale(h, V, N, dsel, dg, have, (double*)Vg, nullptr));
            TLSQ_TRY(gemm_mixed(h, true, false, Vg, 0, N, Z, Prec<T>::f32, M, U_dev, Prec<T>::f32, M, have, M, N, false));
            TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        }
        return converged ? TLSQ_OK : TLSQ_MAXITER;
    }
    if ((S_host || Vt_host || U_dev) && (!large || slow_full_s) && !v_is_full) {
        // The last iteration used a subspace path: the caller wants the complete SVD of the last Z (:194, :238), with small
        // singular values and vectors as accurate as LAPACK's.
        //
        // Deflated form (round 4): the dominant triplets are already known - the loop's last Ritz pairs (X, sigma), converged
        // to 2e-13 lambda_max - and what keeps the Gram matrix of Z from resolving the REST is only their size.  With the
        // dominant part scaled down to s = 2 x (the largest value left),
        //     Z_P = Z - Z X_S diag(1 - s / sigma_i) X_S'   (one tall-skinny product, one rank-|S| panel update),
        // Z_P has the right singular vectors of Z (X_S for an |S|-fold value s, everything else unchanged) and a condition
        // number of s / sigma_min: its Gram matrix is accurate at the scale of the tail, and Cholesky + one-sided Jacobi on it
        // (eig_full: relative accuracy for every value) deliver the tail triplets.  0.3 ms + a Cholesky factorisation instead of
        // the Householder TSQR of the panel (6.5 ms at 20000 x 512); the Jacobi sweeps are the same.  Guards: a spectral gap
        // behind S (leak of span(S) below 1e-10 sigma_top, as on the matrix-function route), the cluster found where it must be,
        // sigma_min >= 2e-4 s (below that the Gram matrix of Z_P loses the value to rounding: tolerance of the returned S) -
        // anything else goes through the TSQR route as before.
        bool deflated_ok = false;
        if (!dev_is(DEV_NO_DEFLATED_SVD, '1') && !large && !Prec<T>::f32 && V && V == (const double*)h->ws[WS_SX].p &&
            s.ncols > 0 && s.ncols < N && mu_iter > 0.0 && N >= 64) {
            const double tau = 1.0 / mu_iter, tau2 = tau * tau;
            const double stop = s.sigma[s.order[0]];
            const double dl = noise_rel * stop * stop;
            int64_t cnt = 0;
            while (cnt < s.ncols && cnt < 32) {
                const double sg = s.sigma[s.order[cnt]];
                if (!(sg * sg >= std::max(1e3 * tau2, tau2 + 2.0 * dl))) break;
                ++cnt;
            }
            while (cnt > 0) {
                const double sg = s.sigma[s.order[cnt - 1]];
                const double sg1 = cnt < s.ncols ? s.sigma[s.order[cnt]] : 0.0;
                if (sg * sg - sg1 * sg1 >= 2e-3 * stop * sg) break;
                --cnt;
            }
            // (every value outside the block is below 1 / mu: the loop's count certificate)
            const double next = cnt < s.ncols ? std::max(s.sigma[s.order[cnt]], tau) : tau;
            const double slev = 2.0 * next;
            if (cnt > 0 && slev < 0.25 * s.sigma[s.order[cnt - 1]]) {
                std::vector<int32_t> sel((size_t)cnt);
                std::vector<double> gg((size_t)cnt), sig_top((size_t)cnt);
                for (int64_t i = 0; i < cnt; ++i) {
                    sel[(size_t)i] = s.order[(size_t)i];
                    sig_top[(size_t)i] = s.sigma[(size_t)sel[(size_t)i]];
                    gg[(size_t)i] = 1.0 - slev / sig_top[(size_t)i];
                }
                const double* X = V;
                const double *TmS = nullptr, *VsS = nullptr;
                TLSQ_TRY(rebuild_factors<T>(h, Z, M, N, M, X, sel, gg, &TmS, &VsS, 3));   // T = Z X_S diag(1 - s / sigma)
                TLSQ_TRY(rebuild_from_factors<T>(h, TmS, VsS, M, N, cnt, R, M));          // R = T X_S'
                TLSQ_TRY(launch_diff<T>(h, Z, R, R, n));                                  // R = Z_P
                double* GP = nullptr;
                TLSQ_TRY(gram_allreduce<T>(h, R, M, N, M, &GP, WS_G2));
                double* V2 = nullptr;
                SmallSvd s2;
                // (hints for the slicer: every eigenvalue of G_P is at most s^2, |S| of them sit there, and everything else is at
                //  most next^2 - Ritz values converged to 1e-13 and the loop's count certificate)
                // Round 6: the normwise sliced solver on G_P itself (sliced.hip) where it applies - the guards below hold its
                // result to the same conditions - otherwise Cholesky + Jacobi as before.
                bool nw = false;
                {
                    void *Vn, *lamn;
                    TLSQ_TRY(ws_get(h, WS_V, (size_t)N * N * 8, &Vn));
                    TLSQ_TRY(ws_get(h, WS_LAM, (size_t)N * 8, &lamn));
                    int64_t sw_n = 0;
                    TLSQ_TRY(symeig_sliced_normwise_f64(h, GP, N, (double*)Vn, (double*)lamn, &sw_n, slev * slev, (int)cnt, slev * slev,
                                                        next * next, &nw));
                    if (nw) {
                        s2.sigma.resize((size_t)N);
                        TLSQ_HIP(h, hipMemcpyAsync(s2.sigma.data(), lamn, (size_t)N * 8, hipMemcpyDeviceToHost, h->stream));
                        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
                        for (auto& v : s2.sigma) v = std::sqrt(std::max(v, 0.0));
                        s2.ncols = N;
                        sort_desc(s2);
                        V2 = (double*)Vn;
                        sweeps += sw_n;
                    }
                }
                if (!nw) TLSQ_TRY(eig_full(h, GP, N, &V2, s2, &sweeps, false, WS_V, false, slev * slev, (int)cnt, slev * slev, next * next));
                bool good = s2.ncols == N;
                for (int64_t i = 0; i < cnt && good; ++i)
                    good = std::fabs(s2.sigma[(size_t)s2.order[(size_t)i]] - slev) <= 1e-6 * slev;
                if (good && cnt < N) good = s2.sigma[(size_t)s2.order[(size_t)cnt]] <= 0.75 * slev;
                if (good) good = s2.sigma[(size_t)s2.order[(size_t)(N - 1)]] >= 2e-4 * slev;
                if (dev_get(DEV_DEBUG))
                    fprintf(stderr, "  returned s, deflated form: |S|=%lld s=%.3e next=%.3e sigma_min=%.3e -> %s\n", (long long)cnt, slev,
                            s2.sigma[(size_t)s2.order[(size_t)std::min<int64_t>(cnt, N - 1)]], s2.sigma[(size_t)s2.order[(size_t)(N - 1)]],
                            good ? "used" : "TSQR route instead");
                if (good) {
                    // the |S|-fold value s: any basis of span(X_S) serves Z_P - for Z it is X_S itself, with the loop's values
                    for (int64_t i = 0; i < cnt; ++i) {
                        const int32_t col = s2.order[(size_t)i];
                        TLSQ_HIP(h, hipMemcpyAsync(V2 + (size_t)col * N, X + (size_t)sel[(size_t)i] * N, (size_t)N * 8,
                                                   hipMemcpyDeviceToDevice, h->stream));
                        s2.sigma[(size_t)col] = sig_top[(size_t)i];
                    }
                    sort_desc(s2);
                    // X_S and the tail vectors come from two different computations: orthogonal to each other only as far as the
                    // Ritz vectors are converged (the guard above: ~1e-10).  Columns into descending order, then the first-order
                    // passes that also serve U (Gram-Schmidt order: the dominant vectors stay as they are) - N x N work only.
                    void* Vsrt;
                    TLSQ_TRY(ws_get(h, WS_CP1, (size_t)N * N * 8, &Vsrt));
                    TLSQ_TRY(gather_cols(h, V2, N, s2.order, (double*)Vsrt));
                    TLSQ_HIP(h, hipMemcpyAsync(V2, Vsrt, (size_t)N * N * 8, hipMemcpyDeviceToDevice, h->stream));
                    std::vector<double> sg((size_t)N);
                    for (int64_t i = 0; i < N; ++i) sg[(size_t)i] = s2.sigma[(size_t)s2.order[(size_t)i]];
                    s2.sigma = sg;
                    std::iota(s2.order.begin(), s2.order.end(), 0);
                    TLSQ_TRY(polish_derived_vectors<double>(h, V2, N, N, sg, N, true));
                    s = s2;
                    V = V2;
                    deflated_ok = true;
                }
            }
        }
        if (!deflated_ok) TLSQ_TRY(svd_via_r<T>(h, Z, M, N, M, &V, s, &sweeps));
        if (info) info->jacobi_sweeps = sweeps;
    }
    if (S_host && V)
        for (int64_t p = 0; p < d; ++p) S_host[p] = s.sigma[s.order[p]];
    if (Vt_host && V && ro.vt_dev && ro.vt_written && d <= N) {
        // Vt[p, j] = V[j, order[p]] straight into the caller's (or the entry layer's) device buffer
        void* ordp;
        TLSQ_TRY(ws_get(h, WS_AUX0, (size_t)d * 16 + 64, &ordp));
        TLSQ_TRY(upload_async(h, ordp, s.order.data(), (size_t)d * 4));
        TLSQ_TRY(launch_vt_out<T>(h, V, N, (const int32_t*)ordp, d, (T*)ro.vt_dev, ro.vt_ld));
        *ro.vt_written = true;
    } else if (Vt_host && V) {
        std::vector<double> hv((size_t)N * N);
        TLSQ_HIP(h, hipMemcpyAsync(hv.data(), V, (size_t)N * N * 8, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        // Vt[p, j] = V[j, order[p]] in 32 x 32 tiles: the plain double loop writes with a stride of ldVt doubles - one cache line
        // (and beyond 512 columns one page) per element, 2.9 ms for 512 x 512 in the kernel trace's host gap, against 0.3 ms tiled
        constexpr int64_t TB = 32;
        for (int64_t p0 = 0; p0 < d; p0 += TB)
            for (int64_t j0 = 0; j0 < N; j0 += TB) {
                const int64_t p1 = std::min(p0 + TB, d), j1 = std::min(j0 + TB, N);
                for (int64_t p = p0; p < p1; ++p) {
                    const double* col = hv.data() + (size_t)s.order[p] * N;
                    for (int64_t j = j0; j < j1; ++j) Vt_host[p + j * ldVt] = col[j];
                }
            }
    }
    if (U_dev && V) {
        // U = Z V diag(1/sigma); columns with sigma == 0 are returned as zeros
        std::vector<int32_t> sel((size_t)d);
        std::vector<double> g((size_t)d);
        for (int64_t p = 0; p < d; ++p) {
            sel[p] = s.order[p];
            const double sg = s.sigma[sel[p]];
            g[p] = sg > 0.0 ? 1.0 / sg : 0.0;
        }
        void *Vg, *aux;
        TLSQ_TRY(ws_get(h, WS_VG, (size_t)N * d * 8, &Vg));
        TLSQ_TRY(ws_get(h, WS_AUX0, (size_t)d * 16, &aux));
        int32_t* dsel = (int32_t*)aux;
        double* dg = (double*)((char*)aux + ((d * 4 + 7) / 8) * 8);
        TLSQ_HIP(h, hipMemcpyAsync(dsel, sel.data(), (size_t)d * 4, hipMemcpyHostToDevice, h->stream));
        TLSQ_HIP(h, hipMemcpyAsync(dg, g.data(), (size_t)d * 8, hipMemcpyHostToDevice, h->stream));
        TLSQ_TRY(launch_gather_scale(h, V, N, dsel, dg, d, (double*)Vg, nullptr));
        TLSQ_TRY(gemm_mixed(h, true, false, Vg, 0, N, Z, Prec<T>::f32, M, U_dev, Prec<T>::f32, M, d, M, N, false));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        std::vector<double> sig_desc((size_t)d);
        for (int64_t p = 0; p < d; ++p) sig_desc[(size_t)p] = s.sigma[sel[(size_t)p]];
        TLSQ_TRY(polish_derived_vectors<T>(h, U_dev, M, d, sig_desc, ro.m_global));
    }
    return converged ? TLSQ_OK : TLSQ_MAXITER;  // :232
}

template int opnorm_gram<double>(Handle*, const double*, int64_t, int64_t, int64_t, double*, int64_t*, double, double, int);
template int svd_via_gram<double>(Handle*, const double*, int64_t, int64_t, int64_t, double**, SmallSvd&, int64_t*,
                                  PhaseTimer*);
template int rebuild_lowrank<double>(Handle*, const double*, int64_t, int64_t, int64_t, const double*,
                                     const std::vector<int32_t>&, const std::vector<double>&, double*, int64_t);
template int rpca_core<double>(Handle*, const double*, int64_t, int64_t, const ResolvedOpts&, const tlsq_rpca_opts*,
                               double*, double*, double*, double*, double*, int64_t, int64_t*, tlsq_rpca_info*);
template int svd_via_gram<float>(Handle*, const float*, int64_t, int64_t, int64_t, double**, SmallSvd&, int64_t*, PhaseTimer*);
template int svd_via_r<double>(Handle*, const double*, int64_t, int64_t, int64_t, double**, SmallSvd&, int64_t*);
template int svd_via_r<float>(Handle*, const float*, int64_t, int64_t, int64_t, double**, SmallSvd&, int64_t*);
template int rebuild_lowrank<float>(Handle*, const float*, int64_t, int64_t, int64_t, const double*,
                                    const std::vector<int32_t>&, const std::vector<double>&, float*, int64_t);
template int rpca_core<float>(Handle*, const float*, int64_t, int64_t, const ResolvedOpts&, const tlsq_rpca_opts*, float*,
                              float*, float*, double*, double*, int64_t, int64_t*, tlsq_rpca_info*);
}  // namespace tlsq
