// Spectrum slicing in front of the one-sided Jacobi of the accurate route (round 6).
//
// The returned `s` of rpca (src/robustPCA.jl:194, :238 under /root/reference) is the complete SVD of the last Z; after the
// dominant triplets are deflated (solver.hip) what is left is an N x N symmetric eigenproblem whose ~N eigenvalues lie in a FLAT
// bulk (a factor of two or so apart, no gaps).  One-sided Jacobi on its Cholesky factor needs 12-14 sweeps there, and a sweep is
// N - 1 sequential rounds whatever the kernel does (k_jacobi_reg: 31 launches of 16 workgroups at N = 512): 10.7 ms of a 9.4 ms
// solve.  The rounds are the cost, so the work is cut across them:
//
//   1. L = chol(G + delta I) as before (cholesky.hip), K = L'L - the matrix whose eigenvectors the Jacobi rotations converge to.
//   2. Split the spectrum of K into slices of at most ~64 eigenvalues with matrix sign functions on the fp64 MFMA (matfun.hip,
//      fixed polynomial schedules, the slices of a level side by side in one batch).  A slice is its spectral projector P and an
//      interval [lo, hi]; it is cut at the MEAN t of its eigenvalues, trace(P K) / trace(P), by the sign function of
//      K_s - t I with K_s = P K + hi (I - P) - everything outside the slice moved to its upper end, so that the iteration is
//      scaled by the slice's own width, not by the spectrum's (a deflated cluster 50 times above a flat bulk would otherwise
//      leave ten bulk eigenvalues within the resolution of every split).  Children: P_low = (I - sign) / 2, P_up = P - P_low;
//      counts and means come from traces, no eigenvalue is ever computed.
//   3. The Cholesky factor of an orthogonal projector has orthonormal columns (P = C C', P^2 = P => C'C = I): a pivoted Cholesky
//      of every P_j, one workgroup per slice, gives J0 = [C_1 | C_2 | ...] - orthogonal up to what the projectors lack, which is
//      measured and polished (J0 <- J0 (1.5 I - 0.5 J0'J0)) to 1e-14 - and B0 = L J0: a factor of G whose column groups are
//      the slices: mutually orthogonal already, flat inside.
//   4. Jacobi sweeps that only pair columns of the same slice (jacobi.hip, k_jacobi_reg's table form): a slice of 64 columns
//      has 3 rounds per sweep instead of 31, and all slices share the launches.
//   5. Ordinary sweeps over all pairs until nothing rotates: they remove what the projectors left between neighbouring slices
//      (eigenvalues closer to a split point than the schedule resolves) - one or two sweeps - and are the accuracy statement:
//      the result is whatever the plain Cholesky + Jacobi route converges to, by the same kernel and the same stopping rule.
//
// Nothing here decides an answer: a slice that comes out wrong only costs sweeps in step 5.  Where a guard fails (a projector
// far from idempotent, J0 not orthogonal) the caller falls back to the plain route.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.hpp"

namespace tlsq {

namespace {

constexpr int SL_MAXSL = 16;     // slices = projector matrices kept
constexpr int SL_MAXK = 256;     // columns of one slice the pivoted Cholesky takes

struct ChildMap {   // per split z of a level: parent's projector slot (becomes the upper child), slot of the lower child
    int32_t parent[8], low[8];
};

// P_low = (I - S_z) / 2 into slot low[z];  P_parent <- P_parent - P_low (the upper child)
__global__ __launch_bounds__(256) void k_slice_children(const double* __restrict__ S, double* __restrict__ PJ, int N, ChildMap m) {
    const int z = blockIdx.y;
    const int64_t total = (int64_t)N * N;
    const double* Sz = S + (int64_t)z * total;
    double* Pp = PJ + (int64_t)m.parent[z] * total;
    double* Pl = PJ + (int64_t)m.low[z] * total;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e % N), j = (int)(e / N);
        const double pl = 0.5 * ((i == j ? 1.0 : 0.0) - Sz[e]);
        Pl[e] = pl;
        Pp[e] -= pl;
    }
}

__global__ __launch_bounds__(256) void k_set_identity(double* __restrict__ P, int N) {
    const int64_t total = (int64_t)N * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
        P[e] = (e % N) == (e / N) ? 1.0 : 0.0;
}

struct ProjMap {
    int32_t slot[SL_MAXSL], start[SL_MAXSL], k[SL_MAXSL];
};

// Pivoted Cholesky of the orthogonal projector in slot[j] (rank k[j]): its factor C (N x k, orthonormal columns for an exact
// projector) to columns start[j].. of J.  One workgroup per slice, thread = row; left-looking: step s picks the largest
// remaining diagonal entry p, column s = (P[:, p] - C[:, :s] C[p, :s]') / sqrt(d_p).  fail[0] != 0: a pivot below 1e-3 (the
// matrix is not a projector of that rank).
__global__ __launch_bounds__(1024) void k_proj_chol(const double* __restrict__ PJ, ProjMap m, int N, double* __restrict__ J,
                                                    int* __restrict__ fail) {
    __shared__ double lp[SL_MAXK];
    __shared__ double redv[16];
    __shared__ int redi[16];
    __shared__ int s_piv;
    __shared__ double s_pd;
    const int j = blockIdx.x, k = m.k[j];
    const int64_t nn = (int64_t)N * N;
    const double* P = PJ + (int64_t)m.slot[j] * nn;
    double* C = J + (int64_t)m.start[j] * N;
    const int i = threadIdx.x, lane = i & 63, w = i >> 6;
    const bool row = i < N;
    double di = row ? P[i + (int64_t)i * N] : -1.0;
    for (int s = 0; s < k; ++s) {
        // argmax of the remaining diagonal (ties: the smallest row index)
        double v = di;
        int vi = i;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_down(v, off, 64);
            const int oi = __shfl_down(vi, off, 64);
            if (ov > v || (ov == v && oi < vi)) {
                v = ov;
                vi = oi;
            }
        }
        if (lane == 0) {
            redv[w] = v;
            redi[w] = vi;
        }
        __syncthreads();
        if (i == 0) {
            double bv = redv[0];
            int bi = redi[0];
            for (int q = 1; q < 16; ++q)
                if (redv[q] > bv || (redv[q] == bv && redi[q] < bi)) {
                    bv = redv[q];
                    bi = redi[q];
                }
            s_piv = bi;
            s_pd = bv;
        }
        __syncthreads();
        const int p = s_piv;
        const double dp = s_pd;
        if (!(dp > 1e-3)) {
            if (i == 0) atomicExch(fail, 1);
            for (int t = s; t < k; ++t)
                if (row) C[i + (int64_t)t * N] = 0.0;
            return;
        }
        if (i < s) lp[i] = C[p + (int64_t)i * N];
        __syncthreads();
        double c = row ? P[i + (int64_t)p * N] : 0.0;
        for (int t = 0; t < s; ++t) c -= (row ? C[i + (int64_t)t * N] : 0.0) * lp[t];
        const double l = c / sqrt(dp);
        if (row) {
            C[i + (int64_t)s * N] = l;
            di -= l * l;
            if (i == p) di = -1.0;
        }
        __syncthreads();   // (column s is visible to the workgroup; lp is free again)
    }
}

struct Slice {
    double lo, hi;    // eigenvalue interval
    int slot;         // its projector
    double cnt, sum, sum2;  // trace(P), trace(P K), trace(P K^2)
};

}   // namespace

int small_mm_batched(Handle* h, const double* A, int64_t sA, const double* B, int64_t sB, double* C, int64_t sC, int64_t N, int nb,
                     double alpha, double beta, bool sym, const double* Add, int64_t sAdd, double gamma);
int small_mm_batched_start(Handle* h, const double* P, const double* K, double* X, int64_t N, int nb, const double* alpha,
                           const double* beta, const double* gamma);
int matfun_sign_batched(Handle* h, const double* K, const double* K2, int64_t N, int nb, const double* t, double hi, double l0, double* X,
                        double* W1, double* W2, double** out, double* stats_dev, int* steps_out);
int jacobi_factor_grouped_f64(Handle* h, double* B, int64_t N, const std::vector<std::pair<int, int>>& groups, double* V,
                              double* sig_dev, double floor_rel, int64_t* sweeps_out);

bool symeig_sliced_ok(int64_t N) { return N >= 256 && N <= 1024 && (N % 128) == 0 && !dev_is(DEV_NO_SLICED_EIG, '1'); }

// Same contract as symeig_chol_f64 (B, V: N x N workspace of the caller; V = normalised eigenvectors, sig_dev = sqrt(lambda +
// delta), unsorted).  Hints, all optional (<= 0: none): lam_hi - an upper bound of the eigenvalues of G; n_out eigenvalues are
// known to sit at val_out, and every other one is at most bulk_hi (the deflated cluster of solver.hip and the certified bound of
// what is left) - they only place the first split and scale the iterations.  *used = false: a guard declined, nothing has been
// computed that the caller may use.
int symeig_sliced_f64(Handle* h, const double* G, int64_t N, int64_t ldG, double* B, double* V, double* sig_dev, double* delta_host,
                      int64_t* sweeps_out, double lam_hi, int n_out, double val_out, double bulk_hi, bool* used) {
    *used = false;
    if (sweeps_out) *sweeps_out = 0;
    if (!symeig_sliced_ok(N)) return TLSQ_OK;
    const bool dbg = dev_get(DEV_DEBUG) != nullptr;
    const int64_t nn = N * N;
    const int target = [] { const char* e = dev_get(DEV_SLICE_TARGET); const int v = e ? atoi(e) : 64; return v >= 16 ? v : 64; }();
    const int maxlev = [] { const char* e = dev_get(DEV_SLICE_LEVELS); const int v = e ? atoi(e) : 6; return v >= 1 && v <= 8 ? v : 6; }();
    const double l0 = [] { const char* e = dev_get(DEV_SLICE_L0); const double v = e ? atof(e) : 1e-4; return v > 0.0 && v < 0.5 ? v : 1e-4; }();
    void *scal, *buf;
    TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &scal));
    double* cstats = reinterpret_cast<double*>(reinterpret_cast<char*>(scal) + 192);    // cholesky: max diagonal, delta
    double* sstats = reinterpret_cast<double*>(reinterpret_cast<char*>(scal) + 3072);   // 4 x 8 doubles of a sign batch (2048..2448 is the SpecCtrl)
    int* failflag = reinterpret_cast<int*>(reinterpret_cast<char*>(scal) + 3328);
    // one slab: K | K^2 | J | J' | E | 3 x 8 iterates | projectors
    const int64_t nbuf = 5 + 24 + SL_MAXSL;
    TLSQ_TRY(ws_get(h, WS_SL_BUF, (size_t)nbuf * nn * 8, &buf));
    double* base = (double*)buf;
    double *K = base, *K2 = base + nn, *J = base + 2 * nn, *J2 = base + 3 * nn, *E = base + 4 * nn;
    double *X = base + 5 * nn, *W1 = X + 8 * nn, *W2 = W1 + 8 * nn, *PJ = W2 + 8 * nn;

    // 1. the Cholesky factor, K = L'L and its square (second moments of the slices)
    TLSQ_TRY(cholesky_shifted(h, G, ldG, N, B, V /* scratch */, cstats));
    TLSQ_TRY(gemm_f64(h, true, true, B, N, B, N, K, N, N, N, N, true));
    TLSQ_TRY(small_mm_batched(h, K, nn, K, nn, K2, nn, N, 1, 1.0, 0.0, true, nullptr, 0, 0.0));
    double st3[3], st3b[3];
    TLSQ_TRY(matfun_stats(h, K, N, st3));   // trace, row-sum norm (host round trip)
    TLSQ_TRY(matfun_stats(h, K2, N, st3b));
    const double trK = st3[1], trK2 = st3b[1];
    // (the eigenvalues of K are those of G + delta: the caller's bound with a little room, or Gershgorin's, or the Frobenius norm)
    double hi = std::min(st3[2], std::sqrt(std::max(trK2, 0.0)) * (1.0 + 1e-12));
    if (lam_hi > 0.0) hi = std::min(hi, lam_hi * (1.0 + 1e-6) + 8.0 * (double)N * 2.3e-16 * st3[2]);
    if (!(trK > 0.0) || !std::isfinite(trK) || !(hi > 0.0) || !std::isfinite(hi) || !std::isfinite(trK2)) return TLSQ_OK;

    // 2. slices
    {
        int64_t g = (nn + 255) / 256;
        if (g > 1024) g = 1024;
        hipLaunchKernelGGL(k_set_identity, dim3((unsigned)g), dim3(256), 0, h->stream, PJ, (int)N);
        TLSQ_HIP(h, hipGetLastError());
    }
    std::vector<Slice> sl;
    sl.push_back(Slice{0.0, hi, 0, (double)N, trK, trK2});
    int nslot = 1, total_steps = 0, levels = 0, nsign = 0;
    const bool hint = n_out > 0 && val_out > 0.0 && bulk_hi > 0.0 && bulk_hi < 0.5 * val_out && n_out < (int)N;
    for (int lev = 0; lev < maxlev; ++lev) {
        // the slices to cut at this level: the largest first, at most 8, while projector slots are left
        std::vector<int> pick;
        for (int j = 0; j < (int)sl.size(); ++j)
            if (sl[(size_t)j].cnt > (double)target + 0.5) pick.push_back(j);
        std::sort(pick.begin(), pick.end(), [&](int x, int y) { return sl[(size_t)x].cnt > sl[(size_t)y].cnt; });
        const int room = std::min(8, SL_MAXSL - nslot);
        if ((int)pick.size() > room) pick.resize((size_t)std::max(0, room));
        if (pick.empty()) break;
        std::sort(pick.begin(), pick.end());
        std::vector<double> t, al, be, ga;
        std::vector<int> cut;
        double lev_l0 = l0;
        for (int j : pick) {
            const Slice& s = sl[(size_t)j];
            double tj = s.sum / s.cnt;
            {
                // The schedule resolves eigenvalues of X_0 down to l0 of the interval's scale; the interval is only a bound of the
                // slice (second moments, the caller's hints): the resolution is asked for relative to the width the eigenvalues
                // themselves show - sqrt(12 var), a uniform density's - at 3.44 per polynomial step a loose bound costs little.
                const double var = std::max(s.sum2 / s.cnt - tj * tj, 0.0);
                const double width = std::sqrt(12.0 * var), nrm_b = std::max(tj - s.lo, s.hi - tj);
                if (nrm_b > 0.0) lev_l0 = std::min(lev_l0, std::max(l0 * std::max(width / nrm_b, 1e-3), 1e-9));
            }
            if (lev == 0 && hint) {
                // the known cluster above the rest: the first cut goes between them, and its iteration only has to resolve that gap
                tj = std::sqrt(bulk_hi * val_out);
                const double nrm0 = std::max(tj - s.lo, s.hi - tj);
                lev_l0 = std::min(0.2, 0.5 * std::min(tj - bulk_hi, val_out - tj) / nrm0);
            }
            // (the mean of at least two distinct eigenvalues lies strictly inside; anything else: leave the slice alone)
            if (!(tj > s.lo + 1e-9 * (s.hi - s.lo)) || !(tj < s.hi - 1e-9 * (s.hi - s.lo))) continue;
            const double nrm = std::max(tj - s.lo, s.hi - tj);
            // X_0 = (K_s - t I) / nrm,  K_s = P K + hi (I - P)
            t.push_back(tj);
            al.push_back(1.0 / nrm);
            ga.push_back(-s.hi / nrm);
            be.push_back((s.hi - tj) / nrm);
            cut.push_back(j);
        }
        if (t.empty()) break;
        const int nb = (int)t.size();
        // the parents' projectors side by side (slots are not contiguous: gather through the iterate buffer W2)
        for (int z = 0; z < nb; ++z)
            TLSQ_HIP(h, hipMemcpyAsync(W2 + (int64_t)z * nn, PJ + (int64_t)sl[(size_t)cut[(size_t)z]].slot * nn, (size_t)nn * 8,
                                       hipMemcpyDeviceToDevice, h->stream));
        TLSQ_TRY(small_mm_batched_start(h, W2, K, X, N, nb, al.data(), be.data(), ga.data()));
        double* out = nullptr;
        int steps = 0;
        TLSQ_TRY(matfun_sign_batched(h, K, K2, N, nb, nullptr, 0.0, lev_l0, X, W1, W2, &out, sstats, &steps));
        total_steps += steps;
        nsign += nb;
        TLSQ_HIP(h, hipMemcpyAsync(h->pinned, sstats, (size_t)nb * 32, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        std::vector<double> hs((size_t)nb * 4);
        memcpy(hs.data(), h->pinned, (size_t)nb * 32);
        ChildMap cm;
        memset(&cm, 0, sizeof(cm));
        for (int q = 0; q < nb; ++q) {
            const double tr = hs[(size_t)q * 4], ip = hs[(size_t)q * 4 + 1], fr = hs[(size_t)q * 4 + 2], ip2 = hs[(size_t)q * 4 + 3];
            if (!std::isfinite(tr) || !std::isfinite(ip) || !std::isfinite(fr) || !std::isfinite(ip2)) return TLSQ_OK;
            // ||S||_F^2 = N for an exact sign matrix; what is missing are eigenvalues the schedule did not carry to +-1
            if (dbg)
                fprintf(stderr, "  slicer: level %d, slice [%.4e, %.4e] of %.1f cut at %.6e: %.2f below, N - ||S||_F^2 = %.3e\n", lev,
                        sl[(size_t)cut[(size_t)q]].lo, sl[(size_t)cut[(size_t)q]].hi, sl[(size_t)cut[(size_t)q]].cnt, t[(size_t)q],
                        0.5 * ((double)N - tr), (double)N - fr);
            if (std::fabs((double)N - fr) > 0.05 * (double)N) return TLSQ_OK;
            cm.parent[q] = sl[(size_t)cut[(size_t)q]].slot;
            cm.low[q] = nslot + q;
        }
        {
            int64_t g = (nn + 255) / 256;
            if (g > 512) g = 512;
            hipLaunchKernelGGL(k_slice_children, dim3((unsigned)g, (unsigned)nb), dim3(256), 0, h->stream, (const double*)out, PJ, (int)N, cm);
            TLSQ_HIP(h, hipGetLastError());
        }
        for (int q = nb - 1; q >= 0; --q) {
            const int j = cut[(size_t)q];
            const Slice s = sl[(size_t)j];
            const double tr = hs[(size_t)q * 4], ip = hs[(size_t)q * 4 + 1], ip2 = hs[(size_t)q * 4 + 3];
            (void)ip;
            // below the cut the sign is -1: P_low = (I - S) / 2
            const double cnt_lo = 0.5 * ((double)N - tr), sum_lo = 0.5 * (trK - ip), sum2_lo = 0.5 * (trK2 - ip2);
            Slice lo_c{s.lo, t[(size_t)q], nslot + q, cnt_lo, sum_lo, sum2_lo};
            Slice hi_c{t[(size_t)q], s.hi, s.slot, s.cnt - cnt_lo, s.sum - sum_lo, s.sum2 - sum2_lo};
            hi_c.hi = std::min(hi_c.hi, std::max(hi_c.lo, std::sqrt(std::max(hi_c.sum2, 0.0)) * (1.0 + 1e-9)));
            if (!(hi_c.hi > hi_c.lo)) hi_c.hi = s.hi;
            if (lev == 0 && hint) {   // what is below the first cut is at most bulk_hi
                lo_c.hi = std::min(lo_c.hi, bulk_hi * (1.0 + 1e-6) + 8.0 * (double)N * 2.3e-16 * hi);
            } else {
                // no eigenvalue of a slice exceeds the root of its second moment
                lo_c.hi = std::min(lo_c.hi, std::max(lo_c.lo, std::sqrt(std::max(sum2_lo, 0.0)) * (1.0 + 1e-9)));
            }
            if (!(lo_c.hi > lo_c.lo)) lo_c.hi = t[(size_t)q];
            sl[(size_t)j] = lo_c;
            sl.insert(sl.begin() + j + 1, hi_c);
        }
        nslot += nb;
        ++levels;
    }
    // columns per slice: the rounded counts, made to add up to N on the largest slices; empty slices go
    std::vector<int> kcol(sl.size());
    int have = 0;
    for (size_t j = 0; j < sl.size(); ++j) {
        kcol[j] = (int)std::max(0.0, std::floor(sl[j].cnt + 0.5));
        have += kcol[j];
    }
    for (int guard = 0; have != (int)N && guard < 4 * (int)N; ++guard) {
        size_t big = 0;
        for (size_t j = 1; j < sl.size(); ++j)
            if (kcol[j] > kcol[big]) big = j;
        const int d = have < (int)N ? 1 : -1;
        kcol[big] += d;
        have += d;
    }
    ProjMap pm;
    memset(&pm, 0, sizeof(pm));
    std::vector<std::pair<int, int>> groups;
    int col = 0, nsl = 0;
    for (size_t j = 0; j < sl.size(); ++j) {
        if (kcol[j] <= 0) continue;
        if (nsl >= SL_MAXSL || kcol[j] > SL_MAXK) return TLSQ_OK;
        pm.slot[nsl] = sl[j].slot;
        pm.start[nsl] = col;
        pm.k[nsl] = kcol[j];
        ++nsl;
        groups.push_back({col, kcol[j]});
        col += kcol[j];
    }
    if (col != (int)N || nsl < 2) return TLSQ_OK;   // (nothing was split: the plain route is the same thing)
    if (dbg) {
        fprintf(stderr, "  slicer: N=%lld, %d levels, %d sign matrices (%d polynomial steps), slices:", (long long)N, levels, nsign, total_steps);
        for (auto& g : groups) fprintf(stderr, " %d", g.second);
        fprintf(stderr, "\n");
    }

    // 3. J0 = the Cholesky factors of the projectors, polished to orthogonality; B0 = L J0
    TLSQ_HIP(h, hipMemsetAsync(failflag, 0, 4, h->stream));
    hipLaunchKernelGGL(k_proj_chol, dim3((unsigned)nsl), dim3(1024), 0, h->stream, (const double*)PJ, pm, (int)N, J, failflag);
    TLSQ_HIP(h, hipGetLastError());
    double *Jc = J, *Jn = J2;
    double dev2 = 0.0;
    for (int pass = 0; pass < 5; ++pass) {
        TLSQ_TRY(gemm_f64(h, true, true, Jc, N, Jc, N, E, N, N, N, N, true));
        TLSQ_TRY(matfun_stats(h, E, N, st3));
        dev2 = st3[0];
        if (pass == 0) {
            int ff = 0;
            TLSQ_HIP(h, hipMemcpyAsync(h->pinned, failflag, 4, hipMemcpyDeviceToHost, h->stream));
            TLSQ_HIP(h, hipStreamSynchronize(h->stream));
            memcpy(&ff, h->pinned, 4);
            if (ff) {
                if (dbg) fprintf(stderr, "  slicer: a projector is not of its rank (pivot below 1e-3): plain route\n");
                return TLSQ_OK;
            }
        }
        if (dbg) fprintf(stderr, "  slicer: ||J0'J0 - I||_F = %.3e\n", std::sqrt(std::max(dev2, 0.0)));
        if (!std::isfinite(dev2) || dev2 > 0.25) return TLSQ_OK;
        if (dev2 <= 1e-27) break;
        TLSQ_TRY(matfun_axpbi(h, E, E, N, -0.5, 1.5));
        TLSQ_TRY(small_mm_batched(h, Jc, nn, E, nn, Jn, nn, N, 1, 1.0, 0.0, false, nullptr, 0, 0.0));
        std::swap(Jc, Jn);
    }
    if (dev2 > 1e-24) return TLSQ_OK;
    // B0 = L J0 into K's place, then back into the caller's B (L is not needed any more)
    TLSQ_TRY(small_mm_batched(h, B, nn, Jc, nn, K, nn, N, 1, 1.0, 0.0, false, nullptr, 0, 0.0));
    TLSQ_HIP(h, hipMemcpyAsync(B, K, (size_t)nn * 8, hipMemcpyDeviceToDevice, h->stream));

    // 4. + 5. sweeps inside the slices, then over all pairs
    int64_t sw[2] = {0, 0};
    const int stj = jacobi_factor_grouped_f64(h, B, N, groups, V, sig_dev, (double)N * 2.220446049250313e-16, sw);
    if (dbg) fprintf(stderr, "  slicer: %lld sweeps inside the slices, %lld over all pairs\n", (long long)sw[0], (long long)sw[1]);
    if (sweeps_out) *sweeps_out = sw[0] + sw[1];
    TLSQ_HIP(h, hipMemcpyAsync(h->pinned, cstats, 16, hipMemcpyDeviceToHost, h->stream));
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    double cs[2];
    memcpy(cs, h->pinned, 16);
    if (delta_host) *delta_host = cs[1];
    if (stj < 0) return stj;
    *used = true;
    return TLSQ_OK;
}

}   // namespace tlsq
