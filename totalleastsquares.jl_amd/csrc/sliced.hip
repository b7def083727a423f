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
//      fixed polynomial schedules).  A slice is its spectral projector P and an interval [lo, hi]; it is cut at the MEAN t of
//      its eigenvalues, trace(P K) / trace(P), scaled by the slice's own width, not by the spectrum's (a deflated cluster 50
//      times above a flat bulk would otherwise leave ten bulk eigenvalues within the resolution of every split).  All the cuts
//      of a level share ONE sign iteration: the slices' projectors commute with K and are mutually orthogonal, so
//          X0 = sum_z P_z (K - t_z I) / nrm_z + (I - sum_z P_z) = A K + B        (k_slice_start builds A and B)
//      is block diagonal in the eigenbasis with every slice scaled by its own nrm_z, sign(X0) = S cuts all of them at once,
//      Q = (I - S) / 2, and the children are P_low_z = P_z Q (one batched, indexed product), P_up_z = P_z - P_low_z.
//      Counts and means come from traces, no eigenvalue is ever computed.  (C2: 4 levels = 4 sign matrices, 46 polynomial
//      steps in all, where one iteration per cut took 15 sign matrices.)
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
// far from idempotent, J0 not orthogonal) the caller falls back to the plain route.  (symeig_sliced_f64: serves eig_full's
// Cholesky route, N a multiple of 128 in [256, 1024].  C2, returned s: 14 sweeps over all pairs -> 9 inside the slices + 1.)
//
// The NORMWISE form (symeig_sliced_normwise_f64, what the returned `s` of a plain call takes): the same slices and basis, of G
// itself - no Cholesky factor (1.0 ms at N = 512), no sweeps over all pairs (0.8 ms each) - then the slices' own k x k problems
// T_jj = J0_j' G J0_j (k <= 96), one workgroup per slice with the whole problem in LDS and every sweep in one launch
// (jacobi.hip, k_jacobi_mid_blocks: 0.6 ms; Cholesky + the register kernel on windows of k rows, 45 launches, took 1.3 -
// SLICE_MIDJ=0), V = J0 blockdiag(W_j), one first-order
// refinement from V'GV for what the projectors left between slices, and a certificate max |offdiag(V'GV)| <= 8 N eps ||G||.
// Measured at C2 (20000 x 512, rank 16; profiles/r06_ws_*): the decomposition after the loop 14.8 -> 4.35 ms (under the profiler:
// slicing 2.06, pivoted Cholesky of the projectors 0.35, polish 0.3, the slices' Jacobi 0.65, refinement + certificate 0.8, U and
// the cluster 1.2).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.hpp"

namespace tlsq {

namespace {

constexpr int SL_MAXSL = 32;     // slices = projector matrices kept
constexpr int SL_MAXK = 256;     // columns of one slice the pivoted Cholesky takes

struct ChildMap {   // per cut z of a level: the parent's projector slot (it becomes the upper child), the slot of the lower child
    int32_t n;
    int32_t parent[SL_MAXSL], low[SL_MAXSL];
};

// P_parent <- P_parent - P_low (the upper child), for every cut of the level
__global__ __launch_bounds__(256) void k_slice_children(double* __restrict__ PJ, int N, ChildMap m) {
    const int z = blockIdx.y;
    const int64_t total = (int64_t)N * N;
    double* Pp = PJ + (int64_t)m.parent[z] * total;
    const double* Pl = PJ + (int64_t)m.low[z] * total;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) Pp[e] -= Pl[e];
}

struct CutMap {   // the cuts of a level: projector slot, 1 / nrm, -t / nrm - 1
    int32_t n;
    int32_t slot[SL_MAXSL];
    double alpha[SL_MAXSL], beta1[SL_MAXSL];
};

// A = sum_z alpha_z P_z,  B = I + sum_z beta1_z P_z  (so that A K + B = sum_z P_z (K - t_z I) / nrm_z + (I - sum_z P_z))
__global__ __launch_bounds__(256) void k_slice_start(const double* __restrict__ PJ, int N, CutMap m, double* __restrict__ A,
                                                     double* __restrict__ B) {
    const int64_t total = (int64_t)N * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        double a = 0.0, bsum = (e % N) == (e / N) ? 1.0 : 0.0;
        for (int z = 0; z < m.n; ++z) {
            const double pv = PJ[(int64_t)m.slot[z] * total + e];
            a += m.alpha[z] * pv;
            bsum += m.beta1[z] * pv;
        }
        A[e] = a;
        B[e] = bsum;
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
int small_mm_batched_idx(Handle* h, const double* Abase, const int32_t* idx, int nb, const double* B, double* C, int64_t N);
int slice_stats_batched(Handle* h, const double* S, const double* K, const double* K2, int64_t N, int nb, double* stats_dev);
int matfun_sign_batched(Handle* h, const double* K, const double* K2, int64_t N, int nb, const double* t, double hi, double l0, double* X,
                        double* W1, double* W2, double** out, double* stats_dev, int* steps_out);
int jacobi_factor_grouped_f64(Handle* h, double* B, int64_t N, const std::vector<std::pair<int, int>>& groups, double* V,
                              double* sig_dev, double floor_rel, int64_t* sweeps_out, bool block_diagonal);

bool symeig_sliced_ok(int64_t N) { return N >= 256 && N <= 1024 && (N % 128) == 0 && !dev_is(DEV_NO_SLICED_EIG, '1'); }

namespace {

struct SlicePlan {
    ProjMap pm;
    int nsl = 0;
    std::vector<std::pair<int, int>> groups;   // (first column, columns) of every slice, ascending eigenvalues
};

struct SliceBufs {
    double *K, *K2, *J, *J2, *E, *X, *W1, *W2, *PJ;
    double *sstats;
    int* failflag;
};

int slice_bufs(Handle* h, int64_t N, SliceBufs* b) {
    const int64_t nn = N * N;
    void *scal, *buf;
    TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &scal));
    b->sstats = reinterpret_cast<double*>(reinterpret_cast<char*>(scal) + 3072);   // 4 x 16 doubles of a level's children (2048..2448 is the SpecCtrl)
    b->failflag = reinterpret_cast<int*>(reinterpret_cast<char*>(scal) + 3584);
    // one slab: K | K^2 | J | J' | E | 3 x 8 iterates | projectors
    const int64_t nbuf = 5 + 24 + SL_MAXSL;
    TLSQ_TRY(ws_get(h, WS_SL_BUF, (size_t)nbuf * nn * 8, &buf));
    double* base = (double*)buf;
    b->K = base;
    b->K2 = base + nn;
    b->J = base + 2 * nn;
    b->J2 = base + 3 * nn;
    b->E = base + 4 * nn;
    b->X = base + 5 * nn;
    b->W1 = b->X + 8 * nn;
    b->W2 = b->W1 + 8 * nn;
    b->PJ = b->W2 + 8 * nn;
    return TLSQ_OK;
}

// Step 2: the spectrum of the symmetric positive semi-definite K (N x N, in b.K; b.K2 receives its square) cut into slices of at
// most `want` eigenvalues (SLICE_TARGET overrides); a slice of more than maxk declines.  Projectors in b.PJ, the plan in *pl.
// *ok = false: a guard declined.
int slice_spectrum(Handle* h, int64_t N, const SliceBufs& b, double lam_hi, int n_out, double val_out, double bulk_hi, int want,
                   int maxk, SlicePlan* pl, bool* ok) {
    *ok = false;
    const bool dbg = dev_get(DEV_DEBUG) != nullptr;
    const int64_t nn = N * N;
    const int target = [&] { const char* e = dev_get(DEV_SLICE_TARGET); const int v = e ? atoi(e) : 0; return v >= 16 ? std::min(v, maxk) : want; }();
    const int maxlev = [] { const char* e = dev_get(DEV_SLICE_LEVELS); const int v = e ? atoi(e) : 6; return v >= 1 && v <= 8 ? v : 6; }();
    const double l0 = [] { const char* e = dev_get(DEV_SLICE_L0); const double v = e ? atof(e) : 1e-4; return v > 0.0 && v < 0.5 ? v : 1e-4; }();
    double *K = b.K, *K2 = b.K2, *X = b.X, *W1 = b.W1, *W2 = b.W2, *PJ = b.PJ;
    TLSQ_TRY(small_mm_batched(h, K, nn, K, nn, K2, nn, N, 1, 1.0, 0.0, true, nullptr, 0, 0.0));
    double st3[3], st3b[3];
    TLSQ_TRY(matfun_stats(h, K, N, st3));   // trace, row-sum norm (host round trip)
    TLSQ_TRY(matfun_stats(h, K2, N, st3b));
    const double trK = st3[1], trK2 = st3b[1];
    // (an upper bound of the eigenvalues: the caller's with a little room, or Gershgorin's, or the Frobenius norm)
    double hi = std::min(st3[2], std::sqrt(std::max(trK2, 0.0)) * (1.0 + 1e-12));
    if (lam_hi > 0.0) hi = std::min(hi, lam_hi * (1.0 + 1e-6) + 8.0 * (double)N * 2.3e-16 * st3[2]);
    if (!(trK > 0.0) || !std::isfinite(trK) || !(hi > 0.0) || !std::isfinite(hi) || !std::isfinite(trK2)) return TLSQ_OK;
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
    double* Am = b.J;    // scratch of a level (the basis buffers are free until the slices exist): sum_z alpha_z P_z
    double* Bm = b.J2;   // I + sum_z (beta_z - 1) P_z
    double* Qm = b.E;    // (I - S) / 2
    for (int lev = 0; lev < maxlev; ++lev) {
        // the slices to cut at this level: every one that is still too large (the largest first while projector slots last)
        std::vector<int> pick;
        for (int j = 0; j < (int)sl.size(); ++j)
            if (sl[(size_t)j].cnt > (double)target + 0.5) pick.push_back(j);
        std::sort(pick.begin(), pick.end(), [&](int x, int y) { return sl[(size_t)x].cnt > sl[(size_t)y].cnt; });
        const int room = std::min(16, SL_MAXSL - nslot);
        if ((int)pick.size() > room) pick.resize((size_t)std::max(0, room));
        if (pick.empty()) break;
        std::sort(pick.begin(), pick.end());
        // ONE sign iteration serves every cut of the level: the slices are invariant subspaces of K with mutually orthogonal
        // projectors, so a function of  X_0 = sum_z P_z (K - t_z I) / nrm_z + (I - sum_z P_z)  acts on slice z as the function of
        // (K - t_z I) / nrm_z - each slice scaled by its own width - and leaves everything else at +1.  (Round 6, first build: one
        // iteration per cut, batched - four 512 x 512 iterations at the third level cost what the four levels cost now.)
        CutMap cmap;
        memset(&cmap, 0, sizeof(cmap));
        std::vector<double> t;
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
            cmap.slot[cmap.n] = s.slot;
            cmap.alpha[cmap.n] = 1.0 / nrm;
            cmap.beta1[cmap.n] = -tj / nrm - 1.0;
            ++cmap.n;
            t.push_back(tj);
            cut.push_back(j);
        }
        if (t.empty()) break;
        const int nb = (int)t.size();
        {
            int64_t g = (nn + 255) / 256;
            if (g > 1024) g = 1024;
            hipLaunchKernelGGL(k_slice_start, dim3((unsigned)g), dim3(256), 0, h->stream, (const double*)PJ, (int)N, cmap, Am, Bm);
            TLSQ_HIP(h, hipGetLastError());
        }
        TLSQ_TRY(small_mm_batched(h, Am, nn, K, nn, X, nn, N, 1, 1.0, 0.0, true, Bm, nn, 1.0));   // X_0 = A K + B
        double* out = nullptr;
        int steps = 0;
        TLSQ_TRY(matfun_sign_batched(h, K, K2, N, 1, nullptr, 0.0, lev_l0, X, W1, W2, &out, nullptr, &steps));
        total_steps += steps;
        nsign += 1;
        // below a cut the sign is -1: Q = (I - S) / 2 is the sum of the lower children's projectors, P_low_z = P_z Q
        TLSQ_TRY(matfun_axpbi(h, out, Qm, N, -0.5, 0.5));
        std::vector<int32_t> idx((size_t)nb);
        for (int q = 0; q < nb; ++q) idx[(size_t)q] = sl[(size_t)cut[(size_t)q]].slot;
        double* kids = PJ + (int64_t)nslot * nn;   // the lower children: consecutive new slots
        TLSQ_TRY(small_mm_batched_idx(h, PJ, idx.data(), nb, Qm, kids, N));
        TLSQ_TRY(slice_stats_batched(h, kids, K, K2, N, nb, b.sstats));
        TLSQ_HIP(h, hipMemcpyAsync(h->pinned, b.sstats, (size_t)nb * 32, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        std::vector<double> hs((size_t)nb * 4);
        memcpy(hs.data(), h->pinned, (size_t)nb * 32);
        ChildMap cm;
        memset(&cm, 0, sizeof(cm));
        double soft = 0.0;
        for (int q = 0; q < nb; ++q) {
            const double tr = hs[(size_t)q * 4], ip = hs[(size_t)q * 4 + 1], fr = hs[(size_t)q * 4 + 2], ip2 = hs[(size_t)q * 4 + 3];
            if (!std::isfinite(tr) || !std::isfinite(ip) || !std::isfinite(fr) || !std::isfinite(ip2)) return TLSQ_OK;
            // ||P||_F^2 = trace(P) for a projector; what is missing are eigenvalues the schedule did not carry to +-1
            if (dbg)
                fprintf(stderr, "  slicer: level %d, slice [%.4e, %.4e] of %.1f cut at %.6e: %.2f below, trace - ||P||_F^2 = %.3e (%d steps)\n", lev,
                        sl[(size_t)cut[(size_t)q]].lo, sl[(size_t)cut[(size_t)q]].hi, sl[(size_t)cut[(size_t)q]].cnt, t[(size_t)q], tr, tr - fr, steps);
            soft += std::fabs(tr - fr);
            cm.parent[q] = sl[(size_t)cut[(size_t)q]].slot;
            cm.low[q] = nslot + q;
        }
        if (soft > 0.05 * (double)N) return TLSQ_OK;
        cm.n = nb;
        {
            int64_t g = (nn + 255) / 256;
            if (g > 512) g = 512;
            hipLaunchKernelGGL(k_slice_children, dim3((unsigned)g, (unsigned)nb), dim3(256), 0, h->stream, PJ, (int)N, cm);
            TLSQ_HIP(h, hipGetLastError());
        }
        for (int q = nb - 1; q >= 0; --q) {
            const int j = cut[(size_t)q];
            const Slice s = sl[(size_t)j];
            const double cnt_lo = hs[(size_t)q * 4], sum_lo = hs[(size_t)q * 4 + 1], sum2_lo = hs[(size_t)q * 4 + 3];
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
    memset(&pl->pm, 0, sizeof(pl->pm));
    pl->groups.clear();
    pl->nsl = 0;
    int col = 0;
    for (size_t j = 0; j < sl.size(); ++j) {
        if (kcol[j] <= 0) continue;
        if (pl->nsl >= SL_MAXSL || kcol[j] > maxk) return TLSQ_OK;
        pl->pm.slot[pl->nsl] = sl[j].slot;
        pl->pm.start[pl->nsl] = col;
        pl->pm.k[pl->nsl] = kcol[j];
        ++pl->nsl;
        pl->groups.push_back({col, kcol[j]});
        col += kcol[j];
    }
    if (col != (int)N || pl->nsl < 2) return TLSQ_OK;   // (nothing was split: the plain route is the same thing)
    if (dbg) {
        fprintf(stderr, "  slicer: N=%lld, %d levels, %d sign matrices (%d polynomial steps), slices:", (long long)N, levels, nsign, total_steps);
        for (auto& g : pl->groups) fprintf(stderr, " %d", g.second);
        fprintf(stderr, "\n");
    }
    *ok = true;
    return TLSQ_OK;
}

// E = Q'Q, then Q <- Q (1.5 I - 0.5 E) while ||E - I||_F > 1e-13.5 (at most `passes` times); *Qc = where the result is (Q or Q2).
// *ok = false: Q was too far from orthogonal to begin with (||E - I||_F > 0.5) or did not get there.
int polish_orthogonal(Handle* h, int64_t N, double* Q, double* Q2, double* E, int passes, double** Qc_out, bool* ok, const char* tag) {
    *ok = false;
    const bool dbg = dev_get(DEV_DEBUG) != nullptr;
    const int64_t nn = N * N;
    double *Jc = Q, *Jn = Q2;
    double dev2 = 0.0, st3[3];
    for (int pass = 0; pass <= passes; ++pass) {
        TLSQ_TRY(gemm_f64(h, true, true, Jc, N, Jc, N, E, N, N, N, N, true));
        TLSQ_TRY(matfun_stats(h, E, N, st3));
        dev2 = st3[0];
        if (dbg) fprintf(stderr, "  slicer: ||%s'%s - I||_F = %.3e\n", tag, tag, std::sqrt(std::max(dev2, 0.0)));
        if (!std::isfinite(dev2) || dev2 > 0.25) return TLSQ_OK;
        if (dev2 <= 1e-27 || pass == passes) break;
        TLSQ_TRY(matfun_axpbi(h, E, E, N, -0.5, 1.5));
        TLSQ_TRY(small_mm_batched(h, Jc, nn, E, nn, Jn, nn, N, 1, 1.0, 0.0, false, nullptr, 0, 0.0));
        std::swap(Jc, Jn);
    }
    if (dev2 > 1e-24) return TLSQ_OK;
    *Qc_out = Jc;
    *ok = true;
    return TLSQ_OK;
}

// Step 3: J0 = the Cholesky factors of the projectors (b.J / b.J2), polished to orthogonality
int slices_to_basis(Handle* h, int64_t N, const SliceBufs& b, const SlicePlan& pl, double** J0, bool* ok) {
    *ok = false;
    TLSQ_HIP(h, hipMemsetAsync(b.failflag, 0, 4, h->stream));
    hipLaunchKernelGGL(k_proj_chol, dim3((unsigned)pl.nsl), dim3(1024), 0, h->stream, (const double*)b.PJ, pl.pm, (int)N, b.J, b.failflag);
    TLSQ_HIP(h, hipGetLastError());
    bool pok = false;
    TLSQ_TRY(polish_orthogonal(h, N, b.J, b.J2, b.E, 4, J0, &pok, "J0"));
    int ff = 0;
    TLSQ_HIP(h, hipMemcpyAsync(h->pinned, b.failflag, 4, hipMemcpyDeviceToHost, h->stream));
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    memcpy(&ff, h->pinned, 4);
    if (ff) {
        if (dev_get(DEV_DEBUG)) fprintf(stderr, "  slicer: a projector is not of its rank (pivot below 1e-3): plain route\n");
        return TLSQ_OK;
    }
    *ok = pok;
    return TLSQ_OK;
}

// ---- the slices' own eigenproblems: Cholesky factors of the diagonal blocks of T = J0' G J0, one workgroup per slice -----------
// C_j C_j' = T_jj (k x k, k <= 96, in LDS; the eigenvalues of a slice lie within a few per cent of each other: no pivoting, no
// shift) written to the diagonal block of Bd (N x N, zero elsewhere - cleared by the caller).  One-sided Jacobi on the columns of
// Bd (jacobi.hip, the register kernel on the blocks' own row windows) then gives the eigenvectors of every T_jj at once.
// fail[0] != 0: a pivot was not positive.
__global__ __launch_bounds__(1024) void k_chol_blocks(const double* __restrict__ T, int N, ProjMap m, double* __restrict__ Bd,
                                                      int* __restrict__ fail) {
    extern __shared__ __attribute__((aligned(16))) double cb_sm[];
    const int blk = blockIdx.x, k = m.k[blk], c0 = m.start[blk];
    const int LD = k + 1, tid = threadIdx.x;
    double* S = cb_sm;   // S[c * LD + r], lower triangle used
    for (int e = tid; e < k * k; e += 1024) {
        const int r = e % k, c = e / k;
        S[c * LD + r] = 0.5 * (T[(c0 + r) + (int64_t)(c0 + c) * N] + T[(c0 + c) + (int64_t)(c0 + r) * N]);
    }
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        const double piv = S[j * LD + j];
        __syncthreads();   // (everybody has read the pivot before column j is scaled)
        if (!(piv > 0.0)) {
            if (tid == 0) atomicExch(fail, 1);
            return;
        }
        const double inv = 1.0 / sqrt(piv);
        for (int r = j + tid; r < k; r += 1024) S[j * LD + r] *= inv;
        __syncthreads();
        // trailing update: S[c][r] -= S[j][r] S[j][c] for j < c <= r < k
        const int nc = k - j - 1;
        for (int e = tid; e < nc * nc; e += 1024) {
            const int c = j + 1 + e / nc, r = j + 1 + e % nc;
            if (r >= c) S[c * LD + r] -= S[j * LD + r] * S[j * LD + c];
        }
        __syncthreads();
    }
    for (int e = tid; e < k * k; e += 1024) {
        const int r = e % k, c = e / k;
        Bd[(c0 + r) + (int64_t)(c0 + c) * N] = r >= c ? S[c * LD + r] : 0.0;
    }
}

// X (N x N) from T = V'GV: X[i, j] = tan of the Jacobi angle of the pair (i, j) - to first order the rotation that removes
// T[i, j] - antisymmetric; zero where |T[i, j]| <= floor_abs (the rounding level of T itself: pairs of equal eigenvalues - the
// deflated cluster - would otherwise be "rotated" by their noise) or below the relative rotation threshold.  V <- V (I + X).
// Pairs inside columns [c0, c1) - the caller's known cluster of equal eigenvalues, where every orthonormal basis is as good as any
// other - and pairs whose tangent would exceed 0.05 (not a first-order correction any more) are left alone.
__global__ __launch_bounds__(256) void k_refine_x(const double* __restrict__ T, int N, double tol, double floor_abs, int c0, int c1,
                                                  double* __restrict__ X) {
    const int64_t total = (int64_t)N * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e % N), j = (int)(e / N);
        double x = 0.0;
        if (i != j && !(i >= c0 && i < c1 && j >= c0 && j < c1)) {
            const double hii = T[i + (int64_t)i * N], hjj = T[j + (int64_t)j * N];
            const double hij = 0.5 * (T[e] + T[j + (int64_t)i * N]);
            if (fabs(hij) > floor_abs && hij * hij > tol * tol * fabs(hii * hjj)) {
                const double d = hjj - hii;
                x = (d >= 0.0 ? 2.0 : -2.0) * hij / (fabs(d) + sqrt(d * d + 4.0 * hij * hij));
                if (fabs(x) > 0.05) x = 0.0;
            }
        }
        X[e] = x;
    }
}

// lam[i] = T[i, i]; out[0] = max over i != j (not both in [c0, c1)) of |T[i, j]| as bits (atomicMax on non-negative doubles)
__global__ __launch_bounds__(256) void k_offdiag_max(const double* __restrict__ T, int N, int c0, int c1, double* __restrict__ lam,
                                                     unsigned long long* __restrict__ out) {
    const int64_t total = (int64_t)N * N;
    double m = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e % N), j = (int)(e / N);
        const double v = T[e];
        if (i == j) lam[i] = v;
        else if (!(i >= c0 && i < c1 && j >= c0 && j < c1)) m = fabs(v) > m ? fabs(v) : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

}   // namespace

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
    SliceBufs b;
    TLSQ_TRY(slice_bufs(h, N, &b));
    void* scal;
    TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &scal));
    double* cstats = reinterpret_cast<double*>(reinterpret_cast<char*>(scal) + 192);    // cholesky: max diagonal, delta

    // 1. the Cholesky factor and K = L'L
    TLSQ_TRY(cholesky_shifted(h, G, ldG, N, B, V /* scratch */, cstats));
    TLSQ_TRY(gemm_f64(h, true, true, B, N, B, N, b.K, N, N, N, N, true));
    // 2. slices, 3. their basis
    SlicePlan pl;
    bool ok = false;
    TLSQ_TRY(slice_spectrum(h, N, b, lam_hi, n_out, val_out, bulk_hi, 80, SL_MAXK, &pl, &ok));
    if (!ok) return TLSQ_OK;
    double* J0 = nullptr;
    TLSQ_TRY(slices_to_basis(h, N, b, pl, &J0, &ok));
    if (!ok) return TLSQ_OK;
    // B0 = L J0 into K's place, then back into the caller's B (L is not needed any more)
    TLSQ_TRY(small_mm_batched(h, B, nn, J0, nn, b.K, nn, N, 1, 1.0, 0.0, false, nullptr, 0, 0.0));
    TLSQ_HIP(h, hipMemcpyAsync(B, b.K, (size_t)nn * 8, hipMemcpyDeviceToDevice, h->stream));

    // 4. + 5. sweeps inside the slices, then over all pairs
    int64_t sw[2] = {0, 0};
    const int stj = jacobi_factor_grouped_f64(h, B, N, pl.groups, V, sig_dev, (double)N * 2.220446049250313e-16, sw, false);
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

// The normwise form, on G itself (no Cholesky factor, no sweeps over all pairs): slices of G, their basis J0, the slices' own
// eigenproblems T_jj = J0_j' G J0_j by one-sided Jacobi on their Cholesky factors (all slices side by side, the register kernel
// on windows of <= 96 rows), V = J0 blockdiag(W_j); what the
// projectors left between the slices is removed to first order from T = V'GV (V <- V (I + X), X the pairwise Jacobi tangents,
// polished back to orthogonality) - twice at most - and the result is CERTIFIED: max_{i != j} |(V'GV)_ij| <= 8 N eps lam_hi, i.e.
// V diagonalises G to the accuracy of a backward-stable eigensolver; lam_dev = diag(V'GV).  (Pairs inside the caller's cluster of
// n_out equal eigenvalues are exempt: its span is what matters, the caller replaces the basis by its own.)  That is an absolute statement
// (eps ||G||): right for the deflated panel of the returned `s` (solver.hip), whose spectrum is one cluster and a flat bulk a
// bounded factor below it - the caller's guards - and not a replacement for the factor route on graded spectra.
// V: N x N result; lam_dev: N eigenvalues (unsorted).  *used = false: a guard or the certificate declined.
int symeig_sliced_normwise_f64(Handle* h, const double* G, int64_t N, double* V, double* lam_dev, int64_t* sweeps_out, double lam_hi,
                               int n_out, double val_out, double bulk_hi, bool* used) {
    *used = false;
    if (sweeps_out) *sweeps_out = 0;
    if (!symeig_sliced_ok(N) || dev_is(DEV_SLICE_NORMWISE, '0') || !(lam_hi > 0.0)) return TLSQ_OK;
    const bool dbg = dev_get(DEV_DEBUG) != nullptr;
    const int64_t nn = N * N;
    const double eps = 2.220446049250313e-16;
    SliceBufs b;
    TLSQ_TRY(slice_bufs(h, N, &b));
    void* scal;
    TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &scal));
    unsigned long long* omax = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(scal) + 3592);
    TLSQ_HIP(h, hipMemcpyAsync(b.K, G, (size_t)nn * 8, hipMemcpyDeviceToDevice, h->stream));
    SlicePlan pl;
    bool ok = false;
    TLSQ_TRY(slice_spectrum(h, N, b, lam_hi, n_out, val_out, bulk_hi, 96, 96, &pl, &ok));
    if (!ok) return TLSQ_OK;
    double* J0 = nullptr;
    TLSQ_TRY(slices_to_basis(h, N, b, pl, &J0, &ok));
    if (!ok) return TLSQ_OK;
    // T = J0' (G J0)
    double *GJ = b.X, *T = b.X + nn, *Wb = b.X + 2 * nn, *Xr = b.X + 3 * nn, *Vn = b.W1, *Vn2 = b.W1 + nn;
    TLSQ_TRY(small_mm_batched(h, b.K, nn, J0, nn, GJ, nn, N, 1, 1.0, 0.0, false, nullptr, 0, 0.0));
    TLSQ_TRY(gemm_f64(h, true, true, J0, N, GJ, N, T, N, N, N, N, false));
    // the slices' eigenproblems, side by side: W = blockdiag(eigenvector matrices of the T_jj)
    int maxk = 0;
    for (auto& g : pl.groups) maxk = std::max(maxk, g.second);
    int64_t swb[2] = {0, 0};
    if (maxk <= 96 && pl.groups.size() <= 32 && !dev_is(DEV_SLICE_MIDJ, '0')) {
        // One workgroup per slice, the whole k x k problem in LDS, every sweep of it in ONE launch (jacobi.hip,
        // k_jacobi_mid_blocks: one-sided Jacobi on T_jj itself - normwise accurate, which is all the certificate below asks of
        // this stage).  The register kernel on the blocks' Cholesky factors (below) needed 9 sweeps x 5 launches of 28 us for
        // eight slices: 1.3 ms against 0.6.
        const int stj = jacobi_mid_blocks_f64(h, T, N, pl.groups, Wb, lam_dev /* scratch */, &swb[0]);
        if (stj == TLSQ_ERR_NOCONV) return TLSQ_OK;
        if (stj < 0) return stj;
    } else {
        // Cholesky factors of the diagonal blocks, one-sided Jacobi on them (the register kernel on the blocks' own row windows),
        // W = the normalised rotated columns (block diagonal)
        const size_t lds = (size_t)maxk * (maxk + 1) * 8;
        double* Bd = Xr;   // (free until the refinement)
        TLSQ_HIP(h, hipMemsetAsync(Bd, 0, (size_t)nn * 8, h->stream));
        TLSQ_HIP(h, hipMemsetAsync(b.failflag, 0, 4, h->stream));
        TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_blocks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_chol_blocks, dim3((unsigned)pl.nsl), dim3(1024), lds, h->stream, (const double*)T, (int)N, pl.pm, Bd, b.failflag);
        TLSQ_HIP(h, hipGetLastError());
        const int stj = jacobi_factor_grouped_f64(h, Bd, N, pl.groups, Wb, lam_dev /* scratch: the column norms */, 0.0, swb, true);
        if (stj == TLSQ_ERR_NOCONV) return TLSQ_OK;
        if (stj < 0) return stj;
        int ff = 0;
        TLSQ_HIP(h, hipMemcpyAsync(h->pinned, b.failflag, 4, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        memcpy(&ff, h->pinned, 4);
        if (ff) return TLSQ_OK;
    }
    // V = J0 W
    TLSQ_TRY(small_mm_batched(h, J0, nn, Wb, nn, Vn, nn, N, 1, 1.0, 0.0, false, nullptr, 0, 0.0));
    if (dbg) {   // diagnosis: orthogonality of W and of V = J0 W as they come out of the slices' eigenproblems
        double st3[3];
        TLSQ_TRY(gemm_f64(h, true, true, Wb, N, Wb, N, b.E, N, N, N, N, true));
        TLSQ_TRY(matfun_stats(h, b.E, N, st3));
        fprintf(stderr, "  slicer (normwise): ||W'W - I||_F = %.3e", std::sqrt(std::max(st3[0], 0.0)));
        TLSQ_TRY(gemm_f64(h, true, true, Vn, N, Vn, N, b.E, N, N, N, N, true));
        TLSQ_TRY(matfun_stats(h, b.E, N, st3));
        fprintf(stderr, ", ||V'V - I||_F = %.3e, %lld sweeps inside the slices\n", std::sqrt(std::max(st3[0], 0.0)), (long long)swb[0]);
    }
    double* Vc = Vn;
    double* Vo = Vn2;
    const double cert = 8.0 * (double)N * eps * lam_hi;
    // the caller's cluster (n_out eigenvalues at val_out): the columns of the top slice when it is exactly that; any orthonormal
    // basis of its span serves (the caller puts its own vectors there), so pairs inside it are neither refined nor certified
    int cl0 = 0, cl1 = 0;
    if (n_out > 0 && !pl.groups.empty() && pl.groups.back().second == n_out) {
        cl0 = pl.groups.back().first;
        cl1 = cl0 + n_out;
    }
    double offmax = 0.0;
    bool certified = false;
    for (int pass = 0; pass < 3 && !certified; ++pass) {
        // T = V'(G V), its largest off-diagonal entry and its diagonal
        TLSQ_TRY(small_mm_batched(h, b.K, nn, Vc, nn, GJ, nn, N, 1, 1.0, 0.0, false, nullptr, 0, 0.0));
        TLSQ_TRY(gemm_f64(h, true, true, Vc, N, GJ, N, T, N, N, N, N, false));
        TLSQ_HIP(h, hipMemsetAsync(omax, 0, 8, h->stream));
        hipLaunchKernelGGL(k_offdiag_max, dim3(256), dim3(256), 0, h->stream, (const double*)T, (int)N, cl0, cl1, lam_dev, omax);
        TLSQ_HIP(h, hipGetLastError());
        TLSQ_HIP(h, hipMemcpyAsync(h->pinned, omax, 8, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        memcpy(&offmax, h->pinned, 8);
        if (dbg) fprintf(stderr, "  slicer (normwise): max off-diagonal of V'GV = %.3e (certificate %.3e)\n", offmax, cert);
        if (!std::isfinite(offmax)) return TLSQ_OK;
        if (offmax <= cert) {
            certified = true;
            break;
        }
        if (pass == 2) break;
        // V <- polish(V (I + X))
        {
            int64_t g = (nn + 255) / 256;
            if (g > 1024) g = 1024;
            hipLaunchKernelGGL(k_refine_x, dim3((unsigned)g), dim3(256), 0, h->stream, (const double*)T, (int)N, 2.0 * eps, 0.125 * cert, cl0, cl1, Xr);
            TLSQ_HIP(h, hipGetLastError());
        }
        if (dbg) {
            double st3[3];
            TLSQ_TRY(matfun_stats(h, Xr, N, st3));
            fprintf(stderr, "  slicer (normwise): ||X||_F = %.3e, row-sum norm %.3e\n", std::sqrt(std::max(st3[0] - (double)N, 0.0)), st3[2]);
        }
        TLSQ_TRY(small_mm_batched(h, Vc, nn, Xr, nn, Vo, nn, N, 1, 1.0, 0.0, false, Vc, nn, 1.0));
        double* Vp = nullptr;
        bool pok = false;
        TLSQ_TRY(polish_orthogonal(h, N, Vo, Vc, b.E, 3, &Vp, &pok, "V"));
        if (!pok) return TLSQ_OK;
        if (Vp == Vo) std::swap(Vc, Vo);   // (the polished matrix is in Vo: it becomes the current one; otherwise it is in Vc already)
    }
    if (!certified) return TLSQ_OK;
    TLSQ_HIP(h, hipMemcpyAsync(V, Vc, (size_t)nn * 8, hipMemcpyDeviceToDevice, h->stream));
    if (sweeps_out) *sweeps_out = swb[0];
    *used = true;
    return TLSQ_OK;
}

}   // namespace tlsq
