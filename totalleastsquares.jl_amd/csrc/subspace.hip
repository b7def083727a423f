// Warm-started subspace iteration for the dominant eigenpairs of the N x N Gram matrix G = Z'Z.
//
// After the first ALM iteration the singular-value threshold step of rpca (/root/reference/src/robustPCA.jl:
// 193-213) only needs (i) the singular pairs with sigma_i >= 1/mu and (ii) their exact count `svp`.  The
// dominant right singular subspace moves slowly between ALM iterations, so instead of a full N x N
// eigen-decomposition we iterate on a block X (N x p, p = svp_prev + pad) taken from the previous iteration:
//     Q = orth(G X)   (CGS2, LDS resident)        H = Q'GQ  (p x p)     H = S Theta S'  (Jacobi, one WG)
//     X <- Q S ;  residuals ||G x_i - theta_i x_i||
// until the wanted pairs are converged to ~1e-13 ||G||, then certify the count with a Lanczos bound on
// lambda_max of the deflated matrix G - X_r Theta_r X_r'.  Any failure (slow convergence, rank growth
// beyond the block, ambiguous certificate) makes the caller fall back to the full Jacobi solver.
#include "common.hpp"

namespace tlsq {

constexpr int SS_THREADS = 1024;

// wave all-reduce without the LDS crossbar (see jacobi.hip): DPP inside each row of 16 lanes, then the four row totals
template <int CTRL>
__device__ __forceinline__ double ss_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ss_lane(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double ss_wsum(double v) {
    v += ss_dpp<0xB1>(v);
    v += ss_dpp<0x4E>(v);
    v += ss_dpp<0x141>(v);
    v += ss_dpp<0x140>(v);
    return (ss_lane(v, 0) + ss_lane(v, 16)) + (ss_lane(v, 32) + ss_lane(v, 48));
}

// In-place orthonormalisation of the p columns of Y (N x p, ld N) by classical Gram-Schmidt with one
// re-orthogonalisation pass (CGS2).  Whole panel in LDS.  status[0] = min over columns of
// ||y_j after projection|| / ||y_j before|| (tiny => numerically dependent column).
template <bool IN_LDS>
__global__ __launch_bounds__(SS_THREADS) void k_cgs2(double* __restrict__ Y, int N, int p,
                                                     double* __restrict__ status, int combine) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    // IN_LDS: the whole panel lives in LDS.  Otherwise it stays in global memory (L2-resident, a single
    // workgroup reads back its own stores after a barrier) and LDS only holds the dot products.
    double* sY = IN_LDS ? sm : Y;
    double* sd = IN_LDS ? sm + (size_t)p * N : sm;    // p dots
    double* red = sd + p;                              // 16
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = SS_THREADS / 64;
    if (IN_LDS) {
        for (int e = tid; e < N * p; e += SS_THREADS) sY[e] = Y[e];
    }
    __syncthreads();
    double minratio = 1.0;
    for (int j = 0; j < p; ++j) {
        double* yj = sY + (size_t)j * N;
        // original norm
        double nn = 0.0;
        for (int r = tid; r < N; r += SS_THREADS) nn += yj[r] * yj[r];
        nn = ss_wsum(nn);
        if (lane == 0) red[w] = nn;
        __syncthreads();
        double n0 = 0.0;
        for (int k = 0; k < nw; ++k) n0 += red[k];
        __syncthreads();
        double n1s = n0;
        for (int pass = 0; pass < 2 && j > 0; ++pass) {
            for (int i = w; i < j; i += nw) {  // d_i = q_i . y_j
                const double* qi = sY + (size_t)i * N;
                double d = 0.0;
                for (int r = lane; r < N; r += 64) d += qi[r] * yj[r];
                d = ss_wsum(d);
                if (lane == 0) sd[i] = d;
            }
            __syncthreads();
            double n1 = 0.0;
            for (int r = tid; r < N; r += SS_THREADS) {  // y_j -= Q_{<j} d
                double acc = yj[r];
                for (int i = 0; i < j; ++i) acc -= sd[i] * sY[(size_t)i * N + r];
                yj[r] = acc;
                n1 += acc * acc;
            }
            n1 = ss_wsum(n1);
            if (lane == 0) red[w] = n1;
            if (!IN_LDS) __threadfence_block();
            __syncthreads();
            const double before = n1s;
            n1s = 0.0;
            for (int k = 0; k < nw; ++k) n1s += red[k];
            __syncthreads();
            // "twice is enough": a second pass only when the first one removed a sizeable part of the column
            if (n1s > 0.5 * before) break;
        }
        if (j == 0) {
            n1s = n0;
        }
        const double ratio = n0 > 0.0 ? sqrt(n1s / n0) : 0.0;
        minratio = ratio < minratio ? ratio : minratio;
        const double inv = n1s > 0.0 ? 1.0 / sqrt(n1s) : 0.0;
        for (int r = tid; r < N; r += SS_THREADS) yj[r] *= inv;
        if (!IN_LDS) __threadfence_block();
        __syncthreads();
    }
    if (IN_LDS) {
        for (int e = tid; e < N * p; e += SS_THREADS) Y[e] = sY[e];
    }
    // (combine: a later block of a blocked orthonormalisation - keep the smallest ratio of all blocks)
    if (tid == 0) status[0] = (combine && status[0] < minratio) ? status[0] : minratio;
}

// res[i] = || GX[:,i] - theta[i] * X[:,i] ||_2   (one wave per column)
__global__ __launch_bounds__(256) void k_ritz_resid(const double* __restrict__ GX,
                                                    const double* __restrict__ X,
                                                    const double* __restrict__ theta, int N, int p,
                                                    double* __restrict__ res) {
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= p) return;
    const double th = theta[col];
    double s = 0.0;
    for (int r = lane; r < N; r += 64) {
        const double v = GX[(size_t)col * N + r] - th * X[(size_t)col * N + r];
        s += v * v;
    }
    s = ss_wsum(s);
    if (lane == 0) res[col] = sqrt(s);
}

// theta[i] = H-eigenvalue i recovered as the Rayleigh quotient x_i . (G x_i)  (one wave per column)
__global__ __launch_bounds__(256) void k_rayleigh(const double* __restrict__ GX, const double* __restrict__ X,
                                                  int N, int p, double* __restrict__ theta) {
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= p) return;
    double s = 0.0;
    for (int r = lane; r < N; r += 64) s += GX[(size_t)col * N + r] * X[(size_t)col * N + r];
    s = ss_wsum(s);
    if (lane == 0) theta[col] = s;
}

// deterministic pseudo-random fill in (-0.5, 0.5) (integer hash of the element index)
__global__ __launch_bounds__(256) void k_fill_hash(double* __restrict__ X, int64_t n, unsigned int seed) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned int x = (unsigned int)i * 2654435761u + seed;
        x ^= x >> 16;
        x *= 2246822519u;
        x ^= x >> 13;
        x *= 3266489917u;
        x ^= x >> 16;
        X[i] = ((double)(x & 0xFFFFFF) + 0.5) / 16777216.0 - 0.5;
    }
}

__device__ __forceinline__ unsigned int hash_u32(unsigned int x) {
    x ^= x >> 16;
    x *= 2246822519u;
    x ^= x >> 13;
    x *= 3266489917u;
    x ^= x >> 16;
    return x;
}

// deterministic standard-normal fill (Box-Muller on two integer hashes of the element index)
__global__ __launch_bounds__(256) void k_fill_gauss(double* __restrict__ X, int64_t n, unsigned int seed) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned int a = hash_u32((unsigned int)i * 2654435761u + seed);
        const unsigned int b = hash_u32((unsigned int)i * 40503u + (seed ^ 0x68E31DA4u) + 0x9E3779B9u);
        const double u1 = ((double)a + 1.0) / 4294967297.0;   // (0,1)
        const double u2 = ((double)b + 0.5) / 4294967296.0;
        X[i] = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    }
}

int launch_fill_gauss(Handle* h, double* X, int64_t n, unsigned int seed) {
    int64_t g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_fill_gauss, dim3((int)g), dim3(256), 0, h->stream, X, n, seed);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// out[c] = sum_r B[r + c*ld]^2   (one workgroup per column, fixed reduction order)
__global__ __launch_bounds__(256) void k_colsumsq(const double* __restrict__ B, int64_t rows, int64_t ld,
                                                  double* __restrict__ out) {
    __shared__ double sw[4];
    const double* __restrict__ col = B + (int64_t)blockIdx.x * ld;
    double s = 0.0;
    for (int64_t r = threadIdx.x; r < rows; r += 256) s += col[r] * col[r];
    s = ss_wsum(s);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

int launch_colsumsq(Handle* h, const double* B, int64_t rows, int64_t ld, int64_t cols, double* out) {
    if (cols <= 0) return TLSQ_OK;
    hipLaunchKernelGGL(k_colsumsq, dim3((int)cols), dim3(256), 0, h->stream, B, rows, ld, out);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

int launch_fill_hash(Handle* h, double* X, int64_t n, unsigned int seed) {
    int64_t g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_fill_hash, dim3((int)g), dim3(256), 0, h->stream, X, n, seed);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// ---- skinny products of the subspace iteration (N x N symmetric G, N x p panels, p <= 64) -------------------
// These are a few MFLOP each: they are latency-bound, so each one is a single small launch (no split-K slabs).

// ---- CholeskyQR2 for panels of up to 32 columns ------------------------------------------------------------
// orth(Y) in four small launches instead of the column-sequential CGS2 (which costs ~4 us per column):
//     W = Y'Y (k_panel_tn)  ->  L L' = D^-1/2 W D^-1/2,  Q1 = Y D^-1/2 L^-T (k_chol_trsm)
// twice.  The first factorisation checks its pivots (= squared distance of each unit-norm column from the span
// of the previous ones): a pivot below 1e-5 means the panel is too ill-conditioned for this route; status[1] is
// then set, every later CholeskyQR launch returns at once with Y untouched, and the host (which reads the status
// with the Ritz residuals anyway) repeats the step with k_cgs2.  status[0] = sqrt(min pivot), the same "min
// ratio" CGS2 reports.
constexpr int CQ_PMAX = 32;
constexpr int CQ_LD = 33;
// smallest pivot of the first factorisation above which one pass is enough: the columns are then so close to
// orthogonal (kappa^2 <~ p / 0.25) that Q1'Q1 - I is O(100 eps) already; the second pass degenerates to a copy
constexpr double CQ_ONEPASS = 0.25;

// One wave: sL <- Cholesky factor (lower) of D^-1/2 W D^-1/2, sD <- D^-1/2.  Returns 0, or 1 when a pivot is
// below `thresh` (or W is not finite / has a non-positive diagonal).  *minpiv_out = smallest pivot.
__device__ __forceinline__ int chol_scaled_wave(const double* __restrict__ W, int p, double* sL, double* sD,
                                                double thresh, double* minpiv_out) {
    const int lane = threadIdx.x;
    const bool act = lane < p;
    for (int e = lane; e < p * p; e += 64) sL[(e % p) + (e / p) * CQ_LD] = W[e];
    __syncthreads();
    const double dii = act ? sL[lane + lane * CQ_LD] : 1.0;
    const bool bad = !(dii > 0.0) || !(dii < 1.0e300);
    const double sc = bad ? 0.0 : 1.0 / sqrt(dii);
    if (act) sD[lane] = sc;
    __syncthreads();
    if (act)
        for (int j = 0; j < p; ++j) sL[lane + j * CQ_LD] *= sc * sD[j];
    int fail = __any(bad && act) ? 1 : 0;
    __syncthreads();
    double minpiv = 1.0;
    for (int k = 0; k < p && !fail; ++k) {
        double sv = 0.0;
        if (act && lane >= k) {
            double s0 = sL[lane + k * CQ_LD], s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int m = 0;
            for (; m + 3 < k; m += 4) {
                const double a0 = sL[lane + m * CQ_LD], b0 = sL[k + m * CQ_LD];
                const double a1 = sL[lane + (m + 1) * CQ_LD], b1 = sL[k + (m + 1) * CQ_LD];
                const double a2 = sL[lane + (m + 2) * CQ_LD], b2 = sL[k + (m + 2) * CQ_LD];
                const double a3 = sL[lane + (m + 3) * CQ_LD], b3 = sL[k + (m + 3) * CQ_LD];
                s0 -= a0 * b0;
                s1 -= a1 * b1;
                s2 -= a2 * b2;
                s3 -= a3 * b3;
            }
            for (; m < k; ++m) s0 -= sL[lane + m * CQ_LD] * sL[k + m * CQ_LD];
            s0 += s2;
            s1 += s3;
            sv = s0 + s1;
        }
        const double piv = __shfl(sv, k, 64);
        if (!(piv >= thresh)) {
            fail = 1;
            break;
        }
        minpiv = piv < minpiv ? piv : minpiv;
        const double rl = rsqrt(piv);
        __syncthreads();
        if (act && lane >= k) sL[lane + k * CQ_LD] = sv * rl;   // diagonal: piv / sqrt(piv) = sqrt(piv)
        __syncthreads();
    }
    *minpiv_out = minpiv;
    return fail;
}

// Qout = Yin * D^-1/2 * L^-T with L L' = D^-1/2 W D^-1/2 (W = Yin'Yin from k_panel_tn): every workgroup (one
// wave) factors the small matrix itself - same code, same bits - and then runs the forward substitution along
// its 64 panel rows (one thread per row).
__global__ __launch_bounds__(64) void k_chol_trsm(const double* __restrict__ Yin, const double* __restrict__ W,
                                                  double* __restrict__ status, double* __restrict__ Qout, int N,
                                                  int p, int pass, int first_block) {
    __shared__ double sL[CQ_PMAX * CQ_LD];
    __shared__ double sD[CQ_PMAX], sI[CQ_PMAX];
    // status[1] is sticky across the column blocks of one orthonormalisation: once a block has failed, every later
    // launch returns at once (the first block's first pass resets it)
    if (!(pass == 1 && first_block) && status[1] != 0.0) return;
    const int lane = threadIdx.x;
    const int r = blockIdx.x * 64 + lane;
    // the row of the panel first: the loads overlap the factorisation
    double yv[CQ_PMAX];
#pragma unroll
    for (int j = 0; j < CQ_PMAX; ++j) yv[j] = (j < p && r < N) ? Yin[r + (size_t)j * N] : 0.0;
    if (pass == 2 && status[2] >= CQ_ONEPASS) {   // one pass was enough: hand the panel over as it is
        if (r < N) {
#pragma unroll
            for (int j = 0; j < CQ_PMAX; ++j)
                if (j < p) Qout[r + (size_t)j * N] = yv[j];
        }
        return;
    }
    double minpiv;
    const int fail = chol_scaled_wave(W, p, sL, sD, pass == 1 ? 1.0e-5 : 1.0e-300, &minpiv);
    if (pass == 1 && blockIdx.x == 0 && lane == 0) {
        const double ratio = fail ? 0.0 : sqrt(minpiv);
        status[0] = first_block ? ratio : (ratio < status[0] ? ratio : status[0]);
        if (first_block || fail) status[1] = fail ? 1.0 : 0.0;
        status[2] = fail ? 0.0 : minpiv;
    }
    if (fail) return;
    if (lane < p) sI[lane] = 1.0 / sL[lane + lane * CQ_LD];
    __syncthreads();
    if (r >= N) return;
    double q[CQ_PMAX];
#pragma unroll
    for (int j = 0; j < CQ_PMAX; ++j) {
        if (j < p) {
            double sv = yv[j] * sD[j];
#pragma unroll
            for (int i = 0; i < j; ++i) sv -= q[i] * sL[j + i * CQ_LD];
            q[j] = sv * sI[j];
            Qout[r + (size_t)j * N] = q[j];
        }
    }
}

// Y (R x p) = A (R x K) * X (K x p), p small: the tall-skinny product behind Y = G X (G symmetric, R = K = N) and
// behind T = Z V_svp (R = M rows of the data panel).  Workgroup = 64 rows x JC panel columns, 8 waves; lane = row,
// so A[r, c] is one coalesced read per wave and c, the X slice (SK_KB rows x JC columns) is staged in LDS as
// [c][j] and read as broadcasts, and no cross-lane reduction is needed.  The waves split the inner dimension;
// their partial sums meet in LDS and are added in a fixed order (deterministic).  A is read exactly once when
// p <= JC.
constexpr int SK_KB = 512;
constexpr int SK_WAVES = 8;
constexpr int SK_CW = SK_KB / SK_WAVES;   // 64 inner indices per wave and round
template <typename TA, int JC>
__global__ __launch_bounds__(SK_WAVES * 64) void k_skinny_mm(const TA* __restrict__ A, int64_t lda,
                                                             const double* __restrict__ X, int64_t ldx,
                                                             double* __restrict__ Y, int64_t ldy, int64_t R,
                                                             int K, int p) {
    extern __shared__ __attribute__((aligned(16))) double sk_sm[];   // max(SK_KB, SK_WAVES * 64) * JC doubles
    double* sX = sk_sm;
    double* sP = sk_sm;   // reused after the last round
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t r = (int64_t)blockIdx.x * 64 + lane;
    const int j0 = blockIdx.y * JC;
    const int nj = (p - j0 < JC) ? p - j0 : JC;
    const bool rin = r < R;
    double acc[JC];
#pragma unroll
    for (int j = 0; j < JC; ++j) acc[j] = 0.0;
    for (int kb = 0; kb < K; kb += SK_KB) {
        const int kn = (K - kb < SK_KB) ? K - kb : SK_KB;
        __syncthreads();
        for (int e = tid; e < SK_KB * JC; e += SK_WAVES * 64) {
            const int c = e % SK_KB, j = e / SK_KB;
            sX[c * JC + j] = (c < kn && j < nj) ? X[(size_t)(j0 + j) * ldx + kb + c] : 0.0;
        }
        __syncthreads();
        const int cb = w * SK_CW;
        const TA* Ar = A + r + (int64_t)(kb + cb) * lda;
        // 16 reads of A in flight per lane; the next 16 are issued before the current ones are consumed.  The
        // scheduling barriers keep the compiler from hoisting every LDS read of the round (=> spills).
        TA gc[16], gn[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) gc[i] = (rin && cb + i < kn) ? Ar[(int64_t)i * lda] : (TA)0;
#pragma unroll 1
        for (int q = 0; q < SK_CW / 16; ++q) {
            if (q + 1 < SK_CW / 16) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int c = cb + (q + 1) * 16 + i;
                    gn[i] = (rin && c < kn) ? Ar[(int64_t)((q + 1) * 16 + i) * lda] : (TA)0;
                }
            }
            constexpr int IB = 32 / JC;
#pragma unroll
            for (int i0 = 0; i0 < 16; i0 += IB) {
#pragma unroll
                for (int i = i0; i < i0 + IB; ++i) {
                    const double* xs = sX + (cb + q * 16 + i) * JC;
                    const double gv = (double)gc[i];
#pragma unroll
                    for (int j = 0; j < JC; ++j) acc[j] += gv * xs[j];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) gc[i] = gn[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < JC; ++j) sP[(w * JC + j) * 64 + lane] = acc[j];
    __syncthreads();
    for (int o = tid; o < JC * 64; o += SK_WAVES * 64) {
        const int j = o >> 6, l = o & 63;
        double sum = 0.0;
#pragma unroll
        for (int ww = 0; ww < SK_WAVES; ++ww) sum += sP[(ww * JC + j) * 64 + l];
        const int64_t rr = (int64_t)blockIdx.x * 64 + l;
        if (rr < R && j < nj) Y[(size_t)(j0 + j) * ldy + rr] = sum;
    }
}

// GD = scale * (G - Vs * Vg')  (N x N, rank-r deflation for the count certificate; Vs, Vg: N x r, ld N)
__global__ __launch_bounds__(256) void k_deflate(const double* __restrict__ G, int64_t ldG,
                                                 const double* __restrict__ Vs, const double* __restrict__ Vg,
                                                 double* __restrict__ GD, int N, int r, double scale) {
    const int64_t total = (int64_t)N * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e % N), j = (int)(e / N);
        double sv = G[i + (int64_t)j * ldG];
        for (int k = 0; k < r; ++k) sv -= Vs[i + (size_t)k * N] * Vg[j + (size_t)k * N];
        GD[e] = sv * scale;
    }
}

// the same with the deflated columns taken straight from the Ritz block X (N x p) through a short selection / weight list
// passed as kernel arguments: GD = scale (G - sum_k w_k X[:, sel_k] X[:, sel_k]') - no gather pass, no Vs / Vg panels
__global__ __launch_bounds__(256) void k_deflate_sel(const double* __restrict__ G, int64_t ldG, const double* __restrict__ X,
                                                     SelWeights sw, double* __restrict__ GD, int N, int r, double scale) {
    const int64_t total = (int64_t)N * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e % N), j = (int)(e / N);
        double sv = G[i + (int64_t)j * ldG];
        for (int k = 0; k < r; ++k) {
            const double* xk = X + (size_t)sw.sel[k] * N;
            sv -= xk[i] * (sw.w[k] * xk[j]);
        }
        GD[e] = sv * scale;
    }
}

// H (p x p, ld p) = A' * B for N x p panels A, B: one wave per entry
__global__ __launch_bounds__(256) void k_panel_tn(const double* __restrict__ A, const double* __restrict__ B,
                                                  double* __restrict__ H, int N, int p,
                                                  const double* __restrict__ skip_status) {
    if (skip_status && (skip_status[1] != 0.0 || skip_status[2] >= CQ_ONEPASS)) return;   // second CholeskyQR pass not needed
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= p * p) return;
    const int i = e % p, j = e / p;
    const double* a = A + (size_t)i * N;
    const double* b = B + (size_t)j * N;
    double s = 0.0;
    for (int r = lane; r < N; r += 64) s += a[r] * b[r];
    s = ss_wsum(s);
    if (lane == 0) H[i + (size_t)j * p] = s;
}

// C (pa x pb, ld pa) = A' * B for panels A (N x pa), B (N x pb): one wave per entry.  Returns at once when the sticky
// failure flag of a blocked orthonormalisation is set.
__global__ __launch_bounds__(256) void k_panel_tn2(const double* __restrict__ A, int pa, const double* __restrict__ B,
                                                   int pb, double* __restrict__ Cm, int N,
                                                   const double* __restrict__ status) {
    if (status && status[1] != 0.0) return;
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= pa * pb) return;
    const int i = e % pa, j = e / pa;
    const double* a = A + (size_t)i * N;
    const double* b = B + (size_t)j * N;
    double s = 0.0;
    for (int r = lane; r < N; r += 64) s += a[r] * b[r];
    s = ss_wsum(s);
    if (lane == 0) Cm[i + (size_t)j * pa] = s;
}

// B (N x pb) -= A (N x pa) * C (pa x pb, ld pa): one wave per 64 rows x 8 columns (grid.y = column groups), its 8
// columns of C in LDS
__global__ __launch_bounds__(64) void k_panel_sub(const double* __restrict__ A, int pa, const double* __restrict__ Cm,
                                                  double* __restrict__ B, int pb, int N,
                                                  const double* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double sC[];   // pa x 8, [i][q]
    if (status && status[1] != 0.0) return;
    const int j0 = blockIdx.y * 8;
    const int nq = (pb - j0 < 8) ? pb - j0 : 8;
    for (int e = threadIdx.x; e < pa * 8; e += 64) {
        const int i = e >> 3, q = e & 7;
        sC[e] = q < nq ? Cm[i + (size_t)(j0 + q) * pa] : 0.0;
    }
    __syncthreads();
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= N) return;
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
#pragma unroll 4
    for (int i = 0; i < pa; ++i) {
        const double a = A[r + (size_t)i * N];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += a * sC[i * 8 + q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (q < nq) B[r + (size_t)(j0 + q) * N] -= acc[q];
}

// X1 = Q * S and X2 = GQ * S  (N x p panels, S p x p, ld p): thread per output element, S from LDS
__global__ __launch_bounds__(256) void k_panel_rot2(const double* __restrict__ Q, const double* __restrict__ GQ,
                                                    const double* __restrict__ S, double* __restrict__ X1,
                                                    double* __restrict__ X2, int N, int p) {
    extern __shared__ __attribute__((aligned(16))) double sS[];
    for (int e = threadIdx.x; e < p * p; e += 256) sS[e] = S[e];
    __syncthreads();
    const int64_t total = (int64_t)N * p;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e % N), c = (int)(e / N);
        double a1 = 0.0, a2 = 0.0;
        for (int k = 0; k < p; ++k) {
            const double sv = sS[k + c * p];
            a1 += Q[(size_t)k * N + r] * sv;
            a2 += GQ[(size_t)k * N + r] * sv;
        }
        X1[e] = a1;
        X2[e] = a2;
    }
}

template <typename TA>
int launch_skinny_mm(Handle* h, const TA* A, int64_t lda, const double* X, int64_t ldx, double* Y, int64_t ldy,
                     int64_t R, int64_t K, int64_t p) {
    if (p <= 0 || R <= 0) return TLSQ_OK;
    const int64_t gx = (R + 63) / 64;
    if (p <= 8) {
        const size_t lds = (size_t)SK_KB * 8 * 8;
        hipLaunchKernelGGL((k_skinny_mm<TA, 8>), dim3((unsigned)gx, 1), dim3(SK_WAVES * 64), lds, h->stream, A, lda, X,
                           ldx, Y, ldy, R, (int)K, (int)p);
    } else {
        const size_t lds = (size_t)SK_KB * 16 * 8;
        // (per device: a multi-GPU group launches this on every GPU of the process)
        TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_skinny_mm<TA, 16>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_skinny_mm<TA, 16>), dim3((unsigned)gx, (unsigned)((p + 15) / 16)), dim3(SK_WAVES * 64), lds,
                           h->stream, A, lda, X, ldx, Y, ldy, R, (int)K, (int)p);
    }
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
template int launch_skinny_mm<double>(Handle*, const double*, int64_t, const double*, int64_t, double*, int64_t,
                                      int64_t, int64_t, int64_t);
template int launch_skinny_mm<float>(Handle*, const float*, int64_t, const double*, int64_t, double*, int64_t, int64_t,
                                     int64_t, int64_t);

// Y (N x p, ld N) = G X for the SYMMETRIC N x N Gram matrix (ld ldG) and a narrow panel X (N x p, ld N): the product
// of the subspace iteration, a few MFLOP that sit in L2 - latency is everything.  One workgroup per 16 x 16 output
// tile; its four waves split the inner dimension in chunks of 128 and feed v_mfma_f64_16x16x4_f64 straight from
// global memory: the A fragment is G[i, k] = G[k, i] (16 consecutive rows of column k: one 128-byte line), and the
// inner index is permuted (k = chunk + 32 fk + u for k-step u) so that every lane reads 32 CONSECUTIVE entries of its
// column of X with 16-byte loads.  All loads of a chunk are issued before the first MFMA; the four partial tiles meet
// in LDS and are added in a fixed order.
typedef double sm_d4 __attribute__((ext_vector_type(4)));
typedef double sm_d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_symm_mfma(const double* __restrict__ G, int64_t ldG, const double* __restrict__ X,
                                                   double* __restrict__ Y, int N, int p) {
    __shared__ double sR[4 * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int i0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
    const int gi = i0 + fr, gj = j0 + fr;
    const bool iok = gi < N, jok = gj < p;
    // (rows / columns beyond the edge read the last valid one instead of being predicated - a guarded load compiles to a branch
    //  each, 48 of them in front of the MFMAs; what they contribute lands in output entries that are not stored)
    const double* Gr = G + (iok ? gi : N - 1);
    const double* Xc = X + (size_t)(jok ? gj : p - 1) * N;
    sm_d4 acc = sm_d4{0.0, 0.0, 0.0, 0.0}, acc2 = acc;
    const bool vec = (N % 2 == 0);
    for (int c0 = w * 128; c0 < N; c0 += 512) {
        const int kb = c0 + fk * 32;
        double a[32], b[32];
        if (c0 + 128 <= N && vec) {   // (whole chunk for every lane of the wave - the MFMAs below must not sit in a divergent branch: two accumulators, no predicates)
#pragma unroll
            for (int u = 0; u < 32; ++u) a[u] = Gr[(int64_t)(kb + u) * ldG];
#pragma unroll
            for (int u = 0; u < 32; u += 2) {
                const sm_d2 v = *reinterpret_cast<const sm_d2*>(Xc + kb + u);
                b[u] = v[0];
                b[u + 1] = v[1];
            }
#pragma unroll
            for (int u = 0; u < 32; u += 2) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u + 1], b[u + 1], acc2, 0, 0, 0);
            }
            continue;
        }
        if (kb + 32 <= N) {
#pragma unroll
            for (int u = 0; u < 32; ++u) a[u] = Gr[(int64_t)(kb + u) * ldG];
#pragma unroll
            for (int u = 0; u < 32; ++u) b[u] = Xc[kb + u];
        } else {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const bool kok = kb + u < N;
                a[u] = (iok && kok) ? Gr[(int64_t)(kb + u) * ldG] : 0.0;
                b[u] = (jok && kok) ? Xc[kb + u] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) sR[w * 256 + q * 64 + lane] = acc[q] + acc2[q];
    __syncthreads();
    // lane holds column j = lane & 15, rows (lane >> 4) + 4 q of the tile
    const int q = tid >> 6, l = tid & 63;
    const double sum = ((sR[q * 64 + l] + sR[256 + q * 64 + l]) + sR[512 + q * 64 + l]) + sR[768 + q * 64 + l];
    const int row = i0 + (l >> 4) + 4 * q, col = j0 + (l & 15);
    if (row < N && col < p) Y[row + (size_t)col * N] = sum;
}

int launch_symm_skinny(Handle* h, const double* G, int64_t ldG, const double* X, double* Y, int64_t N, int64_t p) {
    if (p <= 0) return TLSQ_OK;
    const bool no_mfma = dev_is(DEV_NO_SYMM_MFMA, '1');
    if (!no_mfma && (reinterpret_cast<uintptr_t>(X) % 16) == 0) {
        hipLaunchKernelGGL(k_symm_mfma, dim3((unsigned)((N + 15) / 16), (unsigned)((p + 15) / 16)), dim3(256), 0, h->stream, G,
                           ldG, X, Y, (int)N, (int)p);
        TLSQ_HIP(h, hipGetLastError());
        return TLSQ_OK;
    }
    // symmetric G: column c of G doubles as row c, so the generic kernel's coalesced A[r, c] reads apply as is.
    // 8 columns per workgroup keeps the grid wide for these latency-bound N x N products.
    if (N >= 1024 && p > 8)   // G no longer sits in L2: re-reading it per 8 columns costs more than the narrower grid
        return launch_skinny_mm<double>(h, G, ldG, X, N, Y, N, N, N, p);
    hipLaunchKernelGGL((k_skinny_mm<double, 8>), dim3((unsigned)((N + 63) / 64), (unsigned)((p + 7) / 8)),
                       dim3(SK_WAVES * 64), (size_t)SK_KB * 8 * 8, h->stream, G, ldG, X, N, Y, N, N, (int)N, (int)p);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}


// ---- fused level-1 power certificate (N <= 1024): ||S^2||_F^2 for the symmetric S = GD, straight to the host ------
// One workgroup (16 waves) per 32 x 32 tile of the lower triangle of S^2: wave w owns MFMA tile (w & 3) and the
// K-quarter (w >> 2); both fragments are "16 consecutive rows of column k" of the symmetric S (one 128-byte line per
// 16 lanes), fed to v_mfma_f64_16x16x4_f64 from global memory (S is 2 MB: L2).  The squares of the tile are summed in a
// fixed order; the partial sum goes to the host-visible mailbox slot of the tile (off-diagonal tiles count twice), and
// the workgroup that arrives last publishes the sequence number - the host adds the partials in tile order, so the
// bound is reproducible bit for bit.
__global__ __launch_bounds__(1024) void k_sq_norm(const double* __restrict__ S, int N, int ntile, double* mailbox,
                                                  unsigned int* ticket, double seq) {
    __shared__ double sR[16 * 256];
    __shared__ double red[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    // tile index -> (ti, tj), tj <= ti
    int t = blockIdx.x;
    int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    const int tj = t - ti * (ti + 1) / 2;
    const int sub = w & 3, kq = w >> 2;
    const int i0 = ti * 32 + (sub & 1) * 16, j0 = tj * 32 + (sub >> 1) * 16;
    const int gi = i0 + fr, gj = j0 + fr;
    const bool iok = gi < N, jok = gj < N;
    const double* Si = S + (iok ? gi : 0);
    const double* Sj = S + (jok ? gj : 0);
    sm_d4 acc = sm_d4{0.0, 0.0, 0.0, 0.0};
    const int kper = ((N + 3) / 4 + 3) / 4 * 4;   // inner indices per K-quarter, a multiple of 4
    const int kbeg = kq * kper, kend = (kbeg + kper < N) ? kbeg + kper : N;
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        double a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + fk;
            const bool kok = k < kend;
            a[u] = (iok && kok) ? Si[(int64_t)k * N] : 0.0;
            b[u] = (jok && kok) ? Sj[(int64_t)k * N] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) sR[w * 256 + q * 64 + lane] = acc[q];
    __syncthreads();
    // 4 sub-tiles x 256 entries; thread (sub2, e): sum of the four K-quarters, squared
    const int sub2 = tid >> 8, e = tid & 255;
    const double v = ((sR[(sub2) * 256 + e] + sR[(4 + sub2) * 256 + e]) + sR[(8 + sub2) * 256 + e]) + sR[(12 + sub2) * 256 + e];
    double sq = v * v;
    sq = ss_wsum(sq);
    if (lane == 0) red[w] = sq;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int q = 0; q < 16; ++q) tot += red[q];
        volatile double* mb = mailbox;
        mb[16 + blockIdx.x] = (ti == tj ? 1.0 : 2.0) * tot;
        __threadfence_system();
        if (atomicAdd(ticket, 1u) == (unsigned int)(ntile - 1)) {
            *ticket = 0u;
            __threadfence_system();
            mb[0] = seq;
        }
    }
}

// S (N x N) <- S / ||S||_F, the squared norm given as nb partial sums (the by-product of the slab reduction that
// produced S): keeps a chain of squarings S <- S^2 away from overflow and underflow
__global__ __launch_bounds__(256) void k_scale_by_norm(double* __restrict__ S, int64_t n, const double* __restrict__ part,
                                                       int nb) {
    __shared__ double red[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) v += part[i];
    v = ss_wsum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double tot = (red[0] + red[1]) + (red[2] + red[3]);
    const double sc = tot > 0.0 ? 1.0 / sqrt(tot) : 0.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) S[e] *= sc;
}

// v = S[:, j*] with j* = argmax_j S[j, j] (one workgroup): for S ~ (G / c)^(2^k) this is the dominant eigenvector of G
// up to (lambda_2 / lambda_1)^(2^k)
__global__ __launch_bounds__(1024) void k_dominant_column(const double* __restrict__ S, int N, double* __restrict__ v) {
    __shared__ double sval[16];
    __shared__ int sidx[16];
    double best = -1.0;
    int bi = 0;
    for (int j = threadIdx.x; j < N; j += 1024) {
        const double d = S[(size_t)j * N + j];
        if (d > best) {
            best = d;
            bi = j;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) {
            best = ob;
            bi = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        sval[threadIdx.x >> 6] = best;
        sidx[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    double b = sval[0];
    int jb = sidx[0];
    for (int k = 1; k < 16; ++k)
        if (sval[k] > b || (sval[k] == b && sidx[k] < jb)) {
            b = sval[k];
            jb = sidx[k];
        }
    for (int i = threadIdx.x; i < N; i += 1024) v[i] = S[(size_t)jb * N + i];
}

int launch_scale_by_norm(Handle* h, double* S, int64_t N, const double* part, int nb) {
    int64_t g = (N * N + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_scale_by_norm, dim3((int)g), dim3(256), 0, h->stream, S, N * N, part, nb);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

int launch_dominant_column(Handle* h, const double* S, int64_t N, double* v) {
    hipLaunchKernelGGL(k_dominant_column, dim3(1), dim3(1024), 0, h->stream, S, (int)N, v);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

int launch_sq_norm(Handle* h, const double* S, int64_t N, double* mailbox_dev, unsigned int* ticket, double seq, int* ntile_out) {
    const int nt = (int)((N + 31) / 32);
    const int ntile = nt * (nt + 1) / 2;
    if (ntile_out) *ntile_out = ntile;
    if (sq_norm_blk_ok(N)) return launch_sq_norm_blk(h, S, N, mailbox_dev, ticket, seq, ntile);   // (matfun.hip: the register-blocked tile kernel)
    hipLaunchKernelGGL(k_sq_norm, dim3((unsigned)ntile), dim3(1024), 0, h->stream, S, (int)N, ntile, mailbox_dev, ticket, seq);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// GD = scale * (G - GD)  (second half of the deflation when the rank-r product came from the MFMA GEMM)
__global__ __launch_bounds__(256) void k_sub_scale(const double* __restrict__ G, int64_t ldG, double* __restrict__ GD, int N,
                                                   double scale) {
    const int64_t total = (int64_t)N * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e % N), j = (int)(e / N);
        GD[e] = (G[i + (int64_t)j * ldG] - GD[e]) * scale;
    }
}

int launch_deflate(Handle* h, const double* G, int64_t ldG, const double* Vs, const double* Vg, double* GD, int64_t N,
                   int64_t r, double scale) {
    int64_t g = (N * N + 255) / 256;
    if (g > 2048) g = 2048;
    if (N >= 1024 && r >= 16) {
        // large mode: 2 N^2 r flop (2.1 GFLOP at N = 4096, r = 64) do not belong on one-FMA-per-two-loads code (1.05 ms):
        // the product on the MFMA GEMM, then one pass  GD = scale (G - GD)
        TLSQ_TRY(gemm_f64(h, false, false, Vg, N, Vs, N, GD, N, N, N, r, false));
        hipLaunchKernelGGL(k_sub_scale, dim3((int)g), dim3(256), 0, h->stream, G, ldG, GD, (int)N, scale);
        TLSQ_HIP(h, hipGetLastError());
        return TLSQ_OK;
    }
    hipLaunchKernelGGL(k_deflate, dim3((int)g), dim3(256), 0, h->stream, G, ldG, Vs, Vg, GD, (int)N, (int)r, scale);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

int launch_deflate_sel(Handle* h, const double* G, int64_t ldG, const double* X, const SelWeights& sw, double* GD, int64_t N,
                       int64_t r, double scale) {
    int64_t g = (N * N + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_deflate_sel, dim3((int)g), dim3(256), 0, h->stream, G, ldG, X, sw, GD, (int)N, (int)r, scale);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

int launch_panel_tn(Handle* h, const double* A, const double* B, double* H, int64_t N, int64_t p,
                    const double* skip_status) {
    hipLaunchKernelGGL(k_panel_tn, dim3((int)((p * p + 3) / 4)), dim3(256), 0, h->stream, A, B, H, (int)N, (int)p,
                       skip_status);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

int launch_panel_rot2(Handle* h, const double* Q, const double* GQ, const double* S, double* X1, double* X2,
                      int64_t N, int64_t p) {
    int64_t g = (N * p + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_panel_rot2, dim3((int)g), dim3(256), (size_t)p * p * 8, h->stream, Q, GQ, S, X1, X2, (int)N,
                       (int)p);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

static bool cgs2_fits_lds(int64_t N, int64_t p) { return (size_t)(p * N + p + 16) * 8 <= 144 * 1024; }

int subspace_max_block(int64_t N) {
    // CGS2 keeps the N x p panel in LDS when it fits and works out of L2 otherwise; the p x p Rayleigh-Ritz
    // problem goes to the single-launch Jacobi up to 64 and to the block solver above
    // blocked orthonormalisation and the block Jacobi solver keep a step at ~2 ms up to ~190 columns - still far
    // below a dense N x N decomposition; small N: at most half of the columns
    // large mode (N > 2048) has no dense solver to hand a high rank to (the TSQR route takes ~1 s per decomposition
    // there): blocks of up to 512 columns - a step then costs tens of milliseconds (512 x 512 block Jacobi for the
    // Rayleigh-Ritz problem), still two orders of magnitude below the alternative
    if (N > 2048) return 512;
    // 1024..2048 columns: a dense decomposition costs 0.1-0.3 s there, a 512-column step ~30 ms
    if (N >= 1024) return (int)std::min<int64_t>(512, N / 3);
    return N >= 384 ? 192 : 96;
}

int launch_cgs2(Handle* h, double* Y, int64_t N, int64_t p, double* status_dev) {
    if (cgs2_fits_lds(N, p)) {
        const size_t lds = (size_t)(p * N + p + 16) * 8;
        TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_cgs2<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_cgs2<true>, dim3(1), dim3(SS_THREADS), lds, h->stream, Y, (int)N, (int)p, status_dev, 0);
    } else {
        const size_t lds = (size_t)(p + 16) * 8;
        hipLaunchKernelGGL(k_cgs2<false>, dim3(1), dim3(SS_THREADS), lds, h->stream, Y, (int)N, (int)p, status_dev, 0);
    }
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// Y <- orth(Y) (N x p).  tmp: N x p panel, W: p x p, status: 3 doubles.
// p <= 32: CholeskyQR2 on the whole panel.  32 < p <= 512: block classical Gram-Schmidt with re-orthogonalisation
// (BCGS2) over blocks of 32 columns - each block is projected twice against the finished ones (two small products
// per projection) and then orthonormalised by CholeskyQR2 - a handful of launches per block instead of the
// column-by-column CGS2 (4 us per column at N = 512, 90 us per column at N = 4096).
// `one_pass` (single-block panels only): the caller expects the smallest pivot of the first factorisation to clear
// CQ_ONEPASS again (it did in the previous step of the same warm block), so only the first pass is queued - in place -
// and the two launches of a second pass that would find nothing to do are not.  The caller checks status[2] of THIS
// step (it arrives with the Ritz values) and repeats the step with both passes when the guess was wrong.
// Yb (N x pb) -= X (X' Yb) for the orthonormal columns X (N x c, c <= 512); W: c * pb doubles of scratch
int launch_project_out(Handle* h, const double* X, int64_t c, double* Yb, int64_t pb, double* W, int64_t N) {
    if (c <= 0 || pb <= 0) return TLSQ_OK;
    if (c > 512) return set_err(h, TLSQ_ERR_UNSUPPORTED, "project_out: %lld columns", (long long)c);
    hipLaunchKernelGGL(k_panel_tn2, dim3((unsigned)((c * pb + 3) / 4)), dim3(256), 0, h->stream, X, (int)c, (const double*)Yb, (int)pb,
                       W, (int)N, (const double*)nullptr);
    hipLaunchKernelGGL(k_panel_sub, dim3((unsigned)((N + 63) / 64), (unsigned)((pb + 7) / 8)), dim3(64), (size_t)c * 8 * 8, h->stream,
                       X, (int)c, (const double*)W, Yb, (int)pb, (int)N, (const double*)nullptr);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

int launch_orth(Handle* h, double* Y, double* tmp, double* W, int64_t N, int64_t p, double* status_dev,
                bool allow_cholqr, bool* used_cholqr, bool one_pass, int64_t c_start) {
    const bool no_cholqr = dev_is(DEV_NO_CHOLQR, '1');
    *used_cholqr = allow_cholqr && p <= 512 && !no_cholqr;
    // Yb (N x pb) -= Y[:, 0:c0] (Y[:, 0:c0]' Yb), the finished columns taken PROJ_CHUNK at a time: k_panel_sub keeps its
    // slice of the coefficients in LDS (64 bytes per finished column), and c_start callers (null-space completion of the
    // returned vectors, N up to 4608) start far beyond the 512 columns the block orthonormalisation itself ever reaches.
    // Chunk after chunk is a block modified Gram-Schmidt sweep - at least as accurate as the one-shot projection.
    constexpr int64_t PROJ_CHUNK = 512;
    auto project = [&](double* Yb, int64_t pb, int64_t c0, const double* sticky) {
        for (int64_t cc = 0; cc < c0; cc += PROJ_CHUNK) {
            const int64_t cn = std::min<int64_t>(PROJ_CHUNK, c0 - cc);
            const double* Yc = Y + (size_t)cc * N;
            hipLaunchKernelGGL(k_panel_tn2, dim3((unsigned)((cn * pb + 3) / 4)), dim3(256), 0, h->stream, Yc, (int)cn,
                               (const double*)Yb, (int)pb, W, (int)N, sticky);
            hipLaunchKernelGGL(k_panel_sub, dim3((unsigned)((N + 63) / 64), (unsigned)((pb + 7) / 8)), dim3(64),
                               (size_t)cn * 8 * 8, h->stream, Yc, (int)cn, (const double*)W, Yb, (int)pb, (int)N, sticky);
        }
    };
    if (!*used_cholqr) {
        // Column-sequential CGS2 is one workgroup: 3.4 ms for a 4096 x 76 block (large-mode cold start).  Wide blocks of
        // long vectors go block by block: 16 columns are projected twice against the finished ones (two multi-workgroup
        // products per projection) and orthonormalised among themselves by the one-workgroup kernel, whose cost falls
        // with the square of the block width.
        const bool no_blocked = dev_is(DEV_NO_BLOCKED_CGS2, '1');
        if (no_blocked || ((N < 2048 || p <= 32) && c_start == 0)) return launch_cgs2(h, Y, N, p, status_dev);
        constexpr int64_t GB = 16;
        for (int64_t c0 = c_start; c0 < p; c0 += GB) {
            const int64_t pb = std::min<int64_t>(GB, p - c0);
            double* Yb = Y + (size_t)c0 * N;
            if (c0 > 0) {
                for (int rep = 0; rep < 2; ++rep) project(Yb, pb, c0, nullptr);
            }
            const size_t lds = (size_t)(pb + 16) * 8;
            hipLaunchKernelGGL(k_cgs2<false>, dim3(1), dim3(SS_THREADS), lds, h->stream, Yb, (int)N, (int)pb, status_dev,
                               c0 > c_start ? 1 : 0);
        }
        TLSQ_HIP(h, hipGetLastError());
        return TLSQ_OK;
    }
    const dim3 rows((int)((N + 63) / 64));
    for (int64_t c0 = 0; c0 < p; c0 += CQ_PMAX) {
        const int64_t pb = std::min<int64_t>(CQ_PMAX, p - c0);
        double* Yb = Y + (size_t)c0 * N;
        double* Tb = tmp + (size_t)c0 * N;
        const int first = c0 == 0 ? 1 : 0;
        const double* sticky = first ? nullptr : status_dev;
        if (c0 > 0) {
            for (int rep = 0; rep < 2; ++rep) project(Yb, pb, c0, sticky);   // against the finished columns, twice
        }
        if (one_pass && p <= CQ_PMAX) {   // (k_chol_trsm keeps its rows in registers: in == out is fine)
            TLSQ_TRY(launch_panel_tn(h, Yb, Yb, W, N, pb, nullptr));
            hipLaunchKernelGGL(k_chol_trsm, rows, dim3(64), 0, h->stream, (const double*)Yb, (const double*)W, status_dev, Yb,
                               (int)N, (int)pb, 1, first);
            break;
        }
        for (int pass = 1; pass <= 2; ++pass) {
            const double* in = pass == 1 ? Yb : Tb;
            double* out = pass == 1 ? Tb : Yb;
            if (pass == 1 && !first) {
                hipLaunchKernelGGL(k_panel_tn2, dim3((unsigned)((pb * pb + 3) / 4)), dim3(256), 0, h->stream, in, (int)pb,
                                   in, (int)pb, W, (int)N, sticky);
                TLSQ_HIP(h, hipGetLastError());
            } else {
                TLSQ_TRY(launch_panel_tn(h, in, in, W, N, pb, pass == 2 ? status_dev : nullptr));
            }
            hipLaunchKernelGGL(k_chol_trsm, rows, dim3(64), 0, h->stream, in, (const double*)W, status_dev, out, (int)N,
                               (int)pb, pass, first);
        }
    }
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;   // status[1] != 0: the caller has to redo the step with CGS2 (Y may be partly orthonormalised:
                      // the retry recomputes it from X)
}

int launch_ritz_resid(Handle* h, const double* GX, const double* X, const double* theta, int64_t N, int64_t p,
                      double* res) {
    hipLaunchKernelGGL(k_ritz_resid, dim3((int)((p + 3) / 4)), dim3(256), 0, h->stream, GX, X, theta, (int)N,
                       (int)p, res);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

int launch_rayleigh(Handle* h, const double* GX, const double* X, int64_t N, int64_t p, double* theta) {
    hipLaunchKernelGGL(k_rayleigh, dim3((int)((p + 3) / 4)), dim3(256), 0, h->stream, GX, X, (int)N, (int)p,
                       theta);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// ---- Rayleigh-Ritz of a warm block without an orthonormalisation pass and without Jacobi sweeps ---------------------
// The warm block Y = [G^q X_top, G X_pad] consists of images of the previous Ritz vectors: its columns are nearly
// orthogonal and Y'GY is nearly diagonal once they are normalised.  CholeskyQR2 (two launches with a column-sequential
// factorisation each), the product G Q, H = Q'GQ and the Jacobi solver (57 dependent rotation rounds at p = 20, 33 us)
// are then replaced by two reductions over the panel - B = Y'Y, Hg = Y'(G Y) (k_panel_tn2x) - and ONE workgroup that works
// on p x p matrices in LDS through dense p x p x p products only (k_rr_small, p <= 32):
//   1. D = diag(B)^-1/2, Bh = D B D, Hh = D sym(Hg) D
//   2. Bh = L L' in LDS (the pad columns G X_pad lean towards the dominant directions by O(1): Bh is well conditioned but
//      not close to I, so no Newton-Schulz here); T = L^-1; C_0 = T'
//   3. the refinement of Ogita & Aishima (Japan J. Indust. Appl. Math. 35, 2018) for the pencil (Hh, Bh), started from C_0:
//      R = I - C'Bh C, S = C'Hh C, lambda_i = S_ii / (1 - R_ii), E_ij = (S_ij + lambda_j R_ij) / (lambda_j - lambda_i) - or
//      R_ij / 2 for pairs closer than delta_c = 2 (||S - diag||_F + ||Hh||_F ||R||_F) -, C <- C + C E: quadratic convergence
//      for separated Ritz values, and B-orthonormality is refined along with the vectors - the factorisation only has to
//      provide a starting point.
//   4. D C: the caller's rotation kernel forms X' = Y D C, G X' = (G Y) D C, Ritz values and residuals - which remain the
//      acceptance test, whatever produced C.
// status[0] = sqrt(smallest pivot), status[1] = 0 ok / 1 Y too far from orthogonal or not finite / 2 no convergence /
// 3 a pad column may have reached the threshold (nt = number of wanted columns, tau2 = the threshold: pad columns are
// not rotated among themselves),
// status[2] = delta.  On failure C = D (X' = the normalised columns of Y: same span, nothing is lost).
constexpr int RS_P = 32;
constexpr int RS_LD = 33;
// Four waves, one 16 x 16 tile of the (padded) 32 x 32 matrices each; all p x p x p products run on
// v_mfma_f64_16x16x4_f64 with operands read from LDS (a plain one-entry-per-thread product moves 2 p loads per entry
// through the LDS port: 1.2 us per product at p = 20 with 1024 threads - the port, not the arithmetic, was the limit).
// Lane l of wave w owns the entries (row ti 16 + (l >> 4) + 4 q, q = 0..3; column tj 16 + (l & 15)), ti = w & 1, tj = w >> 1
// - the accumulator layout of the MFMA - in every matrix, so all element-wise steps work on registers.
struct RsLane {
    int ti, tj, fr, fk, cj;
    __device__ __forceinline__ int ri(int q) const { return ti * 16 + fk + 4 * q; }
};
// C tile = op(A) op(B) over nk k-steps of 4 (A, B: 32 x 32 in LDS, column-major, ld RS_LD)
// (all operands of a tile are fetched before the first MFMA: one LDS latency per product instead of one per k-step)
template <bool TA, bool TB>
__device__ __forceinline__ sm_d4 rs_mfma(const double* A, const double* B, const RsLane& L, int nk) {
    sm_d4 acc = sm_d4{0.0, 0.0, 0.0, 0.0};
    const int ar = L.ti * 16 + L.fr, bc = L.tj * 16 + L.fr;
    double a[8], b[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int k = 4 * s + L.fk;
        a[s] = (s < nk) ? (TA ? A[k + ar * RS_LD] : A[ar + k * RS_LD]) : 0.0;
        b[s] = (s < nk) ? (TB ? B[bc + k * RS_LD] : B[k + bc * RS_LD]) : 0.0;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
        if (s < nk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ double rs_wmax(double v) {
    // (DPP inside the rows of 16, then the four row maxima by v_readlane: no LDS crossbar)
    double o;
    o = ss_dpp<0xB1>(v);  v = o > v ? o : v;
    o = ss_dpp<0x4E>(v);  v = o > v ? o : v;
    o = ss_dpp<0x141>(v); v = o > v ? o : v;
    o = ss_dpp<0x140>(v); v = o > v ? o : v;
    const double a = ss_lane(v, 0), b = ss_lane(v, 16), c = ss_lane(v, 32), d = ss_lane(v, 48);
    const double ab = b > a ? b : a, cd = d > c ? d : c;
    return cd > ab ? cd : ab;
}
// block-wide sums of two values and maxima of three (4 waves), the same bits in every thread
__device__ __forceinline__ void rs_reduce(double& s0, double& s1, double& m0, double& m1, double& m2, double* red) {
    s0 = ss_wsum(s0);
    s1 = ss_wsum(s1);
    m0 = rs_wmax(m0);
    m1 = rs_wmax(m1);
    m2 = rs_wmax(m2);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[w * 5 + 0] = s0;
        red[w * 5 + 1] = s1;
        red[w * 5 + 2] = m0;
        red[w * 5 + 3] = m1;
        red[w * 5 + 4] = m2;
    }
    __syncthreads();
    s0 = (red[0] + red[5]) + (red[10] + red[15]);
    s1 = (red[1] + red[6]) + (red[11] + red[16]);
    m0 = fmax(fmax(red[2], red[7]), fmax(red[12], red[17]));
    m1 = fmax(fmax(red[3], red[8]), fmax(red[13], red[18]));
    m2 = fmax(fmax(red[4], red[9]), fmax(red[14], red[19]));
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_rr_small(const double* __restrict__ Bg, const double* __restrict__ Hg, int p,
                                                  double* __restrict__ Cout, double* __restrict__ lam_out,
                                                  double* __restrict__ status, int nt, double tau2, int block_ok) {
    // seven 32 x 32 buffers (59 KB): Bh and Hh stay; the others change roles between the factorisation and the refinement
    __shared__ double sBh[RS_P * RS_LD], sH[RS_P * RS_LD], b2[RS_P * RS_LD], b3[RS_P * RS_LD], b4[RS_P * RS_LD];
    __shared__ double b5[RS_P * RS_LD], b6[RS_P * RS_LD];
    __shared__ double sd[RS_P], slam[RS_P], red[20];
    double* const sZ = b4;   // L                      | H C
    double* const sT = b5;   // T = L^-1
    double* const sU = b6;   // 2 I - L X (Newton)     | E
    double* const sV = b2;   // trailing matrix (even) | C
    double* const sY = b3;   // trailing matrix (odd)  | Bh C
    double* const sW = b4;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    RsLane L;
    L.ti = w & 1;
    L.tj = w >> 1;
    L.fr = lane & 15;
    L.fk = lane >> 4;
    L.cj = L.tj * 16 + L.fr;
    const int cj = L.cj;
    const int nk = (p + 3) / 4;
    const bool cin = cj < p;
    // ---- load, scale: Bh = D B D (unit diagonal), Hh = D sym(Hg) D; identity / zero in the padding ----
    double bq[4], hq[4], badv = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = L.ri(q);
        const bool in = cin && i < p;
        bq[q] = in ? Bg[i + (size_t)cj * p] : (i == cj ? 1.0 : 0.0);
        hq[q] = in ? 0.5 * (Hg[i + (size_t)cj * p] + Hg[cj + (size_t)i * p]) : 0.0;
        if (i == cj) {
            const bool ok = (bq[q] > 0.0) && (bq[q] < 1.0e300);
            sd[i] = ok ? 1.0 / sqrt(bq[q]) : 0.0;
            if (!ok) badv = 1.0;
        }
        if (!(fabs(hq[q]) < 1.0e300)) badv = 1.0;
    }
    double dsum = 0.0, z1 = 0.0, z2 = 0.0, zm = 0.0, zm2 = 0.0, dtt = 0.0, ktp = 0.0;
    rs_reduce(badv, z1, z2, zm, zm2, red);   // (also orders the writes of sd)
    const double dj = sd[cj];
    double a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = L.ri(q);
        const double di = sd[i];
        a[q] = (i == cj) ? 1.0 : bq[q] * di * dj;
        hq[q] *= di * dj;
        const double dv = (i == cj) ? 0.0 : a[q];
        dsum += dv * dv;
        if (i < nt && cj < nt) dtt += dv * dv;
        if ((i < nt) != (cj < nt) && cin && i < p) ktp = fmax(ktp, fabs(dv));
        sBh[i + cj * RS_LD] = a[q];
        sH[i + cj * RS_LD] = hq[q];
        sV[i + cj * RS_LD] = a[q];
        sZ[i + cj * RS_LD] = (i == cj && i >= p) ? 1.0 : 0.0;   // L (identity in the padding)
    }
    z2 = zm2 = 0.0;
    z1 = dtt;
    zm = ktp;
    rs_reduce(dsum, z1, zm, z2, zm2, red);   // (its barriers also order the stores above)
    const double delta = sqrt(dsum);
    const double delta_tt = sqrt(z1), k_tp = zm;
    int fail = (badv != 0.0 || !(delta < 1.0e300)) ? 1 : 0;
    bool blocked = false;
    double minpiv = 1.0;
    int oa_its = 0;
    double last_numax = -1.0;
    double lamv[4] = {0.0, 0.0, 0.0, 0.0};
    if (!fail) {
        // ---- Bh = L L' (right-looking, every thread keeps its entries in registers; the trailing matrix alternates between
        //      two buffers - step k reads the state step k - 1 wrote and writes its own update to the other one: one barrier per
        //      column).  The pivots of the unit-diagonal Bh are the squared distances of each column from the span of the ones
        //      before it (wanted columns first: Gram-Schmidt order).  L only has to be good enough to START the refinement
        //      below, which works in the metric Bh itself and removes what is left of the non-orthogonality: pivots down to
        //      1e-4 are accepted (CholeskyQR proper needs a second pass below 0.25). ----
        // Block form (warm blocks a few iterations old: the wanted columns G^q x_i are orthogonal among themselves to 1e-3 ..
        // 1e-9 once they are normalised, only the pad columns lean on them by O(1)): with Bh = [I K; K' B_pp] up to that
        // deviation, L = [I 0; K' L_S], L_S L_S' = S = B_pp - K'K - one p x p x p product and a factorisation of the pad block
        // (4 columns of 20) instead of p dependent column steps.  What the top block really deviates from I is left to the
        // refinement below, which has to remove deviations of that size from C'Hh C anyway.
        blocked = block_ok && nt >= 1 && nt < p && delta_tt <= 3e-3;
        int k0 = 0;
        if (blocked) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = L.ri(q);
                sU[i + cj * RS_LD] = (cin && i < nt && cj >= nt) ? a[q] : 0.0;   // K (top rows, pad columns), zero elsewhere
            }
            __syncthreads();
            const sm_d4 kk = rs_mfma<true, false>(sU, sU, L, nk);   // K'K: only its pad-pad block is non-zero
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = L.ri(q);
                if (cin && i < p && i >= nt && cj >= nt) {
                    a[q] -= kk[q];
                    sV[i + cj * RS_LD] = a[q];
                    sY[i + cj * RS_LD] = a[q];
                }
                if (cj < nt) sZ[i + cj * RS_LD] = (i == cj) ? 1.0 : ((i >= nt && i < p) ? a[q] : 0.0);   // [I; K']
            }
            __syncthreads();
            k0 = nt;
        }
        for (int k = k0; k < p; ++k) {
            const double* cur = (k & 1) ? sY : sV;
            double* nxt = (k & 1) ? sV : sY;
            const double akk = cur[k + k * RS_LD];
            if (!(akk >= 1e-4)) {   // (the same value in every thread)
                fail = 1;
                break;
            }
            minpiv = akk < minpiv ? akk : minpiv;
            double r = __builtin_amdgcn_rsq(akk);
            r = r * (1.5 - 0.5 * akk * r * r);
            r = r * (1.5 - 0.5 * akk * r * r);
            const double ljk = cur[cj + k * RS_LD] * r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = L.ri(q);
                if (i >= cj && cin && i < p) {
                    if (cj == k) sZ[i + cj * RS_LD] = a[q] * r;
                    else if (cj > k) {
                        a[q] -= (cur[i + k * RS_LD] * r) * ljk;
                        nxt[i + cj * RS_LD] = a[q];
                    }
                }
            }
            __syncthreads();
        }
    }
    if (!fail) {
        // ---- T = L^-1 by Newton's iteration X <- X (2 I - L X) from X = diag(L)^-1: I - L X is strictly lower triangular,
        //      hence nilpotent, and squares in every step - exact after ceil(log2 p) steps ----
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = L.ri(q);
            sT[i + cj * RS_LD] = (i == cj) ? 1.0 / sZ[i + i * RS_LD] : 0.0;
        }
        __syncthreads();
        int nst = 0;
        const int nil = blocked ? p - nt + 1 : p;   // (block form: I - L X is non-zero in the pad rows only)
        while ((1 << nst) < nil) ++nst;
        for (int st = 0; st < nst; ++st) {
            const sm_d4 m = rs_mfma<false, false>(sZ, sT, L, nk);   // L X
#pragma unroll
            for (int q = 0; q < 4; ++q) sU[L.ri(q) + cj * RS_LD] = ((L.ri(q) == cj) ? 2.0 : 0.0) - m[q];
            __syncthreads();
            const sm_d4 xn = rs_mfma<false, false>(sT, sU, L, nk);  // X (2 I - L X)
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) sT[L.ri(q) + cj * RS_LD] = (L.ri(q) >= cj) ? xn[q] : 0.0;
            __syncthreads();
        }
        // ---- C = T' to start with (C' Bh C = I up to the rounding of the factorisation, ~eps cond(Bh)); ||Hh||_F ----
        double hn2 = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sV[L.ri(q) + cj * RS_LD] = sT[cj + L.ri(q) * RS_LD];
            hn2 += hq[q] * hq[q];
            if (L.ri(q) == cj) slam[cj] = hq[q];   // (a first scale for lambda; replaced in the first step)
        }
        z1 = z2 = zm = zm2 = 0.0;
        rs_reduce(hn2, z1, z2, zm, zm2, red);   // (also orders the writes of sV, slam)
        const double hn = sqrt(hn2);
        if (!(hn > 0.0) || !(hn < 1.0e300)) fail = 1;
        if (!fail) {
            // ---- refinement of Ogita & Aishima for the pencil (Hh, Bh): R = I - C'Bh C, S = C'Hh C, lambda_i = S_ii / (1 - R_ii),
            //      E_ij = (S_ij + lambda_j R_ij) / (lambda_j - lambda_i), C <- C (I + E) ----
            // Convergence is judged on what matters downstream, the Ritz residuals: max |S_ij + lambda_j R_ij| over every pair
            // that is not a pad-pad pair <= 1e-14 lambda_max, and max |R_ij| <= 1e-14 - both a small multiple of their rounding
            // floor p eps.  Pairs closer than delta_c = 2 (||S - diag||_F + ||Hh||_F ||R||_F), i.e. closer than their own
            // coupling, are only orthonormalised (E_ij = R_ij / 2): a true multiple Ritz value converges that way, a
            // near-degenerate pair with coupling does not and ends the attempt (status 2: the Jacobi solver rotates those).
            // Pad columns (index >= nt) are never rotated among themselves - see the Gershgorin test at the end.
            bool conv = false;
            double sq[4] = {0.0, 0.0, 0.0, 0.0}, rq[4] = {0.0, 0.0, 0.0, 0.0};
            double numax_prev = 1.0e300;
            // (12 corrections at most - 8 until round 6: the first warm step of a solve, two fresh pad columns and the largest change
            //  of the panel there is, converged in its eighth or not at all depending on the last bit of ||D||_2: a miss is a
            //  second step on the Jacobi path, 150 us against 6 per correction; stalls still leave through the progress test)
            for (int it = 0; it < 12; ++it) {
                oa_its = it + 1;
                const sm_d4 bc = rs_mfma<false, false>(sBh, sV, L, nk);
                const sm_d4 hc = rs_mfma<false, false>(sH, sV, L, nk);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    sY[L.ri(q) + cj * RS_LD] = bc[q];
                    sW[L.ri(q) + cj * RS_LD] = hc[q];
                }
                __syncthreads();
                const sm_d4 gm = rs_mfma<true, false>(sV, sY, L, nk);   // C' Bh C
                const sm_d4 sm = rs_mfma<true, false>(sV, sW, L, nk);   // C' Hh C
                double off2 = 0.0, rn2 = 0.0, lmx = 0.0, rmx = 0.0, numax = 0.0;
                const double ljp = slam[cj];
                double lnew[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {   // (the products only run over k < 4 nk: nothing is known about the padding)
                    const int i = L.ri(q);
                    const bool in = cin && i < p;
                    sq[q] = in ? sm[q] : 0.0;
                    rq[q] = in ? ((i == cj) ? 1.0 : 0.0) - gm[q] : 0.0;
                    lnew[q] = 0.0;
                    if (i == cj) {
                        lnew[q] = sq[q] / (1.0 - rq[q]);
                        if (in) lmx = fabs(lnew[q]);
                    } else {
                        off2 += sq[q] * sq[q];
                        if (in && !(i >= nt && cj >= nt)) numax = fmax(numax, fabs(sq[q] + ljp * rq[q]));
                    }
                    rn2 += rq[q] * rq[q];
                    rmx = fmax(rmx, fabs(rq[q]));
                }
                rs_reduce(off2, rn2, lmx, rmx, numax, red);   // (its first barrier also separates the reads of slam above ...
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (L.ri(q) == cj) slam[cj] = lnew[q];   //  ... from this write; the barrier below orders it)
                last_numax = numax;
                if (!(numax < 1.0e300) || !(lmx < 1.0e300) || !(rmx < 1.0e300)) {
                    fail = 2;
                    break;
                }
                if (it >= 1 && numax <= 1e-14 * lmx && rmx <= 1e-14) {   // (step 0 measures with a provisional lambda)
                    conv = true;
                    break;
                }
                // no progress worth the name after the first corrections: either the couplings sit on their rounding floor -
                // pad columns that lean on the dominant directions by O(1) (late ALM iterations: their own Ritz values are
                // noise) put eps lambda_max ||c_j|| into S_ij, a few 1e-14 lambda_max; the caller's residual test decides what
                // that is worth - or a near-degenerate pair with coupling keeps them up
                if (it >= 2 && numax > 0.1 * numax_prev && rmx <= 1e-10) {
                    if (numax <= 1e-13 * lmx && rmx <= 1e-13) conv = true;
                    else fail = 2;
                    break;
                }
                numax_prev = numax;
                __syncthreads();
                const double dc = 2.0 * (sqrt(off2) + hn * sqrt(rn2));
                const double lj = slam[cj];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = L.ri(q);
                    double e;
                    if (i == cj || !cin || i >= p || (i >= nt && cj >= nt)) e = 0.5 * rq[q];
                    else {
                        const double gap = lj - slam[i];
                        e = fabs(gap) > dc ? (sq[q] + lj * rq[q]) / gap : 0.5 * rq[q];
                    }
                    sU[i + cj * RS_LD] = e;
                }
                __syncthreads();
                const sm_d4 ve = rs_mfma<false, false>(sV, sU, L, nk);
                double vn[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) vn[q] = sV[L.ri(q) + cj * RS_LD] + ve[q];
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q) sV[L.ri(q) + cj * RS_LD] = vn[q];
                __syncthreads();
            }
            if (!fail && !conv) fail = 2;   // no convergence: Jacobi decides
            if (!fail && nt < p) {
                // The pad block of C'Hh C is left as it is: the pad columns are orthonormal and decoupled from the wanted ones,
                // their Rayleigh quotients stand in for Ritz values.  That is all the subspace iteration needs from them AS LONG
                // AS none of them is about to be counted: every eigenvalue of the pad block lies below its largest Gershgorin
                // disc edge, which has to stay clear of the threshold tau^2 - otherwise (the rank is growing) status 3.
#pragma unroll
                for (int q = 0; q < 4; ++q) sU[L.ri(q) + cj * RS_LD] = fabs(sq[q]);
                __syncthreads();
                double gmx = 0.0;
                if (tid >= nt && tid < p) {
                    double g = 0.0;
                    for (int c = nt; c < p; ++c) g += sU[tid + c * RS_LD];
                    gmx = g;
                }
                double z0 = 0.0, za = 0.0, zb = 0.0, zc = 0.0;
                rs_reduce(z0, za, gmx, zb, zc, red);
                if (!(gmx < tau2 * (1.0 - 1e-6))) fail = 3;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (L.ri(q) == cj) lamv[q] = slam[cj];
        }
    }
    if (fail == 1) {
        // C = D: the normalised columns of Y - the same span for the fall-back step
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = L.ri(q);
            if (cin && i < p) Cout[i + (size_t)cj * p] = (i == cj) ? (sd[i] > 0.0 ? sd[i] : 1.0) : 0.0;
            if (cin && i == cj) lam_out[i] = 0.0;
        }
    } else {
        // the rotation for the unscaled columns: D C  (status 2 / 3: the last iterate - still a basis of the same span)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = L.ri(q);
            if (cin && i < p) Cout[i + (size_t)cj * p] = sd[i] * sV[i + cj * RS_LD];
            if (cin && i == cj) lam_out[i] = lamv[q];
        }
    }
    if (tid == 0) {
        status[0] = sqrt(minpiv);
        status[1] = (double)fail;
        status[2] = delta;
        status[3] = (double)oa_its;
        status[4] = last_numax;
        status[5] = delta_tt;
        status[6] = k_tp;
        status[7] = blocked ? 1.0 : 0.0;
    }
}

// B = Y'Y and Hg = Y'(GY) (p x p each, ld p) for N x p panels: one wave per entry of each
__global__ __launch_bounds__(256) void k_panel_tn2x(const double* __restrict__ Y, const double* __restrict__ GY,
                                                    double* __restrict__ Bm, double* __restrict__ Hm, int N, int p) {
    const int lane = threadIdx.x & 63;
    int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= 2 * p * p) return;
    const bool second = e >= p * p;
    if (second) e -= p * p;
    const int i = e % p, j = e / p;
    const double* a = Y + (size_t)i * N;
    const double* b = (second ? GY : Y) + (size_t)j * N;
    double s = 0.0;
    for (int r = lane; r < N; r += 64) s += a[r] * b[r];
    s = ss_wsum(s);
    if (lane == 0) (second ? Hm : Bm)[i + (size_t)j * p] = s;
}

int launch_rr_small(Handle* h, const double* Y, const double* GY, double* Bm, double* Hm, double* Cout, double* lam,
                    double* status, int64_t N, int64_t p, int64_t nt, double tau2) {
    if (p < 1 || p > RS_P) return set_err(h, TLSQ_ERR_ARG, "rr_small: p = %lld (1 .. %d)", (long long)p, RS_P);
    hipLaunchKernelGGL(k_panel_tn2x, dim3((unsigned)((2 * p * p + 3) / 4)), dim3(256), 0, h->stream, Y, GY, Bm, Hm, (int)N, (int)p);
    hipLaunchKernelGGL(k_rr_small, dim3(1), dim3(256), 0, h->stream, (const double*)Bm, (const double*)Hm, (int)p, Cout, lam,
                       status, (int)nt, tau2, dev_is(DEV_NO_RR_BLOCKED, '1') ? 0 : 1);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
// the p x p stage alone (tests): C, lambda, status from given B, Hg
int launch_rr_small_only(Handle* h, const double* Bm, const double* Hm, double* Cout, double* lam, double* status, int64_t p,
                         int64_t nt, double tau2) {
    if (p < 1 || p > RS_P) return set_err(h, TLSQ_ERR_ARG, "rr_small: p = %lld (1 .. %d)", (long long)p, RS_P);
    hipLaunchKernelGGL(k_rr_small, dim3(1), dim3(256), 0, h->stream, Bm, Hm, (int)p, Cout, lam, status, (int)nt, tau2,
                       dev_is(DEV_NO_RR_BLOCKED, '1') ? 0 : 1);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// Rayleigh-Ritz finish in one launch, one workgroup per Ritz vector c:  x = Q S[:,c],  gx = GQ S[:,c],
// theta_c = x . gx,  res_c = ||gx - theta_c x||  (k_panel_rot2 + k_rayleigh + k_ritz_resid for small panels)
__global__ __launch_bounds__(256) void k_ritz_finish(const double* __restrict__ Q, const double* __restrict__ GQ,
                                                     const double* __restrict__ S, double* __restrict__ X,
                                                     double* __restrict__ GX, double* __restrict__ theta,
                                                     double* __restrict__ res, int N, int p,
                                                     const double* __restrict__ status, double* mailbox,
                                                     unsigned int* arrivals, double seq, SpecCtrl* ctrl, double inv_mu,
                                                     int nukeA, const double* __restrict__ keys,
                                                     const double* __restrict__ keys_guard) {
    __shared__ double sS[CQ_PMAX * 16];   // p <= 512
    __shared__ double red[4];
    __shared__ int s_last, s_dest;
    const int cs = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int k = tid; k < p; k += 256) sS[k] = S[k + (size_t)cs * p];
    // keys (round 6): the eigenvalues of the p x p problem, in the order of S's columns.  The Ritz pair of column cs is written
    // to the position its key has in descending order (ties by column index, NaN last: a permutation whatever the keys are),
    // so that the block arrives sorted and the host does not queue two panel copies and two gathers to sort it.  Not when the
    // guard (the status of a CholeskyQR orthonormalisation, where one ran) says the step is going to be repeated.
    if (tid == 0) s_dest = cs;
    __syncthreads();
    if (keys && !(keys_guard && keys_guard[1] != 0.0)) {
        if (tid < 64) {
            const double kc = keys[cs];
            const double mine = kc == kc ? kc : -__builtin_inf();
            int cnt = 0;
            for (int i = tid; i < p; i += 64) {
                const double ki = keys[i];
                const double other = ki == ki ? ki : -__builtin_inf();
                cnt += (other > mine || (other == mine && i < cs)) ? 1 : 0;
            }
            for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
            if (tid == 0) s_dest = cnt;
        }
        __syncthreads();
    }
    const int c = s_dest;
    double dot = 0.0;
    for (int r = tid; r < N; r += 256) {
        double x = 0.0, g = 0.0;
        for (int k = 0; k < p; ++k) {
            x += Q[r + (size_t)k * N] * sS[k];
            g += GQ[r + (size_t)k * N] * sS[k];
        }
        X[r + (size_t)c * N] = x;
        GX[r + (size_t)c * N] = g;
        dot += x * g;
    }
    dot = ss_wsum(dot);
    if (lane == 0) red[w] = dot;
    __syncthreads();
    const double th = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    double rs = 0.0;
    for (int r = tid; r < N; r += 256) {   // own writes, read back by the same thread
        const double v = GX[r + (size_t)c * N] - th * X[r + (size_t)c * N];
        rs += v * v;
    }
    rs = ss_wsum(rs);
    if (lane == 0) red[w] = rs;
    __syncthreads();
    if (tid == 0) {
        const double rr = sqrt((red[0] + red[1]) + (red[2] + red[3]));
        theta[c] = th;
        res[c] = rr;
        s_last = 0;
        if (mailbox) {
            // Results go to host-visible (coherent, pinned) memory: [0] sequence flag, [8..) theta[p], res[p], status[3].  The
            // workgroup that arrives last copies all of them there and publishes the flag; the host polls it instead of paying
            // a copy command plus a stream synchronisation (~40 us) for 2p+2 numbers.  (Every workgroup used to write its own
            // pair and fence at system scope: twenty fences over the host link instead of one.)
            __threadfence();
            s_last = atomicAdd(arrivals, 1u) == (unsigned int)(p - 1) ? 1 : 0;
        }
    }
    __syncthreads();
    if (!s_last) return;
    // ---- last arrival (its first wave): the other workgroups' values were released by their fences; they are read past this
    // CU's L1 ----
    if (tid >= 64) return;
    volatile double* mb = mailbox;
    double* sT = sS;   // (S's column is not needed any more)
    for (int i = tid; i < p; i += 64) {
        const double t = __hip_atomic_load(theta + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double e = __hip_atomic_load(res + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sT[i] = t;
        mb[8 + i] = t;
        mb[8 + p + i] = e;
    }
    if (tid == 0) {
        mb[8 + 2 * p] = status ? status[0] : 0.0;
        mb[8 + 2 * p + 1] = status ? status[1] : 0.0;
        mb[8 + 2 * p + 2] = status ? status[2] : 0.0;
    }
    if (ctrl) {
        // The decision the host takes from these numbers (solver.hip: sorted block, count of sigma >= 1/mu, weights of the
        // thresholded rebuild), in the same arithmetic (IEEE sqrt and division), so that the factor product queued behind
        // this kernel runs without waiting for the host.  One lane per Ritz value (blocks of up to 64 columns; wider ones
        // never take the speculative product: ok = 0).
        bool ok = !(status && status[1] != 0.0) && p <= 64;
        const bool mine = tid < p;
        const double t = mine && p <= 64 ? sT[tid] : 0.0;   // (one wave: its LDS writes above are ordered before this read)
        const double sg = sqrt(t > 0.0 ? t : 0.0);
        const double up = __shfl_up(sg, 1, 64);
        // (the host's sort would move this column - unless it is a pad column far below the threshold: svdstep.hip, pads_only)
        const bool bad = mine && (!(t - t == 0.0) || (tid > 0 && sg > up && !(sg < 0.5 * inv_mu)));
        const bool above = mine && sg >= inv_mu;
        const unsigned long long mb_above = __ballot(above);
        if (__ballot(bad) != 0ull) ok = false;
        const int r = __popcll(mb_above);
        if (above) {
            const int pos = __popcll(mb_above & ((1ull << tid) - 1ull));
            if (pos < 32) {
                ctrl->sw.sel[pos] = tid;
                ctrl->sw.w[pos] = nukeA ? (sg - inv_mu) / sg : 1.0;
            }
        }
        if (r > 32) ok = false;
        if (tid == 0) {
            ctrl->r = r;
            ctrl->ok = ok ? 1 : 0;
            mb[8 + 2 * p + 3] = ok ? 1.0 : 0.0;
            mb[8 + 2 * p + 4] = (double)r;
        }
    }
    if (tid == 0) *arrivals = 0u;
    __threadfence_system();   // (wave-wide: every lane's mailbox writes are out before the flag)
    if (tid == 0) mb[0] = seq;
}

// X = Q S, GX = GQ S, theta, res: one launch (every workgroup reads both panels); the three separate kernels remain
// for blocks of more than 256 columns (k_panel_rot2 keeps S in LDS: p <= 90 there)
int launch_ritz_finish(Handle* h, const double* Q, const double* GQ, const double* S, double* X, double* GX,
                       double* theta, double* res, int64_t N, int64_t p, const double* status, double* mailbox_dev,
                       unsigned int* arrivals, double seq, SpecCtrl* ctrl, double inv_mu, int nukeA, const double* sort_keys,
                       const double* sort_guard) {
    if (p <= 512) {   // (always, for the block sizes in use: at most 2 p^2 N doubles of L2 traffic, 0.5 ms at p = 192, N = 4096)
        hipLaunchKernelGGL(k_ritz_finish, dim3((unsigned)p), dim3(256), 0, h->stream, Q, GQ, S, X, GX, theta, res, (int)N,
                           (int)p, status, mailbox_dev, arrivals, seq, mailbox_dev ? ctrl : nullptr, inv_mu, nukeA, sort_keys, sort_guard);
        TLSQ_HIP(h, hipGetLastError());
        return TLSQ_OK;
    }
    TLSQ_TRY(launch_panel_rot2(h, Q, GQ, S, X, GX, N, p));
    TLSQ_TRY(launch_rayleigh(h, GX, X, N, p, theta));
    return launch_ritz_resid(h, GX, X, theta, N, p, res);
}

// w -= Vs (Vg' q)  for N-vectors q, w and N x r panels Vs, Vg (the deflation term of the count certificate applied to
// one Lanczos vector); c: r doubles of scratch
int launch_deflate_vec(Handle* h, const double* Vs, const double* Vg, int64_t r, const double* q, double* c, double* w,
                       int64_t N) {
    if (r <= 0) return TLSQ_OK;
    hipLaunchKernelGGL(k_panel_tn2, dim3((unsigned)((r + 3) / 4)), dim3(256), 0, h->stream, Vg, (int)r, q, 1, c, (int)N,
                       (const double*)nullptr);
    hipLaunchKernelGGL(k_panel_sub, dim3((unsigned)((N + 63) / 64), 1), dim3(64), (size_t)r * 8 * 8, h->stream, Vs, (int)r,
                       (const double*)c, w, 1, (int)N, (const double*)nullptr);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

}  // namespace tlsq
