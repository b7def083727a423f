// Matrix functions of a small symmetric matrix through matrix products only (fp64 MFMA GEMM, gemm.hip): the sign function by
// Newton-Schulz and the inverse square root by the coupled Newton-Schulz iteration.  Used by the ALM driver where the rank
// count and the thresholded low-rank matrix are needed but not the singular vectors (solver.hip: noisy data, whose late
// iterations would otherwise each cost a dense N x N decomposition):
//     count of eigenvalues of G above t  = trace(P),  P = (I + sign(G - t I)) / 2          (src/robustPCA.jl:198)
//     sum over them of v (1 - sqrt(t / lambda)) v'  = P - sqrt(t) B^(-1/2) P,  B = P G P + t (I - P)     (:205-213)
// Every iterate is a polynomial in the input, so all products are products of commuting symmetric matrices: the GEMM runs in
// its symmetric form (lower tiles computed, mirrored), which also keeps the iterates exactly symmetric.
#include "common.hpp"
#include "quintic.hpp"

namespace tlsq {

// Y = a X + b I   (N x N, ld N)
__global__ __launch_bounds__(256) void k_mf_axpbi(const double* X, double* Y, int N, double a, double b) {   // (Y may be X)
    const int64_t total = (int64_t)N * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e % N), j = (int)(e / N);
        Y[e] = a * X[e] + (i == j ? b : 0.0);
    }
}

// out[0] = ||X - I||_F^2, out[1] = trace(X), out[2] = max_i sum_j |X_ij| in two deterministic stages (N <= 1024):
// stage 1, grid (ceil(N / 256), MF_CHUNKS): thread = row i, block y = a chunk of columns; per-row partials to part
// (layout [3][MF_CHUNKS][N]: deviation, absolute row sum, diagonal).  Stage 2, one workgroup: fixed-order sums.
// ---- product of two SYMMETRIC N x N matrices without split-K slabs (N <= 2048) --------------------------------------------
// The Newton-Schulz iterations multiply N x N iterates a hundred times per noisy ALM iteration.  Through the 128 x 128-tile
// GEMM an N = 512 product is 10-16 tiles: it has to be split over K to fill the chip, and pays for it with 16 MB of slabs, a
// reduction launch and (for X <- 1.5 I - 0.5 X^2) an element-wise launch: 25 us for 0.27 GFLOP.  Here one workgroup (16
// waves) owns a 32 x 32 tile of the result: wave w computes 16 x 16 sub-tile (w & 3) over K-quarter (w >> 2) with
// v_mfma_f64_16x16x4_f64 fed straight from L2 - both operands are symmetric, so the A fragment "rows i0.., column k" and the B
// fragment "rows j0.., column k" (= B[k, j0..]) are 16 consecutive doubles of one column each (one 128-byte line per 16 lanes) -
// the four quarters meet in LDS in a fixed order, and the epilogue applies alpha, beta I.  SYM: the result is known to be
// symmetric (commuting iterates): only the tiles on and below the diagonal are computed, every entry pair written by one thread.
// (layout of the accumulator of v_mfma_f64_16x16x4_f64: lane l, register q hold D[4 q + (l >> 4)][l & 15].)
typedef double mf_d4 __attribute__((ext_vector_type(4)));
template <bool SYM, bool FULL>
__global__ __launch_bounds__(1024) void k_small_mm(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C,
                                                   int N, int nt, double alpha, double beta, const double* __restrict__ Add,
                                                   double gamma) {
    __shared__ double sR[16 * 256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    int ti, tj;
    if (SYM) {
        const int t = blockIdx.x;
        ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        while (ti * (ti + 1) / 2 > t) --ti;
        tj = t - ti * (ti + 1) / 2;
    } else {
        ti = blockIdx.x % nt;
        tj = blockIdx.x / nt;
    }
    const int sub = w & 3, kq = w >> 2;
    const int i0 = ti * 32 + (sub & 1) * 16, j0 = tj * 32 + (sub >> 1) * 16;
    const int gi = i0 + fr, gj = j0 + fr;
    const bool iok = gi < N, jok = gj < N;
    const double* Ai = A + (iok ? gi : 0);
    // SYM: B is symmetric, B[k, j] is read as B[j, k] (16 consecutive doubles per k).  Otherwise the true column j of B: the
    // four lanes of a k-group read 32 consecutive bytes, a line is used up over the eight MFMAs of a trip.  (The coupled
    // inverse-square-root iteration does not survive the symmetric shortcut: its iterates are symmetric only to rounding, and
    // feeding B' for B breaks the commutativity the iteration relies on - residuals of 1e-6 at cond 130.)
    const double* Bj = SYM ? B + (jok ? gj : 0) : B + (int64_t)(jok ? gj : 0) * N;
    mf_d4 acc = mf_d4{0.0, 0.0, 0.0, 0.0};
    const int kper = ((N + 3) / 4 + 3) / 4 * 4;   // inner indices per K-quarter, a multiple of 4
    const int kbeg = kq * kper, kend = (kbeg + kper < N) ? kbeg + kper : N;
    if (FULL) {
        // N a multiple of 128: every tile and every K-quarter is whole (no predicates: the guarded form compiles to a branch per
        // load and one s_waitcnt vmcnt(0) in front of eight dependent MFMAs).  Two rounds of 16 loads are in flight, the MFMAs
        // alternate between two accumulators.
        mf_d4 acc2 = mf_d4{0.0, 0.0, 0.0, 0.0};
        double a[8], b[8];
        const int rounds = (kend - kbeg) / 32;
        if (SYM) {
            // element offsets in 32 bits (N <= 2048): one address register per load on top of the uniform base
            const unsigned int sa = 4u * (unsigned int)N;
            unsigned int oa = (unsigned int)gi + (unsigned int)(kbeg + fk) * (unsigned int)N;
            unsigned int ob = (unsigned int)gj + (unsigned int)(kbeg + fk) * (unsigned int)N;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = A[oa + u * sa];
                b[u] = B[ob + u * sa];
            }
            for (int r = 0; r + 1 < rounds; ++r) {
                oa += 8u * sa;
                ob += 8u * sa;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (u & 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc2, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
                    // (the slot is free again: the next round's pair goes out behind the MFMA that used it - no branch here, the
                    //  compiler's s_waitcnt bookkeeping gives up across one)
                    a[u] = A[oa + u * sa];
                    b[u] = B[ob + u * sa];
                }
            }
        } else {
            // B[k, j] as it stands: column j is contiguous in k.  MFMA u of a round takes the inner index
            //     k = 8 (u >> 1) + 2 fk + (u & 1)
            // (a permutation of the round's 32 indices, the same for both operands), so that a lane's pair (u, u + 1) is ONE
            // 16-byte load and the four fk-lanes of a column read 64 contiguous bytes per instruction: 64 line visits per round
            // for B instead of 128 with one 8-byte load per MFMA (measured: 20.7 -> 14.3 us at N = 512; the kernel is bound by
            // line visits in the vector memory pipe: ~4.7 us + 0.1 us per visit of a wave's round).
            typedef double mf_d2 __attribute__((ext_vector_type(2)));
            const unsigned int n = (unsigned int)N;
            unsigned int oa = (unsigned int)gi + (unsigned int)(kbeg + 2 * fk) * n;
            unsigned int ob = (unsigned int)gj * n + (unsigned int)(kbeg + 2 * fk);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const mf_d2 v = *reinterpret_cast<const mf_d2*>(B + ob + 8u * t);
                b[2 * t] = v[0];
                b[2 * t + 1] = v[1];
                a[2 * t] = A[oa + (8u * t) * n];
                a[2 * t + 1] = A[oa + (8u * t + 1u) * n];
            }
            for (int r = 0; r + 1 < rounds; ++r) {
                oa += 32u * n;
                ob += 32u;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2 * t], b[2 * t], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2 * t + 1], b[2 * t + 1], acc2, 0, 0, 0);
                    const mf_d2 v = *reinterpret_cast<const mf_d2*>(B + ob + 8u * t);
                    b[2 * t] = v[0];
                    b[2 * t + 1] = v[1];
                    a[2 * t] = A[oa + (8u * t) * n];
                    a[2 * t + 1] = A[oa + (8u * t + 1u) * n];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {   // last round: nothing left to fetch
            if (u & 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc2, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += acc2[q];
    } else {
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        double a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u + fk;
            const bool kok = k < kend;
            a[u] = (iok && kok) ? Ai[(int64_t)k * N] : 0.0;
            b[u] = (jok && kok) ? (SYM ? Bj[(int64_t)k * N] : Bj[k]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) sR[w * 256 + q * 64 + lane] = acc[q];
    __syncthreads();
    // thread (sub2, q2, l2): entry D[4 q2 + (l2 >> 4)][l2 & 15] of sub-tile sub2, the four K-quarters added in order
    const int sub2 = tid >> 8, e = tid & 255, q2 = e >> 6, l2 = e & 63;
    const double v = ((sR[sub2 * 256 + e] + sR[(4 + sub2) * 256 + e]) + sR[(8 + sub2) * 256 + e]) + sR[(12 + sub2) * 256 + e];
    const int i = ti * 32 + (sub2 & 1) * 16 + 4 * q2 + (l2 >> 4), j = tj * 32 + (sub2 >> 1) * 16 + (l2 & 15);
    if (i < N && j < N) {
        double out = alpha * v + (i == j ? beta : 0.0);
        if (Add) out += gamma * Add[i + (int64_t)j * N];   // (SYM: Add is symmetric like the result)
        if (SYM) {
            if (ti != tj) {
                C[i + (int64_t)j * N] = out;
                C[j + (int64_t)i * N] = out;
            } else if (j <= i) {   // diagonal tile: both (i, j) and (j, i) are computed here - the lower one is the one that is kept
                C[i + (int64_t)j * N] = out;
                C[j + (int64_t)i * N] = out;
            }
        } else {
            C[i + (int64_t)j * N] = out;
        }
    }
}

// The same product with register blocking, N a multiple of 128 (the sizes the noisy route runs at): 8 waves per 32 x 32 tile,
// every wave owns ALL FOUR 16 x 16 sub-tiles over one eighth of the inner index.  k_small_mm is bound by line visits in the vector
// memory pipe (one operand fragment = 4 lines, two fragments per MFMA: 8 visits per MFMA); here two A and two B fragments feed
// four MFMAs (4 visits per MFMA), four independent accumulators.  Rounds of 16 inner indices, the next round's fragments fetched
// behind the MFMAs that freed their registers.  General product: B[k, j] read as in k_small_mm<false, true> (one 16-byte load
// per pair of inner indices, the round's indices permuted for both operands).
// (the tile's 1024 entries, K-slices added in order: thread tid returns entries tid and tid + 512 in v[0], v[1]; entry e =
//  (sub2, q2, l2) is D[4 q2 + (l2 >> 4)][l2 & 15] of sub-tile sub2 = ih + 2 jh)
template <bool SYM>
__device__ __forceinline__ void smm_blk_tile(const double* __restrict__ A, const double* __restrict__ B, int N, int nt,
                                             double* sR, int& ti, int& tj, double (&v)[2]) {
    typedef double mf_d2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    if (SYM) {
        const int t = blockIdx.x;
        ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        while (ti * (ti + 1) / 2 > t) --ti;
        tj = t - ti * (ti + 1) / 2;
    } else {
        ti = blockIdx.x % nt;
        tj = blockIdx.x / nt;
    }
    const unsigned int n = (unsigned int)N;
    const int kper = N / 8, kbeg = w * kper, rounds = kper / 16;
    mf_d4 acc00 = mf_d4{0.0, 0.0, 0.0, 0.0}, acc10 = acc00, acc01 = acc00, acc11 = acc00;   // acc<ih><jh>
    double a0[4], a1[4], b0[4], b1[4];
    // inner index of MFMA u of a round: SYM 4 u + fk; general 8 (u >> 1) + 2 fk + (u & 1)
    unsigned int oa = (unsigned int)(ti * 32 + fr) + (unsigned int)(kbeg + (SYM ? fk : 2 * fk)) * n;
    unsigned int ob = SYM ? (unsigned int)(tj * 32 + fr) + (unsigned int)(kbeg + fk) * n
                          : (unsigned int)(tj * 32 + fr) * n + (unsigned int)(kbeg + 2 * fk);
    auto fetch = [&](int u) {
        const unsigned int ka = SYM ? 4u * (unsigned int)u : 8u * (unsigned int)(u >> 1) + (unsigned int)(u & 1);
        a0[u] = A[oa + ka * n];
        a1[u] = A[oa + ka * n + 16u];
        if (SYM) {
            b0[u] = B[ob + ka * n];
            b1[u] = B[ob + ka * n + 16u];
        } else if ((u & 1) == 0) {
            const mf_d2 v0 = *reinterpret_cast<const mf_d2*>(B + ob + 8u * (unsigned int)(u >> 1));
            const mf_d2 v1 = *reinterpret_cast<const mf_d2*>(B + ob + 16u * n + 8u * (unsigned int)(u >> 1));
            b0[u] = v0[0];
            b0[u + 1] = v0[1];
            b1[u] = v1[0];
            b1[u + 1] = v1[1];
        }
    };
    auto fma4 = [&](int u) {
        acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc00, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b0[u], acc10, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b1[u], acc01, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc11, 0, 0, 0);
    };
#pragma unroll
    for (int u = 0; u < 4; ++u) fetch(u);
    for (int r = 0; r + 1 < rounds; ++r) {
        oa += 16u * n;
        ob += SYM ? 16u * n : 16u;
        if (SYM) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                fma4(u);
                fetch(u);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; u += 2) {   // (a B pair serves two MFMA groups: refilled once both are through)
                fma4(u);
                fma4(u + 1);
                fetch(u);
                fetch(u + 1);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) fma4(u);
    // sub-tile s = ih + 2 jh, as in k_small_mm
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        sR[w * 1024 + 0 * 256 + q * 64 + lane] = acc00[q];
        sR[w * 1024 + 1 * 256 + q * 64 + lane] = acc10[q];
        sR[w * 1024 + 2 * 256 + q * 64 + lane] = acc01[q];
        sR[w * 1024 + 3 * 256 + q * 64 + lane] = acc11[q];
    }
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int e = tid + 512 * half;
        double acc = 0.0;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) acc += sR[ww * 1024 + e];   // the eight slices in order
        v[half] = acc;
    }
}

template <bool SYM>
__global__ __launch_bounds__(512) void k_small_mm_blk(const double* __restrict__ A, const double* __restrict__ B,
                                                      double* __restrict__ C, int N, int nt, double alpha, double beta,
                                                      const double* __restrict__ Add, double gamma) {
    __shared__ double sR[8 * 1024];
    int ti, tj;
    double v[2];
    smm_blk_tile<SYM>(A, B, N, nt, sR, ti, tj, v);
    const int tid = threadIdx.x;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int e = tid + 512 * half;
        const int sub2 = e >> 8, q2 = (e >> 6) & 3, l2 = e & 63;
        const int i = ti * 32 + (sub2 & 1) * 16 + 4 * q2 + (l2 >> 4), j = tj * 32 + (sub2 >> 1) * 16 + (l2 & 15);
        double out = alpha * v[half] + (i == j ? beta : 0.0);
        if (Add) out += gamma * Add[i + (int64_t)j * N];
        if (SYM) {
            if (ti != tj || j <= i) {   // diagonal tile: (i, j) and (j, i) are both computed here - the lower one is kept
                C[i + (int64_t)j * N] = out;
                C[j + (int64_t)i * N] = out;
            }
        } else {
            C[i + (int64_t)j * N] = out;
        }
    }
}

// The blocked product over a BATCH of N x N problems (blockIdx.y; element strides sA / sB / sC / sAdd, 0 = shared operand): the
// spectrum slicer (sliced.hip) runs its sign iterations for several split points side by side - one matrix is 136 workgroups at
// N = 512, half the chip.  Same tile code, same summation order as k_small_mm_blk.
struct MfBatchCoef3 {   // per-problem coefficients of a batched product (at most 8 problems)
    double alpha[8], beta[8], gamma[8];
};
template <bool SYM, bool PERZ>
__global__ __launch_bounds__(512) void k_small_mm_blk_b(const double* __restrict__ A, int64_t sA, const double* __restrict__ B, int64_t sB,
                                                        double* __restrict__ C, int64_t sC, int N, int nt, double alpha, double beta,
                                                        const double* __restrict__ Add, int64_t sAdd, double gamma, MfBatchCoef3 cz) {
    __shared__ double sR[8 * 1024];
    const int z = blockIdx.y;
    if (PERZ) {
        alpha = cz.alpha[z];
        beta = cz.beta[z];
        gamma = cz.gamma[z];
    }
    A += (int64_t)z * sA;
    B += (int64_t)z * sB;
    C += (int64_t)z * sC;
    if (Add) Add += (int64_t)z * sAdd;
    int ti, tj;
    double v[2];
    smm_blk_tile<SYM>(A, B, N, nt, sR, ti, tj, v);
    const int tid = threadIdx.x;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int e = tid + 512 * half;
        const int sub2 = e >> 8, q2 = (e >> 6) & 3, l2 = e & 63;
        const int i = ti * 32 + (sub2 & 1) * 16 + 4 * q2 + (l2 >> 4), j = tj * 32 + (sub2 >> 1) * 16 + (l2 & 15);
        double out = alpha * v[half] + (i == j ? beta : 0.0);
        if (Add) out += gamma * Add[i + (int64_t)j * N];
        if (SYM) {
            if (ti != tj || j <= i) {
                C[i + (int64_t)j * N] = out;
                C[j + (int64_t)i * N] = out;
            }
        } else {
            C[i + (int64_t)j * N] = out;
        }
    }
}

// C_z = alpha A_z B_z + beta I (+ gamma Add_z), z < nb; N a multiple of 128, N <= 2048.  sym: commuting symmetric operands.
int small_mm_batched(Handle* h, const double* A, int64_t sA, const double* B, int64_t sB, double* C, int64_t sC, int64_t N, int nb,
                     double alpha, double beta, bool sym, const double* Add, int64_t sAdd, double gamma) {
    if ((N % 128) != 0 || N > 2048 || nb < 1) return set_err(h, TLSQ_ERR_ARG, "small_mm_batched: N = %lld", (long long)N);
    const int nt = (int)(N / 32);
    const MfBatchCoef3 none = {};
    if (sym)
        hipLaunchKernelGGL((k_small_mm_blk_b<true, false>), dim3((unsigned)(nt * (nt + 1) / 2), (unsigned)nb), dim3(512), 0, h->stream, A, sA,
                           B, sB, C, sC, (int)N, nt, alpha, beta, Add, sAdd, gamma, none);
    else
        hipLaunchKernelGGL((k_small_mm_blk_b<false, false>), dim3((unsigned)(nt * nt), (unsigned)nb), dim3(512), 0, h->stream, A, sA, B, sB,
                           C, sC, (int)N, nt, alpha, beta, Add, sAdd, gamma, none);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// C_z = A_{idx[z]} B for z < nb <= 32: the first operand of problem z is matrix idx[z] of the array at Abase (stride N^2), B is
// shared, the results are consecutive at C (stride N^2).  Commuting symmetric operands (lower tiles + mirror).  The slicer's
// children P_z Q of a level (sliced.hip): the parents' projectors sit in scattered slots.
struct MfBatchIdx {
    int32_t a[32];
};
__global__ __launch_bounds__(512) void k_small_mm_blk_bi(const double* __restrict__ Abase, MfBatchIdx ix, const double* __restrict__ B,
                                                         double* __restrict__ C, int N, int nt) {
    __shared__ double sR[8 * 1024];
    const int z = blockIdx.y;
    const int64_t nn = (int64_t)N * N;
    const double* A = Abase + (int64_t)ix.a[z] * nn;
    double* Cz = C + (int64_t)z * nn;
    int ti, tj;
    double v[2];
    smm_blk_tile<true>(A, B, N, nt, sR, ti, tj, v);
    const int tid = threadIdx.x;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int e = tid + 512 * half;
        const int sub2 = e >> 8, q2 = (e >> 6) & 3, l2 = e & 63;
        const int i = ti * 32 + (sub2 & 1) * 16 + 4 * q2 + (l2 >> 4), j = tj * 32 + (sub2 >> 1) * 16 + (l2 & 15);
        if (ti != tj || j <= i) {
            Cz[i + (int64_t)j * N] = v[half];
            Cz[j + (int64_t)i * N] = v[half];
        }
    }
}
int small_mm_batched_idx(Handle* h, const double* Abase, const int32_t* idx, int nb, const double* B, double* C, int64_t N) {
    if ((N % 128) != 0 || N > 2048 || nb < 1 || nb > 32) return set_err(h, TLSQ_ERR_ARG, "small_mm_batched_idx: N = %lld, nb = %d", (long long)N, nb);
    MfBatchIdx ix = {};
    for (int z = 0; z < nb; ++z) ix.a[z] = idx[z];
    const int nt = (int)(N / 32);
    hipLaunchKernelGGL(k_small_mm_blk_bi, dim3((unsigned)(nt * (nt + 1) / 2), (unsigned)nb), dim3(512), 0, h->stream, Abase, ix, B, C, (int)N, nt);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// X_z = alpha_z P_z K + beta_z I + gamma_z P_z for z < nb <= 8 (P_z, K commuting symmetric matrices; P_z: stride N^2, K shared):
// the start of the sign iteration of a slice whose spectral projector is P_z (sliced.hip)
int small_mm_batched_start(Handle* h, const double* P, const double* K, double* X, int64_t N, int nb, const double* alpha,
                           const double* beta, const double* gamma) {
    if ((N % 128) != 0 || N > 2048 || nb < 1 || nb > 8) return set_err(h, TLSQ_ERR_ARG, "small_mm_batched_start: N = %lld", (long long)N);
    const int nt = (int)(N / 32);
    MfBatchCoef3 cz = {};
    for (int z = 0; z < nb; ++z) {
        cz.alpha[z] = alpha[z];
        cz.beta[z] = beta[z];
        cz.gamma[z] = gamma[z];
    }
    const int64_t nn = N * N;
    hipLaunchKernelGGL((k_small_mm_blk_b<true, true>), dim3((unsigned)(nt * (nt + 1) / 2), (unsigned)nb), dim3(512), 0, h->stream, P, nn, K,
                       (int64_t)0, X, nn, (int)N, nt, 0.0, 0.0, P, nn, 0.0, cz);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// X_z = a_z K + b_z I for z < nb (nb <= 8): the shifted and scaled starts of a batch of sign iterations
struct MfBatchCoef {
    double a[8], b[8];
};
__global__ __launch_bounds__(256) void k_mf_axpbi_b(const double* __restrict__ K, double* __restrict__ X, int N, MfBatchCoef cf) {
    const int z = blockIdx.y;
    const int64_t total = (int64_t)N * N;
    double* Xz = X + (int64_t)z * total;
    const double a = cf.a[z], b = cf.b[z];
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e % N), j = (int)(e / N);
        Xz[e] = a * K[e] + (i == j ? b : 0.0);
    }
}

// out[4 z + {0, 1, 2, 3}] = trace(S_z), <S_z, K> = sum_ij S_ij K_ij, ||S_z||_F^2, <S_z, K2> (K2 optional: the square of K, for
// second moments) for z < nb: MF_SS_CHUNKS workgroups per matrix, their partial sums added in a fixed order by k_mf_slice_stats2
constexpr int MF_SS_CHUNKS = 32;
__global__ __launch_bounds__(1024) void k_mf_slice_stats(const double* __restrict__ S, const double* __restrict__ K,
                                                         const double* __restrict__ K2, int N, double* __restrict__ part) {
    __shared__ double red[4][16];
    const int z = blockIdx.x, ch = blockIdx.y;
    const int64_t total = (int64_t)N * N;
    const int64_t per = (total + MF_SS_CHUNKS - 1) / MF_SS_CHUNKS, e0 = ch * per, e1 = (e0 + per < total) ? e0 + per : total;
    const double* Sz = S + (int64_t)z * total;
    double tr = 0.0, ip = 0.0, fr = 0.0, ip2 = 0.0;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 1024) {
        const double v = Sz[e];
        const int i = (int)(e % N), j = (int)(e / N);
        if (i == j) tr += v;
        ip += v * K[e];
        fr += v * v;
        if (K2) ip2 += v * K2[e];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        tr += __shfl_down(tr, off, 64);
        ip += __shfl_down(ip, off, 64);
        fr += __shfl_down(fr, off, 64);
        ip2 += __shfl_down(ip2, off, 64);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][w] = tr;
        red[1][w] = ip;
        red[2][w] = fr;
        red[3][w] = ip2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
        for (int k = 0; k < 16; ++k) {
            a += red[0][k];
            b += red[1][k];
            c += red[2][k];
            d += red[3][k];
        }
        double* o = part + ((int64_t)z * MF_SS_CHUNKS + ch) * 4;
        o[0] = a;
        o[1] = b;
        o[2] = c;
        o[3] = d;
    }
}
// the chunks of a matrix added in order (one thread per matrix and statistic)
__global__ __launch_bounds__(64) void k_mf_slice_stats2(const double* __restrict__ part, int nb, double* __restrict__ out) {
    const int t = threadIdx.x;
    if (t >= 4 * nb) return;
    const int z = t >> 2, q = t & 3;
    double a = 0.0;
    for (int ch = 0; ch < MF_SS_CHUNKS; ++ch) a += part[((int64_t)z * MF_SS_CHUNKS + ch) * 4 + q];
    out[4 * z + q] = a;
}

// k_mf_slice_stats of nb <= 16 consecutive matrices at S (trace, <S, K>, ||S||_F^2, <S, K2>) into stats_dev (4 nb doubles)
int slice_stats_batched(Handle* h, const double* S, const double* K, const double* K2, int64_t N, int nb, double* stats_dev) {
    if (nb < 1 || nb > 16) return set_err(h, TLSQ_ERR_ARG, "slice_stats_batched: nb = %d", nb);
    void* part;
    TLSQ_TRY(ws_get(h, WS_MFP, (size_t)3 * 16 * 2048 * 8, &part));
    hipLaunchKernelGGL(k_mf_slice_stats, dim3((unsigned)nb, MF_SS_CHUNKS), dim3(1024), 0, h->stream, S, K, K2, (int)N, (double*)part);
    hipLaunchKernelGGL(k_mf_slice_stats2, dim3(1), dim3(64), 0, h->stream, (const double*)part, nb, stats_dev);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// sign(K - t_z I) for nb <= 8 split points t_z side by side, on a FIXED schedule (no convergence tests, no host round trip):
// X_0 = (K - t I) / max(hi - t, t) for a symmetric K with eigenvalues in [0, hi]; the growth quintic of matfun_sign until an
// eigenvalue that started at l0 has reached 0.6, minimax quintics (quintic.hpp) down to 1e-6, one Newton-Schulz step.  An
// eigenvalue of X_0 closer to zero than l0 arrives late: its direction is left between the two sides (the caller's Jacobi sweeps
// deal with it - sliced.hip).  X: nb N^2 doubles (the starts are written here; t == nullptr: they are there already, symmetric
// with eigenvalues in [-1, 1]), W1, W2: the same size each; *out = the buffer the signs are in (one of the three).  stats_dev
// (4 nb doubles) receives k_mf_slice_stats of the result (K2 optional).
int matfun_sign_batched(Handle* h, const double* K, const double* K2, int64_t N, int nb, const double* t, double hi, double l0, double* X,
                        double* W1, double* W2, double** out, double* stats_dev, int* steps_out) {
    if (nb < 1 || nb > 8 || (N % 128) != 0 || N > 2048) return set_err(h, TLSQ_ERR_ARG, "matfun_sign_batched: nb = %d, N = %lld", nb, (long long)N);
    const int64_t nn = N * N;
    if (t) {   // (t == nullptr: the caller has written the starts to X itself)
        MfBatchCoef cf;
        for (int z = 0; z < nb; ++z) {
            const double nrm = std::max(hi - t[z], t[z]);
            if (!(nrm > 0.0) || !std::isfinite(nrm)) return set_err(h, TLSQ_ERR_ARG, "matfun_sign_batched: split point %g of [0, %g]", t[z], hi);
            cf.a[z] = 1.0 / nrm;
            cf.b[z] = -t[z] / nrm;
        }
        int64_t g = (nn + 255) / 256;
        if (g > 1024) g = 1024;
        hipLaunchKernelGGL(k_mf_axpbi_b, dim3((unsigned)g, (unsigned)nb), dim3(256), 0, h->stream, K, X, (int)N, cf);
        TLSQ_HIP(h, hipGetLastError());
    }
    double *cur = X, *f1 = W1, *f2 = W2;
    int it = 0;
    auto quintic_step = [&](double a, double b, double c) -> int {
        TLSQ_TRY(small_mm_batched(h, cur, nn, cur, nn, f1, nn, N, nb, 1.0, 0.0, true, nullptr, 0, 0.0));   // S = X^2
        TLSQ_TRY(small_mm_batched(h, f1, nn, f1, nn, f2, nn, N, nb, c, a, true, f1, nn, b));               // T = c S^2 + b S + a I
        TLSQ_TRY(small_mm_batched(h, cur, nn, f2, nn, f1, nn, N, nb, 1.0, 0.0, true, nullptr, 0, 0.0));    // X T
        std::swap(cur, f1);
        ++it;
        return TLSQ_OK;
    };
    // image of [l, u] under p(x) = a x + b x^3 + c x^5 (sampled: the schedule only needs it to a per cent)
    auto image = [](double a, double b, double c, double& l, double& u) {
        double lo = 1e300, hi2 = -1e300;
        for (int k = 0; k <= 4000; ++k) {
            const double x = l + (u - l) * (double)k / 4000.0, x2 = x * x;
            const double v = x * (a + x2 * (b + c * x2));
            lo = std::min(lo, v);
            hi2 = std::max(hi2, v);
        }
        l = lo;
        u = hi2;
    };
    double l = std::min(std::max(l0, 1e-12), 0.5), u = 1.0;
    const double ga = 3.4445, gb = -4.7750, gc = 2.0315;
    const bool minimax_all = dev_is(DEV_SLICE_SCHED, 'm');
    if (!minimax_all)
        while (l < 0.60 && it < 40) {   // (the growth quintic takes [0.22, 1.2] into [0.68, 1.2]: its own minimum near x = 1.05)
            TLSQ_TRY(quintic_step(ga, gb, gc));
            image(ga, gb, gc, l, u);
        }
    for (int s = 0; s < 24 && it < 60; ++s) {
        if (std::max(1.0 - l, u - 1.0) < 1e-6) break;
        OddQuintic p;
        if (!odd_quintic(l, u, &p)) break;
        TLSQ_TRY(quintic_step(p.a, p.b, p.c));
        l = 1.0 - p.E;
        u = 1.0 + p.E;
    }
    // one Newton-Schulz step: X <- X (1.5 I - 0.5 X^2)
    TLSQ_TRY(small_mm_batched(h, cur, nn, cur, nn, f1, nn, N, nb, -0.5, 1.5, true, nullptr, 0, 0.0));
    TLSQ_TRY(small_mm_batched(h, cur, nn, f1, nn, f2, nn, N, nb, 1.0, 0.0, true, nullptr, 0, 0.0));
    std::swap(cur, f2);
    ++it;
    if (stats_dev) {
        void* part;
        TLSQ_TRY(ws_get(h, WS_MFP, (size_t)3 * 16 * 2048 * 8, &part));   // (the slot of mf_stats' partials, at its largest size: >= 8 x 32 x 4 doubles)
        hipLaunchKernelGGL(k_mf_slice_stats, dim3((unsigned)nb, MF_SS_CHUNKS), dim3(1024), 0, h->stream, (const double*)cur, K, K2, (int)N,
                           (double*)part);
        hipLaunchKernelGGL(k_mf_slice_stats2, dim3(1), dim3(64), 0, h->stream, (const double*)part, nb, stats_dev);
        TLSQ_HIP(h, hipGetLastError());
    }
    *out = cur;
    if (steps_out) *steps_out = it;
    return TLSQ_OK;
}

// ||S^2||_F^2 of the symmetric S, tile sums to the host-visible mailbox: k_sq_norm (subspace.hip) on the blocked tile kernel, for
// N a multiple of 128 - same mailbox layout and ticket protocol (slot 16 + tile, off-diagonal tiles count twice, the workgroup
// that arrives last publishes the sequence number)
__global__ __launch_bounds__(512) void k_sq_norm_blk(const double* __restrict__ S, int N, int nt, int ntile, double* mailbox,
                                                     unsigned int* ticket, double seq) {
    __shared__ double sR[8 * 1024];
    __shared__ double red[8];
    int ti, tj;
    double v[2];
    smm_blk_tile<true>(S, S, N, nt, sR, ti, tj, v);
    double sq = v[0] * v[0] + v[1] * v[1];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int q = 0; q < 8; ++q) tot += red[q];
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

bool sq_norm_blk_ok(int64_t N) { return (N % 128) == 0 && N <= 2048 && !dev_is(DEV_NO_SMALL_MM, '1') && !dev_is(DEV_NO_SMALL_MM, 'b'); }
int launch_sq_norm_blk(Handle* h, const double* S, int64_t N, double* mailbox_dev, unsigned int* ticket, double seq, int ntile) {
    const int nt = (int)(N / 32);
    hipLaunchKernelGGL(k_sq_norm_blk, dim3((unsigned)ntile), dim3(512), 0, h->stream, S, (int)N, nt, ntile, mailbox_dev, ticket, seq);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// C = alpha A B + beta I (N <= 2048; C must not alias A or B).  sym_out: A, B symmetric and commuting - the result is symmetric,
// lower tiles + mirror; otherwise a general product (A(i, k), B(k, j) as they stand)
// (+ gamma Add with Add != nullptr: one more N x N term in the epilogue; Add may be A or B)
static int small_mm(Handle* h, const double* A, const double* B, double* C, int64_t N, double alpha, double beta, bool sym_out,
                    const double* Add = nullptr, double gamma = 0.0) {
    const int nt = (int)((N + 31) / 32);
    const bool full = (N % 128) == 0;
    const dim3 gs((unsigned)(nt * (nt + 1) / 2)), gg((unsigned)(nt * nt));
    if (full && !dev_is(DEV_NO_SMALL_MM, 'b')) {   // (NO_SMALL_MM=b: the unblocked whole-tile kernels)
        if (sym_out) hipLaunchKernelGGL((k_small_mm_blk<true>), gs, dim3(512), 0, h->stream, A, B, C, (int)N, nt, alpha, beta, Add, gamma);
        else hipLaunchKernelGGL((k_small_mm_blk<false>), gg, dim3(512), 0, h->stream, A, B, C, (int)N, nt, alpha, beta, Add, gamma);
        TLSQ_HIP(h, hipGetLastError());
        return TLSQ_OK;
    }
    if (sym_out && full) hipLaunchKernelGGL((k_small_mm<true, true>), gs, dim3(1024), 0, h->stream, A, B, C, (int)N, nt, alpha, beta, Add, gamma);
    else if (sym_out) hipLaunchKernelGGL((k_small_mm<true, false>), gs, dim3(1024), 0, h->stream, A, B, C, (int)N, nt, alpha, beta, Add, gamma);
    else if (full) hipLaunchKernelGGL((k_small_mm<false, true>), gg, dim3(1024), 0, h->stream, A, B, C, (int)N, nt, alpha, beta, Add, gamma);
    else hipLaunchKernelGGL((k_small_mm<false, false>), gg, dim3(1024), 0, h->stream, A, B, C, (int)N, nt, alpha, beta, Add, gamma);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

constexpr int MF_CHUNKS = 16;
__global__ __launch_bounds__(256) void k_mf_stats1(const double* __restrict__ X, int N, double* __restrict__ part) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int per = (N + MF_CHUNKS - 1) / MF_CHUNKS;
    const int j0 = blockIdx.y * per, j1 = (j0 + per < N) ? j0 + per : N;
    double dev = 0.0, rs = 0.0, dg = 0.0;
    for (int j = j0; j < j1; ++j) {
        const double v = X[(size_t)j * N + i];   // row i read as X[i + j N]: consecutive threads, consecutive addresses
        const double dd = v - (i == j ? 1.0 : 0.0);
        dev += dd * dd;
        rs += fabs(v);
        if (i == j) dg = v;
    }
    const size_t o = (size_t)blockIdx.y * N + i;
    part[o] = dev;
    part[(size_t)MF_CHUNKS * N + o] = rs;
    part[(size_t)2 * MF_CHUNKS * N + o] = dg;
}
__global__ __launch_bounds__(1024) void k_mf_stats2(const double* __restrict__ part, int N, double* __restrict__ out) {
    __shared__ double red[3][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double dev = 0.0, tr = 0.0, rmax = 0.0;
    for (int i = tid; i < N; i += 1024) {
        double d = 0.0, r = 0.0, g = 0.0;
        for (int c = 0; c < MF_CHUNKS; ++c) {
            d += part[(size_t)c * N + i];
            r += part[(size_t)MF_CHUNKS * N + (size_t)c * N + i];
            g += part[(size_t)2 * MF_CHUNKS * N + (size_t)c * N + i];
        }
        dev += d;
        tr += g;
        rmax = r > rmax ? r : rmax;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        dev += __shfl_down(dev, off, 64);
        tr += __shfl_down(tr, off, 64);
        const double o = __shfl_down(rmax, off, 64);
        rmax = o > rmax ? o : rmax;
    }
    if (lane == 0) {
        red[0][w] = dev;
        red[1][w] = tr;
        red[2][w] = rmax;
    }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0, c = 0.0;
        for (int k = 0; k < 16; ++k) {
            a += red[0][k];
            b += red[1][k];
            c = red[2][k] > c ? red[2][k] : c;
        }
        out[0] = a;
        out[1] = b;
        out[2] = c;
    }
}

// Y = a1 X1 + a2 X2 + b I   (Y may be X1 or X2)
__global__ __launch_bounds__(256) void k_mf_lin2(const double* X1, double a1, const double* X2, double a2, double b, double* Y, int N) {
    const int64_t total = (int64_t)N * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e % N), j = (int)(e / N);
        Y[e] = a1 * X1[e] + a2 * X2[e] + (i == j ? b : 0.0);
    }
}

// PhiT[i, k] = F[i, k] + sum_s (w_s Xs[i, s] - Y[i, s]) Xs[k, s]: the transpose of Phi = (I - Xs Xs') F + Xs diag(w) Xs' for a
// symmetric F and Y = F Xs (N x r panels, ld N, r <= 32) - right-multiplying a panel by Phi takes the Xs directions out of F
// exactly (to the orthonormality of Xs) and puts the weighted dominant part back
__global__ __launch_bounds__(256) void k_mf_phi(const double* __restrict__ F, const double* __restrict__ Xs,
                                                const double* __restrict__ Y, SelWeights sw, int r, int N,
                                                double* __restrict__ PhiT) {
    const int64_t total = (int64_t)N * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e % N), k = (int)(e / N);
        double v = F[e];
        for (int s = 0; s < r; ++s) v += (sw.w[s] * Xs[i + (size_t)s * N] - Y[i + (size_t)s * N]) * Xs[k + (size_t)s * N];
        PhiT[e] = v;
    }
}

static int mf_axpbi(Handle* h, const double* X, double* Y, int64_t N, double a, double b) {
    int64_t g = (N * N + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_mf_axpbi, dim3((int)g), dim3(256), 0, h->stream, X, Y, (int)N, a, b);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// host copies of { ||X - I||_F^2, trace(X), ||X||_inf }
static int mf_stats(Handle* h, const double* X, int64_t N, double out[3]) {
    void* scal;
    TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &scal));
    double* dev = reinterpret_cast<double*>(reinterpret_cast<char*>(scal) + 1728);   // (behind the two sets of bound slots)
    void* part;
    TLSQ_TRY(ws_get(h, WS_MFP, (size_t)3 * MF_CHUNKS * N * 8, &part));
    hipLaunchKernelGGL(k_mf_stats1, dim3((unsigned)((N + 255) / 256), MF_CHUNKS), dim3(256), 0, h->stream, X, (int)N, (double*)part);
    hipLaunchKernelGGL(k_mf_stats2, dim3(1), dim3(1024), 0, h->stream, (const double*)part, (int)N, dev);
    TLSQ_HIP(h, hipGetLastError());
    TLSQ_HIP(h, hipMemcpyAsync(h->pinned, dev, 24, hipMemcpyDeviceToHost, h->stream));
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    memcpy(out, h->pinned, 24);
    return TLSQ_OK;
}

// C = A B for commuting symmetric N x N matrices (C symmetric)
static int mf_mul(Handle* h, const double* A, const double* B, double* C, int64_t N) {
    if (N <= 2048 && C != A && C != B && !dev_is(DEV_NO_SMALL_MM, '1')) return small_mm(h, A, B, C, N, 1.0, 0.0, true);
    return gemm_f64(h, true, true, A, N, B, N, C, N, N, N, N, true);
}
// C = A B in full (no symmetry assumed): C[i, j] = sum_k A[i, k] B[k, j], all three column-major.  The coupled iteration below
// is stable as written (Higham, Functions of Matrices, sec. 6.4); mirroring one triangle of Y T and T Z would replace its
// small commutator errors by errors of the same size in the iterates themselves (TLSQ_MATFUN_SYM=1: the symmetric form).
static int mf_mul_full(Handle* h, const double* A, const double* B, double* C, int64_t N) {
    const bool sym = dev_is(DEV_MATFUN_SYM, '1');
    if (sym) return mf_mul(h, A, B, C, N);
    // gemm convention: Cm[j + i ldc] = sum_k Aop(i, k) Bop(k, j); with Aop(i, k) = B[k + i ld] (B's column i) and
    // Bop(k, j) = A[j + k ld] (A's row j):  Cm[j + i ld] = sum_k A[j, k] B[k, i] = (A B)[j, i]
    return gemm_f64(h, true, false, B, N, A, N, C, N, N, N, N, false);
}

// X = sign(C) for the symmetric C (N x N): X_0 = C / ||C||_inf, X <- X (1.5 I - 0.5 X^2) until ||X^2 - I||_F <= 1e-4, then
// two more steps (quadratic convergence: 1e-8, 1e-16).  W1, W2: N x N scratch.  *iters: steps taken; *ok = false when the
// matrix has an eigenvalue so close to zero that `max_iters` steps do not separate it (each step moves it by 1.5x).
int matfun_sign(Handle* h, const double* C, int64_t N, double* X, double* W1, double* W2, int max_iters, int* iters, bool* ok) {
    *ok = false;
    *iters = 0;
    double st[3];
    TLSQ_TRY(mf_stats(h, C, N, st));
    const double nrm = st[2];
    if (!(nrm > 0.0) || !std::isfinite(nrm)) return TLSQ_OK;
    TLSQ_TRY(mf_axpbi(h, C, X, N, 1.0 / nrm, 0.0));
    int extra = -1;   // steps still to do after the error dropped below 1e-4 (-1: not yet)
    // the iterate moves through the three buffers (one copy at the end at most): cur, and two free ones
    double *cur = X, *f1 = W1, *f2 = W2;
    const bool small = N <= 2048 && !dev_is(DEV_NO_SMALL_MM, '1');   // (k_small_mm: no slabs, the element-wise step in the epilogue)
    int it = 0;
    int first_test = 6;   // the convergence test costs a host round trip: not before the linear phase can be over
    // How close to zero an eigenvalue of X_0 may lie and still be given a sign: `max_iters` Newton-Schulz steps multiply it by
    // 1.5^max_iters at most (70 steps: 2e12) - anything closer does not converge and the caller decides by other means (an
    // eigenvalue within rounding of the threshold must not be counted by its rounding error).  The quintics grow 3.44 per step:
    // the same bound is kept on the product of the slopes at zero, not on the step count (tools/fuzz_parity.py 0, case 199:
    // 300 x 64, a value 1e-12 from the threshold "converged" in 36 steps and was counted on the wrong side).
    const double growth_cap = std::pow(1.5, (double)max_iters);
    double growth = 1.0;
    // ---- phase 1: odd quintics ----
    // The spectrum of X_0 lies in [-1, 1]; what is known about its distance from zero is the cluster at -1 / nrm (the null
    // directions of the deflated panel, C = -I there) - and that a noise bulk crosses the threshold with a few hundred
    // eigenvalues, the nearest of which sits ~1e-3 / mu^2 from it (measured: Newton-Schulz needed 12 more steps behind a
    // schedule for 0.1 / nrm).  Eigenvalues closer to zero than l = 1e-3 / nrm just arrive later, in phase 2.
    // Three products per step:   S = X^2,   T = c S^2 + b S + a I,   X <- X T.
    //  (a) growth: p(x) = 3.4445 x - 4.7750 x^3 + 2.0315 x^5 multiplies an eigenvalue near zero by 3.44 and keeps whatever has
    //      reached [0.22, 1.13] inside [0.70, 1.13] (Newton-Schulz: 1.5 per step of two products).  The minimax quintic of the
    //      whole interval [l, 1] (quintic.hpp) would grow by 8.5, then 4.26 per step - but it equioscillates: it sends
    //      eigenvalues from the top of the spectrum down to the level of the smallest one, where rounding mixes their
    //      eigenvectors with the other side of zero (measured: A to 5e-11 instead of 3e-13 at 500 x 200, rank 45);
    //  (b) the minimax quintics of [0.70, 1.13] and of its image: two steps to ~1e-7.
    if (small && !dev_is(DEV_NO_QUINTIC, '1')) {
        auto quintic_step = [&](double a, double b, double c) -> int {
            TLSQ_TRY(small_mm(h, cur, cur, f1, N, 1.0, 0.0, true));          // S
            TLSQ_TRY(small_mm(h, f1, f1, f2, N, c, a, true, f1, b));         // T
            TLSQ_TRY(small_mm(h, cur, f2, f1, N, 1.0, 0.0, true));           // X T
            std::swap(cur, f1);
            ++it;
            growth *= a;
            return TLSQ_OK;
        };
        double l = std::min(std::max(1e-3 / nrm, 1e-14), 0.25);
        const bool minimax_all = dev_is(DEV_NO_QUINTIC, 'm');   // (experiment: the equioscillating schedule from the start)
        double u = 1.0;
        if (!minimax_all) {
            const double ga = 3.4445, gb = -4.7750, gc = 2.0315;
            while (l < 0.22 && it < max_iters) {
                TLSQ_TRY(quintic_step(ga, gb, gc));
                l = l * (ga + l * l * (gb + gc * l * l));
            }
            l = 0.70;
            u = 1.135;
        }
        for (int s = 0; s < 24 && it < max_iters; ++s) {
            OddQuintic p;
            if (!odd_quintic(l, u, &p)) break;
            TLSQ_TRY(quintic_step(p.a, p.b, p.c));
            l = 1.0 - p.E;
            u = 1.0 + p.E;
            if (p.E < 1e-3) break;
        }
        first_test = it + 1;   // (one classical step takes an error of 1e-3 to 1e-6, the tested one to 1e-12)
    }
    // ---- phase 2: Newton-Schulz steps X <- X (1.5 I - 0.5 X^2) with convergence tests ----
    for (; it < max_iters; ++it) {
        if (extra == 0) break;
        if (extra < 0 && growth * 1.5 > growth_cap) break;   // (too close to zero to call: not converged)
        growth *= 1.5;
        const bool test_now = extra < 0 && it >= first_test && ((it - first_test) % 3 == 0 || first_test != 6);
        if (small && !test_now) {
            TLSQ_TRY(small_mm(h, cur, cur, f1, N, -0.5, 1.5, true));   // 1.5 I - 0.5 X^2 in one launch
        } else {
            if (small) TLSQ_TRY(small_mm(h, cur, cur, f1, N, 1.0, 0.0, true));
            else TLSQ_TRY(mf_mul(h, cur, cur, f1, N));        // X^2
            if (test_now) {
                TLSQ_TRY(mf_stats(h, f1, N, st));
                if (!std::isfinite(st[0])) return TLSQ_OK;
                if (st[0] <= 1e-8) extra = st[0] <= 1e-24 ? 0 : (st[0] <= 1e-16 ? 1 : 2);
                if (extra == 0) break;
            }
            TLSQ_TRY(mf_axpbi(h, f1, f1, N, -0.5, 1.5));      // 1.5 I - 0.5 X^2
        }
        if (small) TLSQ_TRY(small_mm(h, cur, f1, f2, N, 1.0, 0.0, true));
        else TLSQ_TRY(mf_mul(h, cur, f1, f2, N));             // X <- X (1.5 I - 0.5 X^2)
        std::swap(cur, f2);
        if (extra > 0) --extra;
    }
    if (extra != 0) return TLSQ_OK;
    if (cur != X) TLSQ_HIP(h, hipMemcpyAsync(X, cur, (size_t)N * N * 8, hipMemcpyDeviceToDevice, h->stream));
    *iters = it;
    *ok = true;
    return TLSQ_OK;
}

// Z = B^(-1/2) for the symmetric positive definite B (N x N) with eigenvalues in [lo, hi] (hi: any upper bound): coupled
// Newton-Schulz  T = 1.5 I - 0.5 Z Y,  Y <- Y T,  Z <- T Z  from Y_0 = B / hi, Z_0 = I; Z -> (B / hi)^(-1/2).  Y, T, W: scratch.
int matfun_invsqrt(Handle* h, const double* B, int64_t N, double hi, double* Z, double* Y, double* T, double* W, int max_iters,
                   int* iters, bool* ok) {
    *ok = false;
    *iters = 0;
    if (!(hi > 0.0) || !std::isfinite(hi)) return TLSQ_OK;
    TLSQ_TRY(mf_axpbi(h, B, Y, N, 1.0 / hi, 0.0));
    TLSQ_TRY(mf_axpbi(h, B, Z, N, 0.0, 1.0));
    int extra = -1;
    double st[3];
    // (k_small_mm, full results: three launches per step - the element-wise step rides in the first product's epilogue and the
    //  iterates change buffers instead of being copied back - against nine through the tiled GEMM)
    const bool small = N <= 2048 && !dev_is(DEV_NO_SMALL_MM, '1') && !dev_is(DEV_MATFUN_SYM, '1');
    double *Zc = Z, *Yc = Y, *Wc = W;   // current Z, current Y, the free buffer
    int it = 0;
    int first_test = 3;
    // ---- phase 1: the coupled iteration with the optimal quintics' even parts ----
    // M = Z Y carries m = x^2 for x on the sign iteration's path (m_0 = the eigenvalues of B / hi, in [lo / hi, 1]):
    // T = q(M) = a I + b M + c M^2,  Y <- Y T,  Z <- T Z  gives  m <- m q(m)^2 = p(sqrt(m))^2  for the odd quintic
    // p(x) = x q(x^2) - four products per step, and the smallest eigenvalue grows by a^2 ~ 18 - 72 instead of 2.25 for three.
    // The caller's B has no eigenvalue below 1 (B = P G2 mu^2 P + I - P): the schedule is made for x in [1 / sqrt(hi), 1].
    if (small && !dev_is(DEV_NO_QUINTIC, '1') && hi >= 4.0) {
        double l = std::min(std::max(0.9 / std::sqrt(hi), 1e-7), 0.5), u = 1.0;
        for (int s = 0; s < 24 && it < max_iters; ++s) {
            OddQuintic p;
            if (!odd_quintic(l, u, &p)) break;
            TLSQ_TRY(small_mm(h, Zc, Yc, T, N, 1.0, 0.0, false));                // M = Z Y
            TLSQ_TRY(small_mm(h, T, T, Wc, N, p.c, p.a, false, T, p.b));         // q(M) = c M^2 + b M + a I
            TLSQ_TRY(small_mm(h, Yc, Wc, T, N, 1.0, 0.0, false));                // Y q(M)
            std::swap(Yc, T);
            TLSQ_TRY(small_mm(h, Wc, Zc, T, N, 1.0, 0.0, false));                // q(M) Z
            std::swap(Zc, T);
            ++it;
            l = 1.0 - p.E;
            u = 1.0 + p.E;
            if (p.E < 1e-3) break;
        }
        first_test = it + 2;
    }
    // ---- phase 2: coupled Newton-Schulz steps with convergence tests ----
    for (; it < max_iters; ++it) {
        if (extra == 0) break;
        const bool test_now = extra < 0 && it >= first_test;
        if (small && !test_now) {
            TLSQ_TRY(small_mm(h, Zc, Yc, T, N, -0.5, 1.5, false));   // T = 1.5 I - 0.5 Z Y
        } else {
            if (small) TLSQ_TRY(small_mm(h, Zc, Yc, T, N, 1.0, 0.0, false));
            else TLSQ_TRY(mf_mul_full(h, Zc, Yc, T, N));         // T = Z Y  (-> I)
            if (test_now) {
                TLSQ_TRY(mf_stats(h, T, N, st));
                if (!std::isfinite(st[0])) return TLSQ_OK;
                if (st[0] <= 1e-8) extra = st[0] <= 1e-24 ? 0 : (st[0] <= 1e-16 ? 1 : 2);
                if (extra == 0) break;
            }
            TLSQ_TRY(mf_axpbi(h, T, T, N, -0.5, 1.5));        // T = 1.5 I - 0.5 Z Y
        }
        if (small) {
            TLSQ_TRY(small_mm(h, Yc, T, Wc, N, 1.0, 0.0, false));    // Y <- Y T
            std::swap(Yc, Wc);
            TLSQ_TRY(small_mm(h, T, Zc, Wc, N, 1.0, 0.0, false));    // Z <- T Z
            std::swap(Zc, Wc);
        } else {
            TLSQ_TRY(mf_mul_full(h, Yc, T, Wc, N));           // Y <- Y T
            TLSQ_HIP(h, hipMemcpyAsync(Yc, Wc, (size_t)N * N * 8, hipMemcpyDeviceToDevice, h->stream));
            TLSQ_TRY(mf_mul_full(h, T, Zc, Wc, N));           // Z <- T Z
            TLSQ_HIP(h, hipMemcpyAsync(Zc, Wc, (size_t)N * N * 8, hipMemcpyDeviceToDevice, h->stream));
        }
        if (extra > 0) --extra;
    }
    if (extra != 0) return TLSQ_OK;
    TLSQ_TRY(mf_axpbi(h, Zc, Z, N, 1.0 / std::sqrt(hi), 0.0));   // (into the caller's buffer, wherever the iterate sits)
    *iters = it;
    *ok = true;
    return TLSQ_OK;
}

// S = G / trace(G) for a symmetric positive semi-definite G (eigenvalues of S in [0, 1], the largest >= 1 / N): the trace in a
// fixed order by every workgroup itself, no host round trip
__global__ __launch_bounds__(256) void k_mf_scale_by_trace(const double* __restrict__ G, double* __restrict__ S, int N) {
    __shared__ double red[4];
    double t = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) t += G[(size_t)i * N + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    const double tr = (red[0] + red[1]) + (red[2] + red[3]);
    const double sc = tr > 0.0 ? 1.0 / tr : 0.0;
    const int64_t total = (int64_t)N * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) S[e] = G[e] * sc;
}

// (G / trace G)^(2^levels) by repeated squaring through k_small_mm (P1, P2: N x N scratch; *out = the buffer the result is in).
// Without rescaling between the squarings: the dominant eigenvalue of G / trace G is at least 1 / N, so five squarings of an
// N <= 1024 matrix stay above 1e-97 (beyond five the power is scaled by its trace again).  false in *ok: N is not one of the sizes k_small_mm_blk serves - the caller keeps its own form.
int matfun_power_start(Handle* h, const double* G, int64_t N, double* P1, double* P2, int levels, const double** out, bool* ok) {
    *ok = false;
    if ((N % 128) != 0 || N > 1024 || levels < 1 || levels > 10 || dev_is(DEV_NO_SMALL_MM, '1')) return TLSQ_OK;
    int64_t g = (N * N + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_mf_scale_by_trace, dim3((int)g), dim3(256), 0, h->stream, G, P1, (int)N);
    TLSQ_HIP(h, hipGetLastError());
    double *src = P1, *dst = P2;
    for (int k = 0; k < levels; ++k) {
        if (k > 0 && k % 5 == 0) {   // (five squarings take the dominant eigenvalue down to N^-32 at worst: scale by the trace again)
            hipLaunchKernelGGL(k_mf_scale_by_trace, dim3((int)g), dim3(256), 0, h->stream, (const double*)src, dst, (int)N);
            TLSQ_HIP(h, hipGetLastError());
            std::swap(src, dst);
        }
        TLSQ_TRY(small_mm(h, src, src, dst, N, 1.0, 0.0, true));
        std::swap(src, dst);
    }
    *out = src;
    *ok = true;
    return TLSQ_OK;
}

// C = A^2 for a symmetric A through k_small_mm (*ok = false: N is not one of its sizes)
int matfun_square(Handle* h, const double* A, double* C, int64_t N, bool* ok) {
    *ok = false;
    if (N > 2048 || dev_is(DEV_NO_SMALL_MM, '1')) return TLSQ_OK;
    *ok = true;
    return small_mm(h, A, A, C, N, 1.0, 0.0, true);
}

int matfun_trace_norm(Handle* h, const double* X, int64_t N, double* trace, double* norm_inf) {
    double st[3];
    TLSQ_TRY(mf_stats(h, X, N, st));
    if (trace) *trace = st[1];
    if (norm_inf) *norm_inf = st[2];
    return TLSQ_OK;
}

int matfun_phi(Handle* h, const double* F, const double* Xs, const double* Y, const SelWeights& sw, int64_t r, int64_t N,
               double* PhiT) {
    int64_t g = (N * N + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_mf_phi, dim3((int)g), dim3(256), 0, h->stream, F, Xs, Y, sw, (int)r, (int)N, PhiT);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
int matfun_stats(Handle* h, const double* X, int64_t N, double out[3]) { return mf_stats(h, X, N, out); }
int matfun_axpbi(Handle* h, const double* X, double* Y, int64_t N, double a, double b) { return mf_axpbi(h, X, Y, N, a, b); }
int matfun_lin2(Handle* h, const double* X1, double a1, const double* X2, double a2, double b, double* Y, int64_t N) {
    int64_t g = (N * N + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_mf_lin2, dim3((int)g), dim3(256), 0, h->stream, X1, a1, X2, a2, b, Y, (int)N);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
int matfun_mul(Handle* h, const double* A, const double* B, double* C, int64_t N) { return mf_mul(h, A, B, C, N); }

}  // namespace tlsq
