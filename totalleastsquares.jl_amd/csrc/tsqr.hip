// Tall-skinny Householder QR (communication-avoiding: TSQR over row blocks inside a blocked right-looking
// factorisation) -> the N x N triangular factor R of an M x N panel, Z = Q R.
//
// Why: LAPACK's `gesdd` behind `svd!(Z)` / `opnorm` (src/robustPCA.jl:194,225 under /root/reference) is backward
// stable: its singular values carry an absolute error of a few eps * sigma_max.  The Gram route (G = Z'Z, gemm.hip)
// squares the condition number — singular values near the rank threshold 1/mu (src/robustPCA.jl:198) are only
// known to eps * sigma_max^2 / sigma.  Orthogonal reduction keeps the full accuracy: the singular values of R are
// those of Z, the right singular vectors too, and the one-sided Jacobi solver (jacobi.hip) runs on R' directly.
// Q is never formed (the loop only needs V and sigma; U = Z V / sigma where it is asked for).
//
// Algorithm (per 32-column panel, left to right; W is a private fp64 copy of Z):
//   level 0   k_qr_factor: every workgroup stages 256 rows x 32 panel columns in LDS, runs 32 Householder steps
//             there (wave-shuffle column norms and v'a dot products, one barrier per step), and leaves
//             Y (reflectors), T (compact-WY factor, dlarft recurrence on Y'Y) in compact buffers and its 32 x 32
//             R_i in place;   k_qr_apply: C <- (I - Y T Y')' C on the trailing columns, 256-row x 32-column tiles,
//             both contractions on v_mfma_f64_16x16x4_f64 with Y and C staged in LDS
//   level l   the R_i of level l-1 (top 32 rows of each 256-row block) form the next, 8x shorter, tall matrix:
//             same two kernels on a strided row map  phys(g) = base + (g / 32) * S_l + g % 32,  S_l = 32 * 8^l
//   until one block is left: its R_i is rows c0..c0+31 of R, its trailing rows are R[c0:c0+32, c0+32:].
// Row shards (communicator): every rank reduces its shard to R_r, one ncclAllGather of the N x N factors, and the
// stacked (nranks N) x N matrix is reduced again — redundantly and bit-identically on every rank.
#include "internal.hpp"

namespace tlsq {

namespace {

constexpr int QB = 32;          // panel width
constexpr int QRB = 256;        // rows per level block
constexpr int QLD = QRB + 2;    // LDS leading dimension: (c * 258 + r) distinct mod 32 over a half-wave of (c, r..r+1)
constexpr int QT = QB + 1;

typedef double d4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ double q_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double q_lane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// wave-wide sum, every lane gets the total: four DPP steps inside each row of 16 lanes, then the four row totals
__device__ __forceinline__ double q_allsum(double v) {
    v += q_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
    v += q_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
    v += q_dpp<0x141>(v);   // row_half_mirror
    v += q_dpp<0x140>(v);   // row_mirror
    return (q_lane(v, 0) + q_lane(v, 16)) + (q_lane(v, 32) + q_lane(v, 48));
}
__device__ __forceinline__ int64_t q_phys(int64_t g, int64_t base, int64_t S) { return base + (g / QB) * S + (g % QB); }

// W (M x Np, ld ldW, fp64) = [Z, 0]: private working copy, columns N..Np-1 are zero padding up to a whole panel
template <typename T>
__global__ __launch_bounds__(256) void k_qr_copy_in(const T* __restrict__ Z, int64_t ldz, int64_t M, int64_t N, int64_t Np,
                                                    double* __restrict__ W, int64_t ldW) {
    const int64_t total = M * Np;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t r = e % M, c = e / M;
        W[r + c * ldW] = c < N ? (double)Z[r + c * ldz] : 0.0;
    }
}

// One level block: Householder QR of (up to) 256 rows x 32 columns in LDS.
// in : the block's rows of panel columns c0..c0+31 (through the row map)
// out: R_i (upper triangular, explicit zeros below) in the block's first min(32, rows) rows of the panel columns,
//      Y_i -> Ybuf[blk][32][256] (unit lower trapezoidal, zero rows beyond the block), T_i -> Tbuf[blk] (32 x 32, ld 32)
__global__ __launch_bounds__(1024) void k_qr_factor(double* __restrict__ W, int64_t ldW, int64_t base, int64_t S, int64_t m,
                                                    int64_t c0, double* __restrict__ Ybuf, double* __restrict__ Tbuf) {
    extern __shared__ __attribute__((aligned(16))) double qsm[];
    double* sA = qsm;                  // [QB][QLD] panel, column-major
    double* sY = sA + QB * QLD;        // [QB][QLD] reflectors
    double* sS = sY + QB * QLD;        // [QB][QT]  Y'Y (strict upper part)
    double* sTau = sS + QB * QT;       // [QB]
    double* sBeta = sTau + QB;         // [QB]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t g0 = (int64_t)blockIdx.x * QRB;
    const int rows = (int)((m - g0) < (int64_t)QRB ? (m - g0) : (int64_t)QRB);
    for (int e = tid; e < QB * QRB; e += 1024) {
        const int r = e % QRB, c = e / QRB;
        double v = 0.0;
        if (r < rows) v = W[q_phys(g0 + r, base, S) + (c0 + c) * ldW];
        sA[c * QLD + r] = v;
        sY[c * QLD + r] = 0.0;
    }
    __syncthreads();
    for (int k = 0; k < QB; ++k) {
        // every wave forms the reflector of column k for itself (no hand-off, no barrier): dlarfg
        double x[4], v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = sA[k * QLD + lane + 64 * i];
        const double alpha = sA[k * QLD + k];
        double sig = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (lane + 64 * i > k) sig += x[i] * x[i];
        sig = q_allsum(sig);
        double tau = 0.0, beta = alpha, scale = 0.0;
        if (sig != 0.0) {
            const double nrm = sqrt(alpha * alpha + sig);
            beta = alpha >= 0.0 ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lane + 64 * i;
            v[i] = r > k ? x[i] * scale : (r == k ? 1.0 : 0.0);
        }
        if (w == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sY[k * QLD + lane + 64 * i] = v[i];
            if (lane == 0) {
                sTau[k] = tau;
                sBeta[k] = beta;
            }
        }
        // H_k applied to the remaining panel columns, one wave per column: a <- a - tau (v'a) v
        for (int j = k + 1 + w; j < QB; j += 16) {
            double a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sA[j * QLD + lane + 64 * i];
            double d = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) d += v[i] * a[i];
            d = q_allsum(d);
            const double f = tau * d;
#pragma unroll
            for (int i = 0; i < 4; ++i) sA[j * QLD + lane + 64 * i] = a[i] - f * v[i];
        }
        __syncthreads();
    }
    // S = Y'Y, strict upper part: one wave per pair
    for (int i = 0; i < QB - 1; ++i) {
        double yi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) yi[q] = sY[i * QLD + lane + 64 * q];
        for (int j = i + 1 + w; j < QB; j += 16) {
            double d = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) d += yi[q] * sY[j * QLD + lane + 64 * q];
            d = q_allsum(d);
            if (lane == 0) sS[i * QT + j] = d;
        }
    }
    __syncthreads();
    const int64_t blk = blockIdx.x;
    if (w == 0) {
        // dlarft, forward / columnwise: T[0:k,k] = -tau_k T[0:k,0:k] (Y[:,0:k]' y_k); lane i owns row i of T in registers
        double trow[QB];
#pragma unroll
        for (int k = 0; k < QB; ++k) trow[k] = 0.0;
#pragma unroll
        for (int k = 0; k < QB; ++k) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < k; ++c) acc += trow[c] * sS[c * QT + k];
            const double tk = sTau[k];
            trow[k] = (lane == k) ? tk : -tk * acc;
        }
        if (lane < QB) {
#pragma unroll
            for (int k = 0; k < QB; ++k) Tbuf[blk * (QB * QB) + k * QB + lane] = trow[k];
        }
    }
    double* Yb = Ybuf + blk * (int64_t)(QB * QRB);
    for (int e = tid; e < QB * QRB; e += 1024) {
        const int r = e % QRB, c = e / QRB;
        Yb[e] = sY[c * QLD + r];
    }
    for (int e = tid; e < QB * QB; e += 1024) {
        const int r = e % QB, c = e / QB;
        if (r < rows) {
            const double v = r < c ? sA[c * QLD + r] : (r == c ? sBeta[c] : 0.0);
            W[q_phys(g0 + r, base, S) + (c0 + c) * ldW] = v;
        }
    }
}

// C <- (I - Y T Y')' C = C - Y (T' (Y' C))  for one level block (256 rows) x 32 trailing columns
__global__ __launch_bounds__(256) void k_qr_apply(double* __restrict__ W, int64_t ldW, int64_t base, int64_t S, int64_t m,
                                                  int64_t col0, const double* __restrict__ Ybuf,
                                                  const double* __restrict__ Tbuf) {
    extern __shared__ __attribute__((aligned(16))) double qsm[];
    double* sY = qsm;                  // [QB][QLD]
    double* sC = sY + QB * QLD;        // [32][QLD]
    double* sT = sC + QB * QLD;        // [QB][QT]
    double* sW = sT + QB * QT;         // [QB][QT]   Y'C
    double* sW2 = sW + QB * QT;        // [QB][QT]   T'(Y'C)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int64_t blk = blockIdx.x;
    const int64_t g0 = blk * QRB;
    const int rows = (int)((m - g0) < (int64_t)QRB ? (m - g0) : (int64_t)QRB);
    const int64_t cb = col0 + (int64_t)blockIdx.y * 32;
    const double* Yb = Ybuf + blk * (int64_t)(QB * QRB);
    for (int e = tid; e < QB * QRB; e += 256) {
        const int r = e % QRB, c = e / QRB;
        sY[c * QLD + r] = Yb[e];
        sC[c * QLD + r] = r < rows ? W[q_phys(g0 + r, base, S) + (cb + c) * ldW] : 0.0;
    }
    for (int e = tid; e < QB * QB; e += 256) {
        const int i = e % QB, k = e / QB;
        sT[i * QT + k] = Tbuf[blk * (QB * QB) + e];
    }
    __syncthreads();
    {   // Wm = Y' C  (32 x 32): one 16 x 16 tile per wave, K = 256 rows
        const int ty = w & 1, tj = w >> 1;
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
        const double* pa = sY + (ty * 16 + fr) * QLD + fk;
        const double* pb = sC + (tj * 16 + fr) * QLD + fk;
#pragma unroll 8
        for (int ks = 0; ks < QRB / 4; ++ks)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * ks], pb[4 * ks], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) sW[(ty * 16 + fk + 4 * q) * QT + tj * 16 + fr] = acc[q];
    }
    __syncthreads();
    {   // Wm2 = T' Wm,  T upper triangular
        const int j = tid & 31;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int a = (tid >> 5) + 8 * i;
            double s = 0.0;
            for (int c = 0; c <= a; ++c) s += sT[c * QT + a] * sW[c * QT + j];
            sW2[a * QT + j] = s;
        }
    }
    __syncthreads();
    // C -= Y Wm2: 16 row tiles x 2 column tiles, K = 32
    for (int t = w; t < 32; t += 4) {
        const int tr = t >> 1, tj = t & 1;
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < QB / 4; ++ks) {
            const int yk = 4 * ks + fk;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sW2[yk * QT + tj * 16 + fr], sY[yk * QLD + tr * 16 + fr], acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) sC[(tj * 16 + fk + 4 * q) * QLD + tr * 16 + fr] -= acc[q];
    }
    __syncthreads();
    for (int e = tid; e < QB * QRB; e += 256) {
        const int r = e % QRB, c = e / QRB;
        if (r < rows) W[q_phys(g0 + r, base, S) + (cb + c) * ldW] = sC[c * QLD + r];
    }
}

// B (N x N, ld N) = R' as a lower triangular matrix (explicit zeros above), R = upper triangle of W[0:N, 0:N]
__global__ __launch_bounds__(256) void k_qr_extract_lt(const double* __restrict__ W, int64_t ldW, int64_t N,
                                                       double* __restrict__ B) {
    const int64_t total = N * N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t r = e % N, c = e / N;
        B[e] = r >= c ? W[c + r * ldW] : 0.0;
    }
}

// gathered factors [rank][Np cols][N rows] -> stacked (nranks N) x Np matrix, ld = nranks N; only the upper triangles
// are meaningful (the rest of a rank's W block is reflector debris): everything below is written as zero
__global__ __launch_bounds__(256) void k_qr_stack(const double* __restrict__ gathered, int64_t N, int64_t Np, int nranks,
                                                  double* __restrict__ Wst) {
    const int64_t per = N * Np, total = per * nranks;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t ld = N * nranks;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t rk = e / per, q = e % per, r = q % N, c = q / N;
        Wst[rk * N + r + c * ld] = r <= c ? gathered[e] : 0.0;
    }
}

// this rank's factor: rows 0..N-1 of W (ld ldW) -> contiguous N x Np block
__global__ __launch_bounds__(256) void k_qr_pack(const double* __restrict__ W, int64_t ldW, int64_t N, int64_t Np,
                                                 double* __restrict__ out) {
    const int64_t total = N * Np;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t r = e % N, c = e / N;
        out[e] = W[r + c * ldW];
    }
}

constexpr size_t kFactorLds = (size_t)(2 * QB * QLD + QB * QT + 2 * QB) * 8;
constexpr size_t kApplyLds = (size_t)(2 * QB * QLD + 3 * QB * QT) * 8;

inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

// in-place reduction of W (Mw x Np, ld ldW, Np a multiple of 32) to its triangular factor: on return the upper
// triangle of W[0:min(Mw,Np), :] is R
int tsqr_inplace(Handle* h, double* W, int64_t ldW, int64_t Mw, int64_t Np) {
    TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_qr_factor), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kFactorLds));
    TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_qr_apply), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kApplyLds));
    const int64_t nblk0 = (Mw + QRB - 1) / QRB;
    void *Yb, *Tb;
    TLSQ_TRY(ws_get(h, WS_QRY, (size_t)nblk0 * QB * QRB * 8, &Yb));
    TLSQ_TRY(ws_get(h, WS_QRT, (size_t)nblk0 * QB * QB * 8, &Tb));
    for (int64_t c0 = 0; c0 < Np && c0 < Mw; c0 += QB) {
        const int64_t ntrail = Np - c0 - QB;
        int64_t m = Mw - c0, S = QB;
        for (;;) {
            const int64_t nblk = (m + QRB - 1) / QRB;
            hipLaunchKernelGGL(k_qr_factor, dim3((unsigned)nblk), dim3(1024), kFactorLds, h->stream, W, ldW, c0, S, m, c0,
                               (double*)Yb, (double*)Tb);
            if (ntrail > 0)
                hipLaunchKernelGGL(k_qr_apply, dim3((unsigned)nblk, (unsigned)(ntrail / QB)), dim3(256), kApplyLds, h->stream,
                                   W, ldW, c0, S, m, c0 + QB, (const double*)Yb, (const double*)Tb);
            TLSQ_HIP(h, hipGetLastError());
            if (nblk == 1) break;
            const int64_t last = m - (nblk - 1) * QRB;
            m = (nblk - 1) * QB + (last < QB ? last : QB);
            S *= (QRB / QB);
        }
    }
    return TLSQ_OK;
}

}  // namespace

// B (N x N, ld N, device) = R' (lower triangular) with Z = Q R; Z is M x N (ld ldz), fp32 or fp64, not modified.
// With a communicator Z is this rank's row shard and R is the factor of the whole stacked matrix.
int tsqr_lt(Handle* h, const void* Z, int z_f32, int64_t M, int64_t N, int64_t ldz, double* B) {
    if (M <= 0 || N <= 0) return set_err(h, TLSQ_ERR_ARG, "tsqr: empty panel");
    const int64_t Np = (N + QB - 1) / QB * QB;
    const int64_t Mw = std::max<int64_t>(M, 1);
    void* Wv;
    TLSQ_TRY(ws_get(h, WS_QRW, (size_t)Mw * Np * 8, &Wv));
    double* W = (double*)Wv;
    if (z_f32)
        hipLaunchKernelGGL((k_qr_copy_in<float>), dim3(grid_for(Mw * Np)), dim3(256), 0, h->stream, (const float*)Z, ldz, M, N,
                           Np, W, Mw);
    else
        hipLaunchKernelGGL((k_qr_copy_in<double>), dim3(grid_for(Mw * Np)), dim3(256), 0, h->stream, (const double*)Z, ldz, M,
                           N, Np, W, Mw);
    TLSQ_HIP(h, hipGetLastError());
    TLSQ_TRY(tsqr_inplace(h, W, Mw, Mw, Np));
    const double* Wr = W;
    int64_t ldr = Mw;
    if (h->comm) {
        // every rank contributes rows 0..N-1 of its reduced shard (shards shorter than N rows: the missing rows are
        // zero); the stack is reduced again, identically on every rank
        const int nr = h->nranks;
        void *pk, *ga, *st;
        TLSQ_TRY(ws_get(h, WS_QRP, (size_t)N * Np * 8, &pk));
        TLSQ_TRY(ws_get(h, WS_QRG, (size_t)N * Np * 8 * nr, &ga));
        TLSQ_TRY(ws_get(h, WS_QRS, (size_t)N * nr * Np * 8, &st));
        if (Mw < N) {
            TLSQ_HIP(h, hipMemsetAsync(pk, 0, (size_t)N * Np * 8, h->stream));
            TLSQ_HIP(h, hipMemcpy2DAsync(pk, (size_t)N * 8, W, (size_t)Mw * 8, (size_t)Mw * 8, (size_t)Np,
                                         hipMemcpyDeviceToDevice, h->stream));
        } else {
            hipLaunchKernelGGL(k_qr_pack, dim3(grid_for(N * Np)), dim3(256), 0, h->stream, (const double*)W, Mw, N, Np,
                               (double*)pk);
        }
        TLSQ_TRY(comm_allgather(h, (const double*)pk, (double*)ga, (size_t)N * Np));
        hipLaunchKernelGGL(k_qr_stack, dim3(grid_for(N * Np * nr)), dim3(256), 0, h->stream, (const double*)ga, N, Np, nr,
                           (double*)st);
        TLSQ_HIP(h, hipGetLastError());
        TLSQ_TRY(tsqr_inplace(h, (double*)st, N * nr, N * nr, Np));
        Wr = (const double*)st;
        ldr = N * nr;
    } else if (Mw < N) {
        return set_err(h, TLSQ_ERR_ARG, "tsqr: needs M >= N (M=%lld N=%lld)", (long long)M, (long long)N);
    }
    hipLaunchKernelGGL(k_qr_extract_lt, dim3(grid_for(N * N)), dim3(256), 0, h->stream, Wr, ldr, N, B);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// R (N x N upper triangular, ld ldR, device) of a device panel (kernel-level entry point, tests)
int tsqr_r(Handle* h, const double* Z, int64_t M, int64_t N, int64_t ldz, double* R, int64_t ldR) {
    void* Bv;
    TLSQ_TRY(ws_get(h, WS_B, (size_t)N * N * 8, &Bv));
    TLSQ_TRY(tsqr_lt(h, Z, 0, M, N, ldz, (double*)Bv));
    // R = B'
    TLSQ_TRY(launch_transpose<double>(h, (const double*)Bv, N, N, N, R, ldR));
    return TLSQ_OK;
}

}  // namespace tlsq
