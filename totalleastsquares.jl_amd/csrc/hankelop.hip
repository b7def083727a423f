// Products with a Hankel matrix that is never stored (one channel, lag 1): H[k, j] = y[k + j], k < K, j < n.
//
// lowrankfilter's plain truncation branch (src/robustPCA.jl:123-126 under /root/reference: `s = svd(H); A = U[:, 1:sv] S V'`)
// needs two things from H: its Gram matrix (right singular vectors, singular values) and the factor T = H V[:, 1:sv] (then
// A = T V' and the anti-diagonal means come from the factors, hankel.hip).  Both follow from the series itself:
//
//   G[i, j] = sum_k y[k + i] y[k + j] = R_d(i),  d = j - i,  R_d(i) = sum_{t = i}^{i + K - 1} y[t] y[t + d]
//           = R_d(0) - sum_{t < i} y[t] y[t + d] + sum_{t < i} y[K + t] y[K + t + d]
//     - n lagged autocorrelation sums of length K (K n multiply-adds: 2.6 GFLOP at K = 1e7, n = 256, against the 1.3 TFLOP of the
//       dense Gram of a 20 GB panel) and n^2 / 2 boundary corrections.  One pass over the 80 MB series; the window of a chunk sits
//       in LDS, thread d reads y[t] (broadcast) and y[t + d] (consecutive).
//   T[k, p] = sum_j y[k + j] X[j, p] - r FIR filters of n taps: thread = row, the taps of a 64-row slice of X in LDS (broadcast
//       reads), the window read with consecutive addresses; K n r multiply-adds, no panel.
//
// Memory: the series and the K x r factor instead of two K x n panels (41 GB at BASELINE config 3's size).
// fp64 arithmetic whatever the element type of the series (the small-matrix convention of the library).
#include "common.hpp"

namespace tlsq {

namespace {

constexpr int HG_CHUNK = 4096;   // samples per workgroup of the autocorrelation pass

// part[blk * n + d] = sum_{t in chunk blk, t < K} y[t] y[t + d]
template <typename T>
__global__ __launch_bounds__(1024) void k_hankel_autocorr(const T* __restrict__ y, int64_t K, int n, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) double sw[];   // HG_CHUNK + n samples
    const int64_t t0 = (int64_t)blockIdx.x * HG_CHUNK;
    const int64_t len = (K - t0 < HG_CHUNK) ? K - t0 : HG_CHUNK;   // samples t of this chunk
    for (int64_t e = threadIdx.x; e < len + n - 1; e += blockDim.x) sw[e] = (double)y[t0 + e];   // (t + d <= K + n - 2: the last sample)
    __syncthreads();
    const int d = threadIdx.x;
    if (d >= n) return;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int64_t t = 0;
    for (; t + 3 < len; t += 4) {
        a0 = __builtin_fma(sw[t], sw[t + d], a0);
        a1 = __builtin_fma(sw[t + 1], sw[t + 1 + d], a1);
        a2 = __builtin_fma(sw[t + 2], sw[t + 2 + d], a2);
        a3 = __builtin_fma(sw[t + 3], sw[t + 3 + d], a3);
    }
    for (; t < len; ++t) a0 = __builtin_fma(sw[t], sw[t + d], a0);
    part[(int64_t)blockIdx.x * n + d] = (a0 + a1) + (a2 + a3);
}

// G (n x n, ld n, both triangles) from the chunk sums: thread d adds them in order (R_d(0)), then walks down the d-th diagonal
template <typename T>
__global__ __launch_bounds__(1024) void k_hankel_gram_fill(const T* __restrict__ y, int64_t K, int n, const double* __restrict__ part,
                                                           int nblk, double* __restrict__ G) {
    __shared__ double sa[2048], sb[2048];   // y[0 .. 2 n - 2], y[K .. K + n - 2] (n <= 1024): what the corrections read
    for (int e = threadIdx.x; e < 2 * n - 1; e += blockDim.x) sa[e] = (double)y[e];
    for (int e = threadIdx.x; e < n - 1; e += blockDim.x) sb[e] = (double)y[K + e];
    __syncthreads();
    const int d = threadIdx.x;
    if (d >= n) return;
    double r0 = 0.0;
    for (int b = 0; b < nblk; ++b) r0 += part[(int64_t)b * n + d];
    double rd = r0;
    for (int i = 0; i + d < n; ++i) {
        G[i + (int64_t)(i + d) * n] = rd;
        G[(i + d) + (int64_t)i * n] = rd;
        // R_d(i + 1) = R_d(i) - y[i] y[i + d] + y[i + K] y[i + K + d]   (i + d <= n - 2 here: K + i + d <= K + n - 2)
        if (i + 1 + d < n) rd = (rd - sa[i] * sa[i + d]) + sb[i] * sb[i + d];
    }
}

// T[k + p ldt] = sum_j y[k + j] X[j + p ldx], k < K, p < r (r <= 32); rows K .. Kp - 1 of T are set to zero
template <typename T, int R>
__global__ __launch_bounds__(256) void k_hankel_times(const T* __restrict__ y, int64_t K, int64_t Kp, int n, const double* __restrict__ X,
                                                      int64_t ldx, int r, double* __restrict__ Tm, int64_t ldt) {
    __shared__ double sy[256 + 64];
    __shared__ __attribute__((aligned(16))) double sx[64 * R];   // a 64-row slice of X, [j][p]
    const int64_t k0 = (int64_t)blockIdx.x * 256, k = k0 + threadIdx.x;
    double acc[R];
#pragma unroll
    for (int p = 0; p < R; ++p) acc[p] = 0.0;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int jl = (n - j0 < 64) ? n - j0 : 64;
        __syncthreads();
        for (int e = threadIdx.x; e < 256 + 64; e += 256) {
            const int64_t t = k0 + j0 + e;
            sy[e] = (t < K + n - 1) ? (double)y[t] : 0.0;
        }
        for (int e = threadIdx.x; e < 64 * R; e += 256) {
            const int j = e / R, p = e % R;
            sx[e] = (j < jl && p < r) ? X[(j0 + j) + (int64_t)p * ldx] : 0.0;
        }
        __syncthreads();
        for (int j = 0; j < jl; ++j) {
            const double yv = sy[threadIdx.x + j];
#pragma unroll
            for (int p = 0; p < R; ++p) acc[p] = __builtin_fma(yv, sx[j * R + p], acc[p]);
        }
    }
    if (k < Kp) {
#pragma unroll
        for (int p = 0; p < R; ++p)
            if (p < r) Tm[k + (int64_t)p * ldt] = k < K ? acc[p] : 0.0;
    }
}

}   // namespace

bool hankel_structured_ok(int64_t K, int64_t n, int64_t r) { return n >= 1 && n <= 1024 && r >= 1 && r <= 32 && K >= 1; }

// G (n x n fp64, ld n) = H'H for H[k, j] = y[k + j], k < K, j < n (y: K + n - 1 samples on the device)
template <typename T>
int hankel_gram(Handle* h, const T* y, int64_t K, int64_t n, double* G) {
    if (!hankel_structured_ok(K, n, 1)) return set_err(h, TLSQ_ERR_ARG, "hankel_gram: n = %lld", (long long)n);
    const int64_t nblk = (K + HG_CHUNK - 1) / HG_CHUNK;
    void* part;
    TLSQ_TRY(ws_get(h, WS_SLAB, (size_t)nblk * n * 8, &part));
    const int threads = (int)std::max<int64_t>(256, (n + 63) / 64 * 64);
    const size_t lds = (size_t)(HG_CHUNK + n) * 8;
    hipLaunchKernelGGL((k_hankel_autocorr<T>), dim3((unsigned)nblk), dim3(threads), lds, h->stream, y, K, (int)n, (double*)part);
    hipLaunchKernelGGL((k_hankel_gram_fill<T>), dim3(1), dim3(threads), 0, h->stream, y, K, (int)n, (const double*)part, (int)nblk, G);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// Tm (Kp x r fp64, ld ldt; rows K.. zero) = H X for X (n x r fp64, ld ldx)
template <typename T>
int hankel_times(Handle* h, const T* y, int64_t K, int64_t Kp, int64_t n, const double* X, int64_t ldx, int64_t r, double* Tm,
                 int64_t ldt) {
    if (!hankel_structured_ok(K, n, r)) return set_err(h, TLSQ_ERR_ARG, "hankel_times: n = %lld, r = %lld", (long long)n, (long long)r);
    const dim3 grid((unsigned)((Kp + 255) / 256));
    if (r <= 8) hipLaunchKernelGGL((k_hankel_times<T, 8>), grid, dim3(256), 0, h->stream, y, K, Kp, (int)n, X, ldx, (int)r, Tm, ldt);
    else if (r <= 16) hipLaunchKernelGGL((k_hankel_times<T, 16>), grid, dim3(256), 0, h->stream, y, K, Kp, (int)n, X, ldx, (int)r, Tm, ldt);
    else hipLaunchKernelGGL((k_hankel_times<T, 32>), grid, dim3(256), 0, h->stream, y, K, Kp, (int)n, X, ldx, (int)r, Tm, ldt);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template int hankel_gram<double>(Handle*, const double*, int64_t, int64_t, double*);
template int hankel_gram<float>(Handle*, const float*, int64_t, int64_t, double*);
template int hankel_times<double>(Handle*, const double*, int64_t, int64_t, int64_t, const double*, int64_t, int64_t, double*, int64_t);
template int hankel_times<float>(Handle*, const float*, int64_t, int64_t, int64_t, const double*, int64_t, int64_t, double*, int64_t);

}   // namespace tlsq
