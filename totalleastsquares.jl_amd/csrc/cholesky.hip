// Cholesky factor of the (shifted) Gram matrix — the preconditioner of the full eigensolver.
//
// One-sided Jacobi converges much faster, and more accurately, on the Cholesky factor L of G = L L' than on G
// itself (Veselic & Hari 1989; Demmel & Veselic 1992): on graded spectra the sweep count drops from ~30 to ~9,
// no eigenvector accumulation is needed (the eigenvectors of G are the normalised columns of the rotated L), and
// the singular values of L are the singular values of Z directly.  G is PSD but may be numerically singular, so
// the factorisation runs on G + delta*I with delta = 2 N eps max_i G_ii (below the resolution of G itself).
//
// Blocked right-looking factorisation, in place in the lower triangle of a private copy:
//   k_chol_panel    one workgroup factors a panel of nb columns held in LDS (rows below the diagonal included)
//   trailing update G22 -= L21 L21'  through the MFMA GEMM (gemm.hip) + k_sub_ld
//   k_chol_finish   zeroes the strict upper triangle
#include "common.hpp"

namespace tlsq {

// panel: columns [k0, k0+nb) of A (N x N, ld N), rows k0..N-1.  LDS image S[c][r], r = row - k0.
__global__ __launch_bounds__(1024) void k_chol_panel(double* __restrict__ A, int N, int k0, int nb, double tiny) {
    extern __shared__ __attribute__((aligned(16))) double S[];
    const int rows = N - k0;
    const int tid = threadIdx.x;
    for (int e = tid; e < rows * nb; e += 1024) {
        const int r = e % rows, c = e / rows;
        S[(size_t)c * rows + r] = A[(size_t)(k0 + c) * N + k0 + r];
    }
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
        const double piv = S[(size_t)j * rows + j];
        // (same value for every thread: read before anybody modifies column j)
        __syncthreads();
        if (piv > tiny) {
            const double inv = 1.0 / sqrt(piv);
            for (int r = j + tid; r < rows; r += 1024) S[(size_t)j * rows + r] *= inv;   // includes the diagonal
        } else {
            for (int r = j + tid; r < rows; r += 1024) S[(size_t)j * rows + r] = 0.0;    // numerically zero pivot
        }
        __syncthreads();
        // rank-1 update of the remaining panel columns: S[r][c] -= S[r][j] * S[c][j], r >= c > j
        const int ncol = nb - j - 1;
        for (int e = tid; e < ncol * rows; e += 1024) {
            const int c = j + 1 + e / rows, r = e % rows;
            if (r >= c) S[(size_t)c * rows + r] -= S[(size_t)j * rows + r] * S[(size_t)j * rows + c];
        }
        __syncthreads();
    }
    for (int e = tid; e < rows * nb; e += 1024) {
        const int r = e % rows, c = e / rows;
        A[(size_t)(k0 + c) * N + k0 + r] = S[(size_t)c * rows + r];
    }
}

// C (P x P, ld ldc) -= T (P x P, ld ldt), lower triangle only
__global__ __launch_bounds__(256) void k_sub_lower(double* __restrict__ C, int64_t ldc, const double* __restrict__ T,
                                                   int64_t ldt, int P) {
    const int64_t total = (int64_t)P * P;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int r = (int)(e % P), c = (int)(e / P);
        if (r >= c) C[r + c * ldc] -= T[r + c * ldt];
    }
}

// L = A + delta*I (copy, ld -> N) ; stats[0] = max diagonal (computed beforehand by k_maxdiag)
__global__ __launch_bounds__(256) void k_copy_shift(const double* __restrict__ G, int64_t ldG, double* __restrict__ L,
                                                    int N, const double* __restrict__ stats, double factor) {
    const double delta = factor * stats[0];
    const int64_t total = (int64_t)N * N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t r = e % N, c = e / N;
        double v = G[r + c * ldG];
        if (r == c) v += delta;
        L[e] = v;
    }
}

// stats[0] = max_i G_ii, stats[1] = delta (shift that k_copy_shift applies)
__global__ __launch_bounds__(1024) void k_maxdiag(const double* __restrict__ G, int64_t ldG, int N,
                                                  double* __restrict__ stats, double factor) {
    __shared__ double sw[16];
    double m = 0.0;
    for (int i = threadIdx.x; i < N; i += 1024) {
        const double v = G[i + (int64_t)i * ldG];
        m = v > m ? v : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int k = 0; k < 16; ++k) r = sw[k] > r ? sw[k] : r;
        stats[0] = r;
        stats[1] = factor * r;
    }
}

__global__ __launch_bounds__(256) void k_zero_upper(double* __restrict__ L, int N) {
    const int64_t total = (int64_t)N * N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t r = e % N, c = e / N;
        if (r < c) L[e] = 0.0;
    }
}

// V[:,c] = B[:,c] / ||B[:,c]||, sig[c] = ||B[:,c]||  (one wave per column); zero columns stay zero
__global__ __launch_bounds__(256) void k_normalize_cols(const double* __restrict__ B, int N, double* __restrict__ V,
                                                        double* __restrict__ sig) {
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= N) return;
    double s = 0.0;
    for (int r = lane; r < N; r += 64) {
        const double v = B[(size_t)col * N + r];
        s += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const double nrm = sqrt(s);
    const double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;
    for (int r = lane; r < N; r += 64) V[(size_t)col * N + r] = B[(size_t)col * N + r] * inv;
    if (lane == 0) sig[col] = nrm;
}

// L (N x N, ld N, workspace) = chol(G + delta I), lower triangular with a zeroed upper part.
// stats_dev[1] receives delta.  T is an N x N scratch for the trailing products.
int cholesky_shifted(Handle* h, const double* G, int64_t ldG, int64_t N, double* L, double* T, double* stats_dev) {
    const double eps = 2.220446049250313e-16;
    const double factor = 2.0 * (double)N * eps;
    hipLaunchKernelGGL(k_maxdiag, dim3(1), dim3(1024), 0, h->stream, G, ldG, (int)N, stats_dev, factor);
    int64_t g = (N * N + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_copy_shift, dim3((int)g), dim3(256), 0, h->stream, G, ldG, L, (int)N,
                       (const double*)stats_dev, factor);
    TLSQ_HIP(h, hipGetLastError());
    // panel width: rows x nb doubles must fit LDS
    int nb = 32;
    while (nb > 1 && (size_t)N * nb * 8 > 144 * 1024) nb >>= 1;
    if ((size_t)N * nb * 8 > 144 * 1024)
        return set_err(h, TLSQ_ERR_UNSUPPORTED, "cholesky: N=%lld too large", (long long)N);
    TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_panel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)N * nb * 8)));
    for (int64_t k0 = 0; k0 < N; k0 += nb) {
        const int w = (int)std::min<int64_t>(nb, N - k0);
        const size_t lds = (size_t)(N - k0) * w * 8;
        hipLaunchKernelGGL(k_chol_panel, dim3(1), dim3(1024), lds, h->stream, L, (int)N, (int)k0, w, 0.0);
        TLSQ_HIP(h, hipGetLastError());
        const int64_t P = N - k0 - w;
        if (P > 0) {
            // T = L21 L21'  (L21 = rows k0+w.., columns k0..k0+w-1), then G22 -= T on the lower triangle
            const double* L21 = L + (size_t)k0 * N + k0 + w;
            TLSQ_TRY(gemm_f64(h, false, false, L21, N, L21, N, T, P, P, P, w, true));
            int64_t g2 = (P * P + 255) / 256;
            if (g2 > 2048) g2 = 2048;
            hipLaunchKernelGGL(k_sub_lower, dim3((int)g2), dim3(256), 0, h->stream, L + (size_t)(k0 + w) * N + k0 + w,
                               N, (const double*)T, P, (int)P);
            TLSQ_HIP(h, hipGetLastError());
        }
    }
    hipLaunchKernelGGL(k_zero_upper, dim3((int)g), dim3(256), 0, h->stream, L, (int)N);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

int launch_normalize_cols(Handle* h, const double* B, int64_t N, double* V, double* sig) {
    hipLaunchKernelGGL(k_normalize_cols, dim3((int)((N + 3) / 4)), dim3(256), 0, h->stream, B, (int)N, V, sig);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

}  // namespace tlsq
