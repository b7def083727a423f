// Hankel lag embedding, anti-diagonal averaging and soft_hankel! — pure HBM-bound index kernels.
//   hankel       /root/reference/src/robustPCA.jl:76-92
//   unhankel     /root/reference/src/robustPCA.jl:28-39 (lag==1,D==1) and :53-68 (general)
//   soft_hankel! /root/reference/src/robustPCA.jl:9-21
// Indices are computed arithmetically (the reference materialises an index-Hankel matrix, :60).
// Every output sample sums its anti-diagonal sequentially in the reference's order (increasing
// column index), so results match the reference's `mean` / scatter-add bit for bit.
#include "common.hpp"

#pragma clang fp contract(off)

namespace tlsq {

// X[k + (d + Dch*l)*ldX] = x[(k*lag + l) + d*ldx]
template <typename T>
__global__ __launch_bounds__(256) void k_hankel(const T* __restrict__ x, int64_t ldx, int64_t K,
                                                int64_t L, int64_t Dch, int64_t lag,
                                                T* __restrict__ X, int64_t ldX) {
    const int64_t c = blockIdx.y;  // column of X
    const int64_t d = c % Dch, l = c / Dch;
    const T* __restrict__ src = x + d * ldx + l;
    T* __restrict__ dst = X + c * ldX;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < K; k += stride)
        dst[k] = src[k * lag];
}

// y[n + d*ldy] = (sum_{l : (n-l)%lag==0, 0<=(n-l)/lag<K} A[(n-l)/lag + (d+Dch*l)*ldA]) / max(count,1)
template <typename T>
__global__ __launch_bounds__(256) void k_unhankel(const T* __restrict__ A, int64_t K, int64_t L,
                                                  int64_t Dch, int64_t ldA, int64_t lag, int64_t Nx,
                                                  T* __restrict__ y, int64_t ldy) {
    const int64_t d = blockIdx.y;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < Nx; n += stride) {
        T tot = T(0);
        int64_t cnt = 0;
        // l ranges over n - lag*k, k in [0,K): l >= 0, l < L
        int64_t lmax = n < L - 1 ? n : L - 1;
        if (lag == 1) {
            int64_t lmin = n - (K - 1);
            if (lmin < 0) lmin = 0;
            for (int64_t l = lmin; l <= lmax; ++l) {
                const T v = A[(n - l) + (d + Dch * l) * ldA];
                tot = (cnt == 0) ? v : tot + v;
                ++cnt;
            }
        } else {
            for (int64_t l = n % lag; l <= lmax; l += lag) {
                const int64_t k = (n - l) / lag;
                if (k < K) {
                    const T v = A[k + (d + Dch * l) * ldA];
                    tot = (cnt == 0) ? v : tot + v;
                    ++cnt;
                }
            }
        }
        y[n + d * ldy] = cnt > 0 ? tot / (T)cnt : T(0);
    }
}

// y = unhankel(A) for A = Tm Vs' (K x L, one channel, lag 1) straight from the factors: entry (k, l) is rebuilt exactly as
// k_rebuild_store does (fp64 fma chain over the r factor columns, rounded to T) and the anti-diagonal is summed in T in the
// order of k_unhankel - bit-identical to materialising A first, without its 2 panel passes (and without the panel).
template <typename T>
__global__ __launch_bounds__(256) void k_unhankel_factors(const double* __restrict__ Tm, int64_t ldT,
                                                          const double* __restrict__ Vs, int64_t ldV, int r, int64_t K,
                                                          int64_t L, int64_t Nx, T* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) double sV[];   // [l][i], L x r
    for (int64_t e = threadIdx.x; e < L * r; e += 256) {
        const int64_t l = e / r, i = e % r;
        sV[e] = Vs[l + i * ldV];
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < Nx; n += stride) {
        T tot = T(0);
        int64_t cnt = 0;
        const int64_t lmax = n < L - 1 ? n : L - 1;
        int64_t lmin = n - (K - 1);
        if (lmin < 0) lmin = 0;
        for (int64_t l = lmin; l <= lmax; ++l) {
            const double* v = sV + l * r;
            const double* t = Tm + (n - l);
            double a = 0.0;
            for (int i = 0; i < r; ++i) a = __builtin_fma(t[(int64_t)i * ldT], v[i], a);
            const T av = (T)a;
            tot = (cnt == 0) ? av : tot + av;
            ++cnt;
        }
        y[n] = cnt > 0 ? tot / (T)cnt : T(0);
    }
}

// Time-window shards (multi-GPU lowrankfilter): the anti-diagonal sums and counts of this rank's rows, written at
// the window's offset `off` into full-length arrays sum / cnt (Nx_glob x Dch, zero-filled by the caller); after the
// all-reduce k_unhankel_finish divides (y ./= max.(counts,1), :66).  Same summation order inside a shard as above.
template <typename T>
__global__ __launch_bounds__(256) void k_unhankel_partial(const T* __restrict__ A, int64_t K, int64_t L, int64_t Dch,
                                                          int64_t ldA, int64_t lag, int64_t Nw, int64_t off,
                                                          double* __restrict__ sum, double* __restrict__ cnt,
                                                          int64_t ldy) {
    const int64_t d = blockIdx.y;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < Nw; n += stride) {
        T tot = (T)0;   // (summed in the element type, like the one-GPU kernel and the reference; fp64 across shards)
        int64_t c = 0;
        const int64_t lmax = n < L - 1 ? n : L - 1;
        for (int64_t l = n % lag; l <= lmax; l += lag) {
            const int64_t k = (n - l) / lag;
            if (k < K) {
                const T v = A[k + (d + Dch * l) * ldA];
                tot = (c == 0) ? v : tot + v;
                ++c;
            }
        }
        sum[off + n + d * ldy] = (double)tot;
        cnt[off + n + d * ldy] = (double)c;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_unhankel_finish(const double* __restrict__ sum, const double* __restrict__ cnt,
                                                         int64_t n, T* __restrict__ y) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = (T)(cnt[i] > 0.0 ? sum[i] / cnt[i] : 0.0);
}

// A[k + l*ldA] = soft_th(A[k + l*ldA], eps, m[k+l]),  soft_th(x,e,m) = max(x-e,m) + min(x+e,m) - m
template <typename T>
__global__ __launch_bounds__(256) void k_soft_toward(T* __restrict__ A, int64_t K, int64_t ldA,
                                                     const T* __restrict__ m, T eps) {
    const int64_t l = blockIdx.y;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < K; k += stride) {
        const T x = A[k + l * ldA];
        const T mm = m[k + l];
        const T a = x - eps, b = x + eps;
        const T hi = (a > mm || a != a) ? a : mm;   // max(x-e, m)
        const T lo = (b < mm || b != b) ? b : mm;   // min(x+e, m)
        A[k + l * ldA] = (hi + lo) - mm;
    }
}

static inline int gx(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > 4096) g = 4096;
    return (int)g;
}

template <typename T>
int launch_hankel(Handle* h, const T* x, int64_t Nx, int64_t Dch, int64_t ldx, int64_t L,
                  int64_t lag, T* X, int64_t ldX) {
    const int64_t K = (Nx - L) / lag + 1;
    const int64_t cols = L * Dch;
    if (K <= 0 || cols <= 0) return TLSQ_OK;
    if (cols > 65535)
        return set_err(h, TLSQ_ERR_UNSUPPORTED, "hankel: L*D=%lld exceeds 65535", (long long)cols);
    hipLaunchKernelGGL((k_hankel<T>), dim3(gx(K), (unsigned)cols), dim3(256), 0, h->stream, x, ldx, K,
                       L, Dch, lag, X, ldX);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template <typename T>
int launch_unhankel(Handle* h, const T* A, int64_t K, int64_t L, int64_t Dch, int64_t ldA,
                    int64_t lag, int64_t Nx, T* y, int64_t ldy) {
    if (Nx <= 0 || Dch <= 0) return TLSQ_OK;
    hipLaunchKernelGGL((k_unhankel<T>), dim3(gx(Nx), (unsigned)Dch), dim3(256), 0, h->stream, A, K, L,
                       Dch, ldA, lag, Nx, y, ldy);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template <typename T>
int launch_soft_hankel(Handle* h, T* A, int64_t K, int64_t L, int64_t ldA, T eps, T* mean_ws) {
    if (K <= 0 || L <= 0) return TLSQ_OK;
    if (L > 65535)
        return set_err(h, TLSQ_ERR_UNSUPPORTED, "soft_hankel: L=%lld exceeds 65535", (long long)L);
    // anti-diagonal means (length K+L-1), then shrink every element towards its diagonal's mean
    TLSQ_TRY(launch_unhankel<T>(h, A, K, L, 1, ldA, 1, K + L - 1, mean_ws, K + L - 1));
    hipLaunchKernelGGL((k_soft_toward<T>), dim3(gx(K), (unsigned)L), dim3(256), 0, h->stream, A, K,
                       ldA, (const T*)mean_ws, eps);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// the second half of soft_hankel! alone: A[k, l] = soft_th(A[k, l], eps, m[k + l]) for given anti-diagonal means m
template <typename T>
int launch_soft_toward(Handle* h, T* A, int64_t K, int64_t L, int64_t ldA, const T* m, T eps) {
    if (K <= 0 || L <= 0) return TLSQ_OK;
    if (L > 65535) return set_err(h, TLSQ_ERR_UNSUPPORTED, "soft_hankel: L=%lld exceeds 65535", (long long)L);
    hipLaunchKernelGGL((k_soft_toward<T>), dim3(gx(K), (unsigned)L), dim3(256), 0, h->stream, A, K, ldA, m, eps);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template <typename T>
int launch_unhankel_partial(Handle* h, const T* A, int64_t K, int64_t L, int64_t Dch, int64_t ldA, int64_t lag,
                            int64_t Nw, int64_t off, double* sum, double* cnt, int64_t ldy) {
    if (Nw <= 0 || Dch <= 0) return TLSQ_OK;
    hipLaunchKernelGGL((k_unhankel_partial<T>), dim3(gx(Nw), (unsigned)Dch), dim3(256), 0, h->stream, A, K, L, Dch, ldA, lag,
                       Nw, off, sum, cnt, ldy);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
template <typename T>
int launch_unhankel_finish(Handle* h, const double* sum, const double* cnt, int64_t n, T* y) {
    if (n <= 0) return TLSQ_OK;
    hipLaunchKernelGGL((k_unhankel_finish<T>), dim3(gx(n)), dim3(256), 0, h->stream, sum, cnt, n, y);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template <typename T>
int launch_unhankel_factors(Handle* h, const double* Tm, int64_t ldT, const double* Vs, int64_t ldV, int64_t r, int64_t K,
                            int64_t L, int64_t Nx, T* y) {
    if (Nx <= 0) return TLSQ_OK;
    if (r < 0 || r > 32 || (size_t)L * (size_t)std::max<int64_t>(r, 1) * 8 > 64 * 1024)
        return set_err(h, TLSQ_ERR_ARG, "unhankel_factors: rank above 32 or window too long for the LDS copy of Vs");
    hipLaunchKernelGGL((k_unhankel_factors<T>), dim3(gx(Nx)), dim3(256), (size_t)L * std::max<int64_t>(r, 1) * 8, h->stream, Tm,
                       ldT, Vs, ldV, (int)r, K, L, Nx, y);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
template int launch_unhankel_factors<double>(Handle*, const double*, int64_t, const double*, int64_t, int64_t, int64_t, int64_t,
                                             int64_t, double*);
template int launch_unhankel_factors<float>(Handle*, const double*, int64_t, const double*, int64_t, int64_t, int64_t, int64_t,
                                            int64_t, float*);

#define INST(T)                                                                                       \
    template int launch_hankel<T>(Handle*, const T*, int64_t, int64_t, int64_t, int64_t, int64_t, T*, \
                                  int64_t);                                                           \
    template int launch_unhankel<T>(Handle*, const T*, int64_t, int64_t, int64_t, int64_t, int64_t,   \
                                    int64_t, T*, int64_t);                                            \
    template int launch_soft_hankel<T>(Handle*, T*, int64_t, int64_t, int64_t, T, T*);                \
    template int launch_unhankel_partial<T>(Handle*, const T*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, \
                                            int64_t, double*, double*, int64_t);                      \
    template int launch_unhankel_finish<T>(Handle*, const double*, const double*, int64_t, T*);        \
    template int launch_soft_toward<T>(Handle*, T*, int64_t, int64_t, int64_t, const T*, T);
INST(double)
INST(float)
#undef INST

}  // namespace tlsq
