// Complex (ComplexF64) rpca support — the `soft_th(x::Complex, eps)` method of the reference
// (/root/reference/src/robustPCA.jl:3-7, exercised by test/runtests.jl:187-199).
//
// The elementwise sweeps run on interleaved complex panels (re, im pairs, column-major, ld in complex elements).
// Everything spectral runs on the REALIFIED panel
//        R(Z) = [ Re Z  -Im Z ]      (2M x 2N real)
//               [ Im Z   Re Z ]
// whose singular values are those of Z, each twice, and for which R(A B) = R(A) R(B), R(Z^H) = R(Z)^T: the Gram /
// eigen / rebuild kernels of the real path apply unchanged (solver.hip: rpca_core_complex groups the eigenvalue
// pairs), and the left block column of R(A) is A.
#include "common.hpp"

#pragma clang fp contract(off)

namespace tlsq {

namespace {
__device__ __forceinline__ double c_pos(double a) { return (a > 0.0 || a != a) ? a : 0.0; }
__device__ __forceinline__ double c_neg(double b) { return (b < 0.0 || b != b) ? b : 0.0; }
// modulus shrink with the phase kept: m' * cis(angle(x)), m' = max(m-eps,0) + min(m+eps,0)   (:3-7)
__device__ __forceinline__ double2 c_soft(double2 x, double eps) {
    const double m = hypot(x.x, x.y);
    const double mn = c_pos(m - eps) + c_neg(m + eps);
    if (!(m > 0.0)) return make_double2(mn, 0.0);   // angle(0) = 0, cis(0) = 1
    const double sc = mn / m;
    return make_double2(x.x * sc, x.y * sc);
}
}  // namespace

// E = soft_th.(D - A + (1/mu) Y, lambda/mu);  Z = D - E + (1/mu) Y        (src/robustPCA.jl:188,192)
__global__ __launch_bounds__(256) void k_cshrink(const double2* __restrict__ D, const double2* __restrict__ A,
                                                 const double2* __restrict__ Y, double2* __restrict__ E,
                                                 double2* __restrict__ Z, int64_t n, double inv_mu, double thr) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double2 d = D[i], a = A[i], y = Y[i];
        const double2 t = make_double2(inv_mu * y.x, inv_mu * y.y);
        const double2 e = c_soft(make_double2((d.x - a.x) + t.x, (d.y - a.y) + t.y), thr);
        E[i] = e;
        Z[i] = make_double2((d.x - e.x) + t.x, (d.y - e.y) + t.y);
    }
}

// Z = D - A - E;  Y = Y + mu Z                                            (src/robustPCA.jl:221-222)
__global__ __launch_bounds__(256) void k_cupdate(const double2* __restrict__ D, const double2* __restrict__ A,
                                                 const double2* __restrict__ E, double2* __restrict__ Y,
                                                 double2* __restrict__ R, int64_t n, double mu) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double2 d = D[i], a = A[i], e = E[i], y = Y[i];
        const double2 z = make_double2((d.x - a.x) - e.x, (d.y - a.y) - e.y);
        R[i] = z;
        Y[i] = make_double2(y.x + mu * z.x, y.y + mu * z.y);
    }
}

// Y = D / s  (complex / real)                                              (src/robustPCA.jl:181)
__global__ __launch_bounds__(256) void k_cdiv(const double2* __restrict__ D, double2* __restrict__ Y, int64_t n,
                                              double s) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        Y[i] = make_double2(D[i].x / s, D[i].y / s);
}

// W (2M x 2N real, ld 2M) = R(Z), Z complex M x N (ld M)
__global__ __launch_bounds__(256) void k_realify(const double2* __restrict__ Z, int64_t M, int64_t N,
                                                 double* __restrict__ W) {
    const int64_t n = M * N, ldw = 2 * M;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i % M, c = i / M;
        const double2 z = Z[i];
        W[r + c * ldw] = z.x;
        W[M + r + c * ldw] = z.y;
        W[r + (N + c) * ldw] = -z.y;
        W[M + r + (N + c) * ldw] = z.x;
    }
}

// A complex M x N = left block column of the realified AR (2M x 2N, ld 2M)
__global__ __launch_bounds__(256) void k_unrealify(const double* __restrict__ AR, int64_t M, int64_t N,
                                                   double2* __restrict__ A) {
    const int64_t n = M * N, ldw = 2 * M;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i % M, c = i / M;
        A[i] = make_double2(AR[r + c * ldw], AR[M + r + c * ldw]);
    }
}

// out[0] = max |x_i| (complex modulus) as the bit pattern of a non-negative double
__global__ __launch_bounds__(256) void k_cmaxabs(const double2* __restrict__ x, int64_t n,
                                                 unsigned long long* __restrict__ out) {
    double m = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double a = hypot(x[i].x, x[i].y);
        m = (a > m || a != a) ? a : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(m, off, 64);
        m = (o > m || o != o) ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

static inline int cgrid(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;
    return (int)g;
}

int launch_cshrink(Handle* h, const double* D, const double* A, const double* Y, double* E, double* Z, int64_t n,
                   double inv_mu, double thr) {
    hipLaunchKernelGGL(k_cshrink, dim3(cgrid(n)), dim3(256), 0, h->stream, (const double2*)D, (const double2*)A,
                       (const double2*)Y, (double2*)E, (double2*)Z, n, inv_mu, thr);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
int launch_cupdate(Handle* h, const double* D, const double* A, const double* E, double* Y, double* R, int64_t n,
                   double mu) {
    hipLaunchKernelGGL(k_cupdate, dim3(cgrid(n)), dim3(256), 0, h->stream, (const double2*)D, (const double2*)A,
                       (const double2*)E, (double2*)Y, (double2*)R, n, mu);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
int launch_cdiv(Handle* h, const double* D, double* Y, int64_t n, double s) {
    hipLaunchKernelGGL(k_cdiv, dim3(cgrid(n)), dim3(256), 0, h->stream, (const double2*)D, (double2*)Y, n, s);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
int launch_realify(Handle* h, const double* Z, int64_t M, int64_t N, double* W) {
    hipLaunchKernelGGL(k_realify, dim3(cgrid(M * N)), dim3(256), 0, h->stream, (const double2*)Z, M, N, W);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
int launch_unrealify(Handle* h, const double* AR, int64_t M, int64_t N, double* A) {
    hipLaunchKernelGGL(k_unrealify, dim3(cgrid(M * N)), dim3(256), 0, h->stream, AR, M, N, (double2*)A);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
// U[m, i] = T[m, i] + i T[M + m, i]: the complex left vectors from the product of the realified panel with the real images
// [Re v; Im v] / sigma of the right vectors (T is 2M x d real, ld 2M; U complex M x d, ld M, interleaved)
__global__ __launch_bounds__(256) void k_pack_complex(const double* __restrict__ T, int64_t M, int64_t d, double2* __restrict__ U) {
    const int64_t total = M * d, stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        const int64_t m = e % M, i = e / M;
        U[e] = double2{T[m + i * 2 * M], T[M + m + i * 2 * M]};
    }
}
int launch_pack_complex(Handle* h, const double* T, int64_t M, int64_t d, double* U) {
    if (M <= 0 || d <= 0) return TLSQ_OK;
    hipLaunchKernelGGL(k_pack_complex, dim3(cgrid(M * d)), dim3(256), 0, h->stream, T, M, d, (double2*)U);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
int launch_cmaxabs(Handle* h, const double* x, int64_t n, double* host_out) {
    void* slot;
    TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &slot));
    unsigned long long* d = reinterpret_cast<unsigned long long*>(slot);
    TLSQ_HIP(h, hipMemsetAsync(d, 0, 8, h->stream));
    hipLaunchKernelGGL(k_cmaxabs, dim3(cgrid(n)), dim3(256), 0, h->stream, (const double2*)x, n, d);
    TLSQ_HIP(h, hipGetLastError());
    unsigned long long bits = 0;
    TLSQ_HIP(h, hipMemcpyAsync(h->pinned, d, 8, hipMemcpyDeviceToHost, h->stream));
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    memcpy(&bits, h->pinned, 8);
    memcpy(host_out, &bits, 8);
    return TLSQ_OK;
}

}  // namespace tlsq
