// Elementwise ALM sweeps of rpca — HBM-bound, 16-byte coalesced loads/stores, grid-stride.
//
//   shrink sweep  = /root/reference/src/robustPCA.jl:188-192  (E-step + Z build, 3 reads / 2 writes)
//   update sweep  = /root/reference/src/robustPCA.jl:217-222  (residual + dual update, 4 reads / 2 writes)
//
// Floating-point contraction is OFF in this file: Julia does not fuse a*b+c, and the parity tests
// compare these kernels bit-for-bit with the oracle (same expression order as the reference).
#include "common.hpp"

#pragma clang fp contract(off)

namespace tlsq {

template <typename T>
__device__ __forceinline__ T pos_part(T a) {  // max(a, 0) with Julia's NaN propagation
    return (a > T(0) || a != a) ? a : T(0);
}
template <typename T>
__device__ __forceinline__ T neg_part(T b) {  // min(b, 0)
    return (b < T(0) || b != b) ? b : T(0);
}
template <typename T>
__device__ __forceinline__ T soft_th(T x, T e) {  // robustPCA.jl:1
    return pos_part(x - e) + neg_part(x + e);
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void k_shrink(const T* __restrict__ D, const T* __restrict__ A,
                                                const T* __restrict__ Y, T* __restrict__ E,
                                                T* __restrict__ Z, int64_t n, T inv_mu, T thr,
                                                int nonnegE) {
    using V = T __attribute__((ext_vector_type(VEC)));
    const int64_t nv = n / VEC;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = tid; i < nv; i += stride) {
        V d = reinterpret_cast<const V*>(D)[i];
        V a = reinterpret_cast<const V*>(A)[i];
        V y = reinterpret_cast<const V*>(Y)[i];
        V e, z;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            T t = inv_mu * y[c];                     // (1/μ) .* Y
            T ee = soft_th((d[c] - a[c]) + t, thr);  // soft_th.(D .- A .+ (1/μ).*Y, λ/μ)   :188
            if (nonnegE) ee = pos_part(ee);          // :189-191
            e[c] = ee;
            z[c] = (d[c] - ee) + t;                  // Z .= D .- E .+ (1/μ).*Y              :192
        }
        reinterpret_cast<V*>(E)[i] = e;
        reinterpret_cast<V*>(Z)[i] = z;
    }
    for (int64_t i = nv * VEC + tid; i < n; i += stride) {  // tail
        T t = inv_mu * Y[i];
        T ee = soft_th((D[i] - A[i]) + t, thr);
        if (nonnegE) ee = pos_part(ee);
        E[i] = ee;
        Z[i] = (D[i] - ee) + t;
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void k_update(const T* __restrict__ D, T* __restrict__ A,
                                                const T* __restrict__ E, T* __restrict__ Y,
                                                T* __restrict__ R, int64_t n, T mu, int nonnegA) {
    using V = T __attribute__((ext_vector_type(VEC)));
    const int64_t nv = n / VEC;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = tid; i < nv; i += stride) {
        V d = reinterpret_cast<const V*>(D)[i];
        V a = reinterpret_cast<const V*>(A)[i];
        V e = reinterpret_cast<const V*>(E)[i];
        V y = reinterpret_cast<const V*>(Y)[i];
        V r;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            if (nonnegA) a[c] = pos_part(a[c]);  // A .= max.(A,0)   :217-219
            T z = (d[c] - a[c]) - e[c];          // @. Z = D - A - E :221
            r[c] = z;
            y[c] = y[c] + mu * z;                // @. Y = Y + μ*Z   :222
        }
        if (nonnegA) reinterpret_cast<V*>(A)[i] = a;
        reinterpret_cast<V*>(R)[i] = r;
        reinterpret_cast<V*>(Y)[i] = y;
    }
    for (int64_t i = nv * VEC + tid; i < n; i += stride) {
        T a = A[i];
        if (nonnegA) {
            a = pos_part(a);
            A[i] = a;
        }
        T z = (D[i] - a) - E[i];
        R[i] = z;
        Y[i] = Y[i] + mu * z;
    }
}

// Fused update(k) + shrink(k+1): one pass reads D, A, E_k, Y and writes R_k, Y_k, E_{k+1}, Z_{k+1}
// (8 array passes instead of the 11 of k_update followed by k_shrink; arithmetic and expression order unchanged:
//  src/robustPCA.jl:217-223 for iteration k, then :188-192 for iteration k+1 with mu_{k+1}).
// E_{k+1}, Z_{k+1} go to buffers of their own: if iteration k turns out to be the last one, E_k and Z_k are
// the results.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void k_update_shrink(const T* __restrict__ D, T* __restrict__ A,
                                                       const T* __restrict__ E, T* __restrict__ Y,
                                                       T* __restrict__ R, T* __restrict__ En, T* __restrict__ Zn,
                                                       int64_t n, T mu, int nonnegA, T inv_mu_n, T thr_n,
                                                       int nonnegE, double* __restrict__ sumsq,
                                                       double* __restrict__ zero_slots) {
    using V = T __attribute__((ext_vector_type(VEC)));
    // the accumulator set of the NEXT sweep is cleared here (its previous contents were read back long ago): no memset
    if (zero_slots && blockIdx.x == 0 && threadIdx.x < 64) zero_slots[threadIdx.x] = 0.0;
    const int64_t nv = n / VEC;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double ss = 0.0;   // ||R||_F^2 of this thread's elements (a bound for the convergence test, not a result)
#ifndef TLSQ_SWEEP_UNROLL
#define TLSQ_SWEEP_UNROLL 1
#endif
#ifndef TLSQ_SWEEP_NT
#define TLSQ_SWEEP_NT 1   // streaming panels: nothing is re-read before ~0.6 GB of other traffic
#endif
#if TLSQ_SWEEP_NT
#define SW_LD(p, i) __builtin_nontemporal_load(reinterpret_cast<const V*>(p) + (i))
#define SW_ST(p, i, v) __builtin_nontemporal_store(v, reinterpret_cast<V*>(p) + (i))
#else
#define SW_LD(p, i) (reinterpret_cast<const V*>(p)[i])
#define SW_ST(p, i, v) (reinterpret_cast<V*>(p)[i] = (v))
#endif
#pragma unroll TLSQ_SWEEP_UNROLL
    for (int64_t i = tid; i < nv; i += stride) {
        V d = SW_LD(D, i);
        V a = SW_LD(A, i);
        V e = SW_LD(E, i);
        V y = SW_LD(Y, i);
        V r, en, zn;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            if (nonnegA) a[c] = pos_part(a[c]);          // A .= max.(A,0)            :217-219
            T z = (d[c] - a[c]) - e[c];                  // @. Z = D - A - E          :221
            ss += (double)z * (double)z;
            r[c] = z;
            y[c] = y[c] + mu * z;                        // @. Y = Y + mu*Z           :222
            T t = inv_mu_n * y[c];                       // next iteration, mu_{k+1}  :188
            T ee = soft_th((d[c] - a[c]) + t, thr_n);
            if (nonnegE) ee = pos_part(ee);              //                           :189-191
            en[c] = ee;
            zn[c] = (d[c] - ee) + t;                     //                           :192
        }
        if (nonnegA) SW_ST(A, i, a);
        if (R) SW_ST(R, i, r);   // R == nullptr: the caller predicted that this residual will not be looked at
        SW_ST(Y, i, y);
        SW_ST(En, i, en);
        SW_ST(Zn, i, zn);
    }
#undef SW_LD
#undef SW_ST
    for (int64_t i = nv * VEC + tid; i < n; i += stride) {
        T a = A[i];
        if (nonnegA) {
            a = pos_part(a);
            A[i] = a;
        }
        T z = (D[i] - a) - E[i];
        ss += (double)z * (double)z;
        if (R) R[i] = z;
        T y = Y[i] + mu * z;
        Y[i] = y;
        T t = inv_mu_n * y;
        T ee = soft_th((D[i] - a) + t, thr_n);
        if (nonnegE) ee = pos_part(ee);
        En[i] = ee;
        Zn[i] = (D[i] - ee) + t;
    }
    if (sumsq) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
        __shared__ double sw[4];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) sw[w] = ss;
        __syncthreads();
        // 64 accumulators (the host adds them up): one address would serialise tens of thousands of atomics
        if (threadIdx.x == 0) atomicAdd(sumsq + (blockIdx.x & 63), (sw[0] + sw[1]) + (sw[2] + sw[3]));
    }
}

// Rebuild + update(k) + shrink(k+1) in one pass: A_k = T Vs' (T = Z V_svp diag(g), M x r; Vs = V_svp, N x r; the
// singular-value-thresholded low-rank matrix of src/robustPCA.jl:205-213) is formed in registers and never stored,
// so an ALM iteration reads D, E_k, Y and writes R_k, Y, E_{k+1}, Z_{k+1}: 7 panel passes, and the skinny GEMM that
// used to write A (one more pass) disappears.  A thread owns two consecutive rows (16-byte accesses for fp64) and
// walks over the 64 columns of its tile; the Vs tile sits in LDS as [column][i] and is read as broadcasts.  The
// caller materialises A from the last T, Vs after the loop.
constexpr int RUS_CT = 64;   // widest column tile
// ROWS = 2: a thread owns two consecutive rows (16-byte accesses for fp64) - the streaming form for tall panels.
// ROWS = 1: one row per thread and narrower column tiles, for panels too small to fill the chip otherwise.
// HK = true: D is a Hankel matrix that is not read at all - D[k, c] = x[k lag + l, d] with c = l Dch + d for k < K (the
// lag embedding of lowrankfilter, src/robustPCA.jl:76-92: `D[k, d:D:L*D] = x[(k-1)*lag .+ (1:L), d]`; one channel and lag 1:
// D[i, j] = y[i + j]), zero pad rows below; `D` then points at the series (channel d at offset d ldx).  One panel pass less.
template <typename T, int RMAX, int ROWS, bool HK>
__global__ __launch_bounds__(256) void k_rebuild_update_shrink(const T* __restrict__ D, const double* __restrict__ Tm,
                                                               const double* __restrict__ Vs,
                                                               const T* __restrict__ E, T* __restrict__ Y,
                                                               T* __restrict__ R, T* __restrict__ En,
                                                               T* __restrict__ Zn, int64_t M, int N, int r, int ct,
                                                               T mu, int nonnegA, T inv_mu_n, T thr_n, int nonnegE,
                                                               double* __restrict__ sumsq,
                                                               double* __restrict__ zero_slots, int64_t hankel_K,
                                                               int64_t row0, int64_t row1, HankelGeom hg) {
    // rows [row0, row1) of the panels (row0 a multiple of ROWS): the whole panel, or one row chunk of it when the caller
    // interleaves the chunks with the Gram kernel of the rows already swept (solver.hip)
    using VR = T __attribute__((ext_vector_type(ROWS)));
    if (zero_slots && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) zero_slots[threadIdx.x] = 0.0;
    __shared__ __attribute__((aligned(16))) double sVs[RUS_CT * RMAX];
    const int c0 = blockIdx.y * ct;
    const int nct = (N - c0 < ct) ? N - c0 : ct;
    for (int e = threadIdx.x; e < ct * RMAX; e += 256) {
        const int c = e / RMAX, i = e % RMAX;
        sVs[e] = (c < nct && i < r) ? Vs[(size_t)(c0 + c) + (size_t)i * N] : 0.0;
    }
    __syncthreads();
    const int64_t row = row0 + ((int64_t)blockIdx.x * 256 + threadIdx.x) * ROWS;   // ROWS = 2: M, row0, row1 are even (launcher)
    double ss = 0.0;
    if (row < row1) {
        double t[ROWS][RMAX];
#pragma unroll
        for (int i = 0; i < RMAX; ++i)
#pragma unroll
            for (int q = 0; q < ROWS; ++q) t[q][i] = i < r ? Tm[row + q + (size_t)i * M] : 0.0;
#pragma unroll 4
        for (int c = 0; c < nct; ++c) {
            const int64_t idx = (row + (int64_t)(c0 + c) * M) / ROWS;
            VR d;
            if constexpr (HK) {
#pragma unroll
                for (int q = 0; q < ROWS; ++q) d[q] = (row + q < hankel_K) ? D[(row + q) * hg.lag + hankel_coff(hg, c0 + c)] : (T)0;
            } else {
                d = __builtin_nontemporal_load(reinterpret_cast<const VR*>(D) + idx);
            }
            const VR e = __builtin_nontemporal_load(reinterpret_cast<const VR*>(E) + idx);
            VR y = __builtin_nontemporal_load(reinterpret_cast<const VR*>(Y) + idx);
            const double* vs = sVs + c * RMAX;
            double acc[ROWS];
#pragma unroll
            for (int q = 0; q < ROWS; ++q) acc[q] = 0.0;
#pragma unroll
            for (int i = 0; i < RMAX; ++i)
#pragma unroll
                for (int q = 0; q < ROWS; ++q) acc[q] = __builtin_fma(t[q][i], vs[i], acc[q]);
            VR rr, en, zn;
#pragma unroll
            for (int q = 0; q < ROWS; ++q) {
                T a = (T)acc[q];
                if (nonnegA) a = pos_part(a);                // A .= max.(A,0)            :217-219
                const T z = (d[q] - a) - e[q];               // @. Z = D - A - E          :221
                ss += (double)z * (double)z;
                rr[q] = z;
                y[q] = y[q] + mu * z;                        // @. Y = Y + mu*Z           :222
                const T tt = inv_mu_n * y[q];                // next iteration, mu_{k+1}  :188
                T ee = soft_th((d[q] - a) + tt, thr_n);
                if (nonnegE) ee = pos_part(ee);              //                           :189-191
                en[q] = ee;
                zn[q] = (d[q] - ee) + tt;                    //                           :192
            }
            if (R) __builtin_nontemporal_store(rr, reinterpret_cast<VR*>(R) + idx);
            __builtin_nontemporal_store(y, reinterpret_cast<VR*>(Y) + idx);
            __builtin_nontemporal_store(en, reinterpret_cast<VR*>(En) + idx);
            __builtin_nontemporal_store(zn, reinterpret_cast<VR*>(Zn) + idx);
        }
    }
    if (sumsq) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
        __shared__ double sw[4];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) sw[w] = ss;
        __syncthreads();
        if (threadIdx.x == 0)
            atomicAdd(sumsq + ((blockIdx.x + blockIdx.y) & 63), (sw[0] + sw[1]) + (sw[2] + sw[3]));
    }
}

// ---- the E-free sweep -------------------------------------------------------------------------------------------------
// The loop state of the reference is (A, E, Y, mu), and Z_k = D - E_k + Y_k / mu_k is what the SVD step reads.  Two
// identities of the iteration (src/robustPCA.jl:188-222) make E redundant while the loop runs:
//     R_k     = D - A_k - E_k            = (Z_k - A_k) - Y_k / mu_k                      (:221)
//     Y_{k+1} = Y_k + mu_k R_k           = mu_k (Z_k - A_k)                              (:222)
// so update(k) + shrink(k+1) needs D, A_k (from its factors, in registers), Y_k and Z_k and produces Y_{k+1} and Z_{k+1}
// (in place): 5 panel passes (4 with an implicit Hankel D) instead of the 7 (+ the store of A) of the sweeps above.  E_{k+1}
// exists only in registers; the caller forms the E it returns once, after the loop, with the reference's own statement
// E_k = soft_th(D - A_{k-1} + Y_k / mu_k, lambda / mu_k) (k_final_e: exact zeros where the reference has them) from the
// previous iteration's factors and Y_k - which is why Y is double-buffered here instead of E and Z.  The two identities
// hold up to the rounding of Z (one ulp of |D|), the same size as the rounding of the reference's own statements.
// ZIP (in place, Z_{k+1} over Z_k): Z is not used and every read of Z_k goes through Zo as well - two restrict-qualified
// pointers to one panel, one of them written, would be undefined behaviour however harmless the access order is.
template <typename T, int RMAX, int ROWS, bool HK, bool ZIP>
__global__ __launch_bounds__(256) void k_zsweep(const T* __restrict__ D, const double* __restrict__ Tm,
                                                const double* __restrict__ Vs, const T* __restrict__ Yin,
                                                T* __restrict__ Yout, const T* __restrict__ Z, T* __restrict__ Zo,
                                                T* __restrict__ R, int64_t M,
                                                int N, int r, int ct, T mu, T inv_mu, int nonnegA, T inv_mu_n, T thr_n,
                                                int nonnegE, double* __restrict__ sumsq, double* __restrict__ zero_slots,
                                                int64_t hankel_K, int64_t row0, int64_t row1, int maxslot, HankelGeom hg) {
    // sumsq: 72 doubles - [0, 64) partial sums of ||R_k||_F^2, [64 + maxslot] (maxslot >= 0) max |R_k[i, j]| as a bit
    // pattern: both are lower bounds of ||R_k||_2 for the convergence test (solver.hip), never results
    using VR = T __attribute__((ext_vector_type(ROWS)));
    if (zero_slots && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 72) zero_slots[threadIdx.x] = 0.0;
    __shared__ __attribute__((aligned(16))) double sVs[RUS_CT * RMAX];
    const int c0 = blockIdx.y * ct;
    const int nct = (N - c0 < ct) ? N - c0 : ct;
    for (int e = threadIdx.x; e < ct * RMAX; e += 256) {
        const int c = e / RMAX, i = e % RMAX;
        sVs[e] = (c < nct && i < r) ? Vs[(size_t)(c0 + c) + (size_t)i * N] : 0.0;
    }
    __syncthreads();
    const int64_t row = row0 + ((int64_t)blockIdx.x * 256 + threadIdx.x) * ROWS;   // ROWS = 2: M, row0, row1 are even (launcher)
    double ss = 0.0;
    T rmax = (T)0;
    if (row < row1) {
        double t[ROWS][RMAX];
#pragma unroll
        for (int i = 0; i < RMAX; ++i)
#pragma unroll
            for (int q = 0; q < ROWS; ++q) t[q][i] = i < r ? Tm[row + q + (size_t)i * M] : 0.0;
#pragma unroll 4
        for (int c = 0; c < nct; ++c) {
            const int64_t idx = (row + (int64_t)(c0 + c) * M) / ROWS;
            VR d;
            if constexpr (HK) {
#pragma unroll
                for (int q = 0; q < ROWS; ++q) d[q] = (row + q < hankel_K) ? D[(row + q) * hg.lag + hankel_coff(hg, c0 + c)] : (T)0;
            } else {
                d = __builtin_nontemporal_load(reinterpret_cast<const VR*>(D) + idx);
            }
            const VR z = __builtin_nontemporal_load(reinterpret_cast<const VR*>(ZIP ? (const T*)Zo : Z) + idx);
            const VR y = __builtin_nontemporal_load(reinterpret_cast<const VR*>(Yin) + idx);
            const double* vs = sVs + c * RMAX;
            double acc[ROWS];
#pragma unroll
            for (int q = 0; q < ROWS; ++q) acc[q] = 0.0;
#pragma unroll
            for (int i = 0; i < RMAX; ++i)
#pragma unroll
                for (int q = 0; q < ROWS; ++q) acc[q] = __builtin_fma(t[q][i], vs[i], acc[q]);
            VR rr, yn, zn;
#pragma unroll
            for (int q = 0; q < ROWS; ++q) {
                T a = (T)acc[q];
                if (nonnegA) a = pos_part(a);                // A .= max.(A,0)                          :217-219
                const T w = z[q] - a;
                const T res = w - inv_mu * y[q];             // R_k = D - A - E                         :221
                ss += (double)res * (double)res;
                const T ares = res < (T)0 ? -res : res;
                rmax = ares > rmax ? ares : rmax;            // (a NaN never wins: the bound stays a bound)
                rr[q] = res;
                const T y1 = mu * w;                         // Y_{k+1} = Y + mu R                      :222
                yn[q] = y1;
                const T tt = inv_mu_n * y1;                  // next iteration, mu_{k+1}                :188
                T ee = soft_th((d[q] - a) + tt, thr_n);
                if (nonnegE) ee = pos_part(ee);              //                                         :189-191
                zn[q] = (d[q] - ee) + tt;                    //                                         :192
            }
            if (R) __builtin_nontemporal_store(rr, reinterpret_cast<VR*>(R) + idx);
            __builtin_nontemporal_store(yn, reinterpret_cast<VR*>(Yout) + idx);
            __builtin_nontemporal_store(zn, reinterpret_cast<VR*>(Zo) + idx);   // (ZIP: in place, every entry read before it is written)
        }
    }
    if (sumsq) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
        __shared__ double sw[4];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) sw[w] = ss;
        __syncthreads();
        if (threadIdx.x == 0)
            atomicAdd(sumsq + ((blockIdx.x + blockIdx.y) & 63), (sw[0] + sw[1]) + (sw[2] + sw[3]));
        if (maxslot >= 0) {
            double m = (double)rmax;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double o = __shfl_down(m, off, 64);
                m = o > m ? o : m;
            }
            __syncthreads();
            if (lane == 0) sw[w] = m;
            __syncthreads();
            if (threadIdx.x == 0) {
                double mm = sw[0];
                for (int k = 1; k < 4; ++k) mm = sw[k] > mm ? sw[k] : mm;
                atomicMax(reinterpret_cast<unsigned long long*>(sumsq + 64 + maxslot),
                          (unsigned long long)__double_as_longlong(mm));   // (non-negative doubles: monotone bit patterns)
            }
        }
    }
}

// the same sweep with A_k read from memory (ranks above 32: A does not fit a thread's registers as factors)
template <typename T, int VEC, bool ZIP>
__global__ __launch_bounds__(256) void k_zsweep_lin(const T* __restrict__ D, T* __restrict__ A, const T* __restrict__ Yin,
                                                    T* __restrict__ Yout, const T* __restrict__ Z, T* __restrict__ Zo,
                                                    T* __restrict__ R, int64_t n,
                                                    T mu, T inv_mu, int nonnegA, T inv_mu_n, T thr_n, int nonnegE,
                                                    double* __restrict__ sumsq, double* __restrict__ zero_slots,
                                                    int maxslot) {
    using V = T __attribute__((ext_vector_type(VEC)));
    if (zero_slots && blockIdx.x == 0 && threadIdx.x < 72) zero_slots[threadIdx.x] = 0.0;
    const int64_t nv = n / VEC;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double ss = 0.0;
    T rmax = (T)0;
    auto one = [&](T d, T& a, T y, T z, T& res, T& y1, T& zn) {
        if (nonnegA) a = pos_part(a);
        const T w = z - a;
        res = w - inv_mu * y;
        ss += (double)res * (double)res;
        const T ares = res < (T)0 ? -res : res;
        rmax = ares > rmax ? ares : rmax;
        y1 = mu * w;
        const T tt = inv_mu_n * y1;
        T ee = soft_th((d - a) + tt, thr_n);
        if (nonnegE) ee = pos_part(ee);
        zn = (d - ee) + tt;
    };
    for (int64_t i = tid; i < nv; i += stride) {
        const V d = __builtin_nontemporal_load(reinterpret_cast<const V*>(D) + i);
        V a = __builtin_nontemporal_load(reinterpret_cast<const V*>(A) + i);
        const V y = __builtin_nontemporal_load(reinterpret_cast<const V*>(Yin) + i);
        const V z = __builtin_nontemporal_load(reinterpret_cast<const V*>(ZIP ? (const T*)Zo : Z) + i);
        V rr, yn, zn;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            T ac = a[c], r1, y1, z1;
            one(d[c], ac, y[c], z[c], r1, y1, z1);
            a[c] = ac;
            rr[c] = r1;
            yn[c] = y1;
            zn[c] = z1;
        }
        if (nonnegA) __builtin_nontemporal_store(a, reinterpret_cast<V*>(A) + i);
        if (R) __builtin_nontemporal_store(rr, reinterpret_cast<V*>(R) + i);
        __builtin_nontemporal_store(yn, reinterpret_cast<V*>(Yout) + i);
        __builtin_nontemporal_store(zn, reinterpret_cast<V*>(Zo) + i);
    }
    for (int64_t i = nv * VEC + tid; i < n; i += stride) {
        T ac = A[i], r1, y1, z1;
        one(D[i], ac, Yin[i], ZIP ? Zo[i] : Z[i], r1, y1, z1);
        if (nonnegA) A[i] = ac;
        if (R) R[i] = r1;
        Yout[i] = y1;
        Zo[i] = z1;
    }
    if (sumsq) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
        __shared__ double sw[4];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) sw[w] = ss;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(sumsq + (blockIdx.x & 63), (sw[0] + sw[1]) + (sw[2] + sw[3]));
        if (maxslot >= 0) {
            double m = (double)rmax;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double o = __shfl_down(m, off, 64);
                m = o > m ? o : m;
            }
            __syncthreads();
            if (lane == 0) sw[w] = m;
            __syncthreads();
            if (threadIdx.x == 0) {
                double mm = sw[0];
                for (int k = 1; k < 4; ++k) mm = sw[k] > mm ? sw[k] : mm;
                atomicMax(reinterpret_cast<unsigned long long*>(sumsq + 64 + maxslot),
                          (unsigned long long)__double_as_longlong(mm));
            }
        }
    }
}

// ---- the E-free sweep for ranks above 32 on fp32 panels: A_k on the fp32 MFMA inside the sweep (round 5) ---------------------
// Above 32 columns A_k = T Vs' does not fit a thread's registers as factors: the loop stored it (a skinny GEMM writing the
// panel: 0.78 ms at 65536 x 4096, r = 64) and k_zsweep_lin read it back (4 reads + 2 writes: 1.42 ms).  Here a wave owns a strip
// of 32 panel columns and walks down a run of 32-row tiles; for every tile A_k[32 x 32] = Vs[strip] T[tile]' comes out of
// 2 KH v_mfma_f32_32x32x2_f32 (the factors as fp32: Vs fragments stay in registers for the whole run, the T fragments of the
// tile are re-read from L2 - T32 is 17 MB) with the ROW index in the lanes: register (g, e) of lane l is the element (row
// m0 + l % 32, column n0 + 8 g + 4 (l / 32) + e), so every D / Y / Z access of the sweep's element-wise part (:217-222 and
// the next :188-192, exactly the statements of k_zsweep_lin) is two 128-byte runs per instruction, and consecutive tiles
// continue the same 32 column streams.  3 reads + 2 writes of the panel (+ R), no stored A.
// Workgroup = 4 waves = 4 neighbouring strips on the same rows (the T fragments are shared through L1/L2).
typedef float zw_f16 __attribute__((ext_vector_type(16)));

// (buffer addressing: a wave-uniform 64-bit base in a descriptor, ONE 32-bit lane offset per tile and the element's column
//  offset as the instruction's scalar offset - 16 global pointers per array would not fit the register file beside the fragments)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t zw_rsrc(const float* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)hi << 32) | lo), 0, 0xFFFFFFFFu, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ float zw_ld(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, AUX));
}
template <int AUX>
__device__ __forceinline__ void zw_st(float v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, (int)voff, (int)soff, AUX);
}

template <int KH, bool ZIP, bool HASR>
__global__ __launch_bounds__(256, 3) void k_zsweep_wide(const float* __restrict__ D, const float* __restrict__ T32, int64_t ldt,
                                                        const float* __restrict__ Vs32, const float* __restrict__ Yin,
                                                        float* __restrict__ Yout, const float* __restrict__ Z, float* __restrict__ Zo,
                                                        float* __restrict__ R, int64_t M, int N, int64_t rows_per_chunk, float mu,
                                                        float inv_mu, int nonnegA, float inv_mu_n, float thr_n, int nonnegE,
                                                        double* __restrict__ sumsq, double* __restrict__ zero_slots, int maxslot,
                                                        unsigned int* __restrict__ zmax_bits) {
    if (zero_slots && blockIdx.x == 0 && threadIdx.x < 72) zero_slots[threadIdx.x] = 0.0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int jl = lane & 31, kh = lane >> 5;
    unsigned int zmax = 0;   // max |Z_{k+1}| as a bit pattern (the scale of the split Gram kernel that reads the panel next)
    const int nsg = N / 128;
    const int64_t chunk = blockIdx.x / nsg;
    const int n0 = ((int)(blockIdx.x % nsg) * 4 + w) * 32;
    float af[KH];
#pragma unroll
    for (int s = 0; s < KH; ++s) af[s] = Vs32[(size_t)(n0 + jl) + (size_t)(2 * s + kh) * N];
    const int64_t mbeg = chunk * rows_per_chunk;
    const int64_t mend = (mbeg + rows_per_chunk < M) ? mbeg + rows_per_chunk : M;
    const int ntile = (int)((mend - mbeg) / 32);
    // element i = 4 g + e of the tile at row m0: (row m0 + jl, column n0 + 4 kh + 8 g + e)
    const int64_t strip = (int64_t)n0 * M + mbeg;                  // first element of the strip's run
    const __amdgpu_buffer_rsrc_t rD = zw_rsrc(D + strip), rY = zw_rsrc(Yin + strip), rYo = zw_rsrc(Yout + strip),
                                 rZ = zw_rsrc((ZIP ? (const float*)Zo : Z) + strip), rZo = zw_rsrc(Zo + strip),
                                 rR = zw_rsrc(HASR ? R + strip : Zo + strip), rT = zw_rsrc(T32 + mbeg);
    const uint32_t m4 = (uint32_t)M * 4u, ldt4 = (uint32_t)ldt * 4u;
    const uint32_t voff0 = (uint32_t)(4 * kh) * m4 + (uint32_t)jl * 4u;   // + 128 t
    const uint32_t toff0 = (uint32_t)kh * ldt4 + (uint32_t)jl * 4u;
    double ss = 0.0;
    float rmax = 0.f;
    // No software pipeline across tiles: a wave requests the 2 KH + 48 values of its tile at once (up to the 63 requests the
    // counter allows in flight), and 8-12 waves per CU x 20 KB cover the HBM latency by themselves (Little: 2048 waves x 20 KB
    // / 2 us = 20 TB/s) - registers carried over the loop's back edge only brought copies and waits.
    for (int t = 0; t < ntile; ++t) {
        const uint32_t to = toff0 + 128u * (uint32_t)t;
        const uint32_t vo = voff0 + 128u * (uint32_t)t;
        float bf[KH];
        float dv[16], yv[16], zv[16];
#pragma unroll
        for (int s = 0; s < KH; ++s) bf[s] = zw_ld<0>(rT, to, (uint32_t)(2 * s) * ldt4);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t so = (uint32_t)(8 * (i >> 2) + (i & 3)) * m4;
            zv[i] = zw_ld<2>(rZ, vo, so);
            yv[i] = zw_ld<2>(rY, vo, so);
            dv[i] = zw_ld<2>(rD, vo, so);
        }
        zw_f16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < KH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc, 0, 0, 0);
        float s32 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t so = (uint32_t)(8 * (i >> 2) + (i & 3)) * m4;
            float a = acc[i];
            if (nonnegA) a = pos_part(a);                   // :217-219
            const float wv = zv[i] - a;
            const float res = wv - inv_mu * yv[i];          // :221 through Z_k = D - E_k + Y_k / mu_k
            s32 += res * res;
            const float ares = res < 0.f ? -res : res;
            rmax = ares > rmax ? ares : rmax;
            const float y1 = mu * wv;                       // :222
            const float tt = inv_mu_n * y1;
            float ee = soft_th((dv[i] - a) + tt, thr_n);    // :188
            if (nonnegE) ee = pos_part(ee);
            const float zn = (dv[i] - ee) + tt;             // :192
            if (HASR) zw_st<2>(res, rR, vo, so);
            zw_st<2>(y1, rYo, vo, so);
            zw_st<2>(zn, rZo, vo, so);
            const unsigned int zb = __float_as_uint(zn) & 0x7FFFFFFFu;
            zmax = zb > zmax ? zb : zmax;
        }
        ss += (double)s32;
        __builtin_amdgcn_sched_barrier(0);
    }
    if (zmax_bits) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned int o = (unsigned int)__shfl_xor((int)zmax, off, 64);
            zmax = o > zmax ? o : zmax;
        }
        if (lane == 0 && zmax) atomicMax(zmax_bits, zmax);
    }
    if (sumsq) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
        __shared__ double sw[4];
        if (lane == 0) sw[w] = ss;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(sumsq + (blockIdx.x & 63), (sw[0] + sw[1]) + (sw[2] + sw[3]));
        if (maxslot >= 0) {
            double m = (double)rmax;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double o = __shfl_down(m, off, 64);
                m = o > m ? o : m;
            }
            __syncthreads();
            if (lane == 0) sw[w] = m;
            __syncthreads();
            if (threadIdx.x == 0) {
                double mm = sw[0];
                for (int k = 1; k < 4; ++k) mm = sw[k] > mm ? sw[k] : mm;
                atomicMax(reinterpret_cast<unsigned long long*>(sumsq + 64 + maxslot),
                          (unsigned long long)__double_as_longlong(mm));
            }
        }
    }
}

bool zsweep_wide_ok(int64_t M, int64_t N, int64_t r) {
    // 32-bit byte offsets: inside a strip 108 M < 2^32, and the T32 scalar offset (2 KH - 1) * 4 * M with KH = 32 (r <= 64) or
    // 40 (r <= 80) - the larger factor, 79 * 4 * M < 2^32, caps M at 13.5e6 for ranks above 64
    const int64_t KH = r <= 64 ? 32 : 40;
    const int64_t mmax = std::min<int64_t>((int64_t)1 << 24, (((int64_t)1 << 32) - 1) / (4 * (2 * KH - 1)));
    return r > 32 && r <= 80 && (M % 32) == 0 && (N % 128) == 0 && M >= 4096 && M <= mmax && N <= 1000000;
}

// T32: M x 2 KH (ld ldt) and Vs32: N x 2 KH (ld N), KH = 32 for r <= 64 and 40 for r <= 80, columns r.. zero
int launch_zsweep_wide(Handle* h, const float* D, const float* T32, int64_t ldt, const float* Vs32, int64_t r, const float* Yin,
                       float* Yout, float* Z, float* Zout, float* R, int64_t M, int64_t N, float mu, float inv_mu, int nonnegA,
                       float inv_mu_n, float thr_n, int nonnegE, double* sumsq, double* zero_slots, int maxslot, bool leave_absmax) {
    if (!zsweep_wide_ok(M, N, r)) return set_err(h, TLSQ_ERR_ARG, "zsweep_wide: shape");
    unsigned int* zmax_bits = nullptr;
    h->absmax_panel = nullptr;
    if (leave_absmax) {
        void* sc;
        TLSQ_TRY(ws_get(h, WS_H16S, 64, &sc));
        zmax_bits = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(sc) + 40);
        TLSQ_HIP(h, hipMemsetAsync(zmax_bits, 0, 4, h->stream));
    }
    if (!Zout) Zout = Z;
    if (!sumsq || maxslot > 7) maxslot = -1;
    const bool zip = Zout == Z;
    const float* Zr = zip ? nullptr : Z;
    // ~16 waves per CU (126 registers: four waves per SIMD): 4096 waves over N / 32 strips
    const int64_t nstrips = N / 32;
    int64_t nchunks = std::max<int64_t>(1, (4096 + nstrips - 1) / nstrips);
    int64_t rpc = (M + nchunks - 1) / nchunks;
    rpc = (rpc + 31) / 32 * 32;
    nchunks = (M + rpc - 1) / rpc;
    const dim3 grid((unsigned)(nchunks * (N / 128)));
#define ZW_LAUNCH(KHV, ZP, HR)                                                                                              \
    hipLaunchKernelGGL((k_zsweep_wide<KHV, ZP, HR>), grid, dim3(256), 0, h->stream, D, T32, ldt, Vs32, Yin, Yout, Zr, Zout, R, M, \
                       (int)N, rpc, mu, inv_mu, nonnegA, inv_mu_n, thr_n, nonnegE, sumsq, zero_slots, maxslot, zmax_bits)
#define ZW_PICK(KHV)                          \
    do {                                      \
        if (zip) {                            \
            if (R) ZW_LAUNCH(KHV, true, true); \
            else ZW_LAUNCH(KHV, true, false);  \
        } else {                              \
            if (R) ZW_LAUNCH(KHV, false, true); \
            else ZW_LAUNCH(KHV, false, false);  \
        }                                     \
    } while (0)
    if (r <= 64) ZW_PICK(32);
    else ZW_PICK(40);
#undef ZW_PICK
#undef ZW_LAUNCH
    ++h->kern_zsweep_wide;
    TLSQ_HIP(h, hipGetLastError());
    if (leave_absmax) h->absmax_panel = Zout;
    return TLSQ_OK;
}

// E = soft_th(D - A_prev + Y / mu, lambda / mu) (:188-191) once after the E-free loop, A_prev = Tm Vs' from the factors of the
// previous iteration (r = 0: A_prev = 0, the first iteration).  E may be the buffer Y lives in (same element read, then
// written): no __restrict__ on the two.
template <typename T, int RMAX, int ROWS, bool HK>
__global__ __launch_bounds__(256) void k_final_e(const T* __restrict__ D, const double* __restrict__ Tm,
                                                 const double* __restrict__ Vs, const T* Y, T* E, int64_t M, int N, int r,
                                                 int ct, T inv_mu, T thr, int nonnegA, int nonnegE, int64_t hankel_K,
                                                 HankelGeom hg) {
    using VR = T __attribute__((ext_vector_type(ROWS)));
    __shared__ __attribute__((aligned(16))) double sVs[RUS_CT * RMAX];
    const int c0 = blockIdx.y * ct;
    const int nct = (N - c0 < ct) ? N - c0 : ct;
    for (int e = threadIdx.x; e < ct * RMAX; e += 256) {
        const int c = e / RMAX, i = e % RMAX;
        sVs[e] = (c < nct && i < r) ? Vs[(size_t)(c0 + c) + (size_t)i * N] : 0.0;
    }
    __syncthreads();
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) * ROWS;
    if (row >= M) return;
    double t[ROWS][RMAX];
#pragma unroll
    for (int i = 0; i < RMAX; ++i)
#pragma unroll
        for (int q = 0; q < ROWS; ++q) t[q][i] = i < r ? Tm[row + q + (size_t)i * M] : 0.0;
#pragma unroll 4
    for (int c = 0; c < nct; ++c) {
        const int64_t idx = (row + (int64_t)(c0 + c) * M) / ROWS;
        VR d;
        if constexpr (HK) {
#pragma unroll
            for (int q = 0; q < ROWS; ++q) d[q] = (row + q < hankel_K) ? D[(row + q) * hg.lag + hankel_coff(hg, c0 + c)] : (T)0;
        } else {
            d = reinterpret_cast<const VR*>(D)[idx];
        }
        const VR y = reinterpret_cast<const VR*>(Y)[idx];
        const double* vs = sVs + c * RMAX;
        double acc[ROWS];
#pragma unroll
        for (int q = 0; q < ROWS; ++q) acc[q] = 0.0;
#pragma unroll
        for (int i = 0; i < RMAX; ++i)
#pragma unroll
            for (int q = 0; q < ROWS; ++q) acc[q] = __builtin_fma(t[q][i], vs[i], acc[q]);
        VR en;
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            T a = (T)acc[q];
            if (nonnegA) a = pos_part(a);
            T ee = soft_th((d[q] - a) + inv_mu * y[q], thr);
            if (nonnegE) ee = pos_part(ee);
            en[q] = ee;
        }
        reinterpret_cast<VR*>(E)[idx] = en;
    }
}

// ... and with A_prev in memory (ranks above 32)
template <typename T>
__global__ __launch_bounds__(256) void k_final_e_lin(const T* __restrict__ D, const T* __restrict__ A, const T* Y, T* E,
                                                     int64_t n, T inv_mu, T thr, int nonnegE) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        T ee = soft_th((D[i] - A[i]) + inv_mu * Y[i], thr);
        if (nonnegE) ee = pos_part(ee);
        E[i] = ee;
    }
}

// The returned E when A_{k-1} has no factor form (the iteration before was served by the matrix-function route, solver.hip):
// E_k = D - A_k - R_k with R_k = (Y_{k+1} - Y_k) / mu_k (:221-222), or E_k = D - Z_k + Y_k / mu_k (:192) when no sweep followed.
// Both equal the reference's soft_th statement up to rounding; entries that are rounding noise of an exact zero (below 32 eps
// of the magnitudes that went into them) are set to zero, so that E keeps the zero pattern of the shrinkage.
// E may be the buffer Y0 / Y lives in.
template <typename T>
__global__ __launch_bounds__(256) void k_e_from_residual(const T* __restrict__ D, const T* __restrict__ A, const T* Y1,
                                                         const T* Y0, T* E, int64_t n, T inv_mu) {
    const T eps32 = (T)32 * std::numeric_limits<T>::epsilon();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const T d = D[i], a = A[i], y1 = Y1[i] * inv_mu, y0 = Y0[i] * inv_mu;
        const T e = (d - a) - (y1 - y0);
        const T mag = fabs(d) + fabs(a) + fabs(y1) + fabs(y0);
        E[i] = fabs(e) <= eps32 * mag ? (T)0 : e;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_e_from_z(const T* __restrict__ D, const T* __restrict__ Z, const T* Y, T* E,
                                                  int64_t n, T inv_mu) {
    const T eps32 = (T)32 * std::numeric_limits<T>::epsilon();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const T d = D[i], z = Z[i], y = Y[i] * inv_mu;
        const T e = (d - z) + y;
        const T mag = fabs(d) + fabs(z) + fabs(y);
        E[i] = fabs(e) <= eps32 * mag ? (T)0 : e;
    }
}

// R_k = (Y_{k+1} - Y_k) / mu_k (:222 solved for the residual): only when an E-free sweep was told not to store R_k and
// the cost evaluation wants it after all
template <typename T>
__global__ __launch_bounds__(256) void k_residual_from_y(const T* __restrict__ Y1, const T* __restrict__ Y0,
                                                         T* __restrict__ R, int64_t n, T inv_mu) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) R[i] = (Y1[i] - Y0[i]) * inv_mu;
}

// Z_k = A_k + Y_{k+1} / mu_k (:222 with the second identity above): the panel whose SVD the call returns, rebuilt once after
// an E-free loop whose last sweep already overwrote it with Z_{k+1}
template <typename T>
__global__ __launch_bounds__(256) void k_z_from_y(const T* __restrict__ A, const T* __restrict__ Y1, T* __restrict__ Z,
                                                  int64_t n, T inv_mu) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) Z[i] = A[i] + Y1[i] * inv_mu;
}

// A (M x N, ld M) = Tm Vs' alone: the store-bound form of the rebuild (src/robustPCA.jl:207-208 / 211-212) for the
// panels that keep A in memory (C2 size).  Same walk as above: a thread owns two consecutive rows of T in registers,
// the Vs tile is broadcast from LDS, every store is 16 bytes per lane and 1 KB contiguous per wave.  The 128 x 128
// tile GEMM needs 24 us for this 82 MB store at 20000 x 512; here the store stream is all there is.
template <typename T, int RMAX>
__global__ __launch_bounds__(256) void k_rebuild_store(const double* __restrict__ Tm, const double* __restrict__ Vs,
                                                       T* __restrict__ A, int64_t M, int N, int r, int ct) {
    using VR = T __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) double sVs[RUS_CT * RMAX];
    const int c0 = blockIdx.y * ct;
    const int nct = (N - c0 < ct) ? N - c0 : ct;
    for (int e = threadIdx.x; e < ct * RMAX; e += 256) {
        const int c = e / RMAX, i = e % RMAX;
        sVs[e] = (c < nct && i < r) ? Vs[(size_t)(c0 + c) + (size_t)i * N] : 0.0;
    }
    __syncthreads();
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (row >= M) return;
    double t[2][RMAX];
#pragma unroll
    for (int i = 0; i < RMAX; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) t[q][i] = i < r ? Tm[row + q + (size_t)i * M] : 0.0;
#pragma unroll 4
    for (int c = 0; c < nct; ++c) {
        const double* vs = sVs + c * RMAX;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int i = 0; i < RMAX; ++i) {
            a0 = __builtin_fma(t[0][i], vs[i], a0);
            a1 = __builtin_fma(t[1][i], vs[i], a1);
        }
        VR out;
        out[0] = (T)a0;
        out[1] = (T)a1;
        *(reinterpret_cast<VR*>(A) + (row + (int64_t)(c0 + c) * M) / 2) = out;
    }
}

template <typename T>
bool rebuild_store_ok(const T* A, int64_t M, int64_t N, int64_t ldA, int64_t r) {
    return r >= 1 && r <= 32 && (M % 2) == 0 && ldA == M && (reinterpret_cast<uintptr_t>(A) % (2 * sizeof(T))) == 0 &&
           N <= 2147483647LL;
}

template <typename T>
int launch_rebuild_store(Handle* h, const double* Tm, const double* Vs, T* A, int64_t M, int64_t N, int64_t r) {
    const int ct = 32;
    const dim3 grid((unsigned)((M / 2 + 255) / 256), (unsigned)((N + ct - 1) / ct));
    if (r <= 8) hipLaunchKernelGGL((k_rebuild_store<T, 8>), grid, dim3(256), 0, h->stream, Tm, Vs, A, M, (int)N, (int)r, ct);
    else if (r <= 16) hipLaunchKernelGGL((k_rebuild_store<T, 16>), grid, dim3(256), 0, h->stream, Tm, Vs, A, M, (int)N, (int)r, ct);
    else hipLaunchKernelGGL((k_rebuild_store<T, 32>), grid, dim3(256), 0, h->stream, Tm, Vs, A, M, (int)N, (int)r, ct);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
template bool rebuild_store_ok<double>(const double*, int64_t, int64_t, int64_t, int64_t);
template bool rebuild_store_ok<float>(const float*, int64_t, int64_t, int64_t, int64_t);
template int launch_rebuild_store<double>(Handle*, const double*, const double*, double*, int64_t, int64_t, int64_t);
template int launch_rebuild_store<float>(Handle*, const double*, const double*, float*, int64_t, int64_t, int64_t);

// The 64 partial sums of ||R||_F^2 go to the host-visible mailbox ([0] flag, [8..72) values) and are published with
// the sequence number: the host polls the flag instead of paying a copy command and an event between two kernels.
__global__ __launch_bounds__(128) void k_publish_slots(const double* __restrict__ slots, double* mailbox, double seq) {
    volatile double* mb = mailbox;
    if (threadIdx.x < 72) mb[8 + threadIdx.x] = slots[threadIdx.x];   // (64 sums + 8 per-rank maxima of |R|)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) mb[0] = seq;
}

// R = (D - A) - E  (the residual statement :221 alone, same expression order as in the sweeps): used when a sweep was
// told not to store R and the residual turns out to be needed after all
template <typename T>
__global__ __launch_bounds__(256) void k_residual(const T* __restrict__ D, const T* __restrict__ A,
                                                  const T* __restrict__ E, T* __restrict__ R, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) R[i] = (D[i] - A[i]) - E[i];
}

// R = D - A - E with D[i, j] = y[i + j] for i < K (zero pad rows below): the Hankel panel is not read (nor kept)
template <typename T>
__global__ __launch_bounds__(256) void k_residual_hankel(const T* __restrict__ y, int64_t K, const T* __restrict__ A,
                                                         const T* __restrict__ E, T* __restrict__ R, int64_t M,
                                                         int64_t N, HankelGeom hg) {
    const int64_t j = blockIdx.y;
    const T* __restrict__ yj = y + hankel_coff(hg, (int)j);
    const int64_t off = j * M;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += stride) {
        const T d = i < K ? yj[i * hg.lag] : (T)0;
        R[off + i] = (d - A[off + i]) - E[off + i];
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void k_div_scalar(const T* __restrict__ D, T* __restrict__ Y,
                                                    int64_t n, T s) {
    using V = T __attribute__((ext_vector_type(VEC)));
    const int64_t nv = n / VEC;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = tid; i < nv; i += stride) {
        V d = reinterpret_cast<const V*>(D)[i];
#pragma unroll
        for (int c = 0; c < VEC; ++c) d[c] = d[c] / s;  // Y ./= dual_norm   :181
        reinterpret_cast<V*>(Y)[i] = d;
    }
    for (int64_t i = nv * VEC + tid; i < n; i += stride) Y[i] = D[i] / s;
}

template <typename T>
__global__ __launch_bounds__(256) void k_clamp_nonneg(T* __restrict__ A, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        A[i] = pos_part(A[i]);
}

// max |x_i| : per-lane running max -> wave shuffle -> LDS across the 4 waves -> one atomicMax per block
// on the bit pattern (monotone for non-negative IEEE-754 values).
template <typename T>
__global__ __launch_bounds__(256) void k_maxabs(const T* __restrict__ x, int64_t n,
                                                unsigned long long* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double m = 0.0;
    auto take = [&](T xv) {
        double v = fabs((double)xv);
        if (!(v <= 1.7976931348623157e308)) v = __longlong_as_double(0x7FF0000000000000ll);   // NaN counts as Inf: "not finite"
        m = v > m ? v : m;
    };
    // 16-byte loads, four of them in flight per thread (8-byte loads one at a time: 2.3 TB/s, 35 us of the C2 set-up);
    // the order of a maximum does not matter
    constexpr int VE = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VE)));
    int64_t i0 = 0;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const int64_t nv = n / VE;
        const vec_t* xv = reinterpret_cast<const vec_t*>(x);
        int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + 3 * stride < nv; i += 4 * stride) {
            const vec_t a = xv[i], b = xv[i + stride], c = xv[i + 2 * stride], d = xv[i + 3 * stride];
#pragma unroll
            for (int q = 0; q < VE; ++q) {
                take(a[q]);
                take(b[q]);
                take(c[q]);
                take(d[q]);
            }
        }
        for (; i < nv; i += stride) {
            const vec_t a = xv[i];
#pragma unroll
            for (int q = 0; q < VE; ++q) take(a[q]);
        }
        i0 = nv * VE;
    }
    for (int64_t i = i0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) take(x[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_down(m, off, 64);
        m = o > m ? o : m;
    }
    __shared__ double sm[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sm[w] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sm[0];
        for (int k = 1; k < 4; ++k) r = sm[k] > r ? sm[k] : r;
        atomicMax(out, (unsigned long long)__double_as_longlong(r));
    }
}

// LDS-tiled transpose: dst (N x M, ld ldd) = src' , src (M x N, ld lds), both column-major
template <typename T>
__global__ __launch_bounds__(256) void k_transpose(const T* __restrict__ src, int64_t lds, int64_t M, int64_t N,
                                                   T* __restrict__ dst, int64_t ldd) {
    __shared__ T tile[32][33];
    const int64_t m0 = (int64_t)blockIdx.x * 32, n0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + tx, n = n0 + ty + 8 * r;
        if (m < M && n < N) tile[ty + 8 * r][tx] = src[m + n * lds];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t n = n0 + tx, m = m0 + ty + 8 * r;
        if (m < M && n < N) dst[n + m * ldd] = tile[tx][ty + 8 * r];
    }
}

// out = a - b (contiguous n)
template <typename T>
__global__ __launch_bounds__(256) void k_diff(const T* __restrict__ a, const T* b, T* out,   // (out may be b)
                                              int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = a[i] - b[i];
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void k_convert(const TS* __restrict__ src, TD* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (TD)src[i];
}

static inline int grid_for(int64_t work_items) {
    const int64_t cap = [] {
        const char* e = dev_get(DEV_SWEEP_GRID);   // tuning knob (tools/kbench.py)
        const long v = e ? atol(e) : 0;
        return (int64_t)(v > 0 ? v : 2048);          // 256 CUs x 8 blocks, grid-stride the rest
    }();
    int64_t g = (work_items + 255) / 256;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T>
int launch_shrink(Handle* h, const T* D, const T* A, const T* Y, T* E, T* Z, int64_t n, T inv_mu,
                  T thr, int nonnegE) {
    if (n <= 0) return TLSQ_OK;
    constexpr int VEC = 16 / sizeof(T);
    if (aligned16(D) && aligned16(A) && aligned16(Y) && aligned16(E) && aligned16(Z)) {
        hipLaunchKernelGGL((k_shrink<T, VEC>), dim3(grid_for(n / VEC + 1)), dim3(256), 0, h->stream,
                           D, A, Y, E, Z, n, inv_mu, thr, nonnegE);
    } else {
        hipLaunchKernelGGL((k_shrink<T, 1>), dim3(grid_for(n)), dim3(256), 0, h->stream, D, A, Y, E,
                           Z, n, inv_mu, thr, nonnegE);
    }
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// Set-up + first shrink in one pass (src/robustPCA.jl:181 and :188-192 at k = 1, where A is still zero):
//   Y = D / s,  E = soft_th(D - 0 + (1/mu) Y, thr),  Z = D - E + (1/mu) Y
// - the same operations in the same order as k_div_scalar followed by k_shrink (bit-identical; D - 0 is D), reading D once
// instead of D twice, A and Y: 4 panel passes instead of 7.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void k_first_shrink(const T* __restrict__ D, T* __restrict__ Y, T* __restrict__ E,
                                                      T* __restrict__ Z, int64_t n, T s, T inv_mu, T thr, int nonnegE) {
    using V = T __attribute__((ext_vector_type(VEC)));
    const int64_t nv = n / VEC;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = tid; i < nv; i += stride) {
        const V d = reinterpret_cast<const V*>(D)[i];
        V y, e, z;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            y[c] = d[c] / s;                          // Y ./= dual_norm                        :181
            const T t = inv_mu * y[c];
            T ee = soft_th(d[c] + t, thr);            // A = 0                                  :188
            if (nonnegE) ee = pos_part(ee);
            e[c] = ee;
            z[c] = (d[c] - ee) + t;                   //                                        :192
        }
        reinterpret_cast<V*>(Y)[i] = y;
        if (E) reinterpret_cast<V*>(E)[i] = e;   // (E == nullptr: the E-free loop below never reads E_1)
        reinterpret_cast<V*>(Z)[i] = z;
    }
    for (int64_t i = nv * VEC + tid; i < n; i += stride) {
        const T y = D[i] / s;
        const T t = inv_mu * y;
        T ee = soft_th(D[i] + t, thr);
        if (nonnegE) ee = pos_part(ee);
        Y[i] = y;
        if (E) E[i] = ee;
        Z[i] = (D[i] - ee) + t;
    }
}

template <typename T>
int launch_first_shrink(Handle* h, const T* D, T* Y, T* E, T* Z, int64_t n, T s, T inv_mu, T thr, int nonnegE) {
    if (n <= 0) return TLSQ_OK;
    constexpr int VEC = 16 / sizeof(T);
    if (aligned16(D) && aligned16(Y) && aligned16(E) && aligned16(Z))
        hipLaunchKernelGGL((k_first_shrink<T, VEC>), dim3(grid_for(n / VEC + 1)), dim3(256), 0, h->stream, D, Y, E, Z, n, s,
                           inv_mu, thr, nonnegE);
    else
        hipLaunchKernelGGL((k_first_shrink<T, 1>), dim3(grid_for(n)), dim3(256), 0, h->stream, D, Y, E, Z, n, s, inv_mu, thr,
                           nonnegE);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template <typename T>
int launch_update(Handle* h, const T* D, T* A, const T* E, T* Y, T* R, int64_t n, T mu,
                  int nonnegA) {
    if (n <= 0) return TLSQ_OK;
    constexpr int VEC = 16 / sizeof(T);
    if (aligned16(D) && aligned16(A) && aligned16(Y) && aligned16(E) && aligned16(R)) {
        hipLaunchKernelGGL((k_update<T, VEC>), dim3(grid_for(n / VEC + 1)), dim3(256), 0, h->stream,
                           D, A, E, Y, R, n, mu, nonnegA);
    } else {
        hipLaunchKernelGGL((k_update<T, 1>), dim3(grid_for(n)), dim3(256), 0, h->stream, D, A, E, Y,
                           R, n, mu, nonnegA);
    }
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template <typename T>
int launch_update_shrink(Handle* h, const T* D, T* A, const T* E, T* Y, T* R, T* En, T* Zn, int64_t n, T mu,
                         int nonnegA, T inv_mu_n, T thr_n, int nonnegE, double* sumsq, double* zero_slots) {
    if (n <= 0) return TLSQ_OK;
    constexpr int VEC = 16 / sizeof(T);
    if (aligned16(D) && aligned16(A) && aligned16(Y) && aligned16(E) && aligned16(R) && aligned16(En) && aligned16(Zn)) {   // (a null R is aligned)
        hipLaunchKernelGGL((k_update_shrink<T, VEC>), dim3(grid_for(n / VEC + 1)), dim3(256), 0, h->stream, D, A, E,
                           Y, R, En, Zn, n, mu, nonnegA, inv_mu_n, thr_n, nonnegE, sumsq, zero_slots);
    } else {
        hipLaunchKernelGGL((k_update_shrink<T, 1>), dim3(grid_for(n)), dim3(256), 0, h->stream, D, A, E, Y, R, En,
                           Zn, n, mu, nonnegA, inv_mu_n, thr_n, nonnegE, sumsq, zero_slots);
    }
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// false: shape not served by the fused kernel (odd M, unaligned panels, r > 32) or too small to gain from it: the
// caller uses the two-step path (skinny GEMM writes A, k_update_shrink streams it).  The column walk of the fused
// kernel sustains 4.5-5 TB/s where the linear sweep reaches 5.5-5.7, so dropping A's write and read only pays
// once the panels are large (measured: +16 % at 1e7 x 256, +9 % at 200000 x 512, +2 % at 20000 x 512).
template <typename T>
bool rebuild_update_shrink_ok(const T* D, const T* E, T* Y, T* R, T* En, T* Zn, int64_t M, int64_t N, int64_t r) {
    const bool force = dev_is(DEV_FUSED_REBUILD, '1');
    if (!force && M * N < ((int64_t)1 << 26)) return false;
    return (M % 2 == 0) && r <= 32 && aligned16(D) && aligned16(E) && aligned16(Y) && aligned16(R) && aligned16(En) &&
           aligned16(Zn);   // (D is not read when the caller passes an implicit Hankel source instead)
}

template <typename T>
int launch_rebuild_update_shrink(Handle* h, const T* D, const double* Tm, const double* Vs, const T* E, T* Y, T* R,
                                 T* En, T* Zn, int64_t M, int64_t N, int64_t r, T mu, int nonnegA, T inv_mu_n, T thr_n,
                                 int nonnegE, double* sumsq, double* zero_slots, const T* hankel_y, int64_t hankel_K,
                                 int64_t row0, int64_t row1, size_t pad_lds, HankelGeom hg) {
    if (M <= 0 || N <= 0) return TLSQ_OK;
    if (row1 <= 0) row1 = M;   // (default: the whole panel)
    if (row0 < 0 || row0 >= row1 || row1 > M || (row0 % 2) != 0 || (row1 % 2) != 0)
        return set_err(h, TLSQ_ERR_ARG, "rebuild_update_shrink: bad row range");
    // tall panels: two rows per thread, 64-column tiles.  Otherwise one row per thread and tiles narrow enough to
    // put ~16 waves on every CU (each tile re-reads its rows of T from L2, so not narrower than needed).
    const int64_t want_waves = 4096;
    const bool two = (M / 128) * ((N + 63) / 64) >= 2 * want_waves;
    int ct = 64;
    if (!two)
        while (ct > 8 && ((M + 63) / 64) * ((N + ct - 1) / ct) < want_waves) ct /= 2;
    const int env_rows = [] { const char* e = dev_get(DEV_RUS_ROWS); return e ? atoi(e) : 0; }();   // tuning knobs
    const int env_ct = [] { const char* e = dev_get(DEV_RUS_CT); return e ? atoi(e) : 0; }();
    bool two2 = two;
    if (env_rows == 1) two2 = false;
    if (env_rows == 2) two2 = true;
    if (env_ct >= 8 && env_ct <= 64) ct = env_ct;
    const int rows = two2 ? 2 : 1;
    const dim3 grid((unsigned)(((row1 - row0) / rows + 255) / 256), (unsigned)((N + ct - 1) / ct));
#define RUS_LAUNCH(RM, RW)                                                                                            \
    do {                                                                                                              \
        if (pad_lds > 32768) {   /* (unused dynamic LDS: caps the resident workgroups per CU, see solver.hip) */      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rebuild_update_shrink<T, RM, RW, true>),       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad_lds);                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rebuild_update_shrink<T, RM, RW, false>),      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad_lds);                      \
        }                                                                                                             \
        if (hankel_y)                                                                                                 \
            hipLaunchKernelGGL((k_rebuild_update_shrink<T, RM, RW, true>), grid, dim3(256), pad_lds, h->stream, hankel_y, Tm, \
                               Vs, E, Y, R, En, Zn, M, (int)N, (int)r, ct, mu, nonnegA, inv_mu_n, thr_n, nonnegE,      \
                               sumsq, zero_slots, hankel_K, row0, row1, hg);                                          \
        else                                                                                                          \
            hipLaunchKernelGGL((k_rebuild_update_shrink<T, RM, RW, false>), grid, dim3(256), pad_lds, h->stream, D, Tm, Vs, \
                               E, Y, R, En, Zn, M, (int)N, (int)r, ct, mu, nonnegA, inv_mu_n, thr_n, nonnegE, sumsq,   \
                               zero_slots, (int64_t)0, row0, row1, hg);                                               \
    } while (0)
    if (two2) {
        if (r <= 8) RUS_LAUNCH(8, 2);
        else if (r <= 16) RUS_LAUNCH(16, 2);
        else RUS_LAUNCH(32, 2);
    } else {
        if (r <= 8) RUS_LAUNCH(8, 1);
        else if (r <= 16) RUS_LAUNCH(16, 1);
        else RUS_LAUNCH(32, 1);
    }
#undef RUS_LAUNCH
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// E-free sweep (k_zsweep): A_k from its factors (A == nullptr, r <= 32) or from memory (A != nullptr; D must be a real panel
// and the row range the whole panel).  Rows [row0, row1), row1 <= 0: all.
template <typename T>
int launch_zsweep(Handle* h, const T* D, const double* Tm, const double* Vs, T* A, const T* Yin, T* Yout, T* Z, T* R,
                  int64_t M, int64_t N, int64_t r, T mu, T inv_mu, int nonnegA, T inv_mu_n, T thr_n, int nonnegE,
                  double* sumsq, double* zero_slots, const T* hankel_y, int64_t hankel_K, int64_t row0, int64_t row1,
                  int maxslot, HankelGeom hg, T* Zout) {
    if (M <= 0 || N <= 0) return TLSQ_OK;
    if (!Zout) Zout = Z;   // in place (the default): Z_{k+1} over Z_k
    if (row1 <= 0) row1 = M;
    if (!sumsq || maxslot > 7) maxslot = -1;
    if (A) {
        if (hankel_y || row0 != 0 || row1 != M) return set_err(h, TLSQ_ERR_ARG, "zsweep: explicit A needs the whole real panel");
        const int64_t n = M * N;
        constexpr int VEC = 16 / sizeof(T);
        const bool zip = Zout == Z;
        const T* Zr = zip ? nullptr : Z;
#define ZL_LAUNCH(VV, ZP, NG)                                                                                          \
    hipLaunchKernelGGL((k_zsweep_lin<T, VV, ZP>), dim3(grid_for(NG)), dim3(256), 0, h->stream, D, A, Yin, Yout, Zr, Zout, R, n, mu, \
                       inv_mu, nonnegA, inv_mu_n, thr_n, nonnegE, sumsq, zero_slots, maxslot)
        if (aligned16(D) && aligned16(A) && aligned16(Yin) && aligned16(Yout) && aligned16(Z) && aligned16(Zout) && aligned16(R)) {
            if (zip) ZL_LAUNCH(VEC, true, n / VEC + 1);
            else ZL_LAUNCH(VEC, false, n / VEC + 1);
        } else {
            if (zip) ZL_LAUNCH(1, true, n);
            else ZL_LAUNCH(1, false, n);
        }
#undef ZL_LAUNCH
        TLSQ_HIP(h, hipGetLastError());
        return TLSQ_OK;
    }
    if (r > 32 || N > 2147483647LL) return set_err(h, TLSQ_ERR_ARG, "zsweep: rank above 32 needs the explicit A");
    const bool pair_ok = (M % 2 == 0) && (row0 % 2 == 0) && (row1 % 2 == 0) && (hankel_y || aligned16(D)) && aligned16(Yin) &&
                         aligned16(Yout) && aligned16(Z) && aligned16(Zout) && aligned16(R);
    if (row0 < 0 || row0 >= row1 || row1 > M) return set_err(h, TLSQ_ERR_ARG, "zsweep: bad row range");
    // two rows per thread (16-byte accesses for fp64) whenever the panels allow it; 64-column tiles for tall panels, narrower
    // ones until ~16 waves sit on every CU (measured at 20000 x 512: 75 us with 2 rows x 16 columns, 80 us with 1 x 32)
    const int64_t want_waves = 4096;
    bool two = pair_ok;
    int ct = 64;
    while (ct > 8 && ((M + (two ? 127 : 63)) / (two ? 128 : 64)) * ((N + ct - 1) / ct) < want_waves) ct /= 2;
    const int env_rows = [] { const char* e = dev_get(DEV_RUS_ROWS); return e ? atoi(e) : 0; }();   // tuning knobs
    const int env_ct = [] { const char* e = dev_get(DEV_RUS_CT); return e ? atoi(e) : 0; }();
    if (env_rows == 1) two = false;
    if (env_rows == 2 && pair_ok) two = true;
    if (env_ct >= 8 && env_ct <= 64) ct = env_ct;
    const int rows = two ? 2 : 1;
    const dim3 grid((unsigned)(((row1 - row0) / rows + 255) / 256), (unsigned)((N + ct - 1) / ct));
    const bool zip = Zout == Z;
    const T* Zr = zip ? nullptr : Z;
#define ZS_LAUNCH2(RM, RW, HKF, ZP)                                                                                   \
    hipLaunchKernelGGL((k_zsweep<T, RM, RW, HKF, ZP>), grid, dim3(256), 0, h->stream, HKF ? hankel_y : D, Tm, Vs, Yin, Yout, Zr, \
                       Zout, R, M, (int)N, (int)r, ct, mu, inv_mu, nonnegA, inv_mu_n, thr_n, nonnegE, sumsq, zero_slots, \
                       HKF ? hankel_K : (int64_t)0, row0, row1, maxslot, hg)
#define ZS_LAUNCH(RM, RW)                                \
    do {                                                 \
        if (hankel_y) {                                  \
            if (zip) ZS_LAUNCH2(RM, RW, true, true);     \
            else ZS_LAUNCH2(RM, RW, true, false);        \
        } else {                                         \
            if (zip) ZS_LAUNCH2(RM, RW, false, true);    \
            else ZS_LAUNCH2(RM, RW, false, false);       \
        }                                                \
    } while (0)
    if (two) {
        if (r <= 8) ZS_LAUNCH(8, 2);
        else if (r <= 16) ZS_LAUNCH(16, 2);
        else ZS_LAUNCH(32, 2);
    } else {
        if (r <= 8) ZS_LAUNCH(8, 1);
        else if (r <= 16) ZS_LAUNCH(16, 1);
        else ZS_LAUNCH(32, 1);
    }
#undef ZS_LAUNCH
#undef ZS_LAUNCH2
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// E = soft_th(D - A_prev + inv_mu Y, thr): A_prev from factors (Aprev == nullptr, r <= 32; r = 0: zero) or from memory
template <typename T>
int launch_final_e(Handle* h, const T* D, const double* Tm, const double* Vs, const T* Aprev, const T* Y, T* E, int64_t M,
                   int64_t N, int64_t r, T inv_mu, T thr, int nonnegA, int nonnegE, const T* hankel_y, int64_t hankel_K,
                   HankelGeom hg) {
    if (M <= 0 || N <= 0) return TLSQ_OK;
    if (Aprev) {
        if (hankel_y) return set_err(h, TLSQ_ERR_ARG, "final_e: explicit A needs the real panel");
        hipLaunchKernelGGL((k_final_e_lin<T>), dim3(grid_for(M * N)), dim3(256), 0, h->stream, D, Aprev, Y, E, M * N, inv_mu,
                           thr, nonnegE);
        TLSQ_HIP(h, hipGetLastError());
        return TLSQ_OK;
    }
    if (r > 32 || N > 2147483647LL) return set_err(h, TLSQ_ERR_ARG, "final_e: rank above 32 needs the explicit A");
    const bool two = (M % 2 == 0) && (hankel_y || aligned16(D)) && aligned16(Y) && aligned16(E);
    const int ct = 32;
    const int rows = two ? 2 : 1;
    const dim3 grid((unsigned)((M / rows + 255) / 256 + 1), (unsigned)((N + ct - 1) / ct));
#define FE_LAUNCH(RM, RW)                                                                                             \
    do {                                                                                                              \
        if (hankel_y)                                                                                                 \
            hipLaunchKernelGGL((k_final_e<T, RM, RW, true>), grid, dim3(256), 0, h->stream, hankel_y, Tm, Vs, Y, E, M,  \
                               (int)N, (int)r, ct, inv_mu, thr, nonnegA, nonnegE, hankel_K, hg);                      \
        else                                                                                                          \
            hipLaunchKernelGGL((k_final_e<T, RM, RW, false>), grid, dim3(256), 0, h->stream, D, Tm, Vs, Y, E, M, (int)N, \
                               (int)r, ct, inv_mu, thr, nonnegA, nonnegE, (int64_t)0, hg);                            \
    } while (0)
    if (two) {
        if (r <= 8) FE_LAUNCH(8, 2);
        else if (r <= 16) FE_LAUNCH(16, 2);
        else FE_LAUNCH(32, 2);
    } else {
        if (r <= 8) FE_LAUNCH(8, 1);
        else if (r <= 16) FE_LAUNCH(16, 1);
        else FE_LAUNCH(32, 1);
    }
#undef FE_LAUNCH
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template <typename T>
int launch_residual_from_y(Handle* h, const T* Y1, const T* Y0, T* R, int64_t n, T inv_mu) {
    if (n <= 0) return TLSQ_OK;
    hipLaunchKernelGGL((k_residual_from_y<T>), dim3(grid_for(n)), dim3(256), 0, h->stream, Y1, Y0, R, n, inv_mu);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template <typename T>
int launch_e_from_residual(Handle* h, const T* D, const T* A, const T* Y1, const T* Y0, T* E, int64_t n, T inv_mu) {
    if (n <= 0) return TLSQ_OK;
    hipLaunchKernelGGL((k_e_from_residual<T>), dim3(grid_for(n)), dim3(256), 0, h->stream, D, A, Y1, Y0, E, n, inv_mu);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template <typename T>
int launch_e_from_z(Handle* h, const T* D, const T* Z, const T* Y, T* E, int64_t n, T inv_mu) {
    if (n <= 0) return TLSQ_OK;
    hipLaunchKernelGGL((k_e_from_z<T>), dim3(grid_for(n)), dim3(256), 0, h->stream, D, Z, Y, E, n, inv_mu);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template <typename T>
int launch_z_from_y(Handle* h, const T* A, const T* Y1, T* Z, int64_t n, T inv_mu) {
    if (n <= 0) return TLSQ_OK;
    hipLaunchKernelGGL((k_z_from_y<T>), dim3(grid_for(n)), dim3(256), 0, h->stream, A, Y1, Z, n, inv_mu);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

int launch_publish_slots(Handle* h, const double* slots, double seq) {
    hipLaunchKernelGGL(k_publish_slots, dim3(1), dim3(128), 0, h->stream, slots, h->mailbox_dev, seq);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template <typename T>
int launch_residual(Handle* h, const T* D, const T* A, const T* E, T* R, int64_t n) {
    if (n <= 0) return TLSQ_OK;
    hipLaunchKernelGGL((k_residual<T>), dim3(grid_for(n)), dim3(256), 0, h->stream, D, A, E, R, n);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template <typename T>
int launch_residual_hankel(Handle* h, const T* y, int64_t K, const T* A, const T* E, T* R, int64_t M, int64_t N, HankelGeom hg) {
    if (M <= 0 || N <= 0) return TLSQ_OK;
    if (N > 65535) return set_err(h, TLSQ_ERR_UNSUPPORTED, "residual: more than 65535 Hankel columns");
    const int64_t gx = std::min<int64_t>((M + 1023) / 1024, 4096);
    hipLaunchKernelGGL((k_residual_hankel<T>), dim3((unsigned)gx, (unsigned)N), dim3(256), 0, h->stream, y, K, A, E, R, M, N, hg);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template <typename T>
int launch_div_scalar(Handle* h, const T* D, T* Y, int64_t n, T s) {
    if (n <= 0) return TLSQ_OK;
    constexpr int VEC = 16 / sizeof(T);
    if (aligned16(D) && aligned16(Y)) {
        hipLaunchKernelGGL((k_div_scalar<T, VEC>), dim3(grid_for(n / VEC + 1)), dim3(256), 0,
                           h->stream, D, Y, n, s);
    } else {
        hipLaunchKernelGGL((k_div_scalar<T, 1>), dim3(grid_for(n)), dim3(256), 0, h->stream, D, Y, n,
                           s);
    }
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template <typename T>
int launch_clamp_nonneg(Handle* h, T* A, int64_t n) {
    if (n <= 0) return TLSQ_OK;
    hipLaunchKernelGGL((k_clamp_nonneg<T>), dim3(grid_for(n)), dim3(256), 0, h->stream, A, n);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template <typename T>
int launch_maxabs(Handle* h, const T* x, int64_t n, double* host_out) {
    void* slot;
    TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &slot));
    unsigned long long* d = reinterpret_cast<unsigned long long*>(slot);
    TLSQ_HIP(h, hipMemsetAsync(d, 0, 8, h->stream));
    if (n > 0) {
        hipLaunchKernelGGL((k_maxabs<T>), dim3(grid_for(n)), dim3(256), 0, h->stream, x, n, d);
        TLSQ_HIP(h, hipGetLastError());
    }
    TLSQ_HIP(h, hipMemcpyAsync(h->pinned, d, 8, hipMemcpyDeviceToHost, h->stream));
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    memcpy(host_out, h->pinned, 8);
    return TLSQ_OK;
}

// The same in two halves: the pass is queued on the handle's SECOND stream (it is memory-bound and overlaps the Gram matrix of the
// set-up, which is not) and its result is read when the caller needs it - no host round trip between the two kernels of the
// set-up.  x must not be written by anything queued on the main stream in between.
int second_stream(Handle* h);   // (runtime.hip)
template <typename T>
int launch_maxabs_begin(Handle* h, const T* x, int64_t n) {
    TLSQ_TRY(second_stream(h));
    void* slot;
    TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &slot));
    unsigned long long* d = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(slot) + 3600);   // (3072..3600: the slicer's statistics and flags, sliced.hip)
    // (the second stream may only start once everything queued so far on the main one is done: the workspace slot, x)
    TLSQ_HIP(h, hipEventRecord(h->ev_b[8], h->stream));
    TLSQ_HIP(h, hipStreamWaitEvent(h->stream_b, h->ev_b[8], 0));
    TLSQ_HIP(h, hipMemsetAsync(d, 0, 8, h->stream_b));
    if (n > 0) {
        hipLaunchKernelGGL((k_maxabs<T>), dim3(grid_for(n)), dim3(256), 0, h->stream_b, x, n, d);
        TLSQ_HIP(h, hipGetLastError());
    }
    TLSQ_HIP(h, hipMemcpyAsync(reinterpret_cast<char*>(h->pinned) + 2048, d, 8, hipMemcpyDeviceToHost, h->stream_b));
    return TLSQ_OK;
}
int launch_maxabs_end(Handle* h, double* host_out) {
    TLSQ_HIP(h, hipStreamSynchronize(h->stream_b));
    memcpy(host_out, reinterpret_cast<char*>(h->pinned) + 2048, 8);
    return TLSQ_OK;
}
template int launch_maxabs_begin<double>(Handle*, const double*, int64_t);
template int launch_maxabs_begin<float>(Handle*, const float*, int64_t);

template <typename T>
int launch_transpose(Handle* h, const T* src, int64_t lds, int64_t M, int64_t N, T* dst, int64_t ldd) {
    if (M <= 0 || N <= 0) return TLSQ_OK;
    dim3 grid((unsigned)((M + 31) / 32), (unsigned)((N + 31) / 32));
    if (grid.y > 65535) return set_err(h, TLSQ_ERR_UNSUPPORTED, "transpose: N too large");
    hipLaunchKernelGGL((k_transpose<T>), grid, dim3(256), 0, h->stream, src, lds, M, N, dst, ldd);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

template <typename T>
int launch_diff(Handle* h, const T* a, const T* b, T* out, int64_t n) {
    if (n <= 0) return TLSQ_OK;
    hipLaunchKernelGGL((k_diff<T>), dim3(grid_for(n)), dim3(256), 0, h->stream, a, b, out, n);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
template int launch_diff<double>(Handle*, const double*, const double*, double*, int64_t);
template int launch_diff<float>(Handle*, const float*, const float*, float*, int64_t);

template <typename TS, typename TD>
int launch_convert(Handle* h, const TS* src, TD* dst, int64_t n) {
    if (n <= 0) return TLSQ_OK;
    hipLaunchKernelGGL((k_convert<TS, TD>), dim3(grid_for(n)), dim3(256), 0, h->stream, src, dst, n);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
template int launch_convert<double, float>(Handle*, const double*, float*, int64_t);
template int launch_convert<float, double>(Handle*, const float*, double*, int64_t);
template int launch_convert<double, double>(Handle*, const double*, double*, int64_t);
template int launch_convert<float, float>(Handle*, const float*, float*, int64_t);

#define INST(T)                                                                                   \
    template int launch_shrink<T>(Handle*, const T*, const T*, const T*, T*, T*, int64_t, T, T, int); \
    template int launch_first_shrink<T>(Handle*, const T*, T*, T*, T*, int64_t, T, T, T, int);    \
    template int launch_update<T>(Handle*, const T*, T*, const T*, T*, T*, int64_t, T, int);      \
    template int launch_update_shrink<T>(Handle*, const T*, T*, const T*, T*, T*, T*, T*, int64_t, T, int, T, T, \
                                         int, double*, double*);                                                     \
    template bool rebuild_update_shrink_ok<T>(const T*, const T*, T*, T*, T*, T*, int64_t, int64_t, int64_t); \
    template int launch_rebuild_update_shrink<T>(Handle*, const T*, const double*, const double*, const T*, T*, T*, \
                                                 T*, T*, int64_t, int64_t, int64_t, T, int, T, T, int, double*, double*, \
                                                 const T*, int64_t, int64_t, int64_t, size_t, HankelGeom);            \
    template int launch_zsweep<T>(Handle*, const T*, const double*, const double*, T*, const T*, T*, T*, T*, int64_t, \
                                  int64_t, int64_t, T, T, int, T, T, int, double*, double*, const T*, int64_t, int64_t, \
                                  int64_t, int, HankelGeom, T*);                                                           \
    template int launch_final_e<T>(Handle*, const T*, const double*, const double*, const T*, const T*, T*, int64_t, \
                                   int64_t, int64_t, T, T, int, int, const T*, int64_t, HankelGeom);                  \
    template int launch_residual_from_y<T>(Handle*, const T*, const T*, T*, int64_t, T);                              \
    template int launch_z_from_y<T>(Handle*, const T*, const T*, T*, int64_t, T);                                     \
    template int launch_e_from_residual<T>(Handle*, const T*, const T*, const T*, const T*, T*, int64_t, T);          \
    template int launch_e_from_z<T>(Handle*, const T*, const T*, const T*, T*, int64_t, T);                           \
    template int launch_residual<T>(Handle*, const T*, const T*, const T*, T*, int64_t);          \
    template int launch_residual_hankel<T>(Handle*, const T*, int64_t, const T*, const T*, T*, int64_t, int64_t, HankelGeom); \
    template int launch_div_scalar<T>(Handle*, const T*, T*, int64_t, T);                         \
    template int launch_clamp_nonneg<T>(Handle*, T*, int64_t);                                    \
    template int launch_maxabs<T>(Handle*, const T*, int64_t, double*);                           \
    template int launch_transpose<T>(Handle*, const T*, int64_t, int64_t, int64_t, T*, int64_t);
INST(double)
INST(float)
#undef INST

}  // namespace tlsq
