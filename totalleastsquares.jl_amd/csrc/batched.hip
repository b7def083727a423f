// Batched rpca for many tiny problems: one workgroup runs the whole inexact-ALM loop of
// one M x N problem (N <= 32) with every panel resident in LDS.
//
// The reference's real workloads are thousands of independent rtls / rpca calls on 50x4 ... 500x6 matrices
// (/root/reference/test/runtests.jl:205-235, total_vs_robust_demo.jl:7-18): far too small for the streaming
// kernels of the large path (a launch costs more than the arithmetic), but embarrassingly parallel across
// problems.  Per problem this kernel follows src/robustPCA.jl:156-239 statement by statement (same expression
// order in the sweeps as sweeps.hip, contraction off); the two LAPACK calls of an iteration (`svd!(Z)` :194,
// `opnorm(Z)` :225) are one-sided (Hestenes) Jacobi SVDs carried out directly on the panel in LDS — no Gram
// matrix, so singular values keep full relative accuracy however small they are.
//
// Layout: D, A, E, Y, Z (M x N, column-major, ld = M) + V (N x N) in LDS when 5*M*N*8 B fits, otherwise the five
// panels live in a per-problem global scratch area (L2 resident) and only V and the small vectors stay in LDS.
#include <limits>
#include "common.hpp"

#pragma clang fp contract(off)

namespace tlsq {

namespace {

constexpr int BT = 256;          // threads per problem
constexpr int BW = BT / 64;      // waves
// NB (template parameter of the kernel): leading dimension of V and length of the small vectors - 16 for N <= 16 (more
// problems per CU: the reference's own uses are 4 ... 6 columns), 32 for 16 < N <= 32

template <int CTRL>
__device__ __forceinline__ double b_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float b_dpp(float v) {
    const int x = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ double b_lane(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ float b_lane(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
template <typename T>
__device__ __forceinline__ T b_wsum(T v) {   // wave all-reduce, fixed order
    v += b_dpp<0xB1>(v);
    v += b_dpp<0x4E>(v);
    v += b_dpp<0x141>(v);
    v += b_dpp<0x140>(v);
    return (b_lane(v, 0) + b_lane(v, 16)) + (b_lane(v, 32) + b_lane(v, 48));
}
template <typename T>
__device__ __forceinline__ T b_pos(T a) { return (a > (T)0 || a != a) ? a : (T)0; }
template <typename T>
__device__ __forceinline__ T b_neg(T b) { return (b < (T)0 || b != b) ? b : (T)0; }
template <typename T>
__device__ __forceinline__ T b_soft(T x, T e) { return b_pos<T>(x - e) + b_neg<T>(x + e); }   // :1

__device__ __forceinline__ void b_pair(int n, int r, int idx, int& p, int& q) {   // round-robin tournament
    const int m = n - 1;
    if (idx == 0) {
        p = r % m;
        q = m;
    } else {
        p = (r + idx) % m;
        q = (r - idx + m) % m;
    }
}

template <typename T>
struct Small {            // per-problem LDS bookkeeping
    T* V;                 // N x N (ld BN)
    T* nrm;               // squared column norms
    T* sig;               // singular values, sorted descending
    int* ord;             // column index of the i-th largest
    T* red;               // BW partials
    unsigned int* cnt;    // rotation counter
};

// One-sided Jacobi SVD of the M x N panel P (ld M) in place: on return the columns of P are mutually orthogonal
// (P = U diag(sigma) in some column order), nrm[j] = ||P[:,j]||^2 and, when V != nullptr, V (N x N, ld BN) holds
// the accumulated rotations (P_in * V = P_out).  All BT threads must call.
template <typename T, int NB>
__device__ void jacobi_svd(T* P, int M, int N, const Small<T>& s, bool want_v) {
    constexpr int BN = NB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (want_v)
        for (int e = tid; e < N * BN; e += BT) s.V[e] = ((e % BN) == (e / BN)) ? (T)1 : (T)0;
    const int nslot = (N + 1) & ~1, npair = nslot / 2;
    const T tol = std::numeric_limits<T>::epsilon() * (T)sqrt((double)M);   // LAPACK gesvj's criterion
    for (int sweep = 0; sweep < 40; ++sweep) {
        __syncthreads();
        for (int c = w; c < N; c += BW) {   // refresh the cached norms once per sweep
            const T* x = P + (size_t)c * M;
            T a = (T)0;
            for (int r = lane; r < M; r += 64) a += x[r] * x[r];
            a = b_wsum<T>(a);
            if (lane == 0) s.nrm[c] = a;
        }
        if (tid == 0) *s.cnt = 0;
        __syncthreads();
        unsigned int my = 0;
        for (int ir = 0; ir < nslot - 1; ++ir) {
            for (int ip = w; ip < npair; ip += BW) {
                int p, q;
                b_pair(nslot, ir, ip, p, q);
                if (p > q) {
                    const int t = p;
                    p = q;
                    q = t;
                }
                if (q >= N) continue;
                T* x = P + (size_t)p * M;
                T* y = P + (size_t)q * M;
                const T a = s.nrm[p], bb = s.nrm[q];
                T c = (T)0;
                for (int r = lane; r < M; r += 64) c += x[r] * y[r];
                c = b_wsum<T>(c);
                if (c * c > tol * tol * a * bb && a > (T)0 && bb > (T)0) {
                    // t = 2 c sgn(d) / (|d| + sqrt(d^2 + 4 c^2)), d = bb - a: the smaller root, one sqrt + one division
                    const T d = bb - a;
                    const T t = (d >= (T)0 ? (T)2 : (T)-2) * c / ((T)fabs(d) + (T)sqrt(d * d + (T)4 * c * c));
                    const T cs = (T)1 / (T)sqrt((T)1 + t * t);
                    const T sn = cs * t;
                    for (int r = lane; r < M; r += 64) {
                        const T u = x[r], v = y[r];
                        x[r] = cs * u - sn * v;
                        y[r] = sn * u + cs * v;
                    }
                    if (want_v && lane < N) {
                        const T u = s.V[lane + p * BN], v = s.V[lane + q * BN];
                        s.V[lane + p * BN] = cs * u - sn * v;
                        s.V[lane + q * BN] = sn * u + cs * v;
                    }
                    if (lane == 0) {
                        const T na = a - t * c, nb = bb + t * c;
                        s.nrm[p] = na > (T)0 ? na : (T)0;
                        s.nrm[q] = nb > (T)0 ? nb : (T)0;
                    }
                    ++my;
                }
            }
            __syncthreads();
        }
        if (lane == 0 && my) atomicAdd(s.cnt, my);
        __syncthreads();
        if (*s.cnt == 0) break;
    }
    __syncthreads();
    // exact norms of the final columns, then the descending order
    for (int c = w; c < N; c += BW) {
        const T* x = P + (size_t)c * M;
        T a = (T)0;
        for (int r = lane; r < M; r += 64) a += x[r] * x[r];
        a = b_wsum<T>(a);
        if (lane == 0) s.nrm[c] = a;
    }
    __syncthreads();
    if (tid < N) {   // rank of column tid (stable: ties keep the column order)
        const T me = s.nrm[tid];
        int rank = 0;
        for (int j = 0; j < N; ++j) rank += (s.nrm[j] > me || (s.nrm[j] == me && j < tid)) ? 1 : 0;
        s.ord[rank] = tid;
        s.sig[rank] = (T)sqrt(me);
    }
    __syncthreads();
}

}  // namespace

struct BatchedArgs {
    int M, N;
    double lambda, tol, rho;
    int iters, maxrank;
    int nonnegA, nonnegE, nukeA;
};

// grid = batch.  Dg/Ag/Eg: batch contiguous M x N problems.  Sg (N per problem), Vtg (N x N per problem, ld N,
// rows sorted by singular value), svg, itg, stg (0 = converged, 1 = iteration limit) may each be nullptr.
template <typename T, bool IN_LDS, int NB>
__global__ __launch_bounds__(BT) void k_rpca_small(const T* __restrict__ Dg, BatchedArgs a,
                                                   T* __restrict__ Ag, T* __restrict__ Eg,
                                                   T* __restrict__ Sg, T* __restrict__ Vtg,
                                                   int64_t* __restrict__ svg, int32_t* __restrict__ itg,
                                                   int32_t* __restrict__ stg, T* __restrict__ costg,
                                                   T* __restrict__ scratch) {
    constexpr int BN = NB;
    extern __shared__ __attribute__((aligned(16))) unsigned char bsm_raw[];
    T* bsm = reinterpret_cast<T*>(bsm_raw);
    const int M = a.M, N = a.N, MN = M * N;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t b = blockIdx.x;
    Small<T> s;
    s.V = bsm;                          // BN*BN
    s.nrm = s.V + BN * BN;              // BN
    s.sig = s.nrm + BN;                 // BN
    s.red = s.sig + BN;                 // 8
    T* g = s.red + 8;                   // BN  rebuild weights
    s.ord = reinterpret_cast<int*>(g + BN);                  // BN ints
    s.cnt = reinterpret_cast<unsigned int*>(s.ord + BN);     // 1 (+ 3 of padding: the panels start 16-byte aligned)
    T* big = IN_LDS ? reinterpret_cast<T*>(s.ord + BN + 4) : (scratch + (size_t)b * 5 * MN);
    T *D = big, *A = D + MN, *E = A + MN, *Y = E + MN, *Z = Y + MN;
    const T* Din = Dg + (size_t)b * MN;

    // ---- setup, src/robustPCA.jl:171-184 ----
    T mx = (T)0;
    for (int e = tid; e < MN; e += BT) {
        const T d = Din[e];
        D[e] = d;
        Z[e] = d;
        A[e] = (T)0;
        E[e] = (T)0;
        T ad = (T)fabs(d);
        if (!(ad <= std::numeric_limits<T>::max())) ad = std::numeric_limits<T>::infinity();   // NaN counts as Inf: "not finite"
        mx = ad > mx ? ad : mx;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if (lane == 0) s.red[w] = mx;
    __syncthreads();
    T maxabs = (T)0;
    for (int k = 0; k < BW; ++k) maxabs = s.red[k] > maxabs ? s.red[k] : maxabs;   // norm(Y, Inf)  :178
    if (!(maxabs <= std::numeric_limits<T>::max())) {
        // Infs / NaNs in this problem: the reference's loop would stop here with LAPACK's ArgumentError (chkfinite behind
        // opnorm, :177).  The other problems of the batch are not affected: status 2, NaN results, no iterations.
        const T qnan = std::numeric_limits<T>::quiet_NaN();
        T* Ao = Ag + (size_t)b * MN;
        T* Eo = Eg + (size_t)b * MN;
        for (int e = tid; e < MN; e += BT) {
            Ao[e] = qnan;
            Eo[e] = qnan;
        }
        if (Sg && tid < N) Sg[(size_t)b * N + tid] = qnan;
        if (Vtg)
            for (int e = tid; e < N * N; e += BT) Vtg[(size_t)b * N * N + e] = qnan;
        if (tid == 0) {
            if (svg) svg[b] = 0;
            if (itg) itg[b] = 0;
            if (stg) stg[b] = 2;
            if (costg) costg[b] = qnan;
        }
        return;
    }
    jacobi_svd<T, NB>(Z, M, N, s, false);
    const T norm2 = s.sig[0];                                  // opnorm(Y)  :177
    const T lam = (T)a.lambda;
    const T norminf = maxabs / lam;
    const T dual_norm = norm2 > norminf ? norm2 : norminf;     // :179
    const T d_norm = norm2;                                    // :180
    for (int e = tid; e < MN; e += BT) Y[e] = D[e] / dual_norm;     // :181
    T mu = (T)1.25 / norm2;                                       // :182
    const T mubar = mu * (T)1.0e7;                                // :183
    int64_t sv = 10;
    int svp = 10;                                                   // :184
    T cost = (T)0;
    int k = 0, converged = 0;
    __syncthreads();
    for (k = 1; k <= a.iters; ++k) {                                // :186
        const T inv_mu = (T)1 / mu, thr = lam / mu;
        for (int e = tid; e < MN; e += BT) {                        // :188-192
            const T t = inv_mu * Y[e];
            T ee = b_soft<T>((D[e] - A[e]) + t, thr);
            if (a.nonnegE) ee = b_pos<T>(ee);
            E[e] = ee;
            Z[e] = (D[e] - ee) + t;
        }
        jacobi_svd<T, NB>(Z, M, N, s, true);                               // :194  (Z <- U*S, V accumulated)
        svp = 0;                                                    // :198
        for (int i = 0; i < N; ++i) svp += (s.sig[i] >= inv_mu) ? 1 : 0;
        {
            int64_t t = svp > 1 ? svp : 1;                          // :199-204
            if (a.maxrank > 0 && t > a.maxrank) t = a.maxrank;
            sv = t;
        }
        if (tid < svp) {
            const T sg = s.sig[tid];
            g[tid] = a.nukeA ? (sg - inv_mu) / sg : (T)1;            // :205-213
        }
        __syncthreads();
        // A = sum_i g_i (U S)[:, o_i] V[:, o_i]'
        for (int e = tid; e < MN; e += BT) {
            const int r = e % M, c = e / M;
            T acc = (T)0;
            for (int i = 0; i < svp; ++i) {
                const int o = s.ord[i];
                acc += (g[i] * Z[r + (size_t)o * M]) * s.V[c + o * BN];
            }
            A[e] = acc;
        }
        __syncthreads();
        for (int e = tid; e < MN; e += BT) {                        // :217-222
            T av = A[e];
            if (a.nonnegA) {
                av = b_pos<T>(av);
                A[e] = av;
            }
            const T z = (D[e] - av) - E[e];
            Z[e] = z;
            Y[e] = Y[e] + mu * z;
        }
        {
            const T mr = mu * (T)a.rho;
            mu = mr < mubar ? mr : mubar;
        }                               // :223
        // keep the decomposition of this iteration's Z (the returned `s`, :194,:238): the values-only Jacobi below
        // leaves V alone but overwrites sig / ord
        T skeep = (T)0;
        int okeep = 0;
        if (tid < N) {
            skeep = s.sig[tid];
            okeep = s.ord[tid];
        }
        __syncthreads();
        jacobi_svd<T, NB>(Z, M, N, s, false);                              // :225 opnorm(Z)
        cost = s.sig[0] / d_norm;
        __syncthreads();
        if (tid < N) {
            s.sig[tid] = skeep;
            s.ord[tid] = okeep;
        }
        __syncthreads();
        if (cost < (T)a.tol) {                                         // :228
            converged = 1;
            break;
        }
    }
    if (k > a.iters) k = a.iters;
    // ---- results ----
    T* Aout = Ag + (size_t)b * MN;
    T* Eout = Eg + (size_t)b * MN;
    for (int e = tid; e < MN; e += BT) {
        Aout[e] = A[e];
        Eout[e] = E[e];
    }
    if (Sg && tid < N) Sg[(size_t)b * N + tid] = s.sig[tid];
    if (Vtg)   // Vt[i, c] = V[c, ord[i]]
        for (int e = tid; e < N * N; e += BT) {
            const int i = e % N, c = e / N;
            Vtg[(size_t)b * N * N + i + (size_t)c * N] = s.V[c + s.ord[i] * BN];
        }
    if (tid == 0) {
        if (svg) svg[b] = sv;
        if (itg) itg[b] = k;
        if (stg) stg[b] = converged ? 0 : 1;
        if (costg) costg[b] = cost;
    }
}

size_t rpca_small_lds_bytes(int64_t M, int64_t N, bool* in_lds, size_t esz) {
    const size_t BN = N <= 16 ? 16 : 32;
    const size_t small = (size_t)(BN * BN + 3 * BN + 8) * esz + (size_t)(BN + 4) * 4;   // (a multiple of 16 bytes)
    const size_t big = (size_t)5 * M * N * esz;
    *in_lds = small + big <= 150 * 1024;
    return *in_lds ? small + big : small;
}

template <typename T>
int launch_rpca_small(Handle* h, const T* D, int64_t M, int64_t N, int64_t batch, double lambda, double tol,
                      double rho, int64_t iters, int64_t maxrank, bool nonnegA, bool nonnegE, bool nukeA, T* A,
                      T* E, T* S, T* Vt, int64_t* sv, int32_t* it, int32_t* st, T* cost, T* scratch) {
    if (batch <= 0) return TLSQ_OK;
    BatchedArgs a;
    a.M = (int)M;
    a.N = (int)N;
    a.lambda = lambda;
    a.tol = tol;
    a.rho = rho;
    a.iters = (int)std::min<int64_t>(iters, 1 << 30);
    a.maxrank = (int)std::min<int64_t>(maxrank, 1 << 30);
    a.nonnegA = nonnegA;
    a.nonnegE = nonnegE;
    a.nukeA = nukeA;
    bool in_lds;
    const size_t lds = rpca_small_lds_bytes(M, N, &in_lds, sizeof(T));
    if (N > 32) return set_err(h, TLSQ_ERR_UNSUPPORTED, "batched rpca: N = %lld > 32", (long long)N);
#define TLSQ_BATCHED_GO(INL, NBV)                                                                                         \
    do {                                                                                                                  \
        if (INL)                                                                                                          \
            TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rpca_small<T, INL, NBV>),                     \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
        hipLaunchKernelGGL((k_rpca_small<T, INL, NBV>), dim3((unsigned)batch), dim3(BT), lds, h->stream, D, a, A, E, S, Vt, \
                           sv, it, st, cost, scratch);                                                                    \
    } while (0)
    if (in_lds) {
        if (N <= 16) TLSQ_BATCHED_GO(true, 16);
        else TLSQ_BATCHED_GO(true, 32);
    } else {
        if (N <= 16) TLSQ_BATCHED_GO(false, 16);
        else TLSQ_BATCHED_GO(false, 32);
    }
#undef TLSQ_BATCHED_GO
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
template int launch_rpca_small<double>(Handle*, const double*, int64_t, int64_t, int64_t, double, double, double, int64_t,
                                       int64_t, bool, bool, bool, double*, double*, double*, double*, int64_t*, int32_t*,
                                       int32_t*, double*, double*);
template int launch_rpca_small<float>(Handle*, const float*, int64_t, int64_t, int64_t, double, double, double, int64_t,
                                      int64_t, bool, bool, bool, float*, float*, float*, float*, int64_t*, int32_t*,
                                      int32_t*, float*, float*);

}  // namespace tlsq
