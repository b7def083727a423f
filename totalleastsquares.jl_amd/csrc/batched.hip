// Batched rpca for many tiny problems: one workgroup runs the whole inexact-ALM loop of
// one M x N problem (N <= 16) with every panel resident in LDS.
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
#include "common.hpp"

#pragma clang fp contract(off)

namespace tlsq {

namespace {

constexpr int BT = 256;          // threads per problem
constexpr int BW = BT / 64;      // waves
constexpr int BN = 16;           // largest N

template <int CTRL>
__device__ __forceinline__ double b_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double b_lane(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double b_wsum(double v) {   // wave all-reduce, fixed order
    v += b_dpp<0xB1>(v);
    v += b_dpp<0x4E>(v);
    v += b_dpp<0x141>(v);
    v += b_dpp<0x140>(v);
    return (b_lane(v, 0) + b_lane(v, 16)) + (b_lane(v, 32) + b_lane(v, 48));
}
__device__ __forceinline__ double b_pos(double a) { return (a > 0.0 || a != a) ? a : 0.0; }
__device__ __forceinline__ double b_neg(double b) { return (b < 0.0 || b != b) ? b : 0.0; }
__device__ __forceinline__ double b_soft(double x, double e) { return b_pos(x - e) + b_neg(x + e); }   // :1

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

struct Small {            // per-problem LDS bookkeeping
    double* V;            // N x N (ld BN)
    double* nrm;          // squared column norms
    double* sig;          // singular values, sorted descending
    int* ord;             // column index of the i-th largest
    double* red;          // BW partials
    unsigned int* cnt;    // rotation counter
};

// One-sided Jacobi SVD of the M x N panel P (ld M) in place: on return the columns of P are mutually orthogonal
// (P = U diag(sigma) in some column order), nrm[j] = ||P[:,j]||^2 and, when V != nullptr, V (N x N, ld BN) holds
// the accumulated rotations (P_in * V = P_out).  All BT threads must call.
__device__ void jacobi_svd(double* P, int M, int N, const Small& s, bool want_v) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (want_v)
        for (int e = tid; e < N * BN; e += BT) s.V[e] = ((e % BN) == (e / BN)) ? 1.0 : 0.0;
    const int nslot = (N + 1) & ~1, npair = nslot / 2;
    const double tol = 2.220446049250313e-16 * sqrt((double)M);   // LAPACK dgesvj's criterion
    for (int sweep = 0; sweep < 40; ++sweep) {
        __syncthreads();
        for (int c = w; c < N; c += BW) {   // refresh the cached norms once per sweep
            const double* x = P + (size_t)c * M;
            double a = 0.0;
            for (int r = lane; r < M; r += 64) a += x[r] * x[r];
            a = b_wsum(a);
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
                double* x = P + (size_t)p * M;
                double* y = P + (size_t)q * M;
                const double a = s.nrm[p], bb = s.nrm[q];
                double c = 0.0;
                for (int r = lane; r < M; r += 64) c += x[r] * y[r];
                c = b_wsum(c);
                if (c * c > tol * tol * a * bb && a > 0.0 && bb > 0.0) {
                    // t = 2 c sgn(d) / (|d| + sqrt(d^2 + 4 c^2)), d = bb - a: the smaller root, one sqrt + one division
                    const double d = bb - a;
                    const double t = (d >= 0.0 ? 2.0 : -2.0) * c / (fabs(d) + sqrt(d * d + 4.0 * c * c));
                    const double cs = 1.0 / sqrt(1.0 + t * t);
                    const double sn = cs * t;
                    for (int r = lane; r < M; r += 64) {
                        const double u = x[r], v = y[r];
                        x[r] = cs * u - sn * v;
                        y[r] = sn * u + cs * v;
                    }
                    if (want_v && lane < N) {
                        const double u = s.V[lane + p * BN], v = s.V[lane + q * BN];
                        s.V[lane + p * BN] = cs * u - sn * v;
                        s.V[lane + q * BN] = sn * u + cs * v;
                    }
                    if (lane == 0) {
                        const double na = a - t * c, nb = bb + t * c;
                        s.nrm[p] = na > 0.0 ? na : 0.0;
                        s.nrm[q] = nb > 0.0 ? nb : 0.0;
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
        const double* x = P + (size_t)c * M;
        double a = 0.0;
        for (int r = lane; r < M; r += 64) a += x[r] * x[r];
        a = b_wsum(a);
        if (lane == 0) s.nrm[c] = a;
    }
    __syncthreads();
    if (tid < N) {   // rank of column tid (stable: ties keep the column order)
        const double me = s.nrm[tid];
        int rank = 0;
        for (int j = 0; j < N; ++j) rank += (s.nrm[j] > me || (s.nrm[j] == me && j < tid)) ? 1 : 0;
        s.ord[rank] = tid;
        s.sig[rank] = sqrt(me);
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
template <bool IN_LDS>
__global__ __launch_bounds__(BT) void k_rpca_small(const double* __restrict__ Dg, BatchedArgs a,
                                                   double* __restrict__ Ag, double* __restrict__ Eg,
                                                   double* __restrict__ Sg, double* __restrict__ Vtg,
                                                   int64_t* __restrict__ svg, int32_t* __restrict__ itg,
                                                   int32_t* __restrict__ stg, double* __restrict__ costg,
                                                   double* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) double bsm[];
    const int M = a.M, N = a.N, MN = M * N;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t b = blockIdx.x;
    Small s;
    s.V = bsm;                          // BN*BN
    s.nrm = s.V + BN * BN;              // BN
    s.sig = s.nrm + BN;                 // BN
    s.red = s.sig + BN;                 // 8
    s.ord = reinterpret_cast<int*>(s.red + 8);           // BN ints = 8 doubles
    s.cnt = reinterpret_cast<unsigned int*>(s.red + 16);   // 1 (+pad) -> 2 doubles
    double* g = s.red + 18;             // BN  rebuild weights
    double* big = IN_LDS ? (bsm + BN * BN + 3 * BN + 24) : (scratch + (size_t)b * 5 * MN);
    double *D = big, *A = D + MN, *E = A + MN, *Y = E + MN, *Z = Y + MN;
    const double* Din = Dg + (size_t)b * MN;

    // ---- setup, src/robustPCA.jl:171-184 ----
    double mx = 0.0;
    for (int e = tid; e < MN; e += BT) {
        const double d = Din[e];
        D[e] = d;
        Z[e] = d;
        A[e] = 0.0;
        E[e] = 0.0;
        const double ad = fabs(d);
        mx = ad > mx ? ad : mx;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if (lane == 0) s.red[w] = mx;
    __syncthreads();
    double maxabs = 0.0;
    for (int k = 0; k < BW; ++k) maxabs = s.red[k] > maxabs ? s.red[k] : maxabs;   // norm(Y, Inf)  :178
    jacobi_svd(Z, M, N, s, false);
    const double norm2 = s.sig[0];                                  // opnorm(Y)  :177
    const double lam = a.lambda;
    const double norminf = maxabs / lam;
    const double dual_norm = norm2 > norminf ? norm2 : norminf;     // :179
    const double d_norm = norm2;                                    // :180
    for (int e = tid; e < MN; e += BT) Y[e] = D[e] / dual_norm;     // :181
    double mu = 1.25 / norm2;                                       // :182
    const double mubar = mu * 1.0e7;                                // :183
    int64_t sv = 10;
    int svp = 10;                                                   // :184
    double cost = 0.0;
    int k = 0, converged = 0;
    __syncthreads();
    for (k = 1; k <= a.iters; ++k) {                                // :186
        const double inv_mu = 1.0 / mu, thr = lam / mu;
        for (int e = tid; e < MN; e += BT) {                        // :188-192
            const double t = inv_mu * Y[e];
            double ee = b_soft((D[e] - A[e]) + t, thr);
            if (a.nonnegE) ee = b_pos(ee);
            E[e] = ee;
            Z[e] = (D[e] - ee) + t;
        }
        jacobi_svd(Z, M, N, s, true);                               // :194  (Z <- U*S, V accumulated)
        svp = 0;                                                    // :198
        for (int i = 0; i < N; ++i) svp += (s.sig[i] >= inv_mu) ? 1 : 0;
        {
            int64_t t = svp > 1 ? svp : 1;                          // :199-204
            if (a.maxrank > 0 && t > a.maxrank) t = a.maxrank;
            sv = t;
        }
        if (tid < svp) {
            const double sg = s.sig[tid];
            g[tid] = a.nukeA ? (sg - inv_mu) / sg : 1.0;            // :205-213
        }
        __syncthreads();
        // A = sum_i g_i (U S)[:, o_i] V[:, o_i]'
        for (int e = tid; e < MN; e += BT) {
            const int r = e % M, c = e / M;
            double acc = 0.0;
            for (int i = 0; i < svp; ++i) {
                const int o = s.ord[i];
                acc += (g[i] * Z[r + (size_t)o * M]) * s.V[c + o * BN];
            }
            A[e] = acc;
        }
        __syncthreads();
        for (int e = tid; e < MN; e += BT) {                        // :217-222
            double av = A[e];
            if (a.nonnegA) {
                av = b_pos(av);
                A[e] = av;
            }
            const double z = (D[e] - av) - E[e];
            Z[e] = z;
            Y[e] = Y[e] + mu * z;
        }
        mu = fmin(mu * a.rho, mubar);                               // :223
        // keep the decomposition of this iteration's Z (the returned `s`, :194,:238): the values-only Jacobi below
        // leaves V alone but overwrites sig / ord
        double skeep = 0.0;
        int okeep = 0;
        if (tid < N) {
            skeep = s.sig[tid];
            okeep = s.ord[tid];
        }
        __syncthreads();
        jacobi_svd(Z, M, N, s, false);                              // :225 opnorm(Z)
        cost = s.sig[0] / d_norm;
        __syncthreads();
        if (tid < N) {
            s.sig[tid] = skeep;
            s.ord[tid] = okeep;
        }
        __syncthreads();
        if (cost < a.tol) {                                         // :228
            converged = 1;
            break;
        }
    }
    if (k > a.iters) k = a.iters;
    // ---- results ----
    double* Aout = Ag + (size_t)b * MN;
    double* Eout = Eg + (size_t)b * MN;
    for (int e = tid; e < MN; e += BT) {
        Aout[e] = A[e];
        Eout[e] = E[e];
    }
    if (Sg && tid < N) Sg[(size_t)b * N + tid] = s.sig[tid];
    if (Vtg && tid < N * N) {   // Vt[i, c] = V[c, ord[i]]
        const int i = tid % N, c = tid / N;
        Vtg[(size_t)b * N * N + i + (size_t)c * N] = s.V[c + s.ord[i] * BN];
    }
    if (tid == 0) {
        if (svg) svg[b] = sv;
        if (itg) itg[b] = k;
        if (stg) stg[b] = converged ? 0 : 1;
        if (costg) costg[b] = cost;
    }
}

size_t rpca_small_lds_bytes(int64_t M, int64_t N, bool* in_lds) {
    const size_t small = (size_t)(BN * BN + 3 * BN + 24) * 8;
    const size_t big = (size_t)5 * M * N * 8;
    *in_lds = small + big <= 150 * 1024;
    return *in_lds ? small + big : small;
}

int launch_rpca_small(Handle* h, const double* D, int64_t M, int64_t N, int64_t batch, double lambda, double tol,
                      double rho, int64_t iters, int64_t maxrank, bool nonnegA, bool nonnegE, bool nukeA, double* A,
                      double* E, double* S, double* Vt, int64_t* sv, int32_t* it, int32_t* st, double* cost,
                      double* scratch) {
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
    const size_t lds = rpca_small_lds_bytes(M, N, &in_lds);
    if (in_lds) {
        TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rpca_small<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_rpca_small<true>, dim3((unsigned)batch), dim3(BT), lds, h->stream, D, a, A, E, S, Vt, sv, it,
                           st, cost, scratch);
    } else {
        hipLaunchKernelGGL(k_rpca_small<false>, dim3((unsigned)batch), dim3(BT), lds, h->stream, D, a, A, E, S, Vt, sv,
                           it, st, cost, scratch);
    }
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

}  // namespace tlsq
