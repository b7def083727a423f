// lambda_max of a symmetric PSD N x N matrix by Lanczos — the default `opnorm` of rpca
// (/root/reference/src/robustPCA.jl:177,225: opnorm(Z) = sigma_max(Z) = sqrt(lambda_max(Z'Z)), which the
// reference gets from a full LAPACK gesdd('N') every iteration).
//
// One persistent 1024-thread workgroup runs a chunk of Lanczos steps without host interaction: the
// matrix (N*N*8 bytes, 2 MB at N=512) is streamed from L2 once per step, 16 waves each own N/16 columns
// (G symmetric: column j == row j, so the dot products read contiguous memory), the Krylov vectors live
// in LDS.  alpha/beta are written to global memory; the host evaluates the largest Ritz value of the
// tridiagonal and a residual bound and decides whether another chunk is needed.
#include "common.hpp"

namespace tlsq {

double now_ms();   // runtime.hip

constexpr int LZ_WGS = 64;       // workgroups per step-kernel up to N = 1024 ...
constexpr int LZ_WGS_MAX = 1024; // ... N / 8 beyond (lz_wgs): a 4096 x 4096 matrix is 134 MB, 64 workgroups stream it at 0.6 TB/s
constexpr int LZ_THREADS = 256;  // 4 waves

// wave all-reduce without the LDS crossbar (see jacobi.hip): DPP inside each row of 16 lanes, then the four row totals
template <int CTRL>
__device__ __forceinline__ double lz_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lz_lane(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wsum(double v) {
    v += lz_dpp<0xB1>(v);
    v += lz_dpp<0x4E>(v);
    v += lz_dpp<0x141>(v);
    v += lz_dpp<0x140>(v);
    return (lz_lane(v, 0) + lz_lane(v, 16)) + (lz_lane(v, 32) + lz_lane(v, 48));
}

__device__ __forceinline__ double block_sum4(double v, double* red) {
    v = wsum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

static inline int lz_wgs(int64_t N) {
    if (N <= 1024) return LZ_WGS;
    return (int)std::min<int64_t>(LZ_WGS_MAX, (N + 7) / 8);
}

// rows [r0, r1) of u = G q for this workgroup (G symmetric: row r == column r, contiguous; q: N doubles in LDS), one row
// per wave at a time, 16-byte loads when the rows are aligned.  Returns this wave's lane-0 share of sum_r u_r * (SQ ? u_r
// : q_r); u_r is stored by lane 0.
typedef double lz_d2 __attribute__((ext_vector_type(2)));
template <bool SQ>
__device__ __forceinline__ double lz_rows(const double* __restrict__ G, int64_t ldG, int N, const double* q,
                                          double* __restrict__ unew, int r0, int r1) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool vec = ((ldG & 1) == 0) && ((reinterpret_cast<uintptr_t>(G) & 15) == 0);
    const int n2 = vec ? (N & ~1) : 0;
    double pacc = 0.0;
    for (int r = r0 + w; r < r1; r += LZ_THREADS / 64) {
        const double* __restrict__ col = G + (int64_t)r * ldG;
        double a0 = 0.0, a1 = 0.0;
        int c = 2 * lane;
        // eight 16-byte loads of the row in flight per lane (one at a time left a 4096-column row latency-bound: 32 dependent
        // round trips to HBM per row, 57-70 us per step at N = 4096 = 134 MB at 2 TB/s)
        for (; c + 7 * 128 < n2; c += 8 * 128) {
            lz_d2 g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) g[u] = *reinterpret_cast<const lz_d2*>(col + c + 128 * u);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const lz_d2 x = *reinterpret_cast<const lz_d2*>(q + c + 128 * u);
                a0 += g[u][0] * x[0];
                a1 += g[u][1] * x[1];
            }
        }
        for (; c < n2; c += 128) {
            const lz_d2 g = *reinterpret_cast<const lz_d2*>(col + c);
            const lz_d2 x = *reinterpret_cast<const lz_d2*>(q + c);
            a0 += g[0] * x[0];
            a1 += g[1] * x[1];
        }
        for (int c = n2 + lane; c < N; c += 64) a0 += col[c] * q[c];
        const double acc = wsum(a0 + a1);
        if (lane == 0) {
            unew[r] = acc;
            pacc += acc * (SQ ? acc : q[r]);
        }
    }
    return pacc;
}

// Workgroup 0 copies the header and the completed (alpha, beta) pairs to the host-visible mailbox
// ([0] flag, [8..16) header, [16..) alpha[cap], beta[cap]) and publishes them with the sequence number.
__device__ __forceinline__ void lz_publish(const double* st, const double* ab, int cap, double* mailbox, double seq) {
    volatile double* mb = mailbox;
    __syncthreads();   // st[1], st[2], ab[j-1] were written by thread 0 just before
    const int m = (int)st[2];
    for (int i = threadIdx.x; i < m; i += LZ_THREADS) {
        mb[16 + i] = ab[i];
        mb[16 + cap + i] = ab[cap + i];
    }
    if (threadIdx.x < 8) mb[8 + threadIdx.x] = st[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) mb[0] = seq;
}

// One Lanczos step per launch (the kernel boundary is the grid-wide dependency; ~5 us per step).
// Launch j:
//   phase A (every workgroup, redundantly and identically): finish step j-1 from the partial dot products
//     of launch j-1:  alpha = q.u,  w = u - alpha q - beta_prev q_prev,  beta = ||w||,  q_new = w/beta
//   phase B: this workgroup's rows of u_new = G q_new (G symmetric: row r == column r, contiguous) and its
//     partial of q_new . u_new.
// Vectors rotate through 3 buffers, u and the partials through 2, so no launch overwrites what its own
// (slower) workgroups still read.  Workgroup 0 records alpha_{j-1}, beta_{j-1}.
// layout of `st` (doubles): [0..8) header {-, breakdown, pairs done}; alpha[cap], beta[cap]; vec[3][N]; u[2][N];
// part[2][LZ_WGS_MAX]
__global__ __launch_bounds__(LZ_THREADS) void k_lanczos_step(const double* __restrict__ G, int64_t ldG,
                                                             int N, double* __restrict__ st,
                                                             double* __restrict__ ab, int maxsteps, int j,
                                                             double* mailbox, double seq,
                                                             const double* __restrict__ v0, int nwg) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* q = sm;           // N  (q_new)
    double* red = sm + N;     // 4
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double* vec = st + 8 + 2 * (size_t)maxsteps;   // (alpha, beta) sit right behind the header: one read-back
    double* ubuf = vec + 3 * (size_t)N;
    double* part = ubuf + 2 * (size_t)N;
    double* vnew = vec + (size_t)(j % 3) * N;
    if (j == 0) {
        if (blockIdx.x == 0 && tid == 0) {   // the first launch of a run resets the header (no separate memset)
            st[1] = 0.0;
            st[2] = 0.0;
        }
    } else if (st[1] != 0.0) {
        if (mailbox && blockIdx.x == 0) lz_publish(st, ab, maxsteps, mailbox, seq);
        return;  // breakdown flagged by an earlier launch
    }
    if (j == 0) {
        // deterministic pseudo-random start vector (integer hash), normalised
        double nrm = 0.0;
        for (int i = tid; i < N; i += LZ_THREADS) {
            double v;
            if (v0) {   // caller's start vector (e.g. the dominant column of a high power of G)
                v = v0[i];
            } else {
                unsigned int x = (unsigned int)i * 2654435761u + 12345u;
                x ^= x >> 16;
                x *= 2246822519u;
                x ^= x >> 13;
                x *= 3266489917u;
                x ^= x >> 16;
                v = ((double)(x & 0xFFFFFF) + 0.5) / 16777216.0 - 0.5;
            }
            q[i] = v;
            nrm += v * v;
        }
        nrm = block_sum4(nrm, red);
        const double inv = 1.0 / sqrt(nrm);
        for (int i = tid; i < N; i += LZ_THREADS) q[i] *= inv;
    } else {
        const double* qc = vec + (size_t)((j + 2) % 3) * N;   // q_{j-1}  (buffer (j-1)%3)
        const double* qp = vec + (size_t)((j + 1) % 3) * N;   // q_{j-2}  (buffer (j-2)%3)
        const double* u = ubuf + (size_t)((j + 1) % 2) * N;   // u of launch j-1
        const double* pp = part + (size_t)((j + 1) % 2) * LZ_WGS_MAX;
        double alpha = 0.0;
        if (nwg <= 64) {
            for (int k = 0; k < nwg; ++k) alpha += pp[k];
        } else {   // (N > 1024: hundreds of partial sums - every thread adding all of them was half of a 4096-column step)
            double a = 0.0;
            for (int k = tid; k < nwg; k += LZ_THREADS) a += pp[k];
            alpha = block_sum4(a, red);
        }
        const double beta_prev = (j >= 2) ? ab[maxsteps + (j - 2)] : 0.0;
        double nn = 0.0;
        for (int i = tid; i < N; i += LZ_THREADS) {
            double v = u[i] - alpha * qc[i];
            if (j >= 2) v -= beta_prev * qp[i];
            q[i] = v;
            nn += v * v;
        }
        const double beta = sqrt(block_sum4(nn, red));
        if (blockIdx.x == 0 && tid == 0) {
            ab[j - 1] = alpha;
            ab[maxsteps + (j - 1)] = beta;
            st[2] = (double)j;  // completed (alpha,beta) pairs
        }
        if (!(beta > 1e-290)) {
            if (blockIdx.x == 0 && tid == 0) st[1] = 1.0;
            if (mailbox && blockIdx.x == 0) lz_publish(st, ab, maxsteps, mailbox, seq);
            return;
        }
        // the last launch of a chunk hands (alpha, beta) to the host as soon as workgroup 0 has them - the host
        // polls the flag instead of paying a copy command and a stream synchronisation
        if (mailbox && blockIdx.x == 0) lz_publish(st, ab, maxsteps, mailbox, seq);
        const double inv = 1.0 / beta;
        for (int i = tid; i < N; i += LZ_THREADS) q[i] *= inv;
    }
    __syncthreads();
    if (blockIdx.x == 0)
        for (int i = tid; i < N; i += LZ_THREADS) vnew[i] = q[i];
    // phase B: rows [r0, r1) of u_new = G q_new
    double* unew = ubuf + (size_t)(j % 2) * N;
    const int rows_per = (N + nwg - 1) / nwg;
    const int r0 = blockIdx.x * rows_per;
    const int r1 = (r0 + rows_per < N) ? r0 + rows_per : N;
    const double pacc = lz_rows<false>(G, ldG, N, q, unew, r0, r1);
    // partial of q_new . u_new over this workgroup's rows (lane 0 of each wave holds a piece)
    __syncthreads();
    if (lane == 0) red[w] = pacc;
    __syncthreads();
    if (tid == 0) part[(size_t)(j % 2) * LZ_WGS_MAX + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- several Lanczos steps in ONE launch (round 6) --------------------------------------------------------------
// A step of k_lanczos_step is a kernel boundary: ~10 us for 2 MB of matrix that a single CU could multiply in 1.  Up to
// N = 1024 the matrix fits the LDS of 64 workgroups (N / 64 rows each: 32 KB at N = 512, 128 KB at 1024), and what a step
// exchanges is one N-vector: every workgroup publishes its rows of u = G q as data-tagged granules (8 bytes = {tag, half a
// double}, write-through agent-scope stores: the data is the flag, no fence, no counter), every thread polls the granules
// of ITS OWN elements of u (agent-scope loads, never through L1), and alpha, w, beta, q_new are computed by every
// workgroup redundantly and identically - the same threads add the same numbers in the same order as in k_lanczos_step's
// phase A.  Two granule buffers in turn (a workgroup publishes step j + 2 only after it has read every workgroup's j + 1,
// which they publish only after reading j).  tag = (run salt, step + 1): nothing left by an earlier run or step matches.
// Nothing depends on where the workgroups run; they only have to be resident together (64 x 256 threads on 256 CUs), and
// every poll is bounded: a wave that waits longer than ~20 ms flags st[3], the workgroup leaves at its next barrier, the
// host repeats the run with one launch per step and never uses this kernel on the handle again.
typedef __attribute__((address_space(1))) unsigned long long lz_gu64;

__device__ __forceinline__ unsigned long long lz_now() { return wall_clock64(); }   // constant 100 MHz

__global__ __launch_bounds__(LZ_THREADS) void k_lanczos_multi(const double* __restrict__ G, int64_t ldG, int N,
                                                              double* __restrict__ st, double* __restrict__ ab, int cap,
                                                              int j0, int nsteps, double* mailbox, double seq,
                                                              const double* __restrict__ v0, unsigned long long* xch,
                                                              unsigned int salt, int rows_per, int drop_wg) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* q = sm;                       // N
    double* red = sm + N;                 // 8: [0..4) reductions, [4] bail flag
    double* Gs = sm + N + 8;              // rows_per x N
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r0 = blockIdx.x * rows_per;
    if (r0 >= N) return;
    if ((int)blockIdx.x == drop_wg) return;   // (LZ_MULTI=drop, tests: a workgroup that never publishes - the others must give up and say so)
    const int r1 = (r0 + rows_per < N) ? r0 + rows_per : N;
    double* vec = st + 8 + 2 * (size_t)cap;   // [0, N): the current Lanczos vector, [N, 2N): the one before (between launches)
    for (int r = r0 + w; r < r1; r += LZ_THREADS / 64) {
        const double* __restrict__ col = G + (int64_t)r * ldG;
        for (int c = lane; c < N; c += 64) Gs[(size_t)(r - r0) * N + c] = col[c];
    }
    constexpr int EPT = 4;                // elements per thread: N <= 1024
    double qc[EPT], qp[EPT];
    double beta_prev = 0.0;
    if (tid == 0) red[4] = 0.0;
    if (j0 == 0) {
        double nrm = 0.0;
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = tid + k * LZ_THREADS;
            double v = 0.0;
            if (i < N) {
                if (v0) {
                    v = v0[i];
                } else {   // (the start vector of k_lanczos_step)
                    unsigned int x = (unsigned int)i * 2654435761u + 12345u;
                    x ^= x >> 16;
                    x *= 2246822519u;
                    x ^= x >> 13;
                    x *= 3266489917u;
                    x ^= x >> 16;
                    v = ((double)(x & 0xFFFFFF) + 0.5) / 16777216.0 - 0.5;
                }
            }
            qc[k] = v;
            qp[k] = 0.0;
            nrm += v * v;
        }
        nrm = block_sum4(nrm, red);
        const double inv = 1.0 / sqrt(nrm);
#pragma unroll
        for (int k = 0; k < EPT; ++k) qc[k] *= inv;
        if (blockIdx.x == 0 && tid == 0) {
            st[1] = 0.0;
            st[2] = 0.0;
            st[3] = 0.0;
        }
    } else {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = tid + k * LZ_THREADS;
            qc[k] = i < N ? vec[i] : 0.0;
            qp[k] = i < N ? vec[(size_t)N + i] : 0.0;
        }
        beta_prev = ab[cap + (j0 - 1)];
    }
    bool broke = false;
    for (int sidx = 0; sidx < nsteps; ++sidx) {
        const int j = j0 + sidx;
        const unsigned long long epoch = ((unsigned long long)(salt & 0xFFFFFu) << 12) | (unsigned long long)((j + 1) & 0xFFF);
        lz_gu64* xp = (lz_gu64*)(xch + (size_t)(j & 1) * 2 * (size_t)N);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = tid + k * LZ_THREADS;
            if (i < N) q[i] = qc[k];
        }
        __syncthreads();
        // this workgroup's rows of u = G q, published as they are finished
        for (int r = r0 + w; r < r1; r += LZ_THREADS / 64) {
            const double* row = Gs + (size_t)(r - r0) * N;
            double a0 = 0.0, a1 = 0.0;
            int c = 2 * lane;
            for (; c + 1 < N; c += 128) {
                const lz_d2 g = *reinterpret_cast<const lz_d2*>(row + c);
                const lz_d2 x = *reinterpret_cast<const lz_d2*>(q + c);
                a0 += g[0] * x[0];
                a1 += g[1] * x[1];
            }
            if ((N & 1) && lane == 0) a0 += row[N - 1] * q[N - 1];
            const double acc = wsum(a0 + a1);
            if (lane < 2) {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(acc);
                const unsigned long long half = lane ? (bits >> 32) : (bits & 0xFFFFFFFFull);
                __hip_atomic_store(xp + 2 * (size_t)r + lane, (epoch << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // every thread collects its own elements of u
        double u[EPT];
        {
            bool have[EPT];
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                have[k] = tid + k * LZ_THREADS >= N;
                u[k] = 0.0;
            }
            const unsigned long long t_start = lz_now();
            bool timed_out = false;
            for (unsigned int spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < EPT; ++k) {
                    if (!have[k]) {
                        const int i = tid + k * LZ_THREADS;
                        const unsigned long long x0 = __hip_atomic_load(xp + 2 * (size_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long x1 = __hip_atomic_load(xp + 2 * (size_t)i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((x0 >> 32) == epoch && (x1 >> 32) == epoch) {
                            u[k] = __longlong_as_double((long long)(((x1 & 0xFFFFFFFFull) << 32) | (x0 & 0xFFFFFFFFull)));
                            have[k] = true;
                        } else {
                            ok = false;
                        }
                    }
                }
                if (__all(ok)) break;
                if ((spins & 63u) == 63u && lz_now() - t_start > 2000000ull) {   // 20 ms
                    timed_out = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (timed_out && lane == 0) red[4] = 1.0;
        }
        // alpha = q . u,  w = u - alpha q - beta_prev q_prev,  beta = ||w||,  q_new = w / beta
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < EPT; ++k) a += qc[k] * u[k];
        const double alpha = block_sum4(a, red);
        if (red[4] != 0.0) {   // (uniform: read behind block_sum4's barriers)
            if (tid == 0) st[3] = 1.0;
            broke = true;
            break;
        }
        double nn = 0.0, wv[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            double v = u[k] - alpha * qc[k];
            if (j >= 1) v -= beta_prev * qp[k];
            wv[k] = v;
            nn += v * v;
        }
        const double beta = sqrt(block_sum4(nn, red));
        if (blockIdx.x == 0 && tid == 0) {
            ab[j] = alpha;
            ab[cap + j] = beta;
            st[2] = (double)(j + 1);
        }
        if (!(beta > 1e-290)) {
            if (blockIdx.x == 0 && tid == 0) st[1] = 1.0;
            broke = true;
            break;
        }
        const double inv = 1.0 / beta;
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            qp[k] = qc[k];
            qc[k] = wv[k] * inv;
        }
        beta_prev = beta;
    }
    if (blockIdx.x != 0) return;
    if (!broke) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = tid + k * LZ_THREADS;
            if (i < N) {
                vec[i] = qc[k];
                vec[(size_t)N + i] = qp[k];
            }
        }
    }
    if (mailbox) lz_publish(st, ab, cap, mailbox, seq);
}

// ---- host: largest eigenvalue of the symmetric tridiagonal (alpha[0..m), beta[0..m-1)) + residual bound
static int sturm_count_below(const double* a, const double* b, int m, double x) {
    // number of eigenvalues < x
    int cnt = 0;
    double d = 1.0;
    for (int i = 0; i < m; ++i) {
        const double off = (i == 0) ? 0.0 : b[i - 1] * b[i - 1];
        d = (a[i] - x) - (i == 0 ? 0.0 : off / d);
        if (d == 0.0) d = -1e-300;
        if (d < 0.0) ++cnt;
    }
    return cnt;
}

static double tridiag_lmax(const double* a, const double* b, int m, double* last_comp) {
    double lo = a[0], hi = a[0];
    for (int i = 0; i < m; ++i) {
        const double r = (i > 0 ? fabs(b[i - 1]) : 0.0) + (i + 1 < m ? fabs(b[i]) : 0.0);
        lo = std::min(lo, a[i] - r);
        hi = std::max(hi, a[i] + r);
    }
    // bisection for the largest eigenvalue: count_below(x) == m  <=>  x > lambda_max
    for (int it = 0; it < 200; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (!(mid > lo && mid < hi)) break;
        if (sturm_count_below(a, b, m, mid) >= m) hi = mid; else lo = mid;
    }
    const double theta = 0.5 * (lo + hi);
    // eigenvector by the three-term recurrence from the top (stable for the extreme eigenvalue when run
    // towards the small components), normalised; we only need |s_m| / ||s||
    if (last_comp) {
        std::vector<double> s((size_t)m);
        // inverse iteration with a slightly shifted theta (2 steps) using Thomas algorithm w/o pivoting
        std::vector<double> x((size_t)m, 1.0), c((size_t)m), dd((size_t)m);
        const double shift = theta + 1e-14 * std::max(fabs(theta), 1e-300) + 1e-300;
        for (int rep = 0; rep < 3; ++rep) {
            // solve (T - shift I) y = x
            double piv = a[0] - shift;
            if (piv == 0.0) piv = 1e-300;
            dd[0] = piv;
            for (int i = 1; i < m; ++i) {
                c[i - 1] = b[i - 1] / dd[i - 1];
                dd[i] = (a[i] - shift) - c[i - 1] * b[i - 1];
                if (dd[i] == 0.0) dd[i] = 1e-300;
            }
            std::vector<double> y = x;
            for (int i = 1; i < m; ++i) y[i] -= c[i - 1] * y[i - 1];
            y[m - 1] /= dd[m - 1];
            for (int i = m - 2; i >= 0; --i) y[i] = (y[i] - b[i] * y[i + 1]) / dd[i];
            double nn = 0.0;
            for (double v : y) nn += v * v;
            nn = sqrt(nn);
            if (!(nn > 0.0) || !std::isfinite(nn)) break;
            for (int i = 0; i < m; ++i) x[i] = y[i] / nn;
        }
        *last_comp = fabs(x[m - 1]);
    }
    return theta;
}

// ---- explicit matrix: begin / finish ---------------------------------------------------------------------------
// lanczos_begin queues the first chunk of steps and the read-back of (alpha, beta) and records an event;
// lanczos_finish waits for that event only (work queued behind it keeps running), evaluates the Ritz value and the
// residual bound, and - rarely - runs further chunks synchronously.  lanczos_lmax_f64 is begin + finish.
static int lz_launch_chunk(Handle* h, LanczosRun& r) {
    const int n = std::min(r.chunk, r.max_steps + 1 - r.launched);
    r.use_mail = r.mail_ok && n >= 2;
    if (r.use_mail) r.seq = (h->mail_seq += 1.0);
    if (r.multi) {   // (launch j of the one-step kernel completes pair j - 1: the same pairs per chunk here)
        const int pairs_before = std::max(r.launched - 1, 0);
        r.launched += n;
        const int nsteps = (r.launched - 1) - pairs_before;
        if (nsteps > 0) {
            hipLaunchKernelGGL(k_lanczos_multi, dim3(LZ_WGS), dim3(LZ_THREADS), r.lds_multi, h->stream, r.G, r.ldG, (int)r.N, r.st, r.ab,
                               r.cap, pairs_before, nsteps, r.use_mail ? h->mailbox_dev : (double*)nullptr, r.seq, r.v0,
                               (unsigned long long*)r.xch, r.salt, r.rows_per, dev_is(DEV_LZ_MULTI, 'd') ? 1 : -1);
            TLSQ_HIP(h, hipGetLastError());
        }
        if (!r.use_mail)
            TLSQ_HIP(h, hipMemcpyAsync(h->pinned, r.st, 64 + (size_t)2 * r.cap * 8, hipMemcpyDeviceToHost, h->stream));
        return TLSQ_OK;
    }
    for (int k = 0; k < n; ++k, ++r.launched) {
        const bool last = r.use_mail && k == n - 1;
        hipLaunchKernelGGL(k_lanczos_step, dim3(lz_wgs(r.N)), dim3(LZ_THREADS), r.lds, h->stream, r.G, r.ldG, (int)r.N, r.st,
                           r.ab, r.cap, r.launched, last ? h->mailbox_dev : (double*)nullptr, r.seq, r.v0, lz_wgs(r.N));
    }
    TLSQ_HIP(h, hipGetLastError());
    if (!r.use_mail)
        TLSQ_HIP(h, hipMemcpyAsync(h->pinned, r.st, 64 + (size_t)2 * r.cap * 8, hipMemcpyDeviceToHost, h->stream));
    return TLSQ_OK;
}

int lanczos_begin(Handle* h, LanczosRun& r, const double* G, int64_t N, int64_t ldG, double rel_tol, int max_steps,
                  double accept_below, double stop_above, const double* v0) {
    r = LanczosRun();
    r.v0 = v0;
    r.G = G;
    r.N = N;
    r.ldG = ldG;
    r.rel_tol = rel_tol;
    r.accept_below = accept_below;
    r.stop_above = stop_above;
    if (N <= 0) {
        r.trivial = true;
        return TLSQ_OK;
    }
    if (max_steps > (int)N) max_steps = (int)N;
    if (max_steps < 1) max_steps = 1;
    r.max_steps = max_steps;
    r.cap = max_steps + 2;
    void* stv;
    const size_t st_doubles = 8 + 5 * (size_t)N + 2 * LZ_WGS_MAX + 2 * (size_t)r.cap + 16;
    TLSQ_TRY(ws_get(h, WS_AUX4, st_doubles * 8, &stv));
    r.st = (double*)stv;
    r.ab = r.st + 8;
    r.rows_per = (int)((N + LZ_WGS - 1) / LZ_WGS);
    r.lds_multi = ((size_t)N + 8 + (size_t)r.rows_per * N) * 8;
    r.multi = N >= 64 && N <= 1024 && r.lds_multi <= 150 * 1024 && !h->lz_multi_off && !dev_is(DEV_LZ_MULTI, '0');
    if (r.multi) {
        // the granule buffers: a slot nothing else writes, so that whatever they hold carries the tag of an earlier run; cleared
        // when the slot is new and when the run counter wraps (2^20 runs)
        void* xv;
        TLSQ_TRY(ws_get(h, WS_LZX, 4 * (size_t)1024 * 8, &xv));
        r.xch = (double*)xv;
        if (h->lz_xch_clean != xv || h->lz_salt >= 0xFFFFEu) {
            TLSQ_HIP(h, hipMemsetAsync(xv, 0, 4 * (size_t)1024 * 8, h->stream));
            h->lz_xch_clean = xv;
            h->lz_salt = 0;
        }
        r.salt = ++h->lz_salt;
        if (r.lds_multi > 48 * 1024)
            TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_lanczos_multi),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)r.lds_multi));
    }
    r.lds = (size_t)(N + 8) * 8;
    if (r.lds > 150 * 1024 || (size_t)(2 * r.cap) * 8 + 64 > h->pinned_bytes) {
        r.unsupported = true;
        return TLSQ_OK;
    }
    if (r.lds > 48 * 1024)
        TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_lanczos_step),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)r.lds));
    // launch j completes pair j-1.  Yes/no questions (accept_below / stop_above) are usually settled by the
    // first few Ritz values: start with 4 pairs and double
    r.chunk = stop_above > 0.0 ? 5 : (accept_below > 0.0 ? 11 : 16);
    // (a caller that knows the question will not be settled by the first Ritz values - the cost of an ALM iteration whose power
    //  bound has just failed to say "not converged" - asks for a longer first chunk: one launch and one round trip less)
    if (r.multi && h->lz_first_chunk > r.chunk) r.chunk = h->lz_first_chunk;
    h->lz_first_chunk = 0;
    const bool no_mailbox = dev_is(DEV_NO_MAILBOX, '1');
    r.mail_ok = h->mailbox && !no_mailbox && (size_t)(16 + 2 * r.cap) * 8 <= h->mailbox_bytes;
    TLSQ_TRY(lz_launch_chunk(h, r));
    if (!r.use_mail) {
        TLSQ_HIP(h, hipEventRecord(h->ev[33], h->stream));
        r.event_pending = true;
    }
    return TLSQ_OK;
}

// returns TLSQ_OK and *lmax, or 1 if the requested accuracy was not reached in max_steps (caller falls back)
int lanczos_finish(Handle* h, LanczosRun& r, double* lmax, int* steps_used) {
    if (r.trivial) {
        *lmax = 0.0;
        return TLSQ_OK;
    }
    if (r.unsupported) return 1;
    std::vector<double> hab((size_t)2 * r.cap);
    double theta = 0.0;
    for (;;) {
        double hs[8];
        bool from_mail = false;
        if (r.use_mail) {
            volatile double* mb = h->mailbox;
            const double t_poll = now_ms();
            while (mb[0] != r.seq && now_ms() - t_poll < 2000.0) {
            }
            from_mail = mb[0] == r.seq;
            if (from_mail) {
                for (int i = 0; i < 8; ++i) hs[i] = mb[8 + i];
                const int mm = (int)hs[2];
                for (int i = 0; i < mm && i < r.cap; ++i) {
                    hab[(size_t)i] = mb[16 + i];
                    hab[(size_t)r.cap + i] = mb[16 + r.cap + i];
                }
            } else {   // time-out (never seen in practice): classic read-back, and no mailbox on this handle any more
                h->mailbox_bytes = 0;
                r.mail_ok = false;
                TLSQ_HIP(h, hipMemcpyAsync(h->pinned, r.st, 64 + (size_t)2 * r.cap * 8, hipMemcpyDeviceToHost, h->stream));
            }
        }
        if (!from_mail) {
            if (r.event_pending) {
                TLSQ_HIP(h, hipEventSynchronize(h->ev[33]));
                r.event_pending = false;
            } else {
                TLSQ_HIP(h, hipStreamSynchronize(h->stream));
            }
            memcpy(hs, h->pinned, 64);
            memcpy(hab.data(), (char*)h->pinned + 64, (size_t)2 * r.cap * 8);
        }
        if (r.multi && hs[3] != 0.0) {
            // a workgroup of k_lanczos_multi gave up waiting for the others (never seen; they were not resident together):
            // the same run again with one launch per step, and no further use of that kernel on this handle
            h->lz_multi_off = true;
            const LanczosRun o = r;
            TLSQ_TRY(lanczos_begin(h, r, o.G, o.N, o.ldG, o.rel_tol, o.max_steps, o.accept_below, o.stop_above, o.v0));
            continue;
        }
        const int m = (int)hs[2];
        const bool broke = hs[1] != 0.0;
        if (m > 0) {
            const double* a = hab.data();
            const double* b = hab.data() + r.cap;
            double sm = 0.0;
            theta = tridiag_lmax(a, b, m, &sm);
            if (steps_used) *steps_used = m;
            if (broke || m >= (int)r.N) {
                *lmax = theta > 0.0 ? theta : 0.0;
                return TLSQ_OK;
            }
            const double bound = fabs(b[m - 1]) * sm;  // ||G y - theta y|| for the Ritz pair
            if (bound <= r.rel_tol * fabs(theta)) {
                *lmax = theta > 0.0 ? theta : 0.0;
                return TLSQ_OK;
            }
            // the Ritz value is a lower bound of lambda_max: "is lambda_max >= X?" is settled as soon as it passes X
            if (r.stop_above > 0.0 && theta >= r.stop_above) {
                *lmax = theta;
                return TLSQ_OK;
            }
            // early accept for "is lambda_max clearly below X?" questions: the Ritz value is a lower bound that is
            // already within ~15 % of lambda_max after 16 steps even on flat (noise-like) spectra
            // (Kuczynski-Wozniakowski: with a random start vector the Ritz value misses lambda_max by a factor f after m
            // steps with probability <= 1.65 sqrt(N) exp(-sqrt(1 - 1/f) (2m - 1)): 1e-9 for f = 2.5, m = 16 and 1e-6
            // for f = 6, m = 10 at N = 512; anything earlier is not accepted)
            if (r.accept_below > 0.0 &&
                ((m >= 16 && 2.5 * theta < r.accept_below) || (m >= 10 && 6.0 * theta < r.accept_below))) {
                *lmax = theta > 0.0 ? theta : 0.0;
                return TLSQ_OK;
            }
            if (r.chunk < 64) r.chunk *= 2;
        } else if (broke) {
            break;
        }
        if (r.launched >= r.max_steps + 1) break;
        TLSQ_TRY(lz_launch_chunk(h, r));
    }
    *lmax = theta > 0.0 ? theta : 0.0;
    return 1;
}

int lanczos_lmax_f64(Handle* h, const double* G, int64_t N, int64_t ldG, double rel_tol, int max_steps,
                     double* lmax, int* steps_used, double accept_below, double stop_above, const double* v0) {
    LanczosRun r;
    TLSQ_TRY(lanczos_begin(h, r, G, N, ldG, rel_tol, max_steps, accept_below, stop_above, v0));
    return lanczos_finish(h, r, lmax, steps_used);
}


// ---- power steps on a persistent vector: cheap LOWER bounds of lambda_max ----------------------------------------
// The convergence test of rpca (src/robustPCA.jl:225-228) asks "opnorm(R) / opnorm(D) < tol?" once per iteration, and in
// all but the last iteration the answer is no.  For unit v, ||G v|| <= lambda_max(G): a lower bound that costs one
// product.  The vector is kept from one ALM iteration to the next (the dominant direction of the residual moves
// slowly), so three products usually put the bound within a few per cent of lambda_max - enough to say "not converged"
// without a Lanczos run (~25 dependent launches).  Launch j finishes product j - 1 (norm, bound, normalised vector:
// every workgroup redundantly, like k_lanczos_step) and computes its rows of product j; the last launch only finishes
// and hands the bounds to the host through the mailbox.
// layout of `pw` (doubles): [0..8) bounds; v[N] (persistent); u[2][N]; part[2][LZ_WGS_MAX]
__global__ __launch_bounds__(LZ_THREADS) void k_power_step(const double* __restrict__ G, int64_t ldG, int N,
                                                           double* __restrict__ pw, int j, int nsteps, int init,
                                                           double* mailbox, double seq, int nwg) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* q = sm;          // N
    double* red = sm + N;    // 4
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double* v = pw + 8;
    double* ubuf = v + N;
    double* part = ubuf + 2 * (size_t)N;
    if (j == 0) {
        if (init) {   // deterministic pseudo-random start, normalised
            double nrm = 0.0;
            for (int i = tid; i < N; i += LZ_THREADS) {
                unsigned int x = (unsigned int)i * 2654435761u + 777u;
                x ^= x >> 16;
                x *= 2246822519u;
                x ^= x >> 13;
                x *= 3266489917u;
                x ^= x >> 16;
                const double val = ((double)(x & 0xFFFFFF) + 0.5) / 16777216.0 - 0.5;
                q[i] = val;
                nrm += val * val;
            }
            nrm = block_sum4(nrm, red);
            const double inv = 1.0 / sqrt(nrm);
            for (int i = tid; i < N; i += LZ_THREADS) q[i] *= inv;
        } else {
            for (int i = tid; i < N; i += LZ_THREADS) q[i] = v[i];
        }
    } else {
        const double* u = ubuf + (size_t)((j + 1) % 2) * N;
        const double* pp = part + (size_t)((j + 1) % 2) * LZ_WGS_MAX;
        double nn = 0.0;
        if (nwg <= 64) {
            for (int k = 0; k < nwg; ++k) nn += pp[k];
        } else {
            double a = 0.0;
            for (int k = threadIdx.x; k < nwg; k += LZ_THREADS) a += pp[k];
            nn = block_sum4(a, red);
        }
        const double beta = sqrt(nn);                 // ||G v_{j-1}||, v_{j-1} a unit vector
        const double inv = beta > 1e-290 ? 1.0 / beta : 0.0;
        for (int i = tid; i < N; i += LZ_THREADS) q[i] = u[i] * inv;
        if (blockIdx.x == 0 && tid == 0) pw[j - 1] = beta;
    }
    __syncthreads();
    if (j == nsteps) {   // the finishing launch: persist the vector, publish the bounds
        if (blockIdx.x == 0) {
            for (int i = tid; i < N; i += LZ_THREADS) v[i] = q[i];
            if (mailbox) {
                volatile double* mb = mailbox;
                __syncthreads();
                if (tid < nsteps) mb[8 + tid] = pw[tid];
                __threadfence_system();
                __syncthreads();
                if (tid == 0) mb[0] = seq;
            }
        }
        return;
    }
    double* unew = ubuf + (size_t)(j % 2) * N;
    const int rows_per = (N + nwg - 1) / nwg;
    const int r0 = blockIdx.x * rows_per;
    const int r1 = (r0 + rows_per < N) ? r0 + rows_per : N;
    const double pacc = lz_rows<true>(G, ldG, N, q, unew, r0, r1);   // symmetric: row r == column r
    __syncthreads();
    if (lane == 0) red[w] = pacc;
    __syncthreads();
    if (tid == 0) part[(size_t)(j % 2) * LZ_WGS_MAX + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// *lb_out = the best of `nsteps` lower bounds ||G v|| <= lambda_max(G) along power steps on the handle's persistent
// vector (init: start from a fixed pseudo-random vector).  One mailbox read-back.  Returns 1 when this form does not
// apply (N too large for the LDS copy of the vector, no mailbox): the caller runs Lanczos.
int power_lower_bound(Handle* h, const double* G, int64_t N, int64_t ldG, bool init, int nsteps, double* lb_out) {
    *lb_out = 0.0;
    const bool no_mailbox = dev_is(DEV_NO_MAILBOX, '1');
    const size_t lds = (size_t)(N + 8) * 8;
    if (N <= 0 || nsteps < 1 || nsteps > 8 || lds > 150 * 1024 || !h->mailbox || no_mailbox || h->mailbox_bytes < 1024) return 1;
    void* pwv;
    TLSQ_TRY(ws_get(h, WS_PW, (8 + 3 * (size_t)N + 2 * LZ_WGS_MAX) * 8, &pwv));
    if (lds > 48 * 1024)
        TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_power_step), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
    const double seq = (h->mail_seq += 1.0);
    for (int j = 0; j <= nsteps; ++j)
        hipLaunchKernelGGL(k_power_step, dim3(j == nsteps ? 1 : lz_wgs(N)), dim3(LZ_THREADS), lds, h->stream, G, ldG, (int)N,
                           (double*)pwv, j, nsteps, init ? 1 : 0, j == nsteps ? h->mailbox_dev : (double*)nullptr, seq, lz_wgs(N));
    TLSQ_HIP(h, hipGetLastError());
    volatile double* mb = h->mailbox;
    const double t_poll = now_ms();
    while (mb[0] != seq && now_ms() - t_poll < 2000.0) {
    }
    if (mb[0] != seq) {
        h->mailbox_bytes = 0;   // never seen in practice
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        return 1;
    }
    double best = 0.0;
    for (int j = 0; j < nsteps; ++j) {
        const double b = mb[8 + j];
        if (std::isfinite(b) && b > best) best = b;
    }
    *lb_out = best;
    return TLSQ_OK;
}

// ---- Lanczos on an operator that is only available as a product (large mode: G = Z'Z is never formed) ---------
// One single-workgroup kernel per step does the vector part: alpha = q.w, w -= alpha q + beta_prev q_prev,
// beta = ||w||, q_prev <- q, q <- w / beta; alpha/beta go to device arrays and are read back per chunk of steps.
__global__ __launch_bounds__(1024) void k_lz_init(double* __restrict__ q, double* __restrict__ qprev, int N,
                                                  double* __restrict__ st) {
    __shared__ double red[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double nrm = 0.0;
    for (int i = tid; i < N; i += 1024) {
        unsigned int x = (unsigned int)i * 2654435761u + 12345u;
        x ^= x >> 16;
        x *= 2246822519u;
        x ^= x >> 13;
        x *= 3266489917u;
        x ^= x >> 16;
        const double v = ((double)(x & 0xFFFFFF) + 0.5) / 16777216.0 - 0.5;
        q[i] = v;
        qprev[i] = 0.0;
        nrm += v * v;
    }
    nrm = wsum(nrm);
    if (lane == 0) red[w] = nrm;
    __syncthreads();
    double tot = 0.0;
    for (int k = 0; k < 16; ++k) tot += red[k];
    const double inv = 1.0 / sqrt(tot);
    for (int i = tid; i < N; i += 1024) q[i] *= inv;
    if (tid == 0) {
        st[0] = 0.0;   // beta_prev
        st[1] = 0.0;   // breakdown
        st[2] = 0.0;   // completed (alpha, beta) pairs
    }
}

__global__ __launch_bounds__(1024) void k_lz_vec(double* __restrict__ q, double* __restrict__ qprev,
                                                 double* __restrict__ wv, int N, double* __restrict__ st,
                                                 double* __restrict__ ab, int cap, int j) {
    __shared__ double red[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (st[1] != 0.0) return;
    double a = 0.0;
    for (int i = tid; i < N; i += 1024) a += q[i] * wv[i];
    a = wsum(a);
    if (lane == 0) red[w] = a;
    __syncthreads();
    double alpha = 0.0;
    for (int k = 0; k < 16; ++k) alpha += red[k];
    __syncthreads();
    const double beta_prev = st[0];
    double nn = 0.0;
    for (int i = tid; i < N; i += 1024) {
        const double v = wv[i] - alpha * q[i] - beta_prev * qprev[i];
        wv[i] = v;
        nn += v * v;
    }
    nn = wsum(nn);
    if (lane == 0) red[w] = nn;
    __syncthreads();
    double b2 = 0.0;
    for (int k = 0; k < 16; ++k) b2 += red[k];
    const double beta = sqrt(b2);
    __syncthreads();
    if (tid == 0) {
        ab[j] = alpha;
        ab[cap + j] = beta;
        st[2] = (double)(j + 1);
        st[0] = beta;
        if (!(beta > 1e-290)) st[1] = 1.0;
    }
    if (!(beta > 1e-290)) return;
    const double inv = 1.0 / beta;
    for (int i = tid; i < N; i += 1024) {
        qprev[i] = q[i];
        q[i] = wv[i] * inv;
    }
}

int lanczos_lmax_op(Handle* h, int64_t N, const LzApply& apply, double rel_tol, int max_steps, double* lmax,
                    int* steps_used, double accept_below, double stop_above) {
    if (N <= 0) {
        *lmax = 0.0;
        return TLSQ_OK;
    }
    if (max_steps > (int)N) max_steps = (int)N;
    if (max_steps < 1) max_steps = 1;
    const int cap = max_steps + 2;
    if ((size_t)(2 * cap) * 8 + 64 > h->pinned_bytes) return 1;
    void* stv;
    TLSQ_TRY(ws_get(h, WS_LZOP, (8 + 3 * (size_t)N + 2 * (size_t)cap + 16) * 8, &stv));
    double* st = (double*)stv;
    double *q = st + 8, *qprev = q + N, *wv = qprev + N, *ab = wv + N;
    hipLaunchKernelGGL(k_lz_init, dim3(1), dim3(1024), 0, h->stream, q, qprev, (int)N, st);
    TLSQ_HIP(h, hipGetLastError());
    std::vector<double> hab((size_t)2 * cap);
    int done = 0;
    double theta = 0.0;
    int chunk = stop_above > 0.0 ? 4 : (accept_below > 0.0 ? 10 : 16);
    while (done < max_steps) {
        const int n = std::min(chunk, max_steps - done);
        for (int k = 0; k < n; ++k, ++done) {
            TLSQ_TRY(apply(q, wv));
            hipLaunchKernelGGL(k_lz_vec, dim3(1), dim3(1024), 0, h->stream, q, qprev, wv, (int)N, st, ab, cap, done);
        }
        TLSQ_HIP(h, hipGetLastError());
        TLSQ_HIP(h, hipMemcpyAsync(h->pinned, st, 64, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipMemcpyAsync((char*)h->pinned + 64, ab, (size_t)2 * cap * 8, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        double hs[8];
        memcpy(hs, h->pinned, 64);
        memcpy(hab.data(), (char*)h->pinned + 64, (size_t)2 * cap * 8);
        const int m = (int)hs[2];
        const bool broke = hs[1] != 0.0;
        if (m <= 0) {
            if (broke) break;
            continue;
        }
        const double* a = hab.data();
        const double* b = hab.data() + cap;
        double sm = 0.0;
        theta = tridiag_lmax(a, b, m, &sm);
        if (steps_used) *steps_used = m;
        if (broke || m >= (int)N) {
            *lmax = theta > 0.0 ? theta : 0.0;
            return TLSQ_OK;
        }
        const double bound = fabs(b[m - 1]) * sm;
        if (bound <= rel_tol * fabs(theta)) {
            *lmax = theta > 0.0 ? theta : 0.0;
            return TLSQ_OK;
        }
        if (stop_above > 0.0 && theta >= stop_above) {
            *lmax = theta;
            return TLSQ_OK;
        }
        if (accept_below > 0.0 && ((m >= 16 && 2.5 * theta < accept_below) || (m >= 10 && 6.0 * theta < accept_below))) {
            *lmax = theta > 0.0 ? theta : 0.0;
            return TLSQ_OK;
        }
        if (chunk < 64) chunk *= 2;
    }
    *lmax = theta > 0.0 ? theta : 0.0;
    return 1;
}

}  // namespace tlsq
