// Small dense symmetric eigensolver: one-sided (Hestenes) block Jacobi, LDS-resident column pairs.
//
// Replaces the LAPACK work behind `LinearAlgebra.svd!(Z)` / `opnorm` (src/robustPCA.jl:194,225 under
// /root/reference) once Z has been reduced to its N x N Gram matrix G = Z'Z:  the right singular
// vectors of Z are the eigenvectors V of G and sigma_i = sqrt(lambda_i).
//
// Method: iterate on B = G*V (initially B = G, V = I).  A plane rotation of columns (p,q) of B and V
// that makes B[:,p] _|_ B[:,q] is a Jacobi step on B'B = V'G^2V; at convergence the columns of B are
// lambda_i * v_i, so ||B[:,i]|| = lambda_i.  Columns are grouped into blocks of b columns; one
// workgroup owns a block pair per round (round-robin tournament over blocks), keeps its 2b columns of B
// and of V in LDS (2b*N*16 bytes <= 160 KiB/CU), and runs a full inner tournament over those 2b
// columns with one 64-lane wave per column pair (wave-shuffle reductions for the three dot products).
// No two-sided row updates, no LAPACK, deterministic (fixed pair order, fixed reduction trees).
#include "common.hpp"

namespace tlsq {

// Wave-wide all-reduce without the LDS crossbar: four DPP steps inside each row of 16 lanes (quad_perm,
// quad_perm, row_half_mirror, row_mirror), then the four row totals are read with v_readlane and summed.
// (__shfl_xor compiles to ds_bpermute_b32: six dependent LDS round trips per fp64 reduction.)
template <int CTRL>
__device__ __forceinline__ double dpp_perm(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_allsum(double v) {
    v += dpp_perm<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_perm<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_perm<0x141>(v);   // row_half_mirror
    v += dpp_perm<0x140>(v);   // row_mirror
    return (read_lane(v, 0) + read_lane(v, 16)) + (read_lane(v, 32) + read_lane(v, 48));
}

// round-robin tournament: pair `idx` of round `r` among n (even) players
__device__ __forceinline__ void rr_pair(int n, int r, int idx, int& p, int& q) {
    const int m = n - 1;
    if (idx == 0) {
        p = r % m;
        q = m;
    } else {
        p = (r + idx) % m;
        q = (r - idx + m) % m;
    }
}

// params[0] = floor2 (squared column-norm noise floor)
// One workgroup = one block pair (2b column slots).  Waves loop over the b column pairs of each inner round.
// When the whole matrix is a single block pair (nblk == 2) the kernel can run `max_inner_sweeps` complete
// sweeps by itself (stops early after a sweep without rotations) and reports the number in sweeps_done.
template <bool WANT_V>
__global__ __launch_bounds__(1024) void k_jacobi_round(double* __restrict__ B, double* __restrict__ V, int N,
                                                       int b, int nblk, int round, double tol,
                                                       const double* __restrict__ params,
                                                       unsigned int* __restrict__ rot_count,
                                                       int max_inner_sweeps, int* __restrict__ sweeps_done) {
    // dynamic LDS only (keeps the base 16-byte aligned): sB[2b][N], sV[2b][N] (if WANT_V), rotation counter
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int nslot = 2 * b;
    double* sB = sm;
    double* sV = sm + (size_t)nslot * N;
    unsigned int* s_cnt = reinterpret_cast<unsigned int*>(sm + (size_t)nslot * N * (WANT_V ? 2 : 1));
    double* sN = sm + (size_t)nslot * N * (WANT_V ? 2 : 1) + 2;   // cached squared column norms, one per slot
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int bp, bq;
    rr_pair(nblk, round, blockIdx.x, bp, bq);
    if (threadIdx.x == 0) {
        s_cnt[0] = 0;  // rotations in the current sweep
        s_cnt[1] = 0;  // total
    }

    // ---- load the 2b columns (global column-major, ld = N) ----
    for (int s = w; s < nslot; s += nw) {
        const int col = (s < b) ? bp * b + s : bq * b + (s - b);
        if (col < N) {
            for (int row = lane; row < N; row += 64) {
                sB[(size_t)s * N + row] = B[(size_t)col * N + row];
                if (WANT_V) sV[(size_t)s * N + row] = V[(size_t)col * N + row];
            }
        }
    }
    __syncthreads();
    const double floor2 = params[0];

    // ---- inner tournament over the 2b slots ----
    // The squared column norms are computed once per sweep and then carried through the rotations
    // (a' = a - t c, b' = b + t c), so a pair costs one dot product + one wave reduction instead of three.
    const int nin = nslot - 1;
    int sweep = 0;
    for (; sweep < max_inner_sweeps; ++sweep) {
        for (int sl = w; sl < nslot; sl += nw) {
            const double* xcol = sB + (size_t)sl * N;
            double nn = 0.0;
            for (int row = lane; row < N; row += 64) nn += xcol[row] * xcol[row];
            nn = wave_allsum(nn);
            if (lane == 0) sN[sl] = nn;
        }
        __syncthreads();
        unsigned int my_rot = 0;
        for (int ir = 0; ir < nin; ++ir) {
            for (int pi = w; pi < b; pi += nw) {
                int s1, s2;
                rr_pair(nslot, ir, pi, s1, s2);
                if (s1 > s2) {
                    int t = s1;
                    s1 = s2;
                    s2 = t;
                }
                const int c1 = (s1 < b) ? bp * b + s1 : bq * b + (s1 - b);
                const int c2 = (s2 < b) ? bp * b + s2 : bq * b + (s2 - b);
                if (c1 < N && c2 < N) {
                    double* x = sB + (size_t)s1 * N;
                    double* y = sB + (size_t)s2 * N;
                    // Rows are handled in chunks of 8 per lane (512 rows): all 16 LDS reads of a chunk are issued
                    // before the first use, and for N <= 512 the columns stay in registers from the dot products
                    // to the rotation (one LDS round trip instead of three).
                    constexpr int RC = 8;
                    const int nchunk = (N + 64 * RC - 1) / (64 * RC);
                    double xr[RC], yr[RC];
                    const double a = sN[s1], bb = sN[s2];
                    double c = 0.0;
                    for (int ch = nchunk - 1; ch >= 0; --ch) {   // chunk 0 last: it is the one kept in registers
#pragma unroll
                        for (int i = 0; i < RC; ++i) {
                            const int row = lane + 64 * (ch * RC + i);
                            xr[i] = row < N ? x[row] : 0.0;
                            yr[i] = row < N ? y[row] : 0.0;
                        }
#pragma unroll
                        for (int i = 0; i < RC; ++i) c += xr[i] * yr[i];
                    }
                    c = wave_allsum(c);
                    const double mn = a < bb ? a : bb;
                    if (c * c > tol * tol * a * bb && mn > floor2) {
                        // tan(theta) of the Jacobi rotation, smaller root of t^2 + 2 zeta t - 1 = 0 with
                        // zeta = (bb - a)/(2c), written so that it needs one sqrt and one division:
                        //   t = 2 c sgn(d) / (|d| + sqrt(d^2 + 4 c^2)),  d = bb - a
                        const double d = bb - a;
                        const double t = (d >= 0.0 ? 2.0 : -2.0) * c / (fabs(d) + sqrt(d * d + 4.0 * c * c));
                        const double cs = 1.0 / sqrt(1.0 + t * t);
                        const double sn = cs * t;
                        if (lane == 0) {
                            const double na = a - t * c, nb2 = bb + t * c;
                            sN[s1] = na > 0.0 ? na : 0.0;
                            sN[s2] = nb2 > 0.0 ? nb2 : 0.0;
                        }
                        for (int ch = 0; ch < nchunk; ++ch) {
                            if (ch > 0) {
#pragma unroll
                                for (int i = 0; i < RC; ++i) {
                                    const int row = lane + 64 * (ch * RC + i);
                                    xr[i] = row < N ? x[row] : 0.0;
                                    yr[i] = row < N ? y[row] : 0.0;
                                }
                            }
#pragma unroll
                            for (int i = 0; i < RC; ++i) {
                                const int row = lane + 64 * (ch * RC + i);
                                if (row < N) {
                                    x[row] = cs * xr[i] - sn * yr[i];
                                    y[row] = sn * xr[i] + cs * yr[i];
                                }
                            }
                        }
                        if (WANT_V) {
                            double* vx = sV + (size_t)s1 * N;
                            double* vy = sV + (size_t)s2 * N;
                            for (int ch = 0; ch < nchunk; ++ch) {
#pragma unroll
                                for (int i = 0; i < RC; ++i) {
                                    const int row = lane + 64 * (ch * RC + i);
                                    xr[i] = row < N ? vx[row] : 0.0;
                                    yr[i] = row < N ? vy[row] : 0.0;
                                }
#pragma unroll
                                for (int i = 0; i < RC; ++i) {
                                    const int row = lane + 64 * (ch * RC + i);
                                    if (row < N) {
                                        vx[row] = cs * xr[i] - sn * yr[i];
                                        vy[row] = sn * xr[i] + cs * yr[i];
                                    }
                                }
                            }
                        }
                        ++my_rot;
                    }
                }
            }
            __syncthreads();
        }
        if (lane == 0 && my_rot) atomicAdd(&s_cnt[0], my_rot);
        __syncthreads();
        const unsigned int r = s_cnt[0];
        __syncthreads();
        if (threadIdx.x == 0) {
            s_cnt[1] += r;
            s_cnt[0] = 0;
        }
        if (r == 0) {
            ++sweep;
            break;
        }
    }
    __syncthreads();

    // ---- store back ----
    for (int s = w; s < nslot; s += nw) {
        const int col = (s < b) ? bp * b + s : bq * b + (s - b);
        if (col < N) {
            for (int row = lane; row < N; row += 64) {
                B[(size_t)col * N + row] = sB[(size_t)s * N + row];
                if (WANT_V) V[(size_t)col * N + row] = sV[(size_t)s * N + row];
            }
        }
    }
    if (threadIdx.x == 0) {
        if (s_cnt[1]) atomicAdd(rot_count, s_cnt[1]);
        if (sweeps_done) *sweeps_done = sweep;
    }
}

// ---- small problems (N <= 64): the whole eigen-decomposition in ONE launch ---------------------------------
// init (B = G, V = I), noise floor, all sweeps, and the final column norms run inside a single 1024-thread
// workgroup; a 32-lane half-wave owns one column pair per inner round (lane = row, rows <= 64), so a
// 24 x 24 Rayleigh-Ritz problem needs no inter-wave traffic beyond one barrier per inner round.
__device__ __forceinline__ double half_allsum(double v) {
    // sum over the 32 lanes of this half-wave: DPP inside each row of 16, then the two row totals of the half
    v += dpp_perm<0xB1>(v);
    v += dpp_perm<0x4E>(v);
    v += dpp_perm<0x141>(v);
    v += dpp_perm<0x140>(v);
    const double lo = read_lane(v, 0) + read_lane(v, 16);
    const double hi = read_lane(v, 32) + read_lane(v, 48);
    return (threadIdx.x & 32) ? hi : lo;
}

template <bool WANT_V>
__global__ __launch_bounds__(1024) void k_jacobi_small(const double* __restrict__ G, int64_t ldG,
                                                       double* __restrict__ Bout, double* __restrict__ Vout,
                                                       double* __restrict__ lam, int N, double tol, double nfloor,
                                                       int max_sweeps, int* __restrict__ sweeps_done) {
    __shared__ double sB[64 * 65];
    __shared__ double sV[WANT_V ? 64 * 65 : 1];
    __shared__ double red[16];
    __shared__ double sN[64];
    __shared__ unsigned int s_rot;
    __shared__ unsigned long long s_max;
    const int LD = 65;
    const int tid = threadIdx.x;
    const int hw = tid >> 5, hl = tid & 31;   // half-wave index, lane within it
    const int nthr = blockDim.x, nhw = nthr >> 5;   // the launcher sizes the block to the number of pairs
    // init + ||G||_F^2
    double fro = 0.0;
    for (int e = tid; e < N * N; e += nthr) {
        const int r = e % N, c = e / N;
        const double v = G[r + (int64_t)c * ldG];
        sB[c * LD + r] = v;
        if (WANT_V) sV[c * LD + r] = (r == c) ? 1.0 : 0.0;
        fro += v * v;
    }
    fro = wave_allsum(fro);
    if ((tid & 63) == 0) red[tid >> 6] = fro;
    if (tid == 0) {
        s_rot = 0;
        s_max = 0ull;
    }
    __syncthreads();
    double fsum = 0.0;
    for (int k = 0; k < (nthr >> 6); ++k) fsum += red[k];
    const double floor2 = nfloor * nfloor * fsum;
    const int nslot = (N + 1) & ~1;          // even number of tournament players
    const int npair = nslot / 2;             // <= 32
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        unsigned int my_rot = 0;
        double my_max = 0.0;
        // squared column norms, refreshed once per sweep and updated by the rotation formulas in between
        for (int c = hw; c < N; c += nhw) {
            const double* x = sB + c * LD;
            const double v0 = hl < N ? x[hl] : 0.0, v1 = hl + 32 < N ? x[hl + 32] : 0.0;
            const double ssum = half_allsum(v0 * v0 + v1 * v1);
            if (hl == 0) sN[c] = ssum;
        }
        __syncthreads();
        for (int ir = 0; ir < nslot - 1; ++ir) {
            if (hw < npair) {
                int s1, s2;
                rr_pair(nslot, ir, hw, s1, s2);
                if (s1 > s2) {
                    const int t = s1;
                    s1 = s2;
                    s2 = t;
                }
                if (s2 < N) {
                    double* x = sB + s1 * LD;
                    double* y = sB + s2 * LD;
                    const int r0 = hl, r1 = hl + 32;
                    const double x0 = r0 < N ? x[r0] : 0.0, y0 = r0 < N ? y[r0] : 0.0;
                    const double x1 = r1 < N ? x[r1] : 0.0, y1 = r1 < N ? y[r1] : 0.0;
                    const double a = sN[s1], bb = sN[s2];
                    const double c = half_allsum(x0 * y0 + x1 * y1);
                    const double mn = a < bb ? a : bb;
                    if (c * c > tol * tol * a * bb && mn > floor2) {
                        const double ratio2 = c * c / (a * bb);   // cos^2 of the angle between the two columns
                        my_max = ratio2 > my_max ? ratio2 : my_max;
                        // t = 2 c sgn(d) / (|d| + sqrt(d^2 + 4 c^2)), d = bb - a  (one sqrt, one division)
                        const double d = bb - a;
                        const double t = (d >= 0.0 ? 2.0 : -2.0) * c / (fabs(d) + sqrt(d * d + 4.0 * c * c));
                        const double cs = rsqrt(1.0 + t * t);
                        const double sn = cs * t;
                        if (hl == 0) {
                            const double na = a - t * c, nb2 = bb + t * c;
                            sN[s1] = na > 0.0 ? na : 0.0;
                            sN[s2] = nb2 > 0.0 ? nb2 : 0.0;
                        }
                        if (r0 < N) {
                            x[r0] = cs * x0 - sn * y0;
                            y[r0] = sn * x0 + cs * y0;
                        }
                        if (r1 < N) {
                            x[r1] = cs * x1 - sn * y1;
                            y[r1] = sn * x1 + cs * y1;
                        }
                        if (WANT_V) {
                            double* vx = sV + s1 * LD;
                            double* vy = sV + s2 * LD;
                            if (r0 < N) {
                                const double u = vx[r0], w = vy[r0];
                                vx[r0] = cs * u - sn * w;
                                vy[r0] = sn * u + cs * w;
                            }
                            if (r1 < N) {
                                const double u = vx[r1], w = vy[r1];
                                vx[r1] = cs * u - sn * w;
                                vy[r1] = sn * u + cs * w;
                            }
                        }
                        ++my_rot;
                    }
                }
            }
            __syncthreads();
        }
        if (hl == 0 && my_rot) {
            atomicAdd(&s_rot, my_rot);
            atomicMax(&s_max, (unsigned long long)__double_as_longlong(my_max));   // non-negative doubles order like integers
        }
        __syncthreads();
        const unsigned int r = s_rot;
        const double swmax = __longlong_as_double((long long)s_max);
        __syncthreads();
        if (tid == 0) {
            s_rot = 0;
            s_max = 0ull;
        }
        // converged: a sweep without rotations - or one whose largest rotation was so small (|cos| < 1e-8) that,
        // Jacobi converging quadratically, what is left is below the rotation threshold anyway
        if (r == 0 || swmax < (tol > 1e-12 ? tol : 1e-16)) {   // (a caller's loose threshold t: the sweep after cos^2 < t leaves cos ~ t)
            ++sweep;
            break;
        }
    }
    __syncthreads();
    for (int e = tid; e < N * N; e += nthr) {
        const int r = e % N, c = e / N;
        Bout[e] = sB[c * LD + r];
        if (WANT_V) Vout[e] = sV[c * LD + r];
    }
    // lam[c] = ||B[:,c]||: half-wave per column
    for (int c = hw; c < N; c += nhw) {
        const double* x = sB + c * LD;
        const double v0 = hl < N ? x[hl] : 0.0, v1 = hl + 32 < N ? x[hl + 32] : 0.0;
        const double ssum = half_allsum(v0 * v0 + v1 * v1);
        if (hl == 0) lam[c] = sqrt(ssum);
    }
    if (tid == 0 && sweeps_done) *sweeps_done = sweep;
}

// ---- two-sided Jacobi for the Rayleigh-Ritz problems of the subspace iteration (N <= 64, one launch) ----------------------
// H = Q'GQ is symmetric and, on a warm block, nearly diagonal.  The one-sided form above needs a dot product over the
// rows - a DPP reduction in the middle of every rotation's dependent chain - and carries B = H V beside V; rotating H
// itself takes the rotation from three entries (h_pp, h_qq, h_pq), then updates the two columns (lane = row) and, after
// a barrier, the two rows (lane = column) of every pair of the round: ~0.3 us per round instead of ~0.8.  Its accuracy is
// absolute (eps ||H||), which is what the Gram matrix behind H has anyway (SubspaceState::noise_rel); the solvers that
// need relative accuracy (dense Jacobi on a triangular factor) keep the one-sided form.
// Outputs as k_jacobi_small: V = eigenvectors, lam = |eigenvalues| (unsorted), B = H V = V diag(lambda).
template <bool WANT_V>
__global__ __launch_bounds__(1024) void k_jacobi2_small(const double* __restrict__ G, int64_t ldG,
                                                        double* __restrict__ Bout, double* __restrict__ Vout,
                                                        double* __restrict__ lam, int N, double tol, double nfloor,
                                                        int max_sweeps, int* __restrict__ sweeps_done) {
    __shared__ double sH[64 * 65];
    __shared__ double sV[WANT_V ? 64 * 65 : 1];
    __shared__ double red[16];
    __shared__ unsigned int s_rot;
    __shared__ unsigned long long s_max;
    const int LD = 65;
    const int tid = threadIdx.x;
    const int hw = tid >> 5, hl = tid & 31;   // half-wave = one pair of the round
    const int nthr = blockDim.x;
    double fro = 0.0;
    for (int e = tid; e < N * N; e += nthr) {
        const int r = e % N, c = e / N;
        // (symmetrised: H comes from a product Q'(GQ) whose two triangles agree to rounding only)
        const double v = 0.5 * (G[r + (int64_t)c * ldG] + G[c + (int64_t)r * ldG]);
        sH[c * LD + r] = v;
        if (WANT_V) sV[c * LD + r] = (r == c) ? 1.0 : 0.0;
        fro += v * v;
    }
    fro = wave_allsum(fro);
    if ((tid & 63) == 0) red[tid >> 6] = fro;
    if (tid == 0) {
        s_rot = 0;
        s_max = 0ull;
    }
    __syncthreads();
    double fsum = 0.0;
    for (int k = 0; k < (nthr >> 6); ++k) fsum += red[k];
    const double floor2 = nfloor * nfloor * fsum;   // off-diagonal entries below nfloor ||H||_F are rounding noise
    const int nslot = (N + 1) & ~1;
    const int npair = nslot / 2;
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        unsigned int my_rot = 0;
        double my_max = 0.0;
        for (int ir = 0; ir < nslot - 1; ++ir) {
            int p = 0, q = 0;
            bool rotate = false;
            double cs = 1.0, sn = 0.0;
            if (hw < npair) {
                rr_pair(nslot, ir, hw, p, q);
                if (p > q) {
                    const int t = p;
                    p = q;
                    q = t;
                }
                if (q < N) {
                    const double hpp = sH[p * LD + p], hqq = sH[q * LD + q], hpq = sH[q * LD + p];
                    const double prod = fabs(hpp * hqq);
                    if (hpq * hpq > tol * tol * prod && hpq * hpq > floor2) {
                        rotate = true;
                        const double ratio2 = prod > 0.0 ? hpq * hpq / prod : 1.0;
                        my_max = ratio2 > my_max ? ratio2 : my_max;
                        // t = 2 h_pq sgn(d) / (|d| + sqrt(d^2 + 4 h_pq^2)), d = h_qq - h_pp
                        const double d = hqq - hpp;
                        const double t = (d >= 0.0 ? 2.0 : -2.0) * hpq / (fabs(d) + sqrt(d * d + 4.0 * hpq * hpq));
                        cs = rsqrt(1.0 + t * t);
                        sn = cs * t;
                        ++my_rot;
                        // columns p, q of H (and of V): lane = row
                        for (int r = hl; r < N; r += 32) {
                            const double x = sH[p * LD + r], y = sH[q * LD + r];
                            sH[p * LD + r] = cs * x - sn * y;
                            sH[q * LD + r] = sn * x + cs * y;
                            if (WANT_V) {
                                const double u = sV[p * LD + r], w2 = sV[q * LD + r];
                                sV[p * LD + r] = cs * u - sn * w2;
                                sV[q * LD + r] = sn * u + cs * w2;
                            }
                        }
                    }
                }
            }
            __syncthreads();
            if (rotate) {
                // rows p, q of H: lane = column
                for (int c = hl; c < N; c += 32) {
                    const double x = sH[c * LD + p], y = sH[c * LD + q];
                    sH[c * LD + p] = cs * x - sn * y;
                    sH[c * LD + q] = sn * x + cs * y;
                }
                // (same half-wave, program order: the rotated pair is exactly decoupled)
                if (hl == 0) {
                    sH[q * LD + p] = 0.0;
                    sH[p * LD + q] = 0.0;
                }
            }
            __syncthreads();
        }
        if (hl == 0 && my_rot) {
            atomicAdd(&s_rot, my_rot);
            atomicMax(&s_max, (unsigned long long)__double_as_longlong(my_max));
        }
        __syncthreads();
        const unsigned int r = s_rot;
        const double swmax = __longlong_as_double((long long)s_max);
        __syncthreads();
        if (tid == 0) {
            s_rot = 0;
            s_max = 0ull;
        }
        // converged: a sweep without rotations - or one whose largest rotation was so small (h_pq^2 < 1e-16 h_pp h_qq)
        // that, Jacobi converging quadratically, what is left is below the rotation threshold anyway
        if (r == 0 || swmax < (tol > 1e-12 ? tol : 1e-16)) {   // (a caller's loose threshold t: the sweep after cos^2 < t leaves cos ~ t)
            ++sweep;
            break;
        }
    }
    __syncthreads();
    for (int e = tid; e < N * N; e += nthr) {
        const int r = e % N, c = e / N;
        const double v = WANT_V ? sV[c * LD + r] : 0.0;
        if (WANT_V) Vout[e] = v;
        Bout[e] = v * sH[c * LD + c];
    }
    for (int c = tid; c < N; c += nthr) lam[c] = fabs(sH[c * LD + c]);
    if (tid == 0 && sweeps_done) *sweeps_done = sweep;
}

// The same for 64 < N <= 96 (Rayleigh-Ritz problems of blocks of 65..96 columns): B and V in dynamic LDS (2 N (N+1) doubles
// <= 149 KB), one ROW OF 16 LANES per column pair - six rows per lane, the pair's dot product from four DPP steps inside the
// row - so that the up to 48 pairs of a round run at once on the 64 rows of the workgroup (with half-waves, 32 of them, a
// round took two passes: 1.9 us per round, 0.14 ms per sweep at N = 74 - the randomized hook's 1.1 ms eigenproblem).
__device__ __forceinline__ double row_allsum(double v) {   // sum over the 16 lanes of this DPP row
    v += dpp_perm<0xB1>(v);
    v += dpp_perm<0x4E>(v);
    v += dpp_perm<0x141>(v);
    v += dpp_perm<0x140>(v);
    return v;
}
// (the body of k_jacobi_mid and of k_jacobi_mid_blocks: G with leading dimension ldG, symmetrised on the way in when SYMM; Bout /
//  lam may be null; Vout with leading dimension ldV; sweeps_done: the sweep count, or with SW_MAX the maximum over the launch's
//  workgroups)
template <bool WANT_V, bool SYMM, bool SW_MAX, int RL>   // RL: rows per lane, 16 RL >= N
__device__ __forceinline__ void jacobi_mid_body(double* jm_sm, const double* __restrict__ G, int64_t ldG, double* __restrict__ Bout,
                                                double* __restrict__ Vout, int64_t ldV, double* __restrict__ lam, int N, double tol,
                                                double nfloor, int max_sweeps, int* __restrict__ sweeps_done) {
    __shared__ double red[16];
    __shared__ unsigned int s_rot;
    __shared__ unsigned long long s_max;
    const int LD = N + 1;
    double* sB = jm_sm;
    double* sV = jm_sm + (size_t)N * LD;
    double* sN = jm_sm + (size_t)(WANT_V ? 2 : 1) * N * LD;
    const int tid = threadIdx.x;
    const int hw = tid >> 4, hl = tid & 15;   // lane row (one column pair of the round), lane within it
    const int nthr = blockDim.x, nhw = nthr >> 4;
    // init + ||G||_F^2
    double fro = 0.0;
    for (int e = tid; e < N * N; e += nthr) {
        const int r = e % N, c = e / N;
        const double v = SYMM ? 0.5 * (G[r + (int64_t)c * ldG] + G[c + (int64_t)r * ldG]) : G[r + (int64_t)c * ldG];
        sB[c * LD + r] = v;
        if (WANT_V) sV[c * LD + r] = (r == c) ? 1.0 : 0.0;
        fro += v * v;
    }
    fro = wave_allsum(fro);
    if ((tid & 63) == 0) red[tid >> 6] = fro;
    if (tid == 0) {
        s_rot = 0;
        s_max = 0ull;
    }
    __syncthreads();
    double fsum = 0.0;
    for (int k = 0; k < (nthr >> 6); ++k) fsum += red[k];
    const double floor2 = nfloor * nfloor * fsum;
    const int nslot = (N + 1) & ~1;          // even number of tournament players
    const int npair = nslot / 2;             // <= 48
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        unsigned int my_rot = 0;
        double my_max = 0.0;
        // squared column norms, refreshed once per sweep and updated by the rotation formulas in between
        for (int c = hw; c < N; c += nhw) {
            const double* x = sB + c * LD;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < RL; ++k) {
                const int r = hl + 16 * k;
                const double v = r < N ? x[r] : 0.0;
                acc += v * v;
            }
            const double ssum = row_allsum(acc);
            if (hl == 0) sN[c] = ssum;
        }
        __syncthreads();
        for (int ir = 0; ir < nslot - 1; ++ir) {
            for (int ip = hw; ip < npair; ip += nhw) {   // (one pass: 64 lane rows, at most 48 pairs)
                int s1, s2;
                rr_pair(nslot, ir, ip, s1, s2);
                if (s1 > s2) {
                    const int t = s1;
                    s1 = s2;
                    s2 = t;
                }
                if (s2 < N) {
                    double* x = sB + s1 * LD;
                    double* y = sB + s2 * LD;
                    double xv[RL], yv[RL];
                    double dot = 0.0;
#pragma unroll
                    for (int k = 0; k < RL; ++k) {
                        const int r = hl + 16 * k;
                        xv[k] = r < N ? x[r] : 0.0;
                        yv[k] = r < N ? y[r] : 0.0;
                        dot += xv[k] * yv[k];
                    }
                    const double a = sN[s1], bb = sN[s2];
                    const double c = row_allsum(dot);
                    const double mn = a < bb ? a : bb;
                    if (c * c > tol * tol * a * bb && mn > floor2) {
                        const double ratio2 = c * c / (a * bb);   // cos^2 of the angle between the two columns
                        my_max = ratio2 > my_max ? ratio2 : my_max;
                        // t = 2 c sgn(d) / (|d| + sqrt(d^2 + 4 c^2)), d = bb - a  (one sqrt, one division)
                        const double d = bb - a;
                        const double t = (d >= 0.0 ? 2.0 : -2.0) * c / (fabs(d) + sqrt(d * d + 4.0 * c * c));
                        const double cs = rsqrt(1.0 + t * t);
                        const double sn = cs * t;
                        if (hl == 0) {
                            const double na = a - t * c, nb2 = bb + t * c;
                            sN[s1] = na > 0.0 ? na : 0.0;
                            sN[s2] = nb2 > 0.0 ? nb2 : 0.0;
                        }
#pragma unroll
                        for (int k = 0; k < RL; ++k) {
                            const int r = hl + 16 * k;
                            if (r < N) {
                                x[r] = cs * xv[k] - sn * yv[k];
                                y[r] = sn * xv[k] + cs * yv[k];
                            }
                        }
                        if (WANT_V) {
                            double* vx = sV + s1 * LD;
                            double* vy = sV + s2 * LD;
#pragma unroll
                            for (int k = 0; k < RL; ++k) {
                                const int r = hl + 16 * k;
                                if (r < N) {
                                    const double u = vx[r], w = vy[r];
                                    vx[r] = cs * u - sn * w;
                                    vy[r] = sn * u + cs * w;
                                }
                            }
                        }
                        ++my_rot;
                    }
                }
            }
            __syncthreads();
        }
        if (hl == 0 && my_rot) {
            atomicAdd(&s_rot, my_rot);
            atomicMax(&s_max, (unsigned long long)__double_as_longlong(my_max));   // non-negative doubles order like integers
        }
        __syncthreads();
        const unsigned int r = s_rot;
        const double swmax = __longlong_as_double((long long)s_max);
        __syncthreads();
        if (tid == 0) {
            s_rot = 0;
            s_max = 0ull;
        }
        // converged: a sweep without rotations - or one whose largest rotation was so small (|cos| < 1e-8) that,
        // Jacobi converging quadratically, what is left is below the rotation threshold anyway
        if (r == 0 || swmax < (tol > 1e-12 ? tol : 1e-16)) {   // (a caller's loose threshold t: the sweep after cos^2 < t leaves cos ~ t)
            ++sweep;
            break;
        }
    }
    __syncthreads();
    for (int e = tid; e < N * N; e += nthr) {
        const int r = e % N, c = e / N;
        if (Bout) Bout[e] = sB[c * LD + r];
        if (WANT_V) Vout[r + (int64_t)c * ldV] = sV[c * LD + r];
    }
    // lam[c] = ||B[:,c]||: lane row per column
    for (int c = hw; lam && c < N; c += nhw) {
        const double* x = sB + c * LD;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < RL; ++k) {
            const int r = hl + 16 * k;
            const double v = r < N ? x[r] : 0.0;
            acc += v * v;
        }
        const double ssum = row_allsum(acc);
        if (hl == 0) lam[c] = sqrt(ssum);
    }
    if (tid == 0 && sweeps_done) {
        if (SW_MAX) atomicMax(sweeps_done, sweep);
        else *sweeps_done = sweep;
    }
}

template <bool WANT_V, int RL>
__global__ __launch_bounds__(1024) void k_jacobi_mid(const double* __restrict__ G, int64_t ldG,
                                                       double* __restrict__ Bout, double* __restrict__ Vout,
                                                       double* __restrict__ lam, int N, double tol, double nfloor,
                                                       int max_sweeps, int* __restrict__ sweeps_done) {
    extern __shared__ __attribute__((aligned(16))) double jm_sm[];   // sB[N*LD], sV[N*LD] (WANT_V), sN[N]
    jacobi_mid_body<WANT_V, false, false, RL>(jm_sm, G, ldG, Bout, Vout, N, lam, N, tol, nfloor, max_sweeps, sweeps_done);
}

// The same solver on the DIAGONAL BLOCKS of a symmetric T (N x N, ld N), one workgroup per block (round 6: the slices'
// own eigenproblems of the normwise sliced solver, sliced.hip - k <= 96 columns each, side by side): W (N x N, ld N, zero
// outside the blocks: cleared by the caller) receives the blocks' eigenvector matrices on its diagonal, lam (N) the
// eigenvalues' magnitudes in the order of W's columns, sweeps_done the largest sweep count (cleared by the caller).
struct JmBlocks {
    int32_t n;
    int32_t start[32], k[32];
};
template <int RL>
__global__ __launch_bounds__(1024) void k_jacobi_mid_blocks(const double* __restrict__ T, int N, JmBlocks m, double* __restrict__ W,
                                                            double* __restrict__ lam, double tol_scale, double nfloor_scale,
                                                            int max_sweeps, int* __restrict__ sweeps_done) {
    extern __shared__ __attribute__((aligned(16))) double jm_sm[];
    const int c0 = m.start[blockIdx.x], k = m.k[blockIdx.x];
    if (k < 1) return;
    const double eps0 = 2.220446049250313e-16;
    const double tol = tol_scale * 2.0 * eps0 * sqrt((double)k);
    jacobi_mid_body<true, true, true, RL>(jm_sm, T + (int64_t)c0 * (N + 1), N, nullptr, W + (int64_t)c0 * (N + 1), N, lam ? lam + c0 : nullptr, k,
                                      tol, nfloor_scale * (double)k * eps0, max_sweeps, sweeps_done);
}


// ---- register-resident block pair (no eigenvector accumulation, 64 <= N <= 1024): the accurate route's Jacobi ---------------
// k_jacobi_round keeps its 2b columns in LDS and gives every column pair to one wave: a rotation is a chain of LDS round
// trips (load both columns, dot product, wave reduction, parameters, rotate, store: ~3 us), and the full 2b-player
// tournament of every visit rotates the pairs INSIDE a block again and again (961 pair rounds per sweep at N = 512 where
// 511 cover all pairs): 36 ms for the 12 sweeps a flat bulk of singular values needs.  Here the 32 columns of the block
// pair live in REGISTERS - lane = row (NW waves cover the rows), c[0..15] = block p, c[16..31] = block q - so a rotation is
// four FMAs per lane on registers; what crosses lanes is only the 16 dot products of a round (one 16-value butterfly per
// wave, partial sums through LDS, one barrier per round) and the 16 rotation parameters, which every wave works out for
// itself (same numbers, same bits).  The schedule keeps the pairs where they are and rotates the DATA through fixed
// register positions, so one compiled round serves all of them:
//   cross rounds: pairs (c[i], c[16 + i]); after each round block q's registers rotate by one - 16 rounds, all 256 cross pairs;
//   intra rounds (only in outer round 0 of a sweep, where every block takes part exactly once): pairs (c[k], c[15 - k]) and
//     (c[16 + k], c[31 - k]); positions 1..15 of each block rotate, position 0 stays - 15 rounds (the circle method).
// Both rotations are cyclic and complete, so the registers are back in column order when the schedule ends.
// 31 x 16 + 15 = 511 pair rounds per sweep, ~0.6 us each.
template <int NW>
struct JrShared {
    double part[2][NW][16];   // per-wave partial dot products of the 16 pairs (double-buffered by round parity)
    double nrm[NW][32];       // squared column norms by column position, each wave's own copy (set-up, schedule change)
    double npart[NW][32];
};

// totals of v[0..15] over the wave: lanes 4 I .. 4 I + 3 end up with the total of value I (17 exchanges instead of 96)
// (a, b) -> a' + b' after v_permlane32_swap / v_permlane16_swap (gfx950): the upper half (odd rows) of `a` changes places with
// the lower half (even rows) of `b`, so that lanes 0-31 (even rows) end up with both halves' a and the others with both
// halves' b - the first two butterfly stages without the LDS crossbar ds_bpermute goes through, and without the two selects
__device__ __forceinline__ double jr_swap_add32(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double jr_swap_add16(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double jr_reduce16(double (&v)[16], int lane) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = jr_swap_add32(v[k], v[k + 8]);   // lanes < 32 keep value k, the others value k + 8
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = jr_swap_add16(v[k], v[k + 4]);   // even rows keep value k, odd rows value k + 4
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const bool up = (lane & 8) != 0;
        const double send = up ? v[k] : v[k + 2], keep = up ? v[k + 2] : v[k];
        v[k] = keep + dpp_perm<0x128>(send);   // row_ror:8 = lane ^ 8 within the row of 16
    }
    {
        const bool up = (lane & 4) != 0;
        const double send = up ? v[0] : v[1], keep = up ? v[1] : v[0];
        v[0] = keep + __shfl_xor(send, 4, 64);
    }
    v[0] += dpp_perm<0x4E>(v[0]);   // quad_perm [2,3,0,1]
    v[0] += dpp_perm<0xB1>(v[0]);   // quad_perm [1,0,3,2]
    return v[0];
}

// 1 / sqrt(x) for x > 0: the hardware estimate (v_rsq_f64, ~2^-26) and two Newton steps y <- y (3/2 - x y^2 / 2)
__device__ __forceinline__ double jr_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * __builtin_fma(-hx * y, y, 1.5);
    y = y * __builtin_fma(-hx * y, y, 1.5);
    return y;
}

// one round: rotate the 16 pairs (c[P1(k)], c[P2(k)]) given by the position maps of the schedule.  a, bb: squared norms of
// the two members of "this lane's" pair I = lane >> 2 (the same numbers in the four lanes of a group and in every wave).
// (RPL rows per lane: a wave covers 64 RPL rows, so half as many waves meet at the barrier and add up half as many partials)
template <int NW, bool CROSS, int RPL>
__device__ __forceinline__ void jr_round(double (&c)[RPL][32], JrShared<NW>& sh, int w, int lane, int par, double tol2, double floor2,
                                         double& a, double& bb, unsigned int& my_rot, double& my_tmax) {
    // position of the two members of pair k
    auto p1 = [](int k) { return CROSS ? k : (k < 8 ? k : 16 + (k - 8)); };
    auto p2 = [](int k) { return CROSS ? 16 + k : (k < 8 ? 15 - k : 31 - (k - 8)); };
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        v[k] = c[0][p1(k)] * c[0][p2(k)];
#pragma unroll
        for (int q = 1; q < RPL; ++q) v[k] = __builtin_fma(c[q][p1(k)], c[q][p2(k)], v[k]);
    }
    const double tot = jr_reduce16(v, lane);
    const int I = lane >> 2;   // lanes 4 I .. 4 I + 3 hold the wave's partial of pair I:  I = b5 8 + b4 4 + b3 2 + b2
    double cc = 0.0;
    if constexpr (NW == 1) {
        // one wave covers all rows (the slices' own small problems, sliced.hip): the wave's sum IS the dot product - no LDS
        // exchange, no barrier in the round
        cc = tot;
    } else {
        if ((lane & 3) == 0) sh.part[par][w][I] = tot;
        __syncthreads();
        // every wave adds the partials in wave order and works out the rotation of "its lane's" pair: the same bits everywhere
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) cc += sh.part[par][ww][I];
    }
    const double mn = a < bb ? a : bb;
    double cs = 1.0, sn = 0.0;
    if (cc * cc > tol2 * a * bb && mn > floor2) {
        // The rotation from two reciprocal square roots and no division (round 6; the chain sqrt - divide - sqrt - divide of the
        // textbook form was most of a round's latency): with h = sqrt(d^2 + 4 cc^2), cos 2theta = |d| / h, sin 2theta = sgn(d) 2 cc / h,
        //   c = sqrt((1 + cos 2theta) / 2),   s = sin 2theta / (2 c),   t = s / c = 2 cc sgn(d) / (|d| + h)   (the same angle)
        // v_rsq_f64 + two Newton steps: c^2 + s^2 = 1 to a few ulp.
        const double d = bb - a;
        const double r = jr_rsqrt(d * d + 4.0 * cc * cc);   // 1 / h
        const double c2 = 0.5 + 0.5 * (fabs(d) * r);         // c^2, in [1/2, 1]
        const double ic = jr_rsqrt(c2);                      // 1 / c
        cs = c2 * ic;
        sn = (d >= 0.0 ? cc : -cc) * r * ic;
        const double t = sn * ic;
        const double na = a - t * cc, nb = bb + t * cc;
        a = na > 0.0 ? na : 0.0;
        bb = nb > 0.0 ? nb : 0.0;
        if (w == 0 && (lane & 3) == 0) ++my_rot;
        const double at = fabs(t);
        my_tmax = at > my_tmax ? at : my_tmax;   // (largest rotation of the sweep: quadratic convergence lets the host stop early)
    }
    // the parameters of pair k sit in lanes 4 k ..: v_readlane hands them to every lane as scalars (no LDS round trip)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double ck = read_lane(cs, 4 * k), sk = read_lane(sn, 4 * k);
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            const double x = c[q][p1(k)], y = c[q][p2(k)];
            c[q][p1(k)] = ck * x - sk * y;
            c[q][p2(k)] = sk * x + ck * y;
        }
    }
}

// end of a sweep (one thread): the sweep counts as done and, when it rotated nothing - or nothing larger than |tan| = 1e-8:
// Jacobi converges quadratically, what such a sweep leaves behind is below the rotation threshold - the `done` flag stops every
// kernel of the sweeps the host has queued ahead (it queues a few at a time and reads the state once per batch).
// rot_count: [0] rotations, [4..6) largest |tan| as bits (both cleared here for the next sweep); state: [0] done, [1] sweeps
__global__ void k_jr_sweep_end(unsigned int* __restrict__ rot_count, int* __restrict__ state) {
    if (state[0]) return;
    const unsigned int nrot = rot_count[0];
    const double tmax = __longlong_as_double((long long)*reinterpret_cast<unsigned long long*>(rot_count + 4));
    state[1] += 1;
    if (nrot == 0 || tmax < 1e-8) state[0] = 1;
    rot_count[0] = 0;
    *reinterpret_cast<unsigned long long*>(rot_count + 4) = 0ull;
}

// tab (optional): the block pairs of this launch from a table instead of the round-robin schedule over all blocks - entry
// blockIdx.x = (first column, columns) of the two blocks (at most 16 columns each; a first column < 0: nothing to do) and the
// window of rows [row0, row0 + nrows) their columns are non-zero in (the workgroup's waves cover nrows, not N).  The
// spectrum slicer (sliced.hip) sweeps within groups of columns this way; `round` == 0 still means "with the pairs inside
// each block".
struct JrPair {
    int cp0, lp, cq0, lq, row0, nrows, pad0, pad1;
};
template <int NW, int RPL>
__global__ __launch_bounds__(64 * NW) void k_jacobi_reg(double* __restrict__ B, int N, int nblk, int round, double tol,
                                                        const double* __restrict__ params,
                                                        unsigned int* __restrict__ rot_count, const int* __restrict__ done,
                                                        const JrPair* __restrict__ tab) {
    if (*done) return;   // (converged in a sweep the host has not heard of yet)
    __shared__ JrShared<NW> sh;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int row = (w * 64 + lane) * RPL;   // this lane's rows: row .. row + RPL - 1
    int cp0, lp, cq0, lq;   // first column and number of columns of the two blocks
    int nrows = N;          // rows the columns live in: [row0, row0 + nrows) (table form: the block's own window; everything else is zero)
    if (tab) {
        const JrPair e = tab[blockIdx.x];
        if (e.cp0 < 0) return;
        cp0 = e.cp0;
        lp = e.lp;
        cq0 = e.cq0;
        lq = e.lq;
        nrows = e.nrows;
        B += e.row0;        // (column offsets below are in units of N: the window only shifts the row origin)
    } else {
        int bp, bq;
        rr_pair(nblk, round, blockIdx.x, bp, bq);
        if (bp > bq) {   // (keeps the lower block in c[0..15]: the order of the columns inside a pair is that of the LDS kernel)
            const int t = bp;
            bp = bq;
            bq = t;
        }
        cp0 = bp * 16;
        cq0 = bq * 16;
        lp = N - cp0 < 16 ? (N - cp0 > 0 ? N - cp0 : 0) : 16;
        lq = N - cq0 < 16 ? (N - cq0 > 0 ? N - cq0 : 0) : 16;
    }
    const double floor2 = params[0], tol2 = tol * tol;
    double c[RPL][32];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const bool have = (s < 16) ? s < lp : (s - 16) < lq;
        const int col = (s < 16) ? cp0 + s : cq0 + (s - 16);
#pragma unroll
        for (int q = 0; q < RPL; ++q) c[q][s] = (have && row + q < nrows) ? B[(size_t)col * N + row + q] : 0.0;
    }
    // squared column norms (32-value reduction as two butterflies), each wave keeps a full copy by column position
    {
        double v[16];
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                v[k] = c[0][16 * hlf + k] * c[0][16 * hlf + k];
#pragma unroll
                for (int q = 1; q < RPL; ++q) v[k] = __builtin_fma(c[q][16 * hlf + k], c[q][16 * hlf + k], v[k]);
            }
            const double tot = jr_reduce16(v, lane);
            if ((lane & 3) == 0) sh.npart[w][16 * hlf + (lane >> 2)] = tot;
        }
        __syncthreads();
        if (lane < 32) {
            double t = 0.0;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) t += sh.npart[ww][lane];
            sh.nrm[w][lane] = t;
        }
        __syncthreads();
    }
    unsigned int my_rot = 0;
    double my_tmax = 0.0;
    int par = 0;
    const int I = lane >> 2;
    if (round == 0) {
        // the pairs inside each of the two blocks: 15 rounds of the circle method, both blocks at once.  Lane group g of a
        // block (g = I & 7; block = I >> 3) holds the norms of positions g and 15 - g.
        const int g = I & 7, base = (I >> 3) * 16;
        double a = sh.nrm[w][base + g], bb = sh.nrm[w][base + 15 - g];
        for (int r = 0; r < 15; ++r) {
            jr_round<NW, false, RPL>(c, sh, w, lane, par, tol2, floor2, a, bb, my_rot, my_tmax);
            par ^= 1;
            // positions 1..15 of each block move on by one (15 -> 1), position 0 stays; the norms move with them:
            // a_g <- a_(g-1) (g >= 2), a_1 <- bb_0, a_0 stays;  bb_g <- bb_(g+1) (g <= 6), bb_7 <- a_7
            const double a_prev = __shfl(a, (lane - 4) & 63, 64), bb_next = __shfl(bb, (lane + 4) & 63, 64);
            const double bb_0 = __shfl(bb, lane & 32, 64), a_7 = __shfl(a, (lane & 32) + 28, 64);
            a = (g == 0) ? a : (g == 1 ? bb_0 : a_prev);
            bb = (g == 7) ? a_7 : bb_next;
#pragma unroll
            for (int q = 0; q < RPL; ++q)
#pragma unroll
                for (int hb = 0; hb < 32; hb += 16) {
                    const double last = c[q][hb + 15];
#pragma unroll
                    for (int k = 15; k >= 2; --k) c[q][hb + k] = c[q][hb + k - 1];
                    c[q][hb + 1] = last;
                }
        }
        // (15 rotations: every column is back in its position) - the norms go back to the position table
        if ((lane & 3) == 0) {
            sh.nrm[w][base + g] = a;
            sh.nrm[w][base + 15 - g] = bb;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    {
        double a = sh.nrm[w][I], bb = sh.nrm[w][16 + I];
        for (int r = 0; r < 16; ++r) {
            jr_round<NW, true, RPL>(c, sh, w, lane, par, tol2, floor2, a, bb, my_rot, my_tmax);
            par ^= 1;
            // block q's registers move on by one (position 16 + i takes what 16 + i + 1 held): pair i meets q's next column
            bb = __shfl(bb, (lane + 4) & 63, 64);
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
                const double first = c[q][16];
#pragma unroll
                for (int k = 16; k < 31; ++k) c[q][k] = c[q][k + 1];
                c[q][31] = first;
            }
        }
    }
    // (16 cross rotations and 15 intra rotations are full cycles: the registers are in column order again)
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const bool have = (s < 16) ? s < lp : (s - 16) < lq;
        const int col = (s < 16) ? cp0 + s : cq0 + (s - 16);
#pragma unroll
        for (int q = 0; q < RPL; ++q)
            if (have && row + q < nrows) B[(size_t)col * N + row + q] = c[q][s];
    }
    if (w == 0) {
        // lanes 4 I of wave 0 counted the rotations of "their" pairs
        unsigned int t = my_rot;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if (lane == 0 && t) atomicAdd(rot_count, t);
        // [rot_count + 4 .. +12): the largest |tan| of a rotation in this sweep, as the bit pattern of a non-negative double
        double m = my_tmax;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(m, off, 64);
            m = o > m ? o : m;
        }
        if (lane == 0 && m > 0.0)
            atomicMax(reinterpret_cast<unsigned long long*>(rot_count + 4), (unsigned long long)__double_as_longlong(m));
    }
}

// B = G (ld -> N), V = I
__global__ __launch_bounds__(256) void k_jacobi_init(const double* __restrict__ G, int64_t ldG,
                                                     double* __restrict__ B, double* __restrict__ V,
                                                     int N, int want_v) {
    const int64_t total = (int64_t)N * N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t r = e % N, c = e / N;
        B[e] = G[r + c * ldG];
        if (want_v) V[e] = (r == c) ? 1.0 : 0.0;
    }
}

// params[0] = (N*eps)^2 * ||G||_F^2 ; single workgroup, fixed reduction order
__global__ __launch_bounds__(1024) void k_fro_floor(const double* __restrict__ B, int N,
                                                    double* __restrict__ params, double factor2) {
    __shared__ double sw[16];
    const int64_t total = (int64_t)N * N;
    double s = 0.0;
    for (int64_t e = threadIdx.x; e < total; e += 1024) {
        const double v = B[e];
        s += v * v;
    }
    s = wave_allsum(s);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += sw[k];
        params[0] = factor2 * t;
        params[1] = t;
    }
}

// lam[i] = ||B[:,i]||_2, one wave per column
__global__ __launch_bounds__(256) void k_colnorm(const double* __restrict__ B, int N,
                                                 double* __restrict__ lam) {
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= N) return;
    double s = 0.0;
    for (int row = lane; row < N; row += 64) {
        const double v = B[(size_t)col * N + row];
        s += v * v;
    }
    s = wave_allsum(s);
    if (lane == 0) lam[col] = sqrt(s);
}

__global__ __launch_bounds__(256) void k_gather_scale(const double* __restrict__ V, int N,
                                                      const int32_t* __restrict__ sel,
                                                      const double* __restrict__ g, int r,
                                                      double* __restrict__ Vg,
                                                      double* __restrict__ Vs) {
    const int64_t total = (int64_t)N * r;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t row = e % N, p = e / N;
        const double v = V[(int64_t)sel[p] * N + row];
        if (Vs) Vs[e] = v;
        if (Vg) Vg[e] = g[p] * v;
    }
}

// the same with the (short) selection and weights passed as kernel arguments: no upload, no extra command
__global__ __launch_bounds__(256) void k_gather_scale_arg(const double* __restrict__ V, int N, SelWeights sw, int r,
                                                          double* __restrict__ Vg, double* __restrict__ Vs) {
    const int64_t total = (int64_t)N * r;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t row = e % N, p = e / N;
        const double v = V[(int64_t)sw.sel[p] * N + row];
        if (Vs) Vs[e] = v;
        if (Vg) Vg[e] = sw.w[p] * v;
    }
}

static int pick_block(int64_t N, bool want_v, bool* single) {
    // 2b columns of B (and of V) resident in LDS; leave headroom below 160 KiB
    const int64_t budget = 150 * 1024;
    const int64_t per_col = N * 8 * (want_v ? 2 : 1);
    const int64_t half = (N + 1) / 2;
    *single = false;
    if (2 * half * per_col <= budget) {  // the whole matrix is one block pair: one workgroup, no host loop
        *single = true;
        return (int)half;
    }
    int64_t b = budget / (2 * per_col);
    const int64_t bcap = want_v ? 8 : 16;   // waves per workgroup (1024 threads at most)
    if (b > bcap) b = bcap;
    return (int)b;
}

// the diagonal blocks [start_j, start_j + k_j) of the symmetric T (N x N, ld N): W = blockdiag(eigenvector matrices) (the rest of
// W is cleared here), lam (N) the eigenvalue magnitudes in the order of W's columns; at most 32 blocks of at most 96 columns
int jacobi_mid_blocks_f64(Handle* h, const double* T, int64_t N, const std::vector<std::pair<int, int>>& blocks, double* W, double* lam,
                          int64_t* sweeps_out) {
    if (blocks.empty() || blocks.size() > 32) return set_err(h, TLSQ_ERR_ARG, "jacobi_mid_blocks: %zu blocks", blocks.size());
    JmBlocks m = {};
    m.n = (int32_t)blocks.size();
    int maxk = 0;
    for (size_t j = 0; j < blocks.size(); ++j) {
        m.start[j] = blocks[j].first;
        m.k[j] = blocks[j].second;
        maxk = std::max(maxk, blocks[j].second);
        if (blocks[j].second > 96 || blocks[j].first < 0 || blocks[j].first + blocks[j].second > N)
            return set_err(h, TLSQ_ERR_ARG, "jacobi_mid_blocks: block %zu = [%d, +%d)", j, blocks[j].first, blocks[j].second);
    }
    void* scal;
    TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &scal));
    int* sweeps_dev = reinterpret_cast<int*>(reinterpret_cast<char*>(scal) + 136);
    TLSQ_HIP(h, hipMemsetAsync(W, 0, (size_t)N * N * 8, h->stream));
    TLSQ_HIP(h, hipMemsetAsync(sweeps_dev, 0, 4, h->stream));
    const size_t lds = ((size_t)2 * maxk * (maxk + 1) + maxk) * 8;
    const int max_sweeps = 40;
    // (rows per lane by the largest block: the lanes of a column pair walk 16 RL rows whatever the block has)
#define JMB_LAUNCH(RLV)                                                                                                              \
    do {                                                                                                                             \
        TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_jacobi_mid_blocks<RLV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lds));                                                                                  \
        hipLaunchKernelGGL(k_jacobi_mid_blocks<RLV>, dim3((unsigned)blocks.size()), dim3(1024), lds, h->stream, T, (int)N, m, W, lam, 1.0, 1.0, \
                           max_sweeps, sweeps_dev);                                                                                  \
    } while (0)
    if (maxk <= 64) JMB_LAUNCH(4);
    else if (maxk <= 80) JMB_LAUNCH(5);
    else JMB_LAUNCH(6);
#undef JMB_LAUNCH
    TLSQ_HIP(h, hipGetLastError());
    TLSQ_HIP(h, hipMemcpyAsync(h->pinned, sweeps_dev, 4, hipMemcpyDeviceToHost, h->stream));
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    int sd;
    memcpy(&sd, h->pinned, 4);
    if (sweeps_out) *sweeps_out = sd;
    if (sd >= max_sweeps) return TLSQ_ERR_NOCONV;   // (the caller falls back: no error text)
    return TLSQ_OK;
}

int symeig_f64(Handle* h, const double* G, int64_t N, int64_t ldG, double* B, double* V, bool want_v,
               double* lam_dev, int64_t* sweeps_out, bool async_small, bool warm_v, bool two_sided, double rot_tol) {
    if (sweeps_out) *sweeps_out = 0;
    if (N <= 0) return TLSQ_OK;
    void* scal;
    TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &scal));
    double* params = reinterpret_cast<double*>(reinterpret_cast<char*>(scal) + 64);
    unsigned int* rot = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(scal) + 128);
    int* sweeps_dev = reinterpret_cast<int*>(reinterpret_cast<char*>(scal) + 136);

    if (N >= 2 && N <= 64) {
        // one launch: init, noise floor, sweeps, column norms
        const double eps0 = 2.220446049250313e-16;
        double tol0 = 2.0 * eps0 * sqrt((double)N);
        if (tol0 < 4.0 * eps0) tol0 = 4.0 * eps0;
        if (rot_tol > tol0) tol0 = rot_tol;
        const int max_sweeps0 = 40;
        // one half-wave per column pair; whole waves only
        const int npair0 = (int)((N + 1) / 2);
        const int nthr0 = ((npair0 * 32 + 63) / 64) * 64;
        const bool no_two = dev_is(DEV_JACOBI2, '0');
        if (want_v && two_sided && !no_two)
            hipLaunchKernelGGL(k_jacobi2_small<true>, dim3(1), dim3(nthr0), 0, h->stream, G, ldG, B, V, lam_dev,
                               (int)N, tol0, (double)N * eps0, max_sweeps0, sweeps_dev);
        else if (want_v)
            hipLaunchKernelGGL(k_jacobi_small<true>, dim3(1), dim3(nthr0), 0, h->stream, G, ldG, B, V, lam_dev,
                               (int)N, tol0, (double)N * eps0, max_sweeps0, sweeps_dev);
        else
            hipLaunchKernelGGL(k_jacobi_small<false>, dim3(1), dim3(nthr0), 0, h->stream, G, ldG, B, V, lam_dev,
                               (int)N, tol0, (double)N * eps0, max_sweeps0, sweeps_dev);
        TLSQ_HIP(h, hipGetLastError());
        if (async_small) return TLSQ_OK;   // the caller validates the result itself (residuals)
        TLSQ_HIP(h, hipMemcpyAsync(h->pinned, sweeps_dev, 4, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        int sd;
        memcpy(&sd, h->pinned, 4);
        if (sweeps_out) *sweeps_out = sd;
        if (sd >= max_sweeps0)
            return set_err(h, TLSQ_ERR_NOCONV, "Jacobi eigensolver did not converge in %d sweeps (N=%lld)",
                           max_sweeps0, (long long)N);
        return TLSQ_OK;
    }
    if (N > 64 && N <= 96 && !(warm_v && want_v)) {
        // one launch as well: init, noise floor, sweeps, column norms (k_jacobi_mid)
        const double eps0 = 2.220446049250313e-16;
        const double tol0 = std::max(2.0 * eps0 * sqrt((double)N), rot_tol);
        const int max_sweeps0 = 40;
        const size_t lds = ((size_t)(want_v ? 2 : 1) * N * (N + 1) + N) * 8;
        // (rows per lane: 5 up to 80 columns, 6 up to 96 - the lanes of a column pair walk 16 RL rows whatever N is)
#define JM_LAUNCH(WV, RLV)                                                                                                          \
    do {                                                                                                                             \
        TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_jacobi_mid<WV, RLV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lds));                                                                                  \
        hipLaunchKernelGGL((k_jacobi_mid<WV, RLV>), dim3(1), dim3(1024), lds, h->stream, G, ldG, B, V, lam_dev, (int)N, tol0,        \
                           (double)N * eps0, max_sweeps0, sweeps_dev);                                                               \
    } while (0)
        if (want_v) {
            if (N <= 80) JM_LAUNCH(true, 5);
            else JM_LAUNCH(true, 6);
        } else {
            if (N <= 80) JM_LAUNCH(false, 5);
            else JM_LAUNCH(false, 6);
        }
#undef JM_LAUNCH
        TLSQ_HIP(h, hipGetLastError());
        if (async_small) return TLSQ_OK;
        TLSQ_HIP(h, hipMemcpyAsync(h->pinned, sweeps_dev, 4, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        int sd;
        memcpy(&sd, h->pinned, 4);
        if (sweeps_out) *sweeps_out = sd;
        if (sd >= max_sweeps0)
            return set_err(h, TLSQ_ERR_NOCONV, "Jacobi eigensolver did not converge in %d sweeps (N=%lld)", max_sweeps0,
                           (long long)N);
        return TLSQ_OK;
    }
    int64_t g = (N * N + 255) / 256;
    if (g > 1024) g = 1024;
    if (warm_v && want_v && N > 64) {
        // warm start: V already holds an orthogonal matrix that nearly diagonalises G (the previous ALM
        // iteration's eigenvectors): iterate on B = G*V instead of B = G, V = I
        TLSQ_TRY(gemm_f64(h, true, false, (const double*)V, N, G, ldG, B, N, N, N, N, false));
    } else {
        hipLaunchKernelGGL(k_jacobi_init, dim3((int)g), dim3(256), 0, h->stream, G, ldG, B, V, (int)N,
                           want_v ? 1 : 0);
        TLSQ_HIP(h, hipGetLastError());
    }
    if (N == 1) {
        hipLaunchKernelGGL(k_colnorm, dim3(1), dim3(256), 0, h->stream, (const double*)B, 1, lam_dev);
        TLSQ_HIP(h, hipGetLastError());
        return TLSQ_OK;
    }
    const double eps = 2.220446049250313e-16;
    const double nf = (double)N * eps;
    hipLaunchKernelGGL(k_fro_floor, dim3(1), dim3(1024), 0, h->stream, (const double*)B, (int)N, params,
                       nf * nf);
    TLSQ_HIP(h, hipGetLastError());

    bool single = false;
    const int b = pick_block(N, want_v, &single);
    if (b < 1)
        return set_err(h, TLSQ_ERR_UNSUPPORTED,
                       "N=%lld too large for the LDS-resident Jacobi eigensolver", (long long)N);
    int nblk = (int)((N + b - 1) / b);
    if (nblk & 1) ++nblk;
    if (nblk < 2) nblk = 2;
    const size_t lds = (size_t)2 * b * N * 8 * (want_v ? 2 : 1) + 16 + (size_t)2 * b * 8;
    const int nwaves = b < 16 ? b : 16;
    if (want_v) {
        TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_jacobi_round<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    } else {
        TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_jacobi_round<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    double tol = 2.0 * eps * sqrt((double)N);
    if (tol < 4.0 * eps) tol = 4.0 * eps;
    const int max_sweeps = 40;
    int sweep = 0;
    bool converged = false;
    if (single && nblk == 2) {
        TLSQ_HIP(h, hipMemsetAsync(rot, 0, 16, h->stream));
        if (want_v)
            hipLaunchKernelGGL(k_jacobi_round<true>, dim3(1), dim3(64 * nwaves), lds, h->stream, B, V, (int)N,
                               b, nblk, 0, tol, (const double*)params, rot, max_sweeps, sweeps_dev);
        else
            hipLaunchKernelGGL(k_jacobi_round<false>, dim3(1), dim3(64 * nwaves), lds, h->stream, B, V, (int)N,
                               b, nblk, 0, tol, (const double*)params, rot, max_sweeps, sweeps_dev);
        TLSQ_HIP(h, hipGetLastError());
        if (async_small) {
            // caller does not want a host round trip here: it validates the result itself (residuals)
            sweep = 0;
            converged = true;
        } else {
            TLSQ_HIP(h, hipMemcpyAsync(h->pinned, sweeps_dev, 4, hipMemcpyDeviceToHost, h->stream));
            TLSQ_HIP(h, hipStreamSynchronize(h->stream));
            int sd;
            memcpy(&sd, h->pinned, 4);
            sweep = sd;
            converged = sd < max_sweeps;
        }
    } else {
        for (; sweep < max_sweeps; ++sweep) {
            TLSQ_HIP(h, hipMemsetAsync(rot, 0, 4, h->stream));
            for (int r = 0; r < nblk - 1; ++r) {
                if (want_v)
                    hipLaunchKernelGGL(k_jacobi_round<true>, dim3(nblk / 2), dim3(64 * nwaves), lds, h->stream,
                                       B, V, (int)N, b, nblk, r, tol, (const double*)params, rot, 1,
                                       (int*)nullptr);
                else
                    hipLaunchKernelGGL(k_jacobi_round<false>, dim3(nblk / 2), dim3(64 * nwaves), lds,
                                       h->stream, B, V, (int)N, b, nblk, r, tol, (const double*)params, rot, 1,
                                       (int*)nullptr);
            }
            TLSQ_HIP(h, hipGetLastError());
            TLSQ_HIP(h, hipMemcpyAsync(h->pinned, rot, 4, hipMemcpyDeviceToHost, h->stream));
            TLSQ_HIP(h, hipStreamSynchronize(h->stream));
            unsigned int nrot;
            memcpy(&nrot, h->pinned, 4);
            if (nrot == 0) {
                ++sweep;
                converged = true;
                break;
            }
        }
    }
    if (sweeps_out) *sweeps_out = sweep;
    hipLaunchKernelGGL(k_colnorm, dim3((int)((N + 3) / 4)), dim3(256), 0, h->stream, (const double*)B,
                       (int)N, lam_dev);
    TLSQ_HIP(h, hipGetLastError());
    if (!converged)
        return set_err(h, TLSQ_ERR_NOCONV, "Jacobi eigensolver did not converge in %d sweeps (N=%lld)",
                       max_sweeps, (long long)N);
    return TLSQ_OK;
}

// Sweeps of the register-resident kernel until one rotates nothing (or nothing above |tan| = 1e-8).  tab_dev == nullptr: the
// round-robin schedule over all nblk blocks (nblk - 1 launches per sweep); otherwise `nrounds` launches per sweep, launch r taking
// round_cnt[r] block pairs from tab_dev + round_off[r] (k_jacobi_reg's table form; r == 0 rotates inside the blocks as well).
// state (device): [0] done, [1] sweeps performed; the host queues four sweeps at a time (kernels of sweeps behind the converged one
// return at once) and reads the state once per batch instead of once per sweep.  rot: rotation count, largest |tan|.
static int jr_run_sweeps(Handle* h, double* B, int64_t N, double tol_r, const double* params, unsigned int* rot, int* state,
                         const JrPair* tab_dev, const int* round_off, const int* round_cnt, int nrounds, int max_sweeps, int* sweeps_done,
                         bool* converged_out, int64_t rows = 0) {   // rows: the largest row window of the table (0: N)
    int nblk = (int)((N + 15) / 16);
    if (nblk & 1) ++nblk;
    bool converged = false;
    int sweep = 0;
    TLSQ_HIP(h, hipMemsetAsync(rot, 0, 48, h->stream));   // rotation count, largest |tan|, sweeps_dev (unused here), state
    const bool one = dev_is(DEV_JACOBI_RPL, '1');   // (JACOBI_RPL=1: one row per lane, the first form: 8 / 16 waves)
    const int rounds = tab_dev ? nrounds : nblk - 1;
    const int64_t R = rows > 0 ? rows : N;   // rows a workgroup has to cover
    while (sweep < max_sweeps && !converged) {
        const int batch = std::min(4, max_sweeps - sweep);
        for (int sb = 0; sb < batch; ++sb) {
            for (int r = 0; r < rounds; ++r) {
                const int grid = tab_dev ? round_cnt[r] : nblk / 2;
                if (grid <= 0) continue;
                const JrPair* tab = tab_dev ? tab_dev + round_off[r] : nullptr;
                if (R <= 128 && !one)
                    hipLaunchKernelGGL((k_jacobi_reg<1, 2>), dim3(grid), dim3(64), 0, h->stream, B, (int)N, nblk, r, tol_r, params, rot,
                                       (const int*)state, tab);
                else if (R <= 256 && !one)
                    hipLaunchKernelGGL((k_jacobi_reg<2, 2>), dim3(grid), dim3(128), 0, h->stream, B, (int)N, nblk, r, tol_r, params, rot,
                                       (const int*)state, tab);
                else if (R <= 512 && !one)
                    hipLaunchKernelGGL((k_jacobi_reg<4, 2>), dim3(grid), dim3(256), 0, h->stream, B, (int)N, nblk, r, tol_r, params, rot,
                                       (const int*)state, tab);
                else if (R <= 512)
                    hipLaunchKernelGGL((k_jacobi_reg<8, 1>), dim3(grid), dim3(512), 0, h->stream, B, (int)N, nblk, r, tol_r, params, rot,
                                       (const int*)state, tab);
                else if (!one)
                    hipLaunchKernelGGL((k_jacobi_reg<8, 2>), dim3(grid), dim3(512), 0, h->stream, B, (int)N, nblk, r, tol_r, params, rot,
                                       (const int*)state, tab);
                else
                    hipLaunchKernelGGL((k_jacobi_reg<16, 1>), dim3(grid), dim3(1024), 0, h->stream, B, (int)N, nblk, r, tol_r, params, rot,
                                       (const int*)state, tab);
            }
            hipLaunchKernelGGL(k_jr_sweep_end, dim3(1), dim3(1), 0, h->stream, rot, state);
        }
        TLSQ_HIP(h, hipGetLastError());
        TLSQ_HIP(h, hipMemcpyAsync(h->pinned, state, 8, hipMemcpyDeviceToHost, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        int hs[2];
        memcpy(hs, h->pinned, 8);
        sweep = hs[1];
        converged = hs[0] != 0;
    }
    *sweeps_done = sweep;
    *converged_out = converged;
    return TLSQ_OK;
}

// One-sided Jacobi on the columns of a square factor B (N x N, ld N, rotated in place): on return the columns of B are
// mutually orthogonal, V (N x N, ld N) holds them normalised and sig_dev[i] their norms (unsorted).  For B = L with
// G = L L' (Cholesky factor, or R' of a QR factorisation Z = Q R) these are the eigenvectors of G / right singular
// vectors of Z and sqrt(lambda_i) / sigma_i — no eigenvector accumulation.  floor_rel: columns whose norm is below
// floor_rel * ||B||_F take no part in rotations (0: every non-zero column does).  Matrices that fit one block pair
// in LDS (N <= ~130) run all sweeps in a single launch.
int jacobi_factor_f64(Handle* h, double* B, int64_t N, double* V, double* sig_dev, double floor_rel,
                      int64_t* sweeps_out) {
    if (sweeps_out) *sweeps_out = 0;
    if (N <= 0) return TLSQ_OK;
    void* scal;
    TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &scal));
    double* params = reinterpret_cast<double*>(reinterpret_cast<char*>(scal) + 64);
    unsigned int* rot = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(scal) + 128);
    int* sweeps_dev = reinterpret_cast<int*>(reinterpret_cast<char*>(scal) + 136);
    const double eps = 2.220446049250313e-16;
    int sweep = 0;
    bool converged = true;
    if (N >= 2) {
        hipLaunchKernelGGL(k_fro_floor, dim3(1), dim3(1024), 0, h->stream, (const double*)B, (int)N, params,
                           floor_rel * floor_rel);
        TLSQ_HIP(h, hipGetLastError());
        const double tol_r = std::max(2.0 * eps * sqrt((double)N), 4.0 * eps);
        if (N >= 64 && N <= 1024 && !dev_is(DEV_NO_JACOBI_REG, '1')) {
            // register-resident block pairs (k_jacobi_reg): blocks of 16 columns, nblk - 1 launches per sweep
            int* state = reinterpret_cast<int*>(reinterpret_cast<char*>(scal) + 160);
            TLSQ_TRY(jr_run_sweeps(h, B, N, tol_r, params, rot, state, nullptr, nullptr, nullptr, 0, 40, &sweep, &converged));
            if (sweeps_out) *sweeps_out = sweep;
            TLSQ_TRY(launch_normalize_cols(h, (const double*)B, N, V, sig_dev));
            if (!converged)
                return set_err(h, TLSQ_ERR_NOCONV, "one-sided Jacobi did not converge in 40 sweeps (N=%lld)", (long long)N);
            return TLSQ_OK;
        }
        bool single = false;
        int b = pick_block(N, false, &single);
        if (!single && b > 16) b = 16;
        if (b < 1)
            return set_err(h, TLSQ_ERR_UNSUPPORTED, "N=%lld too large for the LDS-resident Jacobi eigensolver",
                           (long long)N);
        int nblk = (int)((N + b - 1) / b);
        if (nblk & 1) ++nblk;
        if (nblk < 2) nblk = 2;
        const size_t lds = (size_t)2 * b * N * 8 + 16 + (size_t)2 * b * 8;
        const int nwaves = b < 16 ? b : 16;
        TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_jacobi_round<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        double tol = 2.0 * eps * sqrt((double)N);
        if (tol < 4.0 * eps) tol = 4.0 * eps;
        const int max_sweeps = 40;
        converged = false;
        if (single && nblk == 2) {
            TLSQ_HIP(h, hipMemsetAsync(rot, 0, 16, h->stream));
            hipLaunchKernelGGL(k_jacobi_round<false>, dim3(1), dim3(64 * nwaves), lds, h->stream, B, (double*)nullptr,
                               (int)N, b, nblk, 0, tol, (const double*)params, rot, max_sweeps, sweeps_dev);
            TLSQ_HIP(h, hipGetLastError());
            TLSQ_HIP(h, hipMemcpyAsync(h->pinned, sweeps_dev, 4, hipMemcpyDeviceToHost, h->stream));
            TLSQ_HIP(h, hipStreamSynchronize(h->stream));
            int sd;
            memcpy(&sd, h->pinned, 4);
            sweep = sd;
            converged = sd < max_sweeps;
        } else {
            for (; sweep < max_sweeps; ++sweep) {
                TLSQ_HIP(h, hipMemsetAsync(rot, 0, 4, h->stream));
                for (int r = 0; r < nblk - 1; ++r)
                    hipLaunchKernelGGL(k_jacobi_round<false>, dim3(nblk / 2), dim3(64 * nwaves), lds, h->stream, B,
                                       (double*)nullptr, (int)N, b, nblk, r, tol, (const double*)params, rot, 1,
                                       (int*)nullptr);
                TLSQ_HIP(h, hipGetLastError());
                TLSQ_HIP(h, hipMemcpyAsync(h->pinned, rot, 4, hipMemcpyDeviceToHost, h->stream));
                TLSQ_HIP(h, hipStreamSynchronize(h->stream));
                unsigned int nrot;
                memcpy(&nrot, h->pinned, 4);
                if (nrot == 0) {
                    ++sweep;
                    converged = true;
                    break;
                }
            }
        }
    }
    if (sweeps_out) *sweeps_out = sweep;
    TLSQ_TRY(launch_normalize_cols(h, (const double*)B, N, V, sig_dev));
    if (!converged)
        return set_err(h, TLSQ_ERR_NOCONV, "one-sided Jacobi did not converge in 40 sweeps (N=%lld)", (long long)N);
    return TLSQ_OK;
}

// The same on a factor whose columns come in GROUPS that are already (nearly) orthogonal to each other - the spectrum slicer's
// start (sliced.hip): first sweeps that only pair columns of the same group (a group of k columns costs ceil(k / 16) - 1
// launches per sweep, and all groups share them), then sweeps over all pairs until nothing rotates.  groups: (first column,
// columns), disjoint, inside [0, N).  sweeps_out[0] / [1]: sweeps of the two phases.
// block_diagonal: B is block diagonal - group g's columns are non-zero in rows [first, first + columns) only (the slices' own
// small factors side by side): the workgroups only cover that window, and the sweeps over all pairs are not needed (nothing
// couples the groups).
int jacobi_factor_grouped_f64(Handle* h, double* B, int64_t N, const std::vector<std::pair<int, int>>& groups, double* V,
                              double* sig_dev, double floor_rel, int64_t* sweeps_out, bool block_diagonal) {
    if (sweeps_out) sweeps_out[0] = sweeps_out[1] = 0;
    if (N < 64 || N > 1024) return set_err(h, TLSQ_ERR_ARG, "jacobi_factor_grouped_f64: N = %lld", (long long)N);
    void* scal;
    TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &scal));
    double* params = reinterpret_cast<double*>(reinterpret_cast<char*>(scal) + 64);
    unsigned int* rot = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(scal) + 128);
    int* state = reinterpret_cast<int*>(reinterpret_cast<char*>(scal) + 160);
    const double eps = 2.220446049250313e-16;
    hipLaunchKernelGGL(k_fro_floor, dim3(1), dim3(1024), 0, h->stream, (const double*)B, (int)N, params, floor_rel * floor_rel);
    TLSQ_HIP(h, hipGetLastError());
    const double tol_r = std::max(2.0 * eps * sqrt((double)N), 4.0 * eps);
    // the table: per launch r of a sweep, the block pairs of every group that still has a round r
    std::vector<JrPair> tab;
    std::vector<int> off, cnt;
    int maxr = 0;
    int64_t maxrows = 0;
    for (const auto& g : groups) maxrows = std::max<int64_t>(maxrows, g.second);
    for (const auto& g : groups) {
        const int nb = (g.second + 15) / 16;
        const int nbe = std::max(2, nb + (nb & 1));
        if (g.second > 1) maxr = std::max(maxr, nbe - 1);
    }
    for (int r = 0; r < maxr; ++r) {
        off.push_back((int)tab.size());
        for (const auto& g : groups) {
            if (g.second <= 1) continue;
            const int nb = (g.second + 15) / 16;
            const int nbe = std::max(2, nb + (nb & 1));
            if (r >= nbe - 1) continue;
            for (int idx = 0; idx < nbe / 2; ++idx) {
                const int m = nbe - 1;
                int bp = idx == 0 ? r % m : (r + idx) % m, bq = idx == 0 ? m : (r - idx + m) % m;
                if (bp > bq) std::swap(bp, bq);
                auto len = [&](int b) { return std::max(0, std::min(16, g.second - 16 * b)); };
                const int lp = len(bp), lq = len(bq);
                if (lp + lq == 0) continue;
                if ((lp == 0 || lq == 0) && r != 0) continue;   // (a lone block only has work in the round with the inner pairs)
                const int row0 = block_diagonal ? g.first : 0, nrows = block_diagonal ? g.second : (int)N;
                if (lp == 0) {   // keep the real block in the first slot
                    tab.push_back(JrPair{g.first + 16 * bq, lq, g.first, 0, row0, nrows, 0, 0});
                } else {
                    tab.push_back(JrPair{g.first + 16 * bp, lp, g.first + 16 * bq, lq, row0, nrows, 0, 0});
                }
            }
        }
        cnt.push_back((int)tab.size() - off.back());
    }
    int sw0 = 0, sw1 = 0;
    bool conv0 = true, conv1 = false;
    if (!tab.empty()) {
        void* tdev;
        TLSQ_TRY(ws_get(h, WS_SL_TAB, tab.size() * sizeof(JrPair), &tdev));
        TLSQ_HIP(h, hipMemcpyAsync(tdev, tab.data(), tab.size() * sizeof(JrPair), hipMemcpyHostToDevice, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));   // (tab is a local: the copy must have left it)
        TLSQ_TRY(jr_run_sweeps(h, B, N, tol_r, params, rot, state, (const JrPair*)tdev, off.data(), cnt.data(), maxr, 40, &sw0, &conv0,
                               block_diagonal ? maxrows : 0));
    }
    if (block_diagonal) conv1 = conv0;
    else TLSQ_TRY(jr_run_sweeps(h, B, N, tol_r, params, rot, state, nullptr, nullptr, nullptr, 0, 40, &sw1, &conv1));
    if (sweeps_out) {
        sweeps_out[0] = sw0;
        sweeps_out[1] = sw1;
    }
    TLSQ_TRY(launch_normalize_cols(h, (const double*)B, N, V, sig_dev));
    if (!conv1) return set_err(h, TLSQ_ERR_NOCONV, "one-sided Jacobi did not converge in 40 sweeps (N=%lld)", (long long)N);
    return TLSQ_OK;
}

// Eigen-decomposition of the PSD matrix G through its Cholesky factor (cholesky.hip): one-sided Jacobi on the
// columns of L = chol(G + delta I) with no eigenvector accumulation; the eigenvectors of G are the normalised
// columns of the rotated L and sig = column norms = sqrt(lambda + delta).
int symeig_chol_f64(Handle* h, const double* G, int64_t N, int64_t ldG, double* B, double* V, double* sig_dev,
                    double* delta_host, int64_t* sweeps_out) {
    if (sweeps_out) *sweeps_out = 0;
    if (N <= 0) return TLSQ_OK;
    void* scal;
    TLSQ_TRY(ws_get(h, WS_SCAL, 4096, &scal));
    double* stats = reinterpret_cast<double*>(reinterpret_cast<char*>(scal) + 192);
    TLSQ_TRY(cholesky_shifted(h, G, ldG, N, B, V /* scratch */, stats));
    const int st = jacobi_factor_f64(h, B, N, V, sig_dev, (double)N * 2.220446049250313e-16, sweeps_out);
    TLSQ_HIP(h, hipMemcpyAsync(h->pinned, stats, 16, hipMemcpyDeviceToHost, h->stream));
    TLSQ_HIP(h, hipStreamSynchronize(h->stream));
    double sv[2];
    memcpy(sv, h->pinned, 16);
    if (delta_host) *delta_host = sv[1];
    return st;
}

// Vt[p + j ld] = V[j + order[p] N]: 32 x 32 tiles through LDS (reads along V's columns, writes along Vt's)
template <typename T>
__global__ __launch_bounds__(256) void k_vt_out(const double* __restrict__ V, int N, const int32_t* __restrict__ order, int d,
                                                T* __restrict__ Vt, int64_t ld) {
    __shared__ double tile[32][33];
    const int p0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int pp = ty; pp < 32; pp += 8) {
        const int p = p0 + pp, j = j0 + tx;
        tile[pp][tx] = (p < d && j < N) ? V[j + (int64_t)order[p] * N] : 0.0;
    }
    __syncthreads();
    for (int jj = ty; jj < 32; jj += 8) {
        const int p = p0 + tx, j = j0 + jj;
        if (p < d && j < N) Vt[p + (int64_t)j * ld] = (T)tile[tx][jj];
    }
}
template <typename T>
int launch_vt_out(Handle* h, const double* V, int64_t N, const int32_t* order_dev, int64_t d, T* Vt, int64_t ld) {
    hipLaunchKernelGGL((k_vt_out<T>), dim3((unsigned)((d + 31) / 32), (unsigned)((N + 31) / 32)), dim3(256), 0, h->stream, V, (int)N,
                       order_dev, (int)d, Vt, ld);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}
template int launch_vt_out<double>(Handle*, const double*, int64_t, const int32_t*, int64_t, double*, int64_t);
template int launch_vt_out<float>(Handle*, const double*, int64_t, const int32_t*, int64_t, float*, int64_t);

int launch_gather_scale(Handle* h, const double* V, int64_t N, const int32_t* sel_dev,
                        const double* g_dev, int64_t r, double* Vg, double* Vs) {
    if (r <= 0) return TLSQ_OK;
    int64_t g = (N * r + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_gather_scale, dim3((int)g), dim3(256), 0, h->stream, V, (int)N, sel_dev, g_dev,
                       (int)r, Vg, Vs);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

int launch_gather_scale_arg(Handle* h, const double* V, int64_t N, const SelWeights& sw, int64_t r, double* Vg,
                            double* Vs) {
    if (r <= 0) return TLSQ_OK;
    int64_t g = (N * r + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_gather_scale_arg, dim3((int)g), dim3(256), 0, h->stream, V, (int)N, sw, (int)r, Vg, Vs);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

}  // namespace tlsq
