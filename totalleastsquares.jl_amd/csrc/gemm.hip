// fp64 MFMA contractions of the rpca loop (gfx950, v_mfma_f64_16x16x4_f64):
//   Gram        G = Z'Z            (stands in for LAPACK gesdd's work on Z, src/robustPCA.jl:194,225)
//   rebuild     A = (Z Vg) Vs'     (the two mul! of src/robustPCA.jl:207-208 / 211-212)
// One LDS-tiled kernel, 128x128x16 workgroup tile, 4 waves x (64x64 = 4x4 MFMA tiles), register-staged
// double buffering (one barrier per K stage), XCD-aware block remap, deterministic split-K via slabs.
//
//   Cm[j + i*ldc] = sum_k Aop(i,k) * Bop(k,j)      (the contiguous output index j sits on the MFMA column)
//     A_KC: Aop(i,k) = A[k + i*lda]   else  A[i + k*lda]
//     B_KC: Bop(k,j) = B[k + j*ldb]   else  B[j + k*ldb]
#include <cstdlib>
#include <vector>

#include "common.hpp"

// development-only ablation builds (tools/kbench.py): 0 = product
#ifndef TLSQ_GEMM_ABLATE
#define TLSQ_GEMM_ABLATE 0
#endif

namespace tlsq {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int TI = 128, TJ = 128, TK = 16;
constexpr int LDK = TK + 2;    // K-contiguous panel  [128][18]: (i*18 + k) distinct mod 32 over a half-wave
constexpr int LDM = TI + 16;   // MN-contiguous panel [16][144]: (k*144 + i) distinct mod 32 over a half-wave
constexpr int PANEL = 2304;    // doubles per panel (128*18 == 16*144)

typedef double d2 __attribute__((ext_vector_type(2)));

// ---- global -> registers -> LDS staging of one 128 x 16 operand panel (8 doubles per thread) ----
// FULL: the panel is entirely inside the matrix and 16-byte aligned -> unguarded double2 loads.
// The operands may be fp32 in memory (the fp32 rpca path): they are widened to fp64 while being staged, the
// contraction itself always runs on the fp64 MFMA.
template <bool KC, bool FULL, typename TX>
__device__ __forceinline__ void panel_load(const TX* __restrict__ X, int64_t ld, int64_t r0,
                                           int64_t rmax, int64_t k0, int64_t kmax, double (&reg)[8]) {
    typedef TX x2 __attribute__((ext_vector_type(2)));
    const int t = threadIdx.x;
    if (FULL) {
        if (KC) {  // X[k + r*ld]: thread takes k = 2*(t&7)..+1 of rows (t>>3) + 32 s
            const int k = (t & 7) * 2, rr = t >> 3;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const x2 v = *reinterpret_cast<const x2*>(X + (k0 + k) + (r0 + rr + 32 * s) * ld);
                reg[2 * s] = (double)v[0];
                reg[2 * s + 1] = (double)v[1];
            }
        } else {  // X[r + k*ld]: thread takes r = 2*(t&63)..+1 of k = (t>>6) + 4 s
            const int r = (t & 63) * 2, kk = t >> 6;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const x2 v = *reinterpret_cast<const x2*>(X + (r0 + r) + (k0 + kk + 4 * s) * ld);
                reg[2 * s] = (double)v[0];
                reg[2 * s + 1] = (double)v[1];
            }
        }
    } else {
        if (KC) {  // X[k + r*ld]: 16 consecutive k per r
            const int k = t & 15, rr = t >> 4;
            const int64_t kg = k0 + k;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int64_t r = r0 + rr + 16 * s;
                const bool ok = (r < rmax && kg < kmax);
                const double v = (double)X[ok ? kg + r * ld : 0];  // always-valid address, then select (no branch)
                reg[s] = ok ? v : 0.0;
            }
        } else {  // X[r + k*ld]: 128 consecutive r per k
            const int r = t & 127, kk = t >> 7;
            const int64_t rg = r0 + r;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int64_t kg = k0 + kk + 2 * s;
                const bool ok = (rg < rmax && kg < kmax);
                const double v = (double)X[ok ? rg + kg * ld : 0];
                reg[s] = ok ? v : 0.0;
            }
        }
    }
}

template <bool KC, bool FULL>
__device__ __forceinline__ void panel_store(double* __restrict__ sm, const double (&reg)[8]) {
    const int t = threadIdx.x;
    if (FULL) {
        if (KC) {
            const int k = (t & 7) * 2, rr = t >> 3;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                *reinterpret_cast<d2*>(sm + (rr + 32 * s) * LDK + k) = d2{reg[2 * s], reg[2 * s + 1]};
        } else {
            const int r = (t & 63) * 2, kk = t >> 6;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                *reinterpret_cast<d2*>(sm + (kk + 4 * s) * LDM + r) = d2{reg[2 * s], reg[2 * s + 1]};
        }
    } else {
        if (KC) {
            const int k = t & 15, rr = t >> 4;
#pragma unroll
            for (int s = 0; s < 8; ++s) sm[(rr + 16 * s) * LDK + k] = reg[s];
        } else {
            const int r = t & 127, kk = t >> 7;
#pragma unroll
            for (int s = 0; s < 8; ++s) sm[(kk + 2 * s) * LDM + r] = reg[s];
        }
    }
}

// element (r, k) of a panel, r in [0,128), k in [0,16)
template <bool KC>
__device__ __forceinline__ double panel_at(const double* __restrict__ sm, int r, int k) {
    return KC ? sm[r * LDK + k] : sm[k * LDM + r];
}

// NA = number of live 16-row MFMA tiles of this wave along i (1, 2 or 4); j always uses 4.
// FULL = interior tile (no bounds checks, vector loads).
template <bool A_KC, bool B_KC, bool FULL, int NA, typename TA, typename TB>
__device__ __forceinline__ void gemm_body(const TA* __restrict__ A, int64_t lda,
                                          const TB* __restrict__ B, int64_t ldb,
                                          void* __restrict__ Cv, int c_f32, int64_t ldc, int64_t P, int64_t Q,
                                          int64_t kbeg, int64_t kend, int64_t i0, int64_t j0,
                                          double* __restrict__ smem) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wi = w & 1, wj = w >> 1;
    d4 acc[NA][4];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};

    double ra[8], rb[8];
    const int64_t nstage = (kend > kbeg) ? (kend - kbeg + TK - 1) / TK : 0;
    if (nstage > 0) {
        panel_load<A_KC, FULL, TA>(A, lda, i0, P, kbeg, kend, ra);
        panel_load<B_KC, FULL, TB>(B, ldb, j0, Q, kbeg, kend, rb);
        panel_store<A_KC, FULL>(smem, ra);
        panel_store<B_KC, FULL>(smem + 2 * PANEL, rb);
    }
    __syncthreads();
    const int fr = lane & 15, fk = lane >> 4;

    for (int64_t s = 0; s < nstage; ++s) {
        const int cur = (int)(s & 1);
        const bool more = (s + 1 < nstage);
        if (more && TLSQ_GEMM_ABLATE != 1) {   // ablation build 1: no global loads after the first stage
            const int64_t kn = kbeg + (s + 1) * TK;
            panel_load<A_KC, FULL, TA>(A, lda, i0, P, kn, kend, ra);
            panel_load<B_KC, FULL, TB>(B, ldb, j0, Q, kn, kend, rb);
        }
        const double* __restrict__ sa = smem + cur * PANEL;
        const double* __restrict__ sb = smem + (2 + cur) * PANEL;
        // fragments of k-step q+1 are fetched from LDS while the 4*NA MFMAs of k-step q issue
        double fa[2][NA], fb[2][4];
#pragma unroll
        for (int a = 0; a < NA; ++a) fa[0][a] = panel_at<A_KC>(sa, wi * 64 + a * 16 + fr, fk);
#pragma unroll
        for (int b = 0; b < 4; ++b) fb[0][b] = panel_at<B_KC>(sb, wj * 64 + b * 16 + fr, fk);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cb = q & 1, nbuf = cb ^ 1;
            if (q < 3) {
#pragma unroll
                for (int a = 0; a < NA; ++a)
                    fa[nbuf][a] = panel_at<A_KC>(sa, wi * 64 + a * 16 + fr, 4 * (q + 1) + fk);
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    fb[nbuf][b] = panel_at<B_KC>(sb, wj * 64 + b * 16 + fr, 4 * (q + 1) + fk);
            }
#if TLSQ_GEMM_ABLATE == 2   // ablation build 2: no MFMAs (keeps the LDS reads alive)
#pragma unroll
            for (int a = 0; a < NA; ++a) asm volatile("" ::"v"(fa[cb][a]), "v"(fb[cb][a]));
#else
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cb][a], fb[cb][b], acc[a][b], 0, 0, 0);
#endif
        }
        if (more) {
            panel_store<A_KC, FULL>(smem + (cur ^ 1) * PANEL, ra);
            panel_store<B_KC, FULL>(smem + (2 + (cur ^ 1)) * PANEL, rb);
        }
        __syncthreads();
    }

    // epilogue: lane holds column j = lane&15, rows i = (lane>>4) + 4*reg of each 16x16 tile
#pragma unroll
    for (int a = 0; a < NA; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int64_t j = j0 + wj * 64 + b * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = i0 + wi * 64 + a * 16 + fk + 4 * r;
                if (FULL || (i < P && j < Q)) {
                    if (c_f32) reinterpret_cast<float*>(Cv)[j + i * ldc] = (float)acc[a][b][r];
                    else reinterpret_cast<double*>(Cv)[j + i * ldc] = acc[a][b][r];
                }
            }
        }
    }
}

template <bool A_KC, bool B_KC, typename TA, typename TB>
__global__ __launch_bounds__(256, 2) void k_gemm_f64(const TA* __restrict__ A, int64_t lda,
                                                     const TB* __restrict__ B, int64_t ldb,
                                                     void* __restrict__ C, int c_f32, int64_t ldc, int64_t P,
                                                     int64_t Q, int64_t K, int64_t kchunk,
                                                     int64_t slab_stride, int nti, int ntj, int nsplit,
                                                     int symmetric, int vec_ok, const double* __restrict__ skip) {
    __shared__ __attribute__((aligned(16))) double smem[4 * PANEL];  // A[2], B[2]
    if (skip && skip[0] != 0.0) return;   // (conditional launches of the count certificate, see power_certificate)
    // Work items = (K split z, active tile t), z-major.  Blocks b and b+8 share an XCD (and its L2), so every
    // XCD gets a contiguous run of work items: neighbours share z (the same rows of the operands) and the
    // eight XCDs carry equal loads — also for symmetric launches, where only tiles with tj <= ti exist.
    const int ntiles = symmetric ? nti * (nti + 1) / 2 : nti * ntj;
    const int64_t nwork = (int64_t)ntiles * nsplit;
    const int64_t cpx = (nwork + 7) / 8;
    const int64_t item = (int64_t)(blockIdx.x % 8) * cpx + (int64_t)(blockIdx.x / 8);
    if (item >= nwork) return;
    const int z = (int)(item / ntiles);
    const int t = (int)(item % ntiles);
    int ti, tj;
    if (symmetric) {   // t -> (ti, tj) with tj <= ti, row-wise enumeration of the lower triangle
        ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        while (ti * (ti + 1) / 2 > t) --ti;
        tj = t - ti * (ti + 1) / 2;
    } else {
        ti = t % nti;
        tj = t / nti;
    }
    const int64_t kbeg = (int64_t)z * kchunk;
    const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    // split-K slabs are always fp64; a direct (nsplit == 1) store may be fp32
    void* __restrict__ Cz = c_f32 ? (void*)(reinterpret_cast<float*>(C) + (int64_t)z * slab_stride)
                                  : (void*)(reinterpret_cast<double*>(C) + (int64_t)z * slab_stride);
    const int64_t i0 = (int64_t)ti * TI, j0 = (int64_t)tj * TJ;
    const bool full = vec_ok && (i0 + TI <= P) && (j0 + TJ <= Q) && ((kend - kbeg) % TK == 0);
    if (full) {
#if TLSQ_GEMM_ABLATE == 3   // diagnostic build: in-kernel clock = d(s_memtime) / d(s_memrealtime) * 100 MHz
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
        gemm_body<A_KC, B_KC, true, 4, TA, TB>(A, lda, B, ldb, Cz, c_f32, ldc, P, Q, kbeg, kend, i0, j0, smem);
#if TLSQ_GEMM_ABLATE == 3
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0 && t == 2 && z == 3 && kend - kbeg > 256)
            printf("gemm clk: %llu cycles, %llu x10ns -> %.3f GHz, %.1f us\n", t1 - t0, r1 - r0,
                   (double)(t1 - t0) / (double)(r1 - r0) * 0.1, (double)(r1 - r0) * 0.01);
#endif
    } else {
        // live 16-row tiles along i for the widest wave (wave-uniform by construction: depends on blockIdx only)
        const int64_t rows = P - i0;  // > 0
        if (rows <= 16)
            gemm_body<A_KC, B_KC, false, 1, TA, TB>(A, lda, B, ldb, Cz, c_f32, ldc, P, Q, kbeg, kend, i0, j0, smem);
        else if (rows <= 32)
            gemm_body<A_KC, B_KC, false, 2, TA, TB>(A, lda, B, ldb, Cz, c_f32, ldc, P, Q, kbeg, kend, i0, j0, smem);
        else
            gemm_body<A_KC, B_KC, false, 4, TA, TB>(A, lda, B, ldb, Cz, c_f32, ldc, P, Q, kbeg, kend, i0, j0, smem);
    }
}

// C[j + i*ldc] = sum_z slab[z][j + i*lds]   (fixed order -> deterministic);
// symmetric: only tiles ti>=tj were computed, mirror into both triangles.  C may be fp32.
// normpart (optional): normpart[block] = this block's share of ||C||_F^2 (mirrored entries counted twice), summed in
// a fixed order - the caller adds the gridDim.x partials in order, so the norm is reproducible bit for bit
__global__ __launch_bounds__(256) void k_slab_reduce(const double* __restrict__ slab, int64_t lds_,
                                                     int64_t slab_stride, int nsplit,
                                                     void* __restrict__ C, int c_f32, int64_t ldc, int64_t P,
                                                     int64_t Q, int symmetric, const double* __restrict__ skip,
                                                     double* __restrict__ normpart, int nsplit_d, int tri) {
    __shared__ double nred[4];
    if (skip && skip[0] != 0.0) return;
    const int64_t total = P * Q;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double nacc = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t j = e % Q, i = e / Q;
        // symmetric: ONE writer per entry pair - the thread of (i, j), j <= i, stores the entry and its mirror image.  (A
        // diagonal tile holds both (i, j) and (j, i); for the product of two different commuting matrices the two sums
        // differ in their last bits, and two threads storing different values to the same two addresses made the result
        // depend on which one came last: the run-to-run differences of the matrix-function route, round 2.)
        if (symmetric && j > i) continue;
        // tri (k_gram_kc): diagonal tiles hold only their 16 x 16 MFMA tiles on and below the diagonal, in nsplit_d slabs
        const bool dtile = tri && (j / TJ) == (i / TI);
        const int ns = dtile ? nsplit_d : nsplit;
        double s = 0.0;
        for (int zz = 0; zz < ns; ++zz) s += slab[(int64_t)zz * slab_stride + j + i * lds_];
        const bool twice = symmetric && j < i;
        nacc += (twice ? 2.0 : 1.0) * s * s;
        if (c_f32) {
            reinterpret_cast<float*>(C)[j + i * ldc] = (float)s;
            if (symmetric) reinterpret_cast<float*>(C)[i + j * ldc] = (float)s;
        } else {
            reinterpret_cast<double*>(C)[j + i * ldc] = s;
            if (symmetric) reinterpret_cast<double*>(C)[i + j * ldc] = s;
        }
    }
    if (normpart) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nacc += __shfl_xor(nacc, off, 64);
        if ((threadIdx.x & 63) == 0) nred[threadIdx.x >> 6] = nacc;
        __syncthreads();
        if (threadIdx.x == 0) normpart[blockIdx.x] = (nred[0] + nred[1]) + (nred[2] + nred[3]);
    }
}


// ---- the Gram kernel proper: slab[z] (lower 128 x 128 tiles) = Z[kbeg:kend, :]' Z[kbeg:kend, :], Z column-major --------
// Same tile, panel layout and work-item map as k_gemm_f64<true, true>, but shaped for the way a CDNA wave issues - in
// order, one MFMA every 64 cycles, nothing queued behind it (tools/ubench/syrk_f64.hip holds the measurements, C2 size:
// 4 waves / classic loop 125.7 us, this form 102.5 us, MFMAs alone 95 us):
//  * 8 waves (two per SIMD), wave tile 64 x 32: while one wave waits at the barrier or on LDS the other one issues;
//  * fragments are read 16 bytes at a time: lane (r, g) holds k = 8q + 2g + {0, 1} of its column and feeds element m to
//    MFMA m of the pair - the same permutation of k on both operands, so the product is unchanged;
//  * software pipeline over the two LDS buffers: the next panel is stored during the first half of a stage, the barrier
//    sits between the halves, and the second half already fetches the first fragments of the next stage - no LDS
//    latency is exposed behind the barrier;
//  * every LDS / global access is pinned between two MFMAs of its half-stage (sched_barrier): a burst of ds_write or
//    global_load instructions ahead of the MFMAs would idle the matrix pipe under both waves of the SIMD at once.
// FULL: tile inside the matrix, whole stages, aligned 2-element loads; otherwise every element is loaded under a guard.
template <typename TZ, bool FULL>
__device__ __forceinline__ void gram_body(const TZ* __restrict__ Z, int64_t ld, double* __restrict__ Cz, int64_t ldc,
                                          int64_t N, int64_t kbeg, int64_t kend, int64_t i0, int64_t j0,
                                          double* __restrict__ smem) {
    typedef TZ x2 __attribute__((ext_vector_type(2)));
    constexpr int SL = 2;   // 2-element slots per thread and panel (128 columns x 8 pairs / 512 threads)
    const int nstage = (kend > kbeg) ? (int)((kend - kbeg + TK - 1) / TK) : 0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wj = w & 3, wi = w >> 2;
    const int fr = lane & 15, fk = lane >> 4;
    d4 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
    const TZ* pa[SL];
    const TZ* pb[SL];
    int so[SL], sr[SL], sk[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        const int e = tid + 512 * s;
        sr[s] = e >> 3;
        sk[s] = (e & 7) * 2;
        pa[s] = Z + kbeg + sk[s] + (i0 + sr[s]) * ld;
        pb[s] = Z + kbeg + sk[s] + (j0 + sr[s]) * ld;
        so[s] = sr[s] * LDK + sk[s];
    }
    const int oa = (wi * 64 + fr) * LDK + 2 * fk, ob = (wj * 32 + fr) * LDK + 2 * fk;
    x2 ra[SL], rb[SL];
    d2 ga[2][4], gb[2][2];

#define G_FR(buf, q, slot, i)                                                                                        \
    do {                                                                                                             \
        if ((i) < 4) ga[slot][(i) & 3] = *reinterpret_cast<const d2*>(smem + (buf) * PANEL + oa + ((i) & 3) * 16 * LDK + 8 * (q)); \
        else gb[slot][(i) & 1] = *reinterpret_cast<const d2*>(smem + (2 + (buf)) * PANEL + ob + ((i) & 1) * 16 * LDK + 8 * (q)); \
    } while (0)
#define G_SW(buf, i)                                                                                                  \
    do {                                                                                                              \
        if ((i) < SL) *reinterpret_cast<d2*>(smem + (buf) * PANEL + so[(i) % SL]) = d2{(double)ra[(i) % SL][0], (double)ra[(i) % SL][1]}; \
        else *reinterpret_cast<d2*>(smem + (2 + (buf)) * PANEL + so[(i) % SL]) = d2{(double)rb[(i) % SL][0], (double)rb[(i) % SL][1]}; \
    } while (0)
#define G_GL(i, koff)                                                                                  \
    do {                                                                                               \
        const int u_ = (i) % SL;                                                                       \
        if (FULL) {                                                                                    \
            if ((i) < SL) ra[u_] = *reinterpret_cast<const x2*>(pa[u_] + (koff));                      \
            else rb[u_] = *reinterpret_cast<const x2*>(pb[u_] + (koff));                               \
        } else {                                                                                       \
            const int64_t k_ = kbeg + (koff) + sk[u_];                                                 \
            const int64_t c_ = ((i) < SL ? i0 : j0) + sr[u_];                                          \
            const TZ* p_ = ((i) < SL ? pa[u_] : pb[u_]) + (koff);                                      \
            const bool o0_ = c_ < N && k_ < kend, o1_ = c_ < N && k_ + 1 < kend;                       \
            const TZ v0_ = o0_ ? p_[0] : (TZ)0, v1_ = o1_ ? p_[1] : (TZ)0;                             \
            if ((i) < SL) ra[u_] = x2{v0_, v1_};                                                       \
            else rb[u_] = x2{v0_, v1_};                                                                \
        }                                                                                              \
    } while (0)
#define G_MF(slot, t)                                                                                           \
    acc[((t) >> 1) & 3][(t) & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[slot][((t) >> 1) & 3][(t) >> 3],     \
                                                                        gb[slot][(t) & 1][(t) >> 3], acc[((t) >> 1) & 3][(t) & 1], 0, 0, 0)
    // one half-stage: 16 MFMAs on fragment slot q, each followed by at most one memory instruction.
    // KIND 0: fragments q = 1 of the same buffer; 1: those + the panel store into the other buffer; 2: fragments q = 0 of
    // the other buffer + the global loads of the stage after the next; 3: MFMAs only
    auto half = [&](auto kind, int cur, int q, int64_t koff) {
        constexpr int KIND = decltype(kind)::value;
        const int slot = q & 1, nslot = slot ^ 1;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            G_MF(slot, t);
            __builtin_amdgcn_sched_barrier(0);
            if (KIND != 3 && t < 6) {
                if (KIND == 2) G_FR(cur ^ 1, 0, nslot, t);
                else G_FR(cur, 1, nslot, t);
            } else if (KIND == 1 && t - 6 < 2 * SL) {
                G_SW(cur ^ 1, t - 6);
            } else if (KIND == 2 && t - 6 < 2 * SL) {
                G_GL(t - 6, koff);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    std::integral_constant<int, 0> K0;
    std::integral_constant<int, 1> K1;
    std::integral_constant<int, 2> K2;
    std::integral_constant<int, 3> K3;

    if (nstage > 0) {
#pragma unroll
        for (int i = 0; i < 2 * SL; ++i) G_GL(i, (int64_t)0);
#pragma unroll
        for (int i = 0; i < 2 * SL; ++i) G_SW(0, i);
    }
    __syncthreads();
    if (nstage > 1) {
#pragma unroll
        for (int i = 0; i < 2 * SL; ++i) G_GL(i, (int64_t)TK);
    }
    if (nstage > 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) G_FR(0, 0, 0, i);
    }
    for (int s = 0; s + 1 < nstage; ++s) {
        const int cur = s & 1;
        // (the last two stages load the final stage again instead of branching: those values are never stored)
        const int64_t koff = (int64_t)(s + 2 < nstage ? s + 2 : nstage - 1) * TK;
        half(K1, cur, 0, (int64_t)0);
        __syncthreads();
        half(K2, cur, 1, koff);
    }
    if (nstage > 0) {
        const int cur = (nstage - 1) & 1;
        half(K0, cur, 0, (int64_t)0);
        half(K3, cur, 1, (int64_t)0);
    }
#undef G_FR
#undef G_SW
#undef G_GL
#undef G_MF
    // epilogue: lane holds column j = lane & 15, rows i = (lane >> 4) + 4 reg of each 16 x 16 tile
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int64_t j = j0 + wj * 32 + b * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = i0 + wi * 64 + a * 16 + fk + 4 * r;
                if (FULL || (i < N && j < N)) Cz[j + i * ldc] = acc[a][b][r];
            }
        }
}

// ---- diagonal 128 x 128 tiles of the Gram matrix: only the 36 MFMA tiles (16 x 16) on and below the diagonal ----------
// Tile rows a, columns b of the 8 x 8 grid, b <= a.  Rows a and 7 - a hold nine tiles together; every pair is split
// 5 + 4 over two waves, the five-tile waves (0..3) and the four-tile waves (4..7) of a pair sit on the same SIMD:
//   wave u     (u = 0..3): row 7 - u, columns 0..4
//   wave 4 + u           : row 7 - u, columns 5..7 - u   and   row u, columns 0..u
// 9 tiles per SIMD and k-step instead of 16: a diagonal work item costs 0.56 of an off-diagonal one per row of Z (the
// host gives it a longer K chunk).  One operand panel (A == B), fragments: two A rows + one B per tile.
constexpr int DTK = 32;            // K rows per stage of a diagonal work item: one panel (A == B) leaves room for twice the
constexpr int DLDK = DTK + 2;      // rows of the off-diagonal stage in the same LDS, and halves the barriers per row
constexpr int DPANEL = 128 * DLDK; // doubles per buffer (2 * DPANEL <= 4 * PANEL)
template <typename TZ, bool FULL, bool FIVE>
__device__ __forceinline__ void gram_diag_body(const TZ* __restrict__ Z, int64_t ld, double* __restrict__ Cz,
                                               int64_t ldc, int64_t N, int64_t kbeg, int64_t kend, int64_t i0,
                                               double* __restrict__ smem) {
    typedef TZ x2 __attribute__((ext_vector_type(2)));
    constexpr int SL = 4;   // 2-element slots per thread and stage (128 columns x 16 pairs / 512 threads)
    constexpr int NTL = FIVE ? 5 : 4;
    const int nstage = (kend > kbeg) ? (int)((kend - kbeg + DTK - 1) / DTK) : 0;
    const int tid = threadIdx.x, lane = tid & 63, u = (tid >> 6) & 3;
    const int fr = lane & 15, fk = lane >> 4;
    const int ar0 = 7 - u, ar1 = u;
    const int c1 = FIVE ? 5 : 3 - u;   // tiles t < c1 lie in row ar0 (columns b0 + t), the others in row ar1 (columns t - c1)
    const int b0 = FIVE ? 0 : 5;
    d4 acc[NTL];
    int ob[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        acc[t] = d4{0.0, 0.0, 0.0, 0.0};
        const int b = t < c1 ? b0 + t : t - c1;
        ob[t] = (16 * b + fr) * DLDK + 2 * fk;
    }
    const int oa0 = (16 * ar0 + fr) * DLDK + 2 * fk, oa1 = (16 * ar1 + fr) * DLDK + 2 * fk;
    const TZ* pa[SL];
    int so[SL], sr[SL], sk[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        const int e = tid + 512 * s;
        sr[s] = e >> 4;
        sk[s] = (e & 15) * 2;
        pa[s] = Z + kbeg + sk[s] + (i0 + sr[s]) * ld;
        so[s] = sr[s] * DLDK + sk[s];
    }
    x2 ra[SL];
    d2 ga[2][2], gb[2][NTL];

#define D_FR(buf, q, slot, i)                                                                                    \
    do {                                                                                                         \
        if ((i) == 0) ga[slot][0] = *reinterpret_cast<const d2*>(smem + (buf) * DPANEL + oa0 + 8 * (q));          \
        else if ((i) == 1) ga[slot][1] = *reinterpret_cast<const d2*>(smem + (buf) * DPANEL + oa1 + 8 * (q));     \
        else gb[slot][((i) - 2) % NTL] = *reinterpret_cast<const d2*>(smem + (buf) * DPANEL + ob[((i) - 2) % NTL] + 8 * (q)); \
    } while (0)
#define D_SW(buf, i) \
    *reinterpret_cast<d2*>(smem + (buf) * DPANEL + so[(i) % SL]) = d2{(double)ra[(i) % SL][0], (double)ra[(i) % SL][1]}
#define D_GL(i, koff)                                                                     \
    do {                                                                                  \
        const int u_ = (i) % SL;                                                          \
        if (FULL) {                                                                       \
            ra[u_] = *reinterpret_cast<const x2*>(pa[u_] + (koff));                       \
        } else {                                                                          \
            const int64_t k_ = kbeg + (koff) + sk[u_];                                    \
            const int64_t c_ = i0 + sr[u_];                                               \
            const TZ* p_ = pa[u_] + (koff);                                               \
            const bool o0_ = c_ < N && k_ < kend, o1_ = c_ < N && k_ + 1 < kend;          \
            ra[u_] = x2{o0_ ? p_[0] : (TZ)0, o1_ ? p_[1] : (TZ)0};                        \
        }                                                                                 \
    } while (0)
    // one quarter of a stage (8 rows of Z): 2 NTL MFMAs on fragment slot q & 1, each followed by at most one memory
    // instruction - first the fragments of the next quarter (fbuf, fq), then the two extra items of this quarter:
    //   EXTRA 0 none | 1 global loads of slots x0, x0 + 1 at row offset koff | 2 LDS stores of slots x0, x0 + 1 into sbuf
    auto quarter = [&](auto extra, auto frags, int q, int fbuf, int fq, int x0, int sbuf, int64_t koff) {
        constexpr int EXTRA = decltype(extra)::value;
        constexpr bool FRAGS = decltype(frags)::value;
        const int slot = q & 1, nslot = slot ^ 1;
#pragma unroll
        for (int n = 0; n < 2 * NTL; ++n) {
            const int t = n % NTL, m = n / NTL;
            const double av = (t < c1) ? ga[slot][0][m] : ga[slot][1][m];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, gb[slot][t][m], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (FRAGS && n < 2 + NTL) {
                D_FR(fbuf, fq, nslot, n);
            } else if (EXTRA == 1 && n - (2 + NTL) < 2 && n >= 2 + NTL) {
                D_GL(x0 + n - (2 + NTL), koff);
            } else if (EXTRA == 2 && n - (2 + NTL) < 2 && n >= 2 + NTL) {
                D_SW(sbuf, x0 + n - (2 + NTL));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    std::integral_constant<int, 0> X0;
    std::integral_constant<int, 1> XL;
    std::integral_constant<int, 2> XS;
    std::true_type FR1;
    std::false_type FR0;

    if (nstage > 0) {
#pragma unroll
        for (int i = 0; i < SL; ++i) D_GL(i, (int64_t)0);
#pragma unroll
        for (int i = 0; i < SL; ++i) D_SW(0, i);
    }
    __syncthreads();
    if (nstage > 1) {   // (slots 0, 1 of the next stage: what the last quarter of a stage does for the one after it)
        D_GL(0, (int64_t)DTK);
        D_GL(1, (int64_t)DTK);
    }
    if (nstage > 0) {
#pragma unroll
        for (int i = 0; i < 2 + NTL; ++i) D_FR(0, 0, 0, i);
    }
    for (int s = 0; s + 1 < nstage; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        const int64_t k1 = (int64_t)(s + 1) * DTK;
        // (the last two stages load the final stage again instead of branching: those values are never stored)
        const int64_t k2 = (int64_t)(s + 2 < nstage ? s + 2 : nstage - 1) * DTK;
        quarter(XL, FR1, 0, cur, 1, 2, 0, k1);      // + loads of slots 2, 3 of stage s + 1
        quarter(XS, FR1, 1, cur, 2, 0, nxt, 0);     // + stores of slots 0, 1 (loaded a stage ago)
        quarter(XS, FR1, 2, cur, 3, 2, nxt, 0);     // + stores of slots 2, 3
        __syncthreads();
        quarter(XL, FR1, 3, nxt, 0, 0, 0, k2);      // fragments of the next stage, loads of slots 0, 1 of stage s + 2
    }
    if (nstage > 0) {
        const int cur = (nstage - 1) & 1;
        quarter(X0, FR1, 0, cur, 1, 0, 0, 0);
        quarter(X0, FR1, 1, cur, 2, 0, 0, 0);
        quarter(X0, FR1, 2, cur, 3, 0, 0, 0);
        quarter(X0, FR0, 3, cur, 0, 0, 0, 0);
    }
#undef D_FR
#undef D_SW
#undef D_GL
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int a = t < c1 ? ar0 : ar1, b = t < c1 ? b0 + t : t - c1;
        const int64_t j = i0 + 16 * b + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t i = i0 + 16 * a + fk + 4 * r;
            if (FULL || (i < N && j < N)) Cz[j + i * ldc] = acc[t][r];
        }
    }
}

// Work items: the strictly lower 128 x 128 tiles, nsplit_o K chunks each (z-major), then the diagonal tiles, nsplit_d
// (longer) chunks each; both kinds cost about the same, and the XCDs get contiguous runs of the list (blocks b and b + 8
// share an XCD and its L2: neighbours share z, i.e. the same rows of Z).
template <typename TZ>
__global__ __launch_bounds__(512) void k_gram_kc(const TZ* __restrict__ Z, int64_t ld, double* __restrict__ slab,
                                                 int64_t ldc, int64_t N, int64_t K, int64_t kchunk_o, int64_t kchunk_d,
                                                 int64_t slab_stride, int nti, int nsplit_o, int nsplit_d, int vec_ok,
                                                 const double* __restrict__ skip, const int32_t* __restrict__ order,
                                                 int zbase_o, int zbase_d, int noff_in) {
    __shared__ __attribute__((aligned(16))) double smem[4 * PANEL];  // A[2], B[2]
    if (skip && skip[0] != 0.0) return;
    // (noff_in > 0: only the first noff_in tiles of the order table - gram_offdiag_launch: the off-diagonal block of a Gram
    //  matrix whose diagonal 256-column blocks come from the fused sweep kernel)
    const int noff = noff_in > 0 ? noff_in : nti * (nti - 1) / 2;
    const int64_t n_o = (int64_t)noff * nsplit_o;
    const int64_t nwork = n_o + (int64_t)nti * nsplit_d;
    const int64_t cpx = (nwork + 7) / 8;
    const int64_t item = (int64_t)(blockIdx.x % 8) * cpx + (int64_t)(blockIdx.x / 8);
    if (item >= nwork) return;
    if (item < n_o) {
        const int z = (int)(item / noff);
        const int t = (int)(item % noff);
        // t -> (ti, tj), tj < ti: from the host's blocked order (large N), else row-wise through the strictly lower triangle
        int ti, tj;
        if (order) {
            ti = order[2 * t];
            tj = order[2 * t + 1];
        } else {
            ti = (int)((sqrtf(8.0f * (float)t + 1.0f) + 1.0f) * 0.5f);
            while (ti * (ti + 1) / 2 <= t) ++ti;
            while (ti * (ti - 1) / 2 > t) --ti;
            tj = t - ti * (ti - 1) / 2;
        }
        const int64_t kbeg = (int64_t)z * kchunk_o;
        const int64_t kend = (kbeg + kchunk_o < K) ? kbeg + kchunk_o : K;
        const int64_t i0 = (int64_t)ti * TI, j0 = (int64_t)tj * TJ;
        double* __restrict__ Cz = slab + (int64_t)(zbase_o + z) * slab_stride;   // (row chunks of a panel stack their slabs)
        const bool full = vec_ok && (i0 + TI <= N) && ((kend - kbeg) % TK == 0);   // (j0 < i0)
        if (full) gram_body<TZ, true>(Z, ld, Cz, ldc, N, kbeg, kend, i0, j0, smem);
        else gram_body<TZ, false>(Z, ld, Cz, ldc, N, kbeg, kend, i0, j0, smem);
    } else {
        const int64_t d = item - n_o;
        const int z = (int)(d / nti);
        const int ti = (int)(d % nti);
        const int64_t kbeg = (int64_t)z * kchunk_d;
        const int64_t kend = (kbeg + kchunk_d < K) ? kbeg + kchunk_d : K;
        const int64_t i0 = (int64_t)ti * TI;
        double* __restrict__ Cz = slab + (int64_t)(zbase_d + z) * slab_stride;
        const bool full = vec_ok && (i0 + TI <= N) && ((kend - kbeg) % DTK == 0);
        const bool five = threadIdx.x < 256;   // waves 0..3 (wave-uniform: both sides meet the same barriers)
        if (full) {
            if (five) gram_diag_body<TZ, true, true>(Z, ld, Cz, ldc, N, kbeg, kend, i0, smem);
            else gram_diag_body<TZ, true, false>(Z, ld, Cz, ldc, N, kbeg, kend, i0, smem);
        } else {
            if (five) gram_diag_body<TZ, false, true>(Z, ld, Cz, ldc, N, kbeg, kend, i0, smem);
            else gram_diag_body<TZ, false, false>(Z, ld, Cz, ldc, N, kbeg, kend, i0, smem);
        }
    }
}

// G = Z'Z through k_gram_kc: split-K slabs + the fixed-order reduction (k_slab_reduce, tri = 1)
// The Gram matrix of one K-contiguous operand in three steps, so that the row chunks of a panel can be queued one by
// one (behind the chunks of the sweep that produces it, on a second stream: solver.hip): plan (K splits for chunks of K
// rows, slabs for `nchunks` of them, tile order), launch of one chunk, fixed-order reduction of all slabs.
int gram_plan(Handle* h, int z_f32, int64_t N, int64_t K, int nchunks, GramPlan* pl) {
    const int64_t nti = (N + TI - 1) / TI, noff = nti * (nti - 1) / 2;
    // relative cost of a diagonal work item per row of Z (9 of 16 MFMA tiles per SIMD + the shared per-stage overhead)
    const double rho = [] { const char* e = dev_get(DEV_GRAM_RHO); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 0.62; }();
    const int64_t target_wgs = [] { const char* e = dev_get(DEV_GEMM_WGS); const long v = e ? atol(e) : 0; return (int64_t)(v > 0 ? v : 256); }();
    // K splits (nsplit_o for the off-diagonal tiles, nsplit_d for the diagonal ones) by a small cost model, in us:
    // an item of kc rows takes kc * c_row (a CU at ~92 % of its MFMA peak: 2 * 128 * 128 flop per row) + c_item
    // (dispatch, first loads, slab store: fitted), items run in rounds of one per CU, and every slab entry is written and read once more
    // (~4 TB/s).  Small N: one round, ~252 items.  Large N: enough items that the last round is nearly full without
    // the slabs growing past what that is worth.
    constexpr double c_row = 0.116, c_item = 18.0, slab_us_per_byte = 2.0 / 4.0e6;
    const int64_t maxsplit = std::max<int64_t>(1, (K + 4 * TK - 1) / (4 * TK));                       // >= four K stages per item
    const int64_t memsplit = std::max<int64_t>(1, (int64_t)(((size_t)2 << 30) / ((size_t)N * N * 8)));   // slabs <= 2 GB
    auto chunk_of = [&](int64_t& ns, int64_t tk) {
        int64_t kc = (K + ns - 1) / ns;
        kc = (kc + tk - 1) / tk * tk;
        if (kc < tk) kc = tk;
        ns = K > 0 ? (K + kc - 1) / kc : 1;
        return kc;
    };
    int64_t nsplit_o = 1, nsplit_d = 1;
    double best = 1e300;
    const int64_t hi = std::min<int64_t>(std::min<int64_t>(maxsplit, memsplit), 4 * target_wgs);
    for (int64_t so = 1; so <= hi; ++so) {
        for (int pass = 0; pass < 2; ++pass) {
            int64_t o = noff > 0 ? so : 1;
            int64_t d = std::min<int64_t>(std::max<int64_t>(1, (int64_t)std::floor(so * rho) + pass), std::min(maxsplit, memsplit));
            const int64_t kco = chunk_of(o, TK), kcd = chunk_of(d, DTK);
            const double t_o = kco * c_row + c_item, t_d = rho * kcd * c_row + c_item;
            // list schedule on target_wgs CUs, off-diagonal items first (the launch order): r_o full rounds, then m CUs
            // take the rest of them while the others start on the diagonal items
            const int64_t n_o = noff * o, n_d = nti * d, W = target_wgs;
            const int64_t r_o = n_o / W, m = n_o % W;
            const double s0 = (double)r_o * t_o, s1 = m > 0 ? s0 + t_o : s0;   // when the two groups of CUs become free
            double t_end = s1;
            for (int64_t k = 0;; ++k) {   // smallest finishing time with room for all diagonal items
                const double c0 = s0 + (double)k * t_d, c1 = s1 + (double)k * t_d;
                bool done = false;
                for (const double T : {std::min(c0, c1), std::max(c0, c1)}) {
                    const double cap = (double)(W - m) * std::floor((T - s0) / t_d + 1e-9) +
                                       (m > 0 ? (double)m * std::max(0.0, std::floor((T - s1) / t_d + 1e-9)) : 0.0);
                    if (cap >= (double)n_d && T >= s1 - 1e-9) {
                        t_end = std::max(T, s1);
                        done = true;
                        break;
                    }
                }
                if (done) break;
            }
            const double slab_bytes = ((double)n_o + 0.5625 * (double)n_d) * (double)(TI * TJ * 8);
            // (several rounds: + a quarter item - the CUs drift out of step, and long items lose more to that than the
            // schedule says: 65536 x 4096 measured 18.0 ms with 4 chunks, 16.6 ms with 16)
            const double t = t_end + (n_o + n_d > W ? 0.25 * std::max(t_o, t_d) : 0.0) + slab_bytes * slab_us_per_byte;
            if (t < best) {
                best = t;
                nsplit_o = o;
                nsplit_d = d;
            }
        }
        if (noff == 0 && so >= hi) break;
    }
    if (const char* e = dev_get(DEV_GRAM_SPLIT)) {   // development: "o,d"
        long o = 0, d = 0;
        if (sscanf(e, "%ld,%ld", &o, &d) == 2 && o > 0 && d > 0) {
            nsplit_o = noff > 0 ? std::min<int64_t>(o, maxsplit) : 1;
            nsplit_d = std::min<int64_t>(d, maxsplit);
        }
    }
    const int64_t kchunk_o = chunk_of(nsplit_o, TK), kchunk_d = chunk_of(nsplit_d, DTK);
    {
        const bool dbg = dev_is(DEV_DEBUG, '2');
        if (dbg) fprintf(stderr, "[tlsq] gram %lld x %lld: nsplit %lld / %lld, model %.0f us\n", (long long)K, (long long)N,
                         (long long)nsplit_o, (long long)nsplit_d, best);
    }
    const int64_t nslab = std::max(nsplit_o, nsplit_d) * nchunks;
    const int64_t slab_stride = N * N;
    void* slab;
    TLSQ_TRY(ws_get(h, WS_SLAB, (size_t)(nslab * slab_stride) * sizeof(double), &slab));
    const int64_t nwork = noff * nsplit_o + nti * nsplit_d;
    const int64_t cpx = (nwork + 7) / 8;
    if (8 * cpx > 2147483647LL) return set_err(h, TLSQ_ERR_UNSUPPORTED, "gram: grid too large");
    // Tile order.  The ~32 work items an XCD runs at a time are neighbours in the list and march through the same rows
    // of Z: row-wise order makes them share one column block of Z and differ in the other (33 blocks through the L2 for
    // 32 tiles); blocks of 8 x 4 tiles share 12.  From N = 2048 on the kernel is otherwise bound by that traffic
    // (65536 x 4096: 70 GB per Gram).
    const int32_t* order = nullptr;
    const bool no_order = dev_is(DEV_GRAM_ROWWISE, '1');
    if (nti > 8 && !no_order) {
        void* tab;
        TLSQ_TRY(ws_get(h, WS_GRAMTAB, (size_t)noff * 8, &tab));
        if (h->gram_tab_nti != nti) {
            std::vector<int32_t> o;
            o.reserve((size_t)noff * 2);
            for (int64_t I = 0; I < nti; I += 8)
                for (int64_t J = 0; J < std::min(I + 8, nti); J += 4)
                    for (int64_t ti = I; ti < std::min(I + 8, nti); ++ti)
                        for (int64_t tj = J; tj < std::min(J + 4, ti); ++tj) {
                            o.push_back((int32_t)ti);
                            o.push_back((int32_t)tj);
                        }
            if ((int64_t)o.size() != 2 * noff) return set_err(h, TLSQ_ERR_UNSUPPORTED, "gram: tile order table");
            TLSQ_HIP(h, hipMemcpyAsync(tab, o.data(), o.size() * 4, hipMemcpyHostToDevice, h->stream));
            TLSQ_HIP(h, hipStreamSynchronize(h->stream));   // (pageable source; once per N)
            h->gram_tab_nti = nti;
        }
        order = (const int32_t*)tab;
    }
    pl->N = N;
    pl->K = K;
    pl->nti = nti;
    pl->nsplit_o = nsplit_o;
    pl->nsplit_d = nsplit_d;
    pl->kchunk_o = kchunk_o;
    pl->kchunk_d = kchunk_d;
    pl->nchunks = nchunks;
    pl->z_f32 = z_f32;
    pl->slab = (double*)slab;
    pl->order = order;
    return TLSQ_OK;
}

// chunk `c` of the plan: rows [0, rows) of Z (rows <= pl.K), slabs c * nsplit .. of each kind
int gram_launch_chunk(Handle* h, hipStream_t st, const GramPlan& pl, const void* Z, int64_t ld, int64_t rows, int c,
                      const double* skip) {
    const int64_t N = pl.N, noff = pl.nti * (pl.nti - 1) / 2;
    const int64_t nwork = noff * pl.nsplit_o + pl.nti * pl.nsplit_d, cpx = (nwork + 7) / 8;
    const uintptr_t am = pl.z_f32 ? 8 : 16;
    const int vec_ok = ((ld % 2) == 0 && (pl.kchunk_o % 2) == 0 && (pl.kchunk_d % 2) == 0 && (reinterpret_cast<uintptr_t>(Z) % am) == 0) ? 1 : 0;
    if (pl.z_f32)
        hipLaunchKernelGGL((k_gram_kc<float>), dim3((unsigned)(8 * cpx)), dim3(512), 0, st, (const float*)Z, ld, pl.slab, N, N, rows,
                           pl.kchunk_o, pl.kchunk_d, N * N, (int)pl.nti, (int)pl.nsplit_o, (int)pl.nsplit_d, vec_ok, skip, pl.order,
                           (int)(c * pl.nsplit_o), (int)(c * pl.nsplit_d), 0);
    else
        hipLaunchKernelGGL((k_gram_kc<double>), dim3((unsigned)(8 * cpx)), dim3(512), 0, st, (const double*)Z, ld, pl.slab, N, N, rows,
                           pl.kchunk_o, pl.kchunk_d, N * N, (int)pl.nti, (int)pl.nsplit_o, (int)pl.nsplit_d, vec_ok, skip, pl.order,
                           (int)(c * pl.nsplit_o), (int)(c * pl.nsplit_d), 0);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

int gram_reduce(Handle* h, hipStream_t st, const GramPlan& pl, double* G, int64_t ldg, const double* skip, double* normpart,
                int* normblocks) {
    const int64_t N = pl.N;
    int64_t g = (N * N + 255) / 256;
    if (g > 2048) g = 2048;
    if (normblocks) *normblocks = (int)g;
    hipLaunchKernelGGL(k_slab_reduce, dim3((int)g), dim3(256), 0, st, (const double*)pl.slab, N, N * N,
                       (int)(pl.nsplit_o * pl.nchunks), (void*)G, 0, ldg, N, N, 1, skip, normpart, (int)(pl.nsplit_d * pl.nchunks), 1);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// ---- N = 512 with the fused sweep kernel (fused.hip): that kernel leaves the partial sums of the two diagonal 256-column
// blocks; the off-diagonal block G[256:512, 0:256] - the 128 x 128 tiles (2,0), (2,1), (3,0), (3,1) - is accumulated here from
// the Z_{k+1} it has written, by k_gram_kc's off-diagonal body on a four-entry tile list, into slabs of its own, and one
// reduction adds both sets in a fixed order.
int gram_offdiag_plan(Handle* h, int64_t N, int64_t K, GramPlan* pl) {
    if (N != 512) return set_err(h, TLSQ_ERR_UNSUPPORTED, "gram_offdiag: N = 512 only");
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device);
    const int64_t maxsplit = std::max<int64_t>(1, K / (4 * TK));
    int64_t ns = std::min<int64_t>(std::max<int64_t>(1, ncu / 4), maxsplit);
    int64_t kc = ((K + ns - 1) / ns + TK - 1) / TK * TK;
    ns = (K + kc - 1) / kc;
    void *slab, *tab;
    TLSQ_TRY(ws_get(h, WS_SLAB2, (size_t)ns * (size_t)N * (size_t)N * sizeof(double), &slab));
    TLSQ_TRY(ws_get(h, WS_GRAMTAB3, 64, &tab));
    if (!h->gram_tab3_ready) {
        static const int32_t o[8] = {2, 0, 2, 1, 3, 0, 3, 1};
        TLSQ_HIP(h, hipMemcpyAsync(tab, o, sizeof(o), hipMemcpyHostToDevice, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));   // (pageable source; once per handle)
        h->gram_tab3_ready = true;
    }
    *pl = GramPlan();
    pl->N = N;
    pl->K = K;
    pl->nti = 4;
    pl->nsplit_o = ns;
    pl->nsplit_d = 0;
    pl->kchunk_o = kc;
    pl->kchunk_d = kc;
    pl->nchunks = 1;
    pl->slab = (double*)slab;
    pl->order = (const int32_t*)tab;
    return TLSQ_OK;
}

int gram_offdiag_launch(Handle* h, hipStream_t st, const GramPlan& pl, const double* Z, int64_t ld, int64_t rows) {
    const int64_t N = pl.N, nwork = 4 * pl.nsplit_o, cpx = (nwork + 7) / 8;
    const int vec_ok = ((ld % 2) == 0 && (pl.kchunk_o % 2) == 0 && (reinterpret_cast<uintptr_t>(Z) % 16) == 0) ? 1 : 0;
    hipLaunchKernelGGL((k_gram_kc<double>), dim3((unsigned)(8 * cpx)), dim3(512), 0, st, Z, ld, pl.slab, N, N, rows, pl.kchunk_o,
                       pl.kchunk_d, N * N, 4, (int)pl.nsplit_o, 0, vec_ok, (const double*)nullptr, pl.order, 0, 0, 4);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// G (both triangles) from the two slab sets: entries inside a diagonal 256-column block from slabA (nA slabs), the others from
// slabB (nB slabs); one writer per entry pair, fixed summation order
__global__ __launch_bounds__(256) void k_slab_reduce2(const double* __restrict__ slabA, int nA, const double* __restrict__ slabB,
                                                      int nB, double* __restrict__ G, int64_t ldg, int N) {
    const int64_t total = (int64_t)N * N, stride = (int64_t)gridDim.x * 256, ss = (int64_t)N * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        const int j = (int)(e % N), i = (int)(e / N);
        if (j > i) continue;
        const bool diag = (i >> 8) == (j >> 8);
        const double* sl = diag ? slabA : slabB;
        const int ns = diag ? nA : nB;
        double s = 0.0;
        for (int z = 0; z < ns; ++z) s += sl[(int64_t)z * ss + j + (int64_t)i * N];
        G[j + (int64_t)i * ldg] = s;
        G[i + (int64_t)j * ldg] = s;
    }
}

int gram_reduce2(Handle* h, hipStream_t st, const GramPlan& plA, const GramPlan& plB, double* G, int64_t ldg) {
    const int64_t N = plA.N;
    hipLaunchKernelGGL(k_slab_reduce2, dim3(1024), dim3(256), 0, st, (const double*)plA.slab, (int)plA.nsplit_o, (const double*)plB.slab,
                       (int)plB.nsplit_o, G, ldg, (int)N);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

static int gram_kc(Handle* h, const void* Z, int z_f32, int64_t ld, double* G, int64_t ldg, int64_t N, int64_t K,
                   const double* skip, double* normpart, int* normblocks) {
    GramPlan pl;
    TLSQ_TRY(gram_plan(h, z_f32, N, K, 1, &pl));
    TLSQ_TRY(gram_launch_chunk(h, h->stream, pl, Z, ld, K, 0, skip));
    return gram_reduce(h, h->stream, pl, G, ldg, skip, normpart, normblocks);
}

static int launch_gemm(Handle* h, bool A_KC, bool B_KC, const void* A, int a_f32, int64_t lda,
                       const void* B, int b_f32, int64_t ldb, void* C, int c_f32, int64_t ldc, int64_t P,
                       int64_t Q, int64_t K, int nsplit, int64_t kchunk, int64_t slab_stride, bool symmetric,
                       const double* skip = nullptr) {
    const int nti = (int)((P + TI - 1) / TI), ntj = (int)((Q + TJ - 1) / TJ);
    const int64_t ntiles = symmetric ? (int64_t)nti * (nti + 1) / 2 : (int64_t)nti * ntj;
    const int64_t nwork = ntiles * nsplit;
    const int64_t cpx = (nwork + 7) / 8;
    if (8 * cpx > 2147483647LL) return set_err(h, TLSQ_ERR_UNSUPPORTED, "gemm: grid too large");
    dim3 grid((unsigned)(8 * cpx)), block(256);
    // 2-element vector loads need even leading dimensions, aligned bases and an even K chunk
    const uintptr_t am = a_f32 ? 8 : 16, bm = b_f32 ? 8 : 16;
    const int vec_ok = ((lda % 2) == 0 && (ldb % 2) == 0 && (kchunk % 2) == 0 &&
                        (reinterpret_cast<uintptr_t>(A) % am) == 0 && (reinterpret_cast<uintptr_t>(B) % bm) == 0)
                           ? 1 : 0;
#define GO2(AK, BK, TA, TB)                                                                              \
    hipLaunchKernelGGL((k_gemm_f64<AK, BK, TA, TB>), grid, block, 0, h->stream, (const TA*)A, lda,       \
                       (const TB*)B, ldb, C, c_f32, ldc, P, Q, K, kchunk, slab_stride, nti, ntj, nsplit, \
                       symmetric ? 1 : 0, vec_ok, skip)
#define GO(TA, TB)                                         \
    do {                                                   \
        if (A_KC && B_KC) GO2(true, true, TA, TB);         \
        else if (A_KC && !B_KC) GO2(true, false, TA, TB);  \
        else if (!A_KC && !B_KC) GO2(false, false, TA, TB); \
        else return set_err(h, TLSQ_ERR_UNSUPPORTED, "gemm: operand layout (MN,KC) is not instantiated"); \
    } while (0)
    if (!a_f32 && !b_f32) GO(double, double);
    else if (a_f32 && b_f32) GO(float, float);
    else if (!a_f32 && b_f32) GO(double, float);
    else return set_err(h, TLSQ_ERR_UNSUPPORTED, "gemm: operand types (f32, f64) are not instantiated");
#undef GO
#undef GO2
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

int gemm_mixed(Handle* h, bool A_KC, bool B_KC, const void* A, int a_f32, int64_t lda, const void* B, int b_f32,
               int64_t ldb, void* C, int c_f32, int64_t ldc, int64_t P, int64_t Q, int64_t K, bool symmetric,
               const double* skip, double* normpart, int* normblocks) {
    if (P <= 0 || Q <= 0) return TLSQ_OK;
    if (normpart && !symmetric) return set_err(h, TLSQ_ERR_ARG, "gemm: the norm by-product needs the symmetric (slab) path");
    const bool old_gram = dev_is(DEV_GRAM_OLD, '1');
    if (symmetric && A_KC && B_KC && A == B && lda == ldb && a_f32 == b_f32 && !c_f32 && P == Q && !old_gram)
        return gram_kc(h, A, a_f32, lda, (double*)C, ldc, P, K, skip, normpart, normblocks);   // the Gram matrix of one operand
    const int64_t nti = (P + TI - 1) / TI, ntj = (Q + TJ - 1) / TJ;
    const int64_t tiles = symmetric ? nti * (nti + 1) / 2 : nti * ntj;
    // split K so that the launch has ~256 workgroups (one per CU), each with >= 4 K stages
    const int64_t target_wgs = [] {
        const char* e = dev_get(DEV_GEMM_WGS);
        const long v = e ? atol(e) : 0;
        return (int64_t)(v > 0 ? v : 256);   // one workgroup per CU (512 and 768 measured the same on C2)
    }();
    int64_t nsplit = 1;
    // skinny outputs (P <= 32 rows of MFMA work per tile) are bandwidth/latency bound, not MFMA bound: they want
    // several workgroups per CU in flight, so split K further
    const int64_t want_wgs = (P <= 32 && !symmetric) ? 4 * target_wgs : target_wgs;
    if (tiles < want_wgs && (tiles <= 64 || K < 64 * TK)) {
        nsplit = want_wgs / tiles;   // floor: never more workgroups than the target (no tail wave)
        if (nsplit < 1) nsplit = 1;
        const int64_t maxsplit = (K + 4 * TK - 1) / (4 * TK);
        if (nsplit > maxsplit) nsplit = maxsplit;
        if (nsplit < 1) nsplit = 1;
    } else if (tiles < 8 * want_wgs && K >= 64 * TK) {
        // 65 .. 2000 long tiles (136 for a 2048-column Gram, 528 for 4096 columns) quantise badly on 256 CUs x 2
        // resident workgroups (528 -> two rounds, the second one almost empty; 136 -> half of the CUs idle): cut K
        // until there are ~8 rounds worth of shorter work items
        nsplit = (8 * want_wgs + tiles - 1) / tiles;
        const int64_t maxsplit = (K + 32 * TK - 1) / (32 * TK);
        if (nsplit > maxsplit) nsplit = maxsplit;
        if (nsplit < 1) nsplit = 1;
    }
    int64_t kchunk = (K + nsplit - 1) / nsplit;
    kchunk = (kchunk + TK - 1) / TK * TK;
    if (kchunk < TK) kchunk = TK;
    nsplit = K > 0 ? (K + kchunk - 1) / kchunk : 1;
    if (nsplit == 1 && !symmetric)
        return launch_gemm(h, A_KC, B_KC, A, a_f32, lda, B, b_f32, ldb, C, c_f32, ldc, P, Q, K, 1, kchunk, 0, false, skip);
    // slabs (always fp64): nsplit x (P rows of Q contiguous)
    const int64_t slab_stride = P * Q;
    void* slab;
    TLSQ_TRY(ws_get(h, WS_SLAB, (size_t)(nsplit * slab_stride) * sizeof(double), &slab));
    TLSQ_TRY(launch_gemm(h, A_KC, B_KC, A, a_f32, lda, B, b_f32, ldb, slab, 0, Q, P, Q, K, (int)nsplit, kchunk,
                         slab_stride, symmetric, skip));
    int64_t g = (P * Q + 255) / 256;
    if (g > 2048) g = 2048;
    if (normblocks) *normblocks = (int)g;
    hipLaunchKernelGGL(k_slab_reduce, dim3((int)g), dim3(256), 0, h->stream, (const double*)slab, Q,
                       slab_stride, (int)nsplit, C, c_f32, ldc, P, Q, symmetric ? 1 : 0, skip, normpart, (int)nsplit, 0);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

int gemm_f64(Handle* h, bool A_KC, bool B_KC, const double* A, int64_t lda, const double* B,
             int64_t ldb, double* C, int64_t ldc, int64_t P, int64_t Q, int64_t K, bool symmetric) {
    return gemm_mixed(h, A_KC, B_KC, A, 0, lda, B, 0, ldb, C, 0, ldc, P, Q, K, symmetric);
}

// ---- Gram matrix of an fp32 panel on the fp32 MFMA, folded into fp64 every 64 rows (large mode, C5) -----------------
// v_mfma_f32_16x16x4_f32 runs at twice the rate of the fp64 form (32 cycles per 2048 flop).  A plain fp32 accumulation
// over M ~ 1e5 rows would lose every eigenvalue below ~4e-3 sigma_max^2 - the count sigma >= 1/mu is taken at
// ~5e-4 sigma_max late in a solve - so the fp32 accumulators only ever hold the sum over FFOLD = 2 stages of 32 rows: at the start
// of every second stage each tile's sum is converted and added to the fp64 accumulators (VALU, between the MFMAs) and the
// tile's first MFMA restarts from zero.  What is left is the rounding of the 64-term sums, ~2e-8 sigma_max^2 in
// norm at 65536 x 4096, i.e. singular values are resolved down to ~1.4e-4 sigma_max.
// Same tile / wave grid / software pipeline as gram_body (8 waves, wave tile 64 x 32, panels of 128 columns x 32 rows of
// fp32 in LDS, 16-byte fragment reads: lane (r, g) holds k = 16 q + 4 g + {0..3} of its column, element m feeds MFMA m).
// Diagonal tiles are computed in full (large N: 32 of 528 tiles at N = 4096).
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int FTK = 32;             // rows of Z per stage
constexpr int FFOLD = 2;            // stages per fp64 fold-in: every 64 rows (1: 11.3 ms per Gram at 65536 x 4096, error ~1e-8 sigma_max^2;
                                    // 2: 10.1 ms, ~2e-8; 4: 9.4 ms, ~4e-8 - too close to the last thresholds of a solve, ~2.5e-7 sigma_max^2)
constexpr int FLDK = FTK + 4;       // floats per panel column: (36 r + 4 g) distinct multiples of 4 mod 64 over a quarter wave
constexpr int FPANEL = 128 * FLDK;  // floats per panel (4 panels = 73.7 KB)

// (Zb, ldb, Nb: the matrix whose columns j0.. form the second operand - Z itself for the Gram matrix, the M x 128 fp32 panel
//  T for the skinny product Z'T of the large-mode operator, k_zt_f32mfma below)
template <bool FULL>
__device__ __forceinline__ void gram32_body(const float* __restrict__ Z, int64_t ld, double* __restrict__ Cz, int64_t ldc,
                                            int64_t N, int64_t kbeg, int64_t kend, int64_t i0, int64_t j0,
                                            float* __restrict__ smem, const float* __restrict__ Zb, int64_t ldb, int64_t Nb) {
    constexpr int SL = 2;   // 4-element slots per thread and panel (128 columns x 8 quads / 512 threads)
    const int nstage = (kend > kbeg) ? (int)((kend - kbeg + FTK - 1) / FTK) : 0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wj = w & 3, wi = w >> 2;
    const int fr = lane & 15, fk = lane >> 4;
    d4 acc[4][2];
    f4 a32[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
            a32[a][b] = f4{0.f, 0.f, 0.f, 0.f};
        }
    const float* pa[SL];
    const float* pb[SL];
    int so[SL], sr[SL], sk[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        const int e = tid + 512 * s;
        sr[s] = e >> 3;
        sk[s] = (e & 7) * 4;
        pa[s] = Z + kbeg + sk[s] + (i0 + sr[s]) * ld;
        pb[s] = Zb + kbeg + sk[s] + (j0 + sr[s]) * ldb;
        so[s] = sr[s] * FLDK + sk[s];
    }
    const int oa = (wi * 64 + fr) * FLDK + 4 * fk, ob = (wj * 32 + fr) * FLDK + 4 * fk;
    f4 ra[SL], rb[SL], ga[2][4], gb[2][2];

#define F_FR(buf, q, slot, i)                                                                                         \
    do {                                                                                                              \
        if ((i) < 4) ga[slot][(i) & 3] = *reinterpret_cast<const f4*>(smem + (buf) * FPANEL + oa + ((i) & 3) * 16 * FLDK + 16 * (q)); \
        else gb[slot][(i) & 1] = *reinterpret_cast<const f4*>(smem + (2 + (buf)) * FPANEL + ob + ((i) & 1) * 16 * FLDK + 16 * (q)); \
    } while (0)
#define F_SW(buf, i)                                                                              \
    do {                                                                                          \
        if ((i) < SL) *reinterpret_cast<f4*>(smem + (buf) * FPANEL + so[(i) % SL]) = ra[(i) % SL]; \
        else *reinterpret_cast<f4*>(smem + (2 + (buf)) * FPANEL + so[(i) % SL]) = rb[(i) % SL];    \
    } while (0)
#define F_GL(i, koff)                                                                                  \
    do {                                                                                               \
        const int u_ = (i) % SL;                                                                       \
        if (FULL) {                                                                                    \
            if ((i) < SL) ra[u_] = *reinterpret_cast<const f4*>(pa[u_] + (koff));                      \
            else rb[u_] = *reinterpret_cast<const f4*>(pb[u_] + (koff));                               \
        } else {                                                                                       \
            const int64_t k_ = kbeg + (koff) + sk[u_];                                                 \
            const int64_t c_ = ((i) < SL ? i0 : j0) + sr[u_];                                          \
            const float* p_ = ((i) < SL ? pa[u_] : pb[u_]) + (koff);                                   \
            f4 v_;                                                                                     \
            for (int x_ = 0; x_ < 4; ++x_) v_[x_] = (c_ < ((i) < SL ? N : Nb) && k_ + x_ < kend) ? p_[x_] : 0.f; \
            if ((i) < SL) ra[u_] = v_;                                                                 \
            else rb[u_] = v_;                                                                          \
        }                                                                                              \
    } while (0)
    // fold the fp32 sums of tile i (0..7) into the fp64 accumulators (the MFMA that follows restarts the tile from zero)
#define F_FOLD(i)                                                                          \
    do {                                                                                   \
        const int a_ = ((i) >> 1) & 3, b_ = (i) & 1;                                       \
        for (int x_ = 0; x_ < 4; ++x_) acc[a_][b_][x_] += (double)a32[a_][b_][x_];         \
    } while (0)
    // one half-stage (16 rows of Z): 32 MFMAs on fragment slot q, each followed by at most one memory instruction.
    // NEW: first half of a stage - every tile's first MFMA starts from zero, right after the tile's sums of the previous
    // stage have been folded (its last MFMA there is eight MFMAs back: no wait).
    // KIND 0: fragments q = 1 of the same buffer | 1: those + the panel store into the other buffer | 2: fragments q = 0
    // of the other buffer + global loads | 3: nothing
    auto half = [&](auto kind, auto fresh, int cur, int q, int64_t koff) {
        constexpr int KIND = decltype(kind)::value;
        constexpr bool NEW = decltype(fresh)::value;
        const int slot = q & 1, nslot = slot ^ 1;
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            const int m = t >> 3, a = (t >> 1) & 3, b = t & 1;
            if (NEW && m == 0) F_FOLD(t);
            const f4 c0 = (NEW && m == 0) ? f4{0.f, 0.f, 0.f, 0.f} : a32[a][b];
            a32[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[slot][a][m], gb[slot][b][m], c0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (KIND != 3 && t < 6) {
                if (KIND == 2) F_FR(cur ^ 1, 0, nslot, t);
                else F_FR(cur, 1, nslot, t);
            } else if (KIND == 1 && t - 6 < 2 * SL) {
                F_SW(cur ^ 1, t - 6);
            } else if (KIND == 2 && t - 6 < 2 * SL) {
                F_GL(t - 6, koff);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    std::integral_constant<int, 0> K0;
    std::integral_constant<int, 1> K1;
    std::integral_constant<int, 2> K2;
    std::integral_constant<int, 3> K3;
    std::true_type ZY;
    std::false_type ZN;

    if (nstage > 0) {
#pragma unroll
        for (int i = 0; i < 2 * SL; ++i) F_GL(i, (int64_t)0);
#pragma unroll
        for (int i = 0; i < 2 * SL; ++i) F_SW(0, i);
    }
    __syncthreads();
    if (nstage > 1) {
#pragma unroll
        for (int i = 0; i < 2 * SL; ++i) F_GL(i, (int64_t)FTK);
    }
    if (nstage > 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) F_FR(0, 0, 0, i);
    }
    for (int s = 0; s + 1 < nstage; ++s) {
        const int cur = s & 1;
        // (the last two stages load the final stage again instead of branching: those values are never stored)
        const int64_t koff = (int64_t)(s + 2 < nstage ? s + 2 : nstage - 1) * FTK;
        if ((s % FFOLD) == 0) half(K1, ZY, cur, 0, (int64_t)0);   // (fold + restart every FFOLD stages)
        else half(K1, ZN, cur, 0, (int64_t)0);
        __syncthreads();
        half(K2, ZN, cur, 1, koff);
    }
    if (nstage > 0) {
        const int cur = (nstage - 1) & 1;
        if (((nstage - 1) % FFOLD) == 0) half(K0, ZY, cur, 0, (int64_t)0);
        else half(K0, ZN, cur, 0, (int64_t)0);
        half(K3, ZN, cur, 1, (int64_t)0);
#pragma unroll
        for (int i = 0; i < 8; ++i) F_FOLD(i);
    }
#undef F_FR
#undef F_SW
#undef F_GL
#undef F_FOLD
    // epilogue: v_mfma_f32_16x16x4_f32 leaves column j = lane & 15, rows i = 4 (lane >> 4) + reg in a lane
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int64_t j = j0 + wj * 32 + b * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = i0 + wi * 64 + a * 16 + 4 * fk + r;
                if (FULL || (i < N && j < Nb)) Cz[j + i * ldc] = acc[a][b][r];
            }
        }
}

// work items: (K split z, lower-triangle tile incl. the diagonal, in the host's blocked order), z-major, XCD runs
template <bool ALLFULL>
__global__ __launch_bounds__(512) void k_gram_f32mfma(const float* __restrict__ Z, int64_t ld, double* __restrict__ slab,
                                                      int64_t ldc, int64_t N, int64_t K, int64_t kchunk, int64_t slab_stride,
                                                      int ntiles, int nsplit, int vec_ok, const int32_t* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) float smem[4 * FPANEL];
    const int64_t nwork = (int64_t)ntiles * nsplit;
    const int64_t cpx = (nwork + 7) / 8;
    const int64_t item = (int64_t)(blockIdx.x % 8) * cpx + (int64_t)(blockIdx.x / 8);
    if (item >= nwork) return;
    const int z = (int)(item / ntiles), t = (int)(item % ntiles);
    const int ti = order[2 * t], tj = order[2 * t + 1];
    const int64_t kbeg = (int64_t)z * kchunk;
    const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    const int64_t i0 = (int64_t)ti * TI, j0 = (int64_t)tj * TJ;
    double* __restrict__ Cz = slab + (int64_t)z * slab_stride;
    // (two kernels rather than one branch: register pressure)
    if (ALLFULL) gram32_body<true>(Z, ld, Cz, ldc, N, kbeg, kend, i0, j0, smem, Z, ld, N);
    else gram32_body<false>(Z, ld, Cz, ldc, N, kbeg, kend, i0, j0, smem, Z, ld, N);
}

// ---- the large-mode operator G X = Z'(Z X) for fp32 panels on the fp32 MFMA (the sketch of the randomized hook, C5) ----------
// Both halves at twice the rate of the fp64 MFMA the widening kernels (k_tsmm<float>, the 128 x 128-tile GEMM) run on:
//   T32 (M x lw, fp32) = Z X      k_tsmm_f32: the layout of k_tsmm (A fragments of Z straight from global memory, four waves
//                                 split K), v_mfma_f32_16x16x4_f32, the fp32 sums folded into fp64 every 64 columns of Z
//   Y (N x p, fp64)   = Z' T32    k_zt_f32mfma: the Gram kernel's pipeline with T32 as the second operand (one 128-column
//                                 tile of it), K split over the rows, fp64 fold-in every 32 rows, slabs + k_zt_reduce
// X is rounded to fp32 on the way in: 6e-8 relative, on top of the fp32 panel's own rounding.
__global__ __launch_bounds__(256) void k_pack_w_f32(const double* __restrict__ W, int64_t ldw, int K, int r, int lw,
                                                    float* __restrict__ Wt) {
    const int total = K * lw;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int j = e % lw, k = e / lw;
        Wt[e] = j < r ? (float)W[(size_t)k + (size_t)j * ldw] : 0.f;
    }
}

template <int NCT, int RT>
__global__ __launch_bounds__(256) void k_tsmm_f32(const float* __restrict__ Z, int64_t ldz, const float* __restrict__ Wt,
                                                  float* __restrict__ Tout, int64_t ldt, int64_t M, int K) {
    constexpr int LW = 16 * NCT;
    __shared__ double sR[4 * RT * NCT * 256];   // [w][t][c][reg][lane]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * (16 * RT);
    d4 acc[RT][NCT];
    f4 a32[RT][NCT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            acc[t][c] = d4{0.0, 0.0, 0.0, 0.0};
            a32[t][c] = f4{0.f, 0.f, 0.f, 0.f};
        }
    const float* zrow[RT];
    bool rok[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int64_t row = r0 + 16 * t + fr;
        rok[t] = row < M;
        zrow[t] = Z + (rok[t] ? row : 0);
    }
    const int nks = (K + 3) / 4;
    const int per = (nks + 3) / 4;
    const int ks0 = w * per, ks1 = (ks0 + per < nks) ? ks0 + per : nks;
    int trips = 0;
    for (int ksb = ks0; ksb < ks1; ksb += 4) {
        float fa[4][RT], fb[4][NCT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kg = (ksb + u) * 4 + fk;
            const bool kok = (ksb + u < ks1) && kg < K;
#pragma unroll
            for (int t = 0; t < RT; ++t) fa[u][t] = (rok[t] && kok) ? zrow[t][(int64_t)kg * ldz] : 0.f;
#pragma unroll
            for (int c = 0; c < NCT; ++c) fb[u][c] = kok ? Wt[(size_t)kg * LW + c * 16 + fr] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int c = 0; c < NCT; ++c)
                    a32[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][t], fb[u][c], a32[t][c], 0, 0, 0);
        if ((++trips & 3) == 0) {   // 64 columns of Z summed in fp32: fold
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int c = 0; c < NCT; ++c) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[t][c][q] += (double)a32[t][c][q];
                    a32[t][c] = f4{0.f, 0.f, 0.f, 0.f};
                }
        }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                sR[(((w * RT + t) * NCT + c) * 4 + q) * 64 + lane] = acc[t][c][q] + (double)a32[t][c][q];
    __syncthreads();
    for (int o = tid; o < RT * NCT * 256; o += 256) {
        const int l = o & 63, q = (o >> 6) & 3, tc = o >> 8;
        const int t = tc / NCT, c = tc % NCT;
        double sum = 0.0;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) sum += sR[(((ww * RT + t) * NCT + c) * 4 + q) * 64 + l];
        const int64_t row = r0 + 16 * t + 4 * (l >> 4) + q;   // v_mfma_f32_16x16x4_f32: column j = lane & 15, rows 4 (lane >> 4) + reg
        const int col = c * 16 + (l & 15);
        if (row < M) Tout[row + (int64_t)col * ldt] = (float)sum;
    }
}

// work items: (K split z, 128-column tile of Z), one 128 x 128 output tile each: slab[z][j + i 128] = (Z' T32)[i, j]
template <bool ALLFULL>
__global__ __launch_bounds__(512) void k_zt_f32mfma(const float* __restrict__ Z, int64_t ld, const float* __restrict__ T32,
                                                    int64_t ldt, double* __restrict__ slab, int64_t N, int64_t K,
                                                    int64_t kchunk, int64_t slab_stride, int ntiles, int nsplit) {
    __shared__ __attribute__((aligned(16))) float smem[4 * FPANEL];
    const int64_t nwork = (int64_t)ntiles * nsplit;
    const int64_t cpx = (nwork + 7) / 8;
    const int64_t item = (int64_t)(blockIdx.x % 8) * cpx + (int64_t)(blockIdx.x / 8);
    if (item >= nwork) return;
    const int z = (int)(item / ntiles), ti = (int)(item % ntiles);
    const int64_t kbeg = (int64_t)z * kchunk;
    const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    double* __restrict__ Cz = slab + (int64_t)z * slab_stride;
    if (ALLFULL) gram32_body<true>(Z, ld, Cz, 128, N, kbeg, kend, (int64_t)ti * TI, 0, smem, T32, ldt, 128);
    else gram32_body<false>(Z, ld, Cz, 128, N, kbeg, kend, (int64_t)ti * TI, 0, smem, T32, ldt, 128);
}

// Y[i + j ldy] = sum over the K splits of slab[z][j + i 128]  (i < N, j < p), fixed order
__global__ __launch_bounds__(256) void k_zt_reduce(const double* __restrict__ slab, int64_t slab_stride, int nsplit,
                                                   double* __restrict__ Y, int64_t ldy, int64_t N, int p) {
    const int64_t total = N * p;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e % N;
        const int j = (int)(e / N);
        double sacc = 0.0;
        for (int z = 0; z < nsplit; ++z) sacc += slab[(int64_t)z * slab_stride + j + i * 128];
        Y[i + (int64_t)j * ldy] = sacc;
    }
}

// Y (N x p, ld ldy, fp64) = Z'(Z X) for an fp32 panel Z (M x N, ld ldz) and X (N x p, ld ldx, fp64), p <= 96
int op_gram_f32(Handle* h, const float* Z, int64_t ldz, int64_t M, int64_t N, const double* X, int64_t ldx, double* Y,
                int64_t ldy, int64_t p) {
    if (p <= 0 || M <= 0) return TLSQ_OK;
    if (p > 96) return set_err(h, TLSQ_ERR_ARG, "op_gram_f32: p > 96");
    const int nct = (int)((p + 15) / 16), lw = 16 * nct;
    void *wt, *t32;
    TLSQ_TRY(ws_get(h, WS_OPW, (size_t)N * lw * 8, &wt));
    TLSQ_TRY(ws_get(h, WS_OPT, (size_t)M * 128 * 4 + 256, &t32));
    hipLaunchKernelGGL(k_pack_w_f32, dim3((unsigned)std::min<int64_t>((N * lw + 255) / 256, 1024)), dim3(256), 0, h->stream, X, ldx,
                       (int)N, (int)p, lw, (float*)wt);
    if (op_gram_f32_fast_ok(Z, ldz, M, N, p) && !dev_is(DEV_OPGRAM_OLD, '1')) {
        // third form (opgram16.hip: fp16 MFMA on operands split in registers) when the sweep that wrote the panel left its maximum
        // (Handle::absmax_panel); OPGRAM_H3=0: never, =2: always, with a pass for the maximum (tests)
        const bool force_h3 = dev_is(DEV_OPGRAM_H3, '2');
        if (ldz == M && !dev_is(DEV_OPGRAM_H3, '0') && (h->absmax_panel == (const void*)Z || force_h3)) {
            void* sc;
            TLSQ_TRY(ws_get(h, WS_H16S, 64, &sc));
            unsigned int* zmax = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(sc) + 40);
            if (h->absmax_panel != (const void*)Z) {
                zmax = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(sc) + 48);
                TLSQ_TRY(absmax_bits_f32(h, Z, M * N, zmax));
            }
            return op_gram_f32_h3(h, Z, ldz, M, N, X, ldx, (float*)t32, Y, ldy, p, zmax);
        }
        return op_gram_f32_fast(h, Z, ldz, M, N, (const float*)wt, (float*)t32, Y, ldy, p);
    }
    {
        const dim3 grid((unsigned)((M + 31) / 32));
#define TSF_LAUNCH(NC)                                                                                              \
    hipLaunchKernelGGL((k_tsmm_f32<NC, 2>), grid, dim3(256), 0, h->stream, Z, ldz, (const float*)wt, (float*)t32, M, M, (int)N)
        switch (nct) {
            case 1: TSF_LAUNCH(1); break;
            case 2: TSF_LAUNCH(2); break;
            case 3: TSF_LAUNCH(3); break;
            case 4: TSF_LAUNCH(4); break;
            case 5: TSF_LAUNCH(5); break;
            default: TSF_LAUNCH(6); break;
        }
#undef TSF_LAUNCH
    }
    TLSQ_HIP(h, hipGetLastError());
    // Z' T32: one 128-column tile of Z per work item, the rows split so that ~512 items (two rounds over the CUs) exist
    const int64_t ntiles = (N + TI - 1) / TI;
    int64_t nsplit = std::max<int64_t>(1, (512 + ntiles - 1) / ntiles);
    const int64_t maxsplit = std::max<int64_t>(1, (M + 8 * FTK - 1) / (8 * FTK));
    nsplit = std::min(nsplit, maxsplit);
    int64_t kchunk = (M + nsplit - 1) / nsplit;
    kchunk = (kchunk + FTK - 1) / FTK * FTK;
    nsplit = (M + kchunk - 1) / kchunk;
    const int64_t slab_stride = N * 128;
    void* slab;
    TLSQ_TRY(ws_get(h, WS_SLAB, (size_t)(nsplit * slab_stride) * 8, &slab));
    const int64_t nwork = ntiles * nsplit, cpx = (nwork + 7) / 8;
    const bool vec_ok = (ldz % 4) == 0 && (M % 4) == 0 && (reinterpret_cast<uintptr_t>(Z) % 16) == 0;
    const bool allfull = vec_ok && (N % TI) == 0 && (M % FTK) == 0 && (kchunk % FTK) == 0;
    if (allfull)
        hipLaunchKernelGGL(k_zt_f32mfma<true>, dim3((unsigned)(8 * cpx)), dim3(512), 0, h->stream, Z, ldz, (const float*)t32, M,
                           (double*)slab, N, M, kchunk, slab_stride, (int)ntiles, (int)nsplit);
    else
        hipLaunchKernelGGL(k_zt_f32mfma<false>, dim3((unsigned)(8 * cpx)), dim3(512), 0, h->stream, Z, ldz, (const float*)t32, M,
                           (double*)slab, N, M, kchunk, slab_stride, (int)ntiles, (int)nsplit);
    hipLaunchKernelGGL(k_zt_reduce, dim3((unsigned)std::min<int64_t>((N * p + 255) / 256, 2048)), dim3(256), 0, h->stream,
                       (const double*)slab, slab_stride, (int)nsplit, Y, ldy, N, (int)p);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// the lower-triangle 128 x 128 tiles (diagonal ones included) of an nti x nti tile grid as pairs (ti, tj) in blocked order -
// 8 x 4 tiles share 8 + 4 panels, so a run of work items on one XCD finds them in its L2 -, on the device (kept per handle)
int gram_tile_table(Handle* h, int64_t nti, const int32_t** tab_out) {
    const int64_t ntiles = nti * (nti + 1) / 2;
    void* tab;
    TLSQ_TRY(ws_get(h, WS_GRAMTAB2, (size_t)ntiles * 8, &tab));
    if (h->gram_tab2_nti != nti) {
        std::vector<int32_t> o;
        o.reserve((size_t)ntiles * 2);
        for (int64_t I = 0; I < nti; I += 8)
            for (int64_t J = 0; J < std::min(I + 8, nti); J += 4)
                for (int64_t ti = I; ti < std::min(I + 8, nti); ++ti)
                    for (int64_t tj = J; tj < std::min(J + 4, ti + 1); ++tj) {
                        o.push_back((int32_t)ti);
                        o.push_back((int32_t)tj);
                    }
        if ((int64_t)o.size() != 2 * ntiles) return set_err(h, TLSQ_ERR_UNSUPPORTED, "gram: tile order table");
        TLSQ_HIP(h, hipMemcpyAsync(tab, o.data(), o.size() * 4, hipMemcpyHostToDevice, h->stream));
        TLSQ_HIP(h, hipStreamSynchronize(h->stream));
        h->gram_tab2_nti = nti;
    }
    *tab_out = (const int32_t*)tab;
    return TLSQ_OK;
}

static int gram_f32mfma(Handle* h, const float* Z, int64_t ld, double* G, int64_t ldg, int64_t N, int64_t K) {
    // (aligned tall panels: the fp16 MFMA on the two-plane split of the panel, gram16.hip - 3.4x faster at 65536 x 4096)
    if (gram_h3_ok(Z, ld, N, K) && !dev_is(DEV_GRAM_H3, '0')) return gram_h3(h, Z, ld, G, ldg, N, K);
    const int64_t nti = (N + TI - 1) / TI, ntiles = nti * (nti + 1) / 2;
    static const int64_t target_wgs = 256;
    // uniform K split: rounds of one item per CU, + a quarter item for the drift, + the slab traffic (see gram_kc)
    constexpr double c_row = 0.060, c_item = 18.0, slab_us_per_byte = 2.0 / 4.0e6;
    const int64_t maxsplit = std::max<int64_t>(1, (K + 4 * FTK - 1) / (4 * FTK));
    const int64_t memsplit = std::max<int64_t>(1, (int64_t)(((size_t)2 << 30) / ((size_t)N * N * 8)));
    int64_t nsplit = 1, kchunk = 0;
    double best = 1e300;
    for (int64_t ns0 = 1; ns0 <= std::min<int64_t>(std::min(maxsplit, memsplit), 1024); ++ns0) {
        int64_t kc = (K + ns0 - 1) / ns0;
        kc = (kc + FTK - 1) / FTK * FTK;
        const int64_t ns = (K + kc - 1) / kc;
        const double t_item = kc * c_row + c_item;
        const double rounds = std::ceil((double)(ntiles * ns) / (double)target_wgs);
        const double t = rounds * t_item + (ntiles * ns > target_wgs ? 0.25 * t_item : 0.0) +
                         (double)(ntiles * ns) * (double)(TI * TJ * 8) * slab_us_per_byte;
        if (t < best) {
            best = t;
            nsplit = ns;
            kchunk = kc;
        }
    }
    const int64_t slab_stride = N * N;
    void* slab;
    const int32_t* tab = nullptr;
    TLSQ_TRY(ws_get(h, WS_SLAB, (size_t)(nsplit * slab_stride) * sizeof(double), &slab));
    TLSQ_TRY(gram_tile_table(h, nti, &tab));
    const int64_t nwork = ntiles * nsplit, cpx = (nwork + 7) / 8;
    if (8 * cpx > 2147483647LL) return set_err(h, TLSQ_ERR_UNSUPPORTED, "gram: grid too large");
    const int vec_ok = ((ld % 4) == 0 && (reinterpret_cast<uintptr_t>(Z) % 16) == 0) ? 1 : 0;
    // every tile inside the matrix, every chunk whole stages, aligned 16-byte loads: the unguarded kernel
    const bool allfull = vec_ok && (N % TI) == 0 && (K % FTK) == 0;
    if (allfull)
        hipLaunchKernelGGL(k_gram_f32mfma<true>, dim3((unsigned)(8 * cpx)), dim3(512), 0, h->stream, Z, ld, (double*)slab, N, N, K,
                           kchunk, slab_stride, (int)ntiles, (int)nsplit, vec_ok, tab);
    else
        hipLaunchKernelGGL(k_gram_f32mfma<false>, dim3((unsigned)(8 * cpx)), dim3(512), 0, h->stream, Z, ld, (double*)slab, N, N, K,
                           kchunk, slab_stride, (int)ntiles, (int)nsplit, vec_ok, tab);
    TLSQ_HIP(h, hipGetLastError());
    int64_t g = (N * N + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_slab_reduce, dim3((int)g), dim3(256), 0, h->stream, (const double*)slab, N, slab_stride, (int)nsplit,
                       (void*)G, 0, ldg, N, N, 1, (const double*)nullptr, (double*)nullptr, (int)nsplit, 0);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// G (fp64, N x N) = Z'Z for Z of either precision
int gram_any(Handle* h, const void* Z, int z_f32, int64_t M, int64_t N, int64_t ldZ, double* G, int64_t ldG, int mfma32) {
    if (M <= 0) {
        TLSQ_HIP(h, hipMemset2DAsync(G, ldG * sizeof(double), 0, N * sizeof(double), N, h->stream));
        return TLSQ_OK;
    }
    // fp32 panels in large mode (no dense eigen-solver behind the count anyway): the fp32 MFMA with fp64 fold-in
    const bool no_f32mfma = dev_is(DEV_GRAM_F32MFMA, '0');
    const bool all_f32mfma = dev_is(DEV_GRAM_F32MFMA, '2');
    if (z_f32 && (mfma32 == 1 || (mfma32 < 0 && !no_f32mfma && (N > 2048 || all_f32mfma))))
        return gram_f32mfma(h, (const float*)Z, ldZ, G, ldG, N, M);
    return gemm_mixed(h, true, true, Z, z_f32, ldZ, Z, z_f32, ldZ, G, 0, ldG, N, N, M, true);
}

int gram_f64(Handle* h, const double* Z, int64_t M, int64_t N, int64_t ldZ, double* G, int64_t ldG) {
    return gram_any(h, Z, 0, M, N, ldZ, G, ldG);
}

// ---- tall-skinny product T (M x r) = Z (M x K) * W (K x r), r <= 32 -----------------------------------------
// The factor GEMM of the rebuild (src/robustPCA.jl:205-213: U[:,1:svp] * S = Z * V_svp * diag(g)).  On the
// 128 x 128-tile kernel above a 16-column output wastes 7/8 of the MFMA work and the panel streams at 2.3 TB/s.  Here
// a workgroup owns 16*RT rows, its four waves split K, and every wave feeds v_mfma_f64_16x16x4_f64 straight from
// global memory: the A fragment is 16 consecutive rows x 4 columns of Z (four full 128-byte lines per load), the B
// fragment 4 consecutive rows of the row-major, zero-padded copy of W (512 contiguous bytes, L2 resident).  No LDS
// and no barrier in the main loop; the four partial tiles meet in LDS at the end and are added in a fixed order.
// Z is read exactly once.
__global__ __launch_bounds__(256) void k_pack_w(const double* __restrict__ W, int64_t ldw, int K, int r, int lw,
                                                double* __restrict__ Wt) {
    const int total = K * lw;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int j = e % lw, k = e / lw;
        Wt[e] = j < r ? W[(size_t)k + (size_t)j * ldw] : 0.0;
    }
}

// Vs (K x r, ld K) = V[:, sel] and Wt (row-major K x lw, zero-padded) = the packed form of V[:, sel] diag(w) in one pass:
// what the rebuild needs from the eigenvectors (the factor product's operand and the sweep's Vs tile) without the
// intermediate Vg panel and its re-packing
__global__ __launch_bounds__(256) void k_gather_pack_w(const double* __restrict__ V, int K, SelWeights sw, int r, int lw,
                                                       double* __restrict__ Vs, double* __restrict__ Wt) {
    const int total = K * lw;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int j = e % lw, k = e / lw;
        double v = 0.0;
        if (j < r) {
            v = V[(size_t)sw.sel[j] * K + k];
            if (Vs) Vs[(size_t)j * K + k] = v;
            v *= sw.w[j];
        }
        Wt[e] = v;
    }
}

// The same product with W = V[:, sel] diag(w) read straight from V (K x ., ld K, K a multiple of 4): no packed copy and no
// second launch. Each lane reads four consecutive k of its column at once (k = 16*trip + 4*fk + u for the MFMA of step u; the
// sum over k does not care about the assignment as long as both operands use it), so a trip touches as many cache lines
// of V as it would of the packed panel. Vs (K x r) receives V[:, sel], spread over the first workgroups.
template <typename TA, int NCT, int RT, bool DEVLIST = false>
__global__ __launch_bounds__(256) void k_tsmm_selv(const TA* __restrict__ Z, int64_t ldz, const double* __restrict__ V,
                                                   SelWeights sw_arg, double* __restrict__ Vs, double* __restrict__ Tout,
                                                   int64_t ldt, int64_t M, int K, int r, const SpecCtrl* __restrict__ ctrl) {
    __shared__ double sR[4 * RT * NCT * 256];   // [w][t][c][reg][lane]
    // DEVLIST: selection, weights and r come from device memory (written by k_ritz_finish one launch earlier); nothing to do
    // when that kernel did not vouch for them or the count needs more accumulator tiles than this instance has
    if constexpr (DEVLIST) {
        if (ctrl->ok == 0 || ctrl->r > 16 * NCT) return;
        r = ctrl->r;
    }
    const SelWeights& sw = DEVLIST ? ctrl->sw : sw_arg;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * (16 * RT);
    if (Vs)
        for (int e = blockIdx.x * 256 + tid; e < K * r; e += gridDim.x * 256) Vs[e] = V[(size_t)sw.sel[e / K] * K + (e % K)];
    d4 acc[RT][NCT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[t][c] = d4{0.0, 0.0, 0.0, 0.0};
    const TA* zrow[RT];
    bool rok[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int64_t row = r0 + 16 * t + fr;
        rok[t] = row < M;
        zrow[t] = Z + (rok[t] ? row : 0);
    }
    const double* vcol[NCT];
    double wj[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        const int j = c * 16 + fr;
        vcol[c] = V + (size_t)(j < r ? sw.sel[j] : 0) * K;
        wj[c] = j < r ? sw.w[j] : 0.0;
    }
    const int nks = K / 4;
    const int per = (nks + 3) / 4;
    const int ks0 = w * per, ks1 = (ks0 + per < nks) ? ks0 + per : nks;
    for (int ksb = ks0; ksb < ks1; ksb += 4) {
        const bool kok = ksb + fk < ks1;
        const int kb = (ksb + fk) * 4;
        double fa[4][RT];
        d4 fb[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) fb[c] = kok ? *(const d4*)(vcol[c] + kb) : d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < RT; ++t) fa[u][t] = (rok[t] && kok) ? (double)zrow[t][(int64_t)(kb + u) * ldz] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int c = 0; c < NCT; ++c)
                    acc[t][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[u][t], fb[c][u] * wj[c], acc[t][c], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) sR[(((w * RT + t) * NCT + c) * 4 + q) * 64 + lane] = acc[t][c][q];
    __syncthreads();
    for (int o = tid; o < RT * NCT * 256; o += 256) {
        const int l = o & 63, q = (o >> 6) & 3, tc = o >> 8;
        const int t = tc / NCT, c = tc % NCT;
        double sum = 0.0;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) sum += sR[(((ww * RT + t) * NCT + c) * 4 + q) * 64 + l];
        const int64_t row = r0 + 16 * t + (l >> 4) + 4 * q;
        const int col = c * 16 + (l & 15);
        if (row < M && col < r) Tout[row + (int64_t)col * ldt] = sum;
    }
}

template <typename TA, int NCT, int RT>
__global__ __launch_bounds__(256) void k_tsmm(const TA* __restrict__ Z, int64_t ldz, const double* __restrict__ Wt,
                                              double* __restrict__ Tout, int64_t ldt, int64_t M, int K, int r) {
    constexpr int LW = 16 * NCT;
    __shared__ double sR[4 * RT * NCT * 256];   // [w][t][c][reg][lane]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * (16 * RT);
    d4 acc[RT][NCT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[t][c] = d4{0.0, 0.0, 0.0, 0.0};
    const TA* zrow[RT];
    bool rok[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int64_t row = r0 + 16 * t + fr;
        rok[t] = row < M;
        zrow[t] = Z + (rok[t] ? row : 0);
    }
    // this wave's share of K, in whole k-steps of 4
    const int nks = (K + 3) / 4;
    const int per = (nks + 3) / 4;
    const int ks0 = w * per, ks1 = (ks0 + per < nks) ? ks0 + per : nks;
    // four k-steps per trip: all 4*(RT+NCT) loads are issued before the first MFMA consumes one
    for (int ksb = ks0; ksb < ks1; ksb += 4) {
        double fa[4][RT], fb[4][NCT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kg = (ksb + u) * 4 + fk;
            const bool kok = (ksb + u < ks1) && kg < K;
#pragma unroll
            for (int t = 0; t < RT; ++t) fa[u][t] = (rok[t] && kok) ? (double)zrow[t][(int64_t)kg * ldz] : 0.0;
#pragma unroll
            for (int c = 0; c < NCT; ++c) fb[u][c] = kok ? Wt[(size_t)kg * LW + c * 16 + fr] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int c = 0; c < NCT; ++c)
                    acc[t][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[u][t], fb[u][c], acc[t][c], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) sR[(((w * RT + t) * NCT + c) * 4 + q) * 64 + lane] = acc[t][c][q];
    __syncthreads();
    for (int o = tid; o < RT * NCT * 256; o += 256) {
        const int l = o & 63, q = (o >> 6) & 3, tc = o >> 8;
        const int t = tc / NCT, c = tc % NCT;
        double sum = 0.0;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) sum += sR[(((ww * RT + t) * NCT + c) * 4 + q) * 64 + l];
        const int64_t row = r0 + 16 * t + (l >> 4) + 4 * q;   // lane holds column j = lane&15, rows (lane>>4) + 4*reg
        const int col = c * 16 + (l & 15);
        if (row < M && col < r) Tout[row + (int64_t)col * ldt] = sum;
    }
}

// T (M x r, ldt, fp64) = Z (M x K, ldz; fp32 when z_f32) * W (K x r, ldw), r <= 96.
// W == nullptr: W = V[:, sel] diag(w) for the K x . matrix V (ld K) and a short selection (r <= 32) passed by value; Vs
// (optional, K x r, ld K) receives V[:, sel] on the way (tsmm_sel below).
static int tsmm_impl(Handle* h, const void* Z, int z_f32, int64_t ldz, const double* W, int64_t ldw, const double* V,
                     const SelWeights* sw, double* Vs, double* Tout, int64_t ldt, int64_t M, int64_t K, int64_t r) {
    if (M <= 0 || r <= 0) return TLSQ_OK;
    if (r > 96) return set_err(h, TLSQ_ERR_ARG, "tsmm: r > 96");
    const int nct = (int)((r + 15) / 16);
    const int lw = 16 * nct;
    if (!W && (K & 3) == 0 && nct <= 2 && !dev_is(DEV_NO_TSMM_SELV, '1')) {
        const bool tall = M >= 65536;
        const dim3 grid((unsigned)((M + (tall ? 64 : 32) - 1) / (tall ? 64 : 32)));
#define TSV_LAUNCH(TA, NC, RTT)                                                                                     \
    hipLaunchKernelGGL((k_tsmm_selv<TA, NC, RTT>), grid, dim3(256), 0, h->stream, (const TA*)Z, ldz, V, *sw, Vs, Tout, ldt, \
                       M, (int)K, (int)r, (const SpecCtrl*)nullptr)
#define TSV_TYPE(TA)                                                              \
    if (nct == 1) { if (tall) TSV_LAUNCH(TA, 1, 4); else TSV_LAUNCH(TA, 1, 2); } \
    else { if (tall) TSV_LAUNCH(TA, 2, 4); else TSV_LAUNCH(TA, 2, 2); }
        if (z_f32) { TSV_TYPE(float) } else { TSV_TYPE(double) }
#undef TSV_TYPE
#undef TSV_LAUNCH
        TLSQ_HIP(h, hipGetLastError());
        return TLSQ_OK;
    }
    void* wt;
    TLSQ_TRY(ws_get(h, WS_OPW, (size_t)K * lw * 8, &wt));
    if (W)
        hipLaunchKernelGGL(k_pack_w, dim3((unsigned)std::min<int64_t>((K * lw + 255) / 256, 1024)), dim3(256), 0, h->stream, W,
                           ldw, (int)K, (int)r, lw, (double*)wt);
    else
        hipLaunchKernelGGL(k_gather_pack_w, dim3((unsigned)std::min<int64_t>((K * lw + 255) / 256, 1024)), dim3(256), 0,
                           h->stream, V, (int)K, *sw, (int)r, lw, Vs, (double*)wt);
    // rows per workgroup: 64 for tall panels, 32 otherwise (fewer than ~1000 workgroups would leave CUs idle); the
    // wide forms (more than 32 columns) keep 16 rows so that the accumulators and the reduction buffer stay small
    const bool tall = M >= 65536;
    const int rt = nct > 2 ? 1 : (tall ? 4 : 2);
    const dim3 grid((unsigned)((M + 16 * rt - 1) / (16 * rt)));
#define TS_LAUNCH(TA, NC, RTT)                                                                                      \
    hipLaunchKernelGGL((k_tsmm<TA, NC, RTT>), grid, dim3(256), 0, h->stream, (const TA*)Z, ldz, (const double*)wt,    \
                       Tout, ldt, M, (int)K, (int)r)
#define TS_TYPE(TA)                                                  \
    switch (nct) {                                                   \
        case 1: if (tall) TS_LAUNCH(TA, 1, 4); else TS_LAUNCH(TA, 1, 2); break; \
        case 2: if (tall) TS_LAUNCH(TA, 2, 4); else TS_LAUNCH(TA, 2, 2); break; \
        case 3: TS_LAUNCH(TA, 3, 1); break;                          \
        case 4: TS_LAUNCH(TA, 4, 1); break;                          \
        case 5: TS_LAUNCH(TA, 5, 1); break;                          \
        default: TS_LAUNCH(TA, 6, 1); break;                         \
    }
    if (z_f32) { TS_TYPE(float) } else { TS_TYPE(double) }
#undef TS_TYPE
#undef TS_LAUNCH
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

int tsmm_mixed(Handle* h, const void* Z, int z_f32, int64_t ldz, const double* W, int64_t ldw, double* Tout, int64_t ldt,
               int64_t M, int64_t K, int64_t r) {
    return tsmm_impl(h, Z, z_f32, ldz, W, ldw, nullptr, nullptr, nullptr, Tout, ldt, M, K, r);
}

int tsmm_sel(Handle* h, const void* Z, int z_f32, int64_t ldz, const double* V, const SelWeights& sw, double* Vs, double* Tout,
             int64_t ldt, int64_t M, int64_t K, int64_t r) {
    if (r > 32) return set_err(h, TLSQ_ERR_ARG, "tsmm_sel: r > 32");
    return tsmm_impl(h, Z, z_f32, ldz, nullptr, 0, V, &sw, Vs, Tout, ldt, M, K, r);
}

int tsmm_sel_dev(Handle* h, const void* Z, int z_f32, int64_t ldz, const double* V, const SpecCtrl* ctrl, int nct, double* Vs,
                 double* Tout, int64_t ldt, int64_t M, int64_t K) {
    if ((K & 3) != 0 || nct < 1 || nct > 2 || !ctrl) return set_err(h, TLSQ_ERR_ARG, "tsmm_sel_dev: bad argument");
    const bool tall = M >= 65536;
    const dim3 grid((unsigned)((M + (tall ? 64 : 32) - 1) / (tall ? 64 : 32)));
    const SelWeights none{};
#define TSD_LAUNCH(TA, NC, RTT)                                                                                       \
    hipLaunchKernelGGL((k_tsmm_selv<TA, NC, RTT, true>), grid, dim3(256), 0, h->stream, (const TA*)Z, ldz, V, none, Vs, Tout, \
                       ldt, M, (int)K, 0, ctrl)
#define TSD_TYPE(TA)                                                              \
    if (nct == 1) { if (tall) TSD_LAUNCH(TA, 1, 4); else TSD_LAUNCH(TA, 1, 2); } \
    else { if (tall) TSD_LAUNCH(TA, 2, 4); else TSD_LAUNCH(TA, 2, 2); }
    if (z_f32) { TSD_TYPE(float) } else { TSD_TYPE(double) }
#undef TSD_TYPE
#undef TSD_LAUNCH
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

// Y (N x pc, ld N) = Z' * T for a few columns (pc <= 8): one wave per column of Z (coalesced reads), T (M x pc) from L2.
// The large-mode operator product G X = Z'(Z X) for Lanczos vectors and other narrow blocks (the 128 x 128-tile kernel
// would spend 128/pc times the necessary MFMA work here).
template <typename TA, int PC>
__global__ __launch_bounds__(256) void k_zt_small(const TA* __restrict__ Z, int64_t ldz, const double* __restrict__ Tm,
                                                  int64_t ldt, double* __restrict__ Y, int64_t ldy, int64_t M, int N,
                                                  int pc) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const TA* z = Z + (int64_t)n * ldz;
    double acc[PC];
#pragma unroll
    for (int c = 0; c < PC; ++c) acc[c] = 0.0;
    for (int64_t m = lane; m < M; m += 64) {
        const double zv = (double)z[m];
#pragma unroll
        for (int c = 0; c < PC; ++c)
            if (c < pc) acc[c] += zv * Tm[m + (int64_t)c * ldt];
    }
#pragma unroll
    for (int c = 0; c < PC; ++c) {
        if (c < pc) {
            double v = acc[c];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) Y[n + (int64_t)c * ldy] = v;
        }
    }
}

// Y (N x p, ld N) = Z' * T,  Z: M x N (ldz, fp32 when z_f32), T: M x p (ldt, fp64)
int ztmm_mixed(Handle* h, const void* Z, int z_f32, int64_t ldz, const double* Tm, int64_t ldt, double* Y, int64_t ldy,
               int64_t M, int64_t N, int64_t p) {
    if (p <= 0 || N <= 0) return TLSQ_OK;
    if (p > 8)   // C[j + i*ldc] = sum_k A(i,k) B(j,k): A = T (P = p), B = Z (Q = N), K = M
        return gemm_mixed(h, true, true, Tm, 0, ldt, Z, z_f32, ldz, Y, 0, ldy, p, N, M, false);
    const dim3 grid((unsigned)((N + 3) / 4));
    if (z_f32) {
        if (p == 1) hipLaunchKernelGGL((k_zt_small<float, 1>), grid, dim3(256), 0, h->stream, (const float*)Z, ldz, Tm, ldt, Y, ldy, M, (int)N, (int)p);
        else hipLaunchKernelGGL((k_zt_small<float, 8>), grid, dim3(256), 0, h->stream, (const float*)Z, ldz, Tm, ldt, Y, ldy, M, (int)N, (int)p);
    } else {
        if (p == 1) hipLaunchKernelGGL((k_zt_small<double, 1>), grid, dim3(256), 0, h->stream, (const double*)Z, ldz, Tm, ldt, Y, ldy, M, (int)N, (int)p);
        else hipLaunchKernelGGL((k_zt_small<double, 8>), grid, dim3(256), 0, h->stream, (const double*)Z, ldz, Tm, ldt, Y, ldy, M, (int)N, (int)p);
    }
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

}  // namespace tlsq
