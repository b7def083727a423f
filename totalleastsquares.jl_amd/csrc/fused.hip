// The E-free sweep (sweeps.hip, k_zsweep: /root/reference/src/robustPCA.jl:205-213 product, :217-223 of iteration k,
// :188-192 of iteration k + 1) and the Gram matrix of the Z_{k+1} it writes (the gesdd work of :194) in ONE kernel.
//
// Why: at N = 256 / 512 the sweep is HBM-bound and the Gram of its result MFMA-bound, each alone takes about the same time,
// and the Gram reads back (once per 128-column tile row) what the sweep has just written.  Here the rows of Z_{k+1} go from
// the sweep's registers into LDS as MFMA operands while they are on their way to memory: the Gram reads nothing from HBM,
// and the matrix pipe works under the memory stream instead of after it.
//
// Work item = a contiguous chunk of rows x a set of columns, one workgroup per CU (12 waves):
//   * waves 8..11 ("sweep waves", one per SIMD) stream the chunk in stages of 16 rows: thread = row pair x every 32nd
//     column, the same statements in the same order as k_zsweep (bit-identical Y_{k+1}, Z_{k+1}, R_k), loads kept three
//     columns ahead in a register ring, A_k = T Vs' from the factors (T stage and Vs in LDS), Z_{k+1} stored to memory and,
//     as a [column][k] panel with an XOR swizzle of k, to one of two LDS stage buffers;
//   * waves 0..7 ("MFMA waves", two per SIMD) accumulate Z_{k+1}' Z_{k+1} of the previous stage from the other buffer:
//     v_mfma_f64_16x16x4_f64, 17 (16) accumulator tiles per wave, fragments by conflict-free ds_read_b64;
//   * one s_barrier per stage (LDS traffic only: the global loads stay in flight across it).
// N = 256: every workgroup owns all 136 tiles of the lower triangle (16 x 16 tiles of 16 x 16).  N = 512: four kinds of
// workgroup per row chunk - two "diagonal" ones (columns 0..255 / 256..511: they store Y, Z, R and own the 136 tiles of their
// diagonal block) and two "rectangular" ones (columns 0..255 + one 128-column block of the other half, 8 x 16 tiles of
// the off-diagonal block; they sweep their 384 columns again but store nothing) - dealt to the XCDs so that the four kinds
// of a chunk share one L2 and march through the same rows together.  The partial Gram matrices go to split-K slabs in
// the layout of gemm.hip's k_gram_kc and are summed in a fixed order by its k_slab_reduce (deterministic).
//
// Floating-point contraction is OFF in this file for the same reason as in sweeps.hip: the sweep statements are compared
// bit for bit with the oracle's.
#include "common.hpp"

#pragma clang fp contract(off)

namespace tlsq {

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ T fz_pos(T a) {   // max(a, 0) with Julia's NaN propagation (sweeps.hip: pos_part)
    return (a > T(0) || a != a) ? a : T(0);
}
template <typename T>
__device__ __forceinline__ T fz_neg(T b) {
    return (b < T(0) || b != b) ? b : T(0);
}
template <typename T>
__device__ __forceinline__ T fz_soft(T x, T e) {   // src/robustPCA.jl:1
    return fz_pos(x - e) + fz_neg(x + e);
}

constexpr int FZ_R = 16;                      // rows of the panels per stage
constexpr int FZ_MW = 8, FZ_SW = 4;           // MFMA waves, sweep waves
constexpr int FZ_THREADS = 64 * (FZ_MW + FZ_SW);
constexpr int FZ_PF = 3, FZ_RS = 4;           // columns a sweep thread keeps in flight ahead of the one it works on / ring slots

struct FusedArgs {
    const double* D;      // the panel D (ld), or the series y of an implicit Hankel D (HK)
    const double* Tm;     // M x r, ld M
    const double* Vs;     // N x r, ld N
    const double* Yin;
    double* Yout;
    const double* Zin;
    double* Zout;
    double* R;            // may be nullptr
    int64_t M, ld;
    int N, r;
    double mu, inv_mu, inv_mu_n, thr_n;
    int nonnegA, nonnegE;
    double* sumsq;        // 72 doubles (k_zsweep): [0, 64) partial sums of ||R_k||_F^2, [64 + maxslot] max |R_k[i, j]|
    double* zero_slots;
    int maxslot;
    double* slab;         // nz slabs of N x N (ld N)
    int64_t kchunk;       // rows per chunk (a multiple of 16)
    int nz;
    int64_t hankel_K;
    HankelGeom hg;
    int ablate;           // development (FUSED_ABLATE): 1 no MFMAs, 2 no global stores, 4 no factor product, 8 no global loads
};

// one barrier of the workgroup that waits for this wave's LDS traffic only: __syncthreads() would also drain the global
// loads the sweep waves keep in flight across the stage boundary
__device__ __forceinline__ void fz_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS (doubles): two stage buffers [column][16 k, swizzled], Vs [column][RMAX + 2], two T stages [i][16 rows], reductions
template <int NCL, int RMAX>
struct FzLds {
    static constexpr int ZS = NCL * FZ_R;
    static constexpr int VSP = RMAX + 2;
    static constexpr int VS = NCL * VSP;
    static constexpr int TS = RMAX * FZ_R;
    static constexpr int RED = 16;
    static constexpr int TOTAL = 2 * ZS + VS + 2 * TS + RED;
};

// NCL local columns in 128-column blocks, block lb at global column gcb[lb].  RECT: the tiles of local columns [256, 384) x
// [0, 256) (8 x 16); otherwise the lower triangle of local columns [0, 256) (136 tiles).  store: this workgroup writes
// Y_{k+1}, Z_{k+1}, R_k of its columns and accounts their residual sums.
template <int NCL, bool RECT, int RMAX, bool HK>
__device__ __forceinline__ void fused_body(const FusedArgs& P, int64_t kbeg, int64_t kend, int gcb0, int gcb1, int gcb2,
                                           bool store, double* __restrict__ Cz, double* __restrict__ smem) {
    using L = FzLds<NCL, RMAX>;
    constexpr int J = NCL / 32;    // columns per sweep thread and stage
    static_assert(J % FZ_RS == 0, "ring slots are static");
    double* const Zs = smem;
    double* const sVs = smem + 2 * L::ZS;
    double* const sT = sVs + L::VS;
    double* const sRed = sT + 2 * L::TS;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int nst = (int)((kend - kbeg + FZ_R - 1) / FZ_R);
    const int64_t M = P.M, ld = P.ld;
    const int r = P.r;
    // (local 128-column blocks 0 and 1 are adjacent in every kind of workgroup; plain arithmetic - a table indexed at run
    //  time would live in scratch memory)
    auto gcb = [&](int lb) { return lb < 2 ? gcb0 + 128 * lb : gcb2; };
    (void)gcb1;

    // ---- Vs -> LDS (all waves), zero beyond r -------------------------------------------------------------------------
    for (int e = tid; e < NCL * RMAX; e += FZ_THREADS) {
        const int c = e % NCL, i = e / NCL;
        const int gc = gcb(c >> 7) + (c & 127);
        sVs[c * L::VSP + i] = i < r ? P.Vs[(size_t)gc + (size_t)i * P.N] : 0.0;
    }

    if (wave < FZ_MW) {
        // =================================== MFMA waves ===================================
        const int fr = lane & 15, fk = lane >> 4;
        const int swz = 2 * ((fr >> 1) & 7);
        int lq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) lq[q] = fr * FZ_R + ((4 * q + fk) ^ swz);
        constexpr int NT = RECT ? 16 : 17;
        d4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
        // tile of position t: rows (A operand) from fragment a(t), columns (B operand) from fragment b(t)
        //   triangle: wave w owns tile rows 15 - w (columns 0..15 - w: positions 0..15 - w) and w (columns 0..w: the rest)
        //   rectangle: wave w owns tile row 16 + w, columns 0..15
        const int hi_row = RECT ? 16 + wave : 15 - wave, lo_row = wave, nhi = RECT ? 16 : 16 - wave;
        fz_barrier();   // B0: Vs, T stage 0
        fz_barrier();   // B1: stage 0 in Zs[0]
        for (int s = 0; s < nst; ++s) {
            const double* zb = Zs + (s & 1) * L::ZS;
            if (!(P.ablate & 1))
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double* zq = zb + lq[q];
                const double fa_hi = zq[hi_row * 256];
                const double fa_lo = RECT ? 0.0 : zq[lo_row * 256];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bool hi = RECT || t < 9 || t < nhi;
                    const int b = hi ? t : t - nhi;
                    const double fb = zq[b * 256];
                    const double fa = hi ? fa_hi : fa_lo;
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc[t], 0, 0, 0);
                }
            }
            fz_barrier();
        }
        fz_barrier();   // (the sweep waves' reductions)
        // epilogue: lane holds column j = lane & 15, rows i = (lane >> 4) + 4 reg of each 16 x 16 tile (gemm.hip, gram_body)
        auto gtile = [&](int x) { return (gcb(x >> 3) >> 4) + (x & 7); };
        const int64_t ldc = P.N;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bool hi = RECT || t < 9 || t < nhi;
            const int a = hi ? hi_row : lo_row, b = hi ? t : t - nhi;
            const int64_t j = 16 * (int64_t)gtile(b) + fr;
            const int64_t i0 = 16 * (int64_t)gtile(a) + fk;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) Cz[j + (i0 + 4 * rr) * ldc] = acc[t][rr];
        }
    } else {
        // =================================== sweep waves ===================================
        const int st = tid - 64 * FZ_MW;
        const int p = st & 7, cg = st >> 3;                   // row pair of the stage, first column
        const int swc = 2 * ((cg >> 1) & 7);                  // (columns cg + 32 j: the swizzle does not depend on j)
        const int zoff = cg * FZ_R + ((2 * p) ^ swc);         // + 32 j * 16 per column
        const int ti = st >> 4, tr = st & 15;                 // element of a T stage this thread fetches
        const double mu = P.mu, inv_mu = P.inv_mu, inv_mu_n = P.inv_mu_n, thr_n = P.thr_n;
        const bool nonnegA = P.nonnegA != 0, nonnegE = P.nonnegE != 0;
        double* const Rp = store ? P.R : nullptr;
        double ss = 0.0, rmax = 0.0;

        struct Col {
            d2 d, y, z;
        };
        Col ring[FZ_RS];
        // loads of column j of the stage that starts at row r0 -> ring slot j % RS
        // loads of column j (0 <= j < J, possibly a runtime value) of the stage that starts at row r0 -> ring slot `slot`
        auto issue = [&](Col& c, int j, int64_t r0) {
            const int64_t row = r0 + 2 * p;
            const int gc = gcb(j >> 2) + cg + 32 * (j & 3);
            const int64_t off = (int64_t)gc * ld + row;
            if (row < kend && !(P.ablate & 8)) {
                c.y = *reinterpret_cast<const d2*>(P.Yin + off);
                c.z = *reinterpret_cast<const d2*>(P.Zin + off);
                if constexpr (HK) {
                    const int64_t co = hankel_coff(P.hg, gc);
                    c.d[0] = (row < P.hankel_K) ? P.D[row * P.hg.lag + co] : 0.0;
                    c.d[1] = (row + 1 < P.hankel_K) ? P.D[(row + 1) * P.hg.lag + co] : 0.0;
                } else {
                    c.d = *reinterpret_cast<const d2*>(P.D + off);
                }
            } else {
                c.y = c.z = c.d = d2{0.0, 0.0};
            }
        };
        auto t_fetch = [&](int64_t r0) -> double {
            const int64_t row = r0 + tr;
            return (ti < r && ti < RMAX && row < kend) ? P.Tm[row + (size_t)ti * M] : 0.0;
        };
        // stage s of this chunk: reads T stage buffer s & 1, writes Zs[s & 1]; T of stage s + 1 goes to the other T buffer.
        // The columns are worked through strictly one after the other (sched_barrier): left to itself the scheduler
        // interleaves all of a stage's columns and spills half of them.
        auto sweep_stage = [&](int s) {
            const int64_t r0 = kbeg + (int64_t)s * FZ_R;
            const bool more = s + 1 < nst;
            double tnext = 0.0;
            if (st < L::TS && more) tnext = t_fetch(r0 + FZ_R);
            // rows of T: the first eight columns stay in registers for the stage, the rest is read again per panel column
            // (the register file is what limits this kernel: 168 per wave)
            constexpr int TLO = RMAX < 8 ? RMAX : 8;
            const double* tb = sT + (s & 1) * L::TS + 2 * p;
            d2 t[TLO];
#pragma unroll
            for (int i = 0; i < TLO; ++i) t[i] = *reinterpret_cast<const d2*>(tb + i * FZ_R);
            double* zs = Zs + (s & 1) * L::ZS + zoff;
            const int64_t row = r0 + 2 * p;
            const bool ok = row < kend;
#pragma unroll 1
            for (int g = 0; g < J / FZ_RS; ++g) {
#pragma unroll
                for (int jj = 0; jj < FZ_RS; ++jj) {
                    const int j = FZ_RS * g + jj;
                    {
                        int jn = j + FZ_PF;
                        const bool wrap = jn >= J;
                        if (wrap) jn -= J;
                        if (!wrap || more) issue(ring[(jj + FZ_PF) % FZ_RS], jn, wrap ? r0 + FZ_R : r0);
                    }
                    const Col c = ring[jj];
                    const double* vs = sVs + (cg + 32 * j) * L::VSP;
                    double a0 = 0.0, a1 = 0.0;
                    if (!(P.ablate & 4))
#pragma unroll
                    for (int i = 0; i < RMAX; i += 2) {
                        const d2 v = *reinterpret_cast<const d2*>(vs + i);
                        const d2 t0 = i < TLO ? t[i < TLO ? i : 0] : *reinterpret_cast<const d2*>(tb + i * FZ_R);
                        const d2 t1 = i + 1 < TLO ? t[i + 1 < TLO ? i + 1 : 0] : *reinterpret_cast<const d2*>(tb + (i + 1) * FZ_R);
                        a0 = __builtin_fma(t0[0], v[0], a0);
                        a1 = __builtin_fma(t0[1], v[0], a1);
                        a0 = __builtin_fma(t1[0], v[1], a0);
                        a1 = __builtin_fma(t1[1], v[1], a1);
                    }
                    d2 rr, yn, zn;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        double a = q == 0 ? a0 : a1;
                        if (nonnegA) a = fz_pos(a);                       // A .= max.(A,0)                          :217-219
                        const double w = c.z[q] - a;
                        const double res = w - inv_mu * c.y[q];          // R_k = D - A - E                         :221
                        ss += res * res;
                        const double ares = res < 0.0 ? -res : res;
                        rmax = ares > rmax ? ares : rmax;                 // (a NaN never wins: the bound stays a bound)
                        rr[q] = res;
                        const double y1 = mu * w;                         // Y_{k+1} = Y + mu R                      :222
                        yn[q] = y1;
                        const double tt = inv_mu_n * y1;                  // next iteration, mu_{k+1}                :188
                        double ee = fz_soft((c.d[q] - a) + tt, thr_n);
                        if (nonnegE) ee = fz_pos(ee);                     //                                         :189-191
                        zn[q] = (c.d[q] - ee) + tt;                       //                                         :192
                    }
                    if (!ok) zn = d2{0.0, 0.0};
                    *reinterpret_cast<d2*>(zs + 32 * j * FZ_R) = zn;
                    if (store && ok && !(P.ablate & 2)) {
                        const int gc = gcb(j >> 2) + cg + 32 * (j & 3);
                        const int64_t off = (int64_t)gc * ld + row;
                        if (Rp) __builtin_nontemporal_store(rr, reinterpret_cast<d2*>(Rp + off));
                        __builtin_nontemporal_store(yn, reinterpret_cast<d2*>(P.Yout + off));
                        __builtin_nontemporal_store(zn, reinterpret_cast<d2*>(P.Zout + off));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (st < L::TS && more) sT[((s + 1) & 1) * L::TS + ti * FZ_R + tr] = tnext;
        };

        // prologue: T stage 0, first loads
        if (st < L::TS) sT[ti * FZ_R + tr] = t_fetch(kbeg);
#pragma unroll
        for (int j = 0; j < FZ_PF; ++j) issue(ring[j], j, kbeg);
        fz_barrier();   // B0
        sweep_stage(0);
        fz_barrier();   // B1
        for (int s = 0; s < nst; ++s) {
            if (s + 1 < nst) sweep_stage(s + 1);
            fz_barrier();
        }
        // ||R_k||_F^2 and max |R_k| of this workgroup's columns (bounds for the convergence test, never results)
        if (P.sumsq && store) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                ss += __shfl_down(ss, off, 64);
                const double o = __shfl_down(rmax, off, 64);
                rmax = o > rmax ? o : rmax;
            }
            if (lane == 0) {
                sRed[wave - FZ_MW] = ss;
                sRed[4 + wave - FZ_MW] = rmax;
            }
        }
        fz_barrier();
        if (P.sumsq && store && st == 0) {
            atomicAdd(P.sumsq + (blockIdx.x & 63), (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]));
            if (P.maxslot >= 0) {
                double mm = sRed[4];
                for (int k = 1; k < 4; ++k) mm = sRed[4 + k] > mm ? sRed[4 + k] : mm;
                atomicMax(reinterpret_cast<unsigned long long*>(P.sumsq + 64 + P.maxslot),
                          (unsigned long long)__double_as_longlong(mm));   // (non-negative doubles: monotone bit patterns)
            }
        }
    }
}

// N = 256: one kind of workgroup, chunk z = blockIdx.x
template <int RMAX, bool HK>
__global__ __launch_bounds__(FZ_THREADS) void k_fused_zgram_256(const FusedArgs P) {
    extern __shared__ __attribute__((aligned(16))) double fz_smem[];
    if (P.zero_slots && blockIdx.x == 0 && threadIdx.x < 72) P.zero_slots[threadIdx.x] = 0.0;
    const int z = blockIdx.x;
    const int64_t kbeg = (int64_t)z * P.kchunk;
    if (z >= P.nz || kbeg >= P.M) return;
    const int64_t kend = kbeg + P.kchunk < P.M ? kbeg + P.kchunk : P.M;
    fused_body<256, false, RMAX, HK>(P, kbeg, kend, 0, 128, 0, true, P.slab + (size_t)z * P.N * P.N, fz_smem);
}

// N = 512: blocks b and b + 8 share an XCD (round-robin dispatch: speed only, nothing depends on it); an XCD takes the
// chunks z = x, x + 8, ... and, for each, the four kinds in consecutive slots
template <int RMAX, bool HK>
__global__ __launch_bounds__(FZ_THREADS) void k_fused_zgram_512(const FusedArgs P) {
    extern __shared__ __attribute__((aligned(16))) double fz_smem[];
    if (P.zero_slots && blockIdx.x == 0 && threadIdx.x < 72) P.zero_slots[threadIdx.x] = 0.0;
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int z = x + 8 * (q >> 2), kind = q & 3;
    const int64_t kbeg = (int64_t)z * P.kchunk;
    if (z >= P.nz || kbeg >= P.M) return;
    const int64_t kend = kbeg + P.kchunk < P.M ? kbeg + P.kchunk : P.M;
    double* Cz = P.slab + (size_t)z * P.N * P.N;
    if (kind < 2) fused_body<256, false, RMAX, HK>(P, kbeg, kend, 256 * kind, 256 * kind + 128, 0, true, Cz, fz_smem);
    else fused_body<384, true, RMAX, HK>(P, kbeg, kend, 0, 128, 128 * kind, false, Cz, fz_smem);
}

inline bool fz_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Can the fused kernel serve this sweep?  fp64 panels of 256 or 512 columns, contiguous (ld = M), even M, rank <= 16,
// 16-byte aligned panels; below ~16 stages per workgroup the separate kernels are the better schedule.
bool fused_zgram_ok(int64_t M, int64_t N, int64_t r, const void* D, const void* Yin, const void* Yout, const void* Zin,
                    const void* Zout, const void* R, bool hankel) {
    if (dev_is(DEV_NO_FUSED_ZGRAM, '1')) return false;
    if (N != 256 && N != 512) return false;
    if (r < 0 || r > 16 || (M & 1)) return false;
    const int64_t min_rows = [] { const char* e = dev_get(DEV_FUSED_ZGRAM_MINROWS); return e ? atol(e) : 0L; }();
    if (M < (min_rows > 0 ? min_rows : (N == 256 ? 65536 : 16384))) return false;
    if (!(hankel || fz_aligned16(D)) || !fz_aligned16(Yin) || !fz_aligned16(Yout) || !fz_aligned16(Zin) || !fz_aligned16(Zout) ||
        !fz_aligned16(R))
        return false;
    return true;
}

// K split and slabs of the fused kernel, as a GramPlan whose reduction is gram_reduce (gemm.hip)
int fused_zgram_plan(Handle* h, int64_t M, int64_t N, GramPlan* pl) {
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device);
    if (ncu < 8) ncu = 8;
    const int per = N == 256 ? 1 : 4;            // workgroups per row chunk
    int64_t nz = std::max<int64_t>(1, ncu / per);
    int64_t kc = ((M + nz - 1) / nz + FZ_R - 1) / FZ_R * FZ_R;
    nz = (M + kc - 1) / kc;
    void* slab;
    TLSQ_TRY(ws_get(h, WS_SLAB, (size_t)nz * (size_t)N * (size_t)N * sizeof(double), &slab));
    *pl = GramPlan();
    pl->N = N;
    pl->K = M;
    pl->nti = (N + 127) / 128;
    pl->nsplit_o = pl->nsplit_d = nz;
    pl->kchunk_o = pl->kchunk_d = kc;
    pl->nchunks = 1;
    pl->z_f32 = 0;
    pl->slab = (double*)slab;
    return TLSQ_OK;
}

int launch_fused_zgram(Handle* h, const GramPlan& pl, const double* D, const double* Tm, const double* Vs, const double* Yin,
                       double* Yout, const double* Zin, double* Zout, double* R, int64_t M, int64_t N, int64_t r, double mu,
                       double inv_mu, int nonnegA, double inv_mu_n, double thr_n, int nonnegE, double* sumsq, double* zero_slots,
                       const double* hankel_y, int64_t hankel_K, int maxslot, HankelGeom hg) {
    if (!sumsq || maxslot > 7) maxslot = -1;
    FusedArgs a;
    a.D = hankel_y ? hankel_y : D;
    a.Tm = Tm;
    a.Vs = Vs;
    a.Yin = Yin;
    a.Yout = Yout;
    a.Zin = Zin;
    a.Zout = Zout;
    a.R = R;
    a.M = M;
    a.ld = M;
    a.N = (int)N;
    a.r = (int)r;
    a.mu = mu;
    a.inv_mu = inv_mu;
    a.inv_mu_n = inv_mu_n;
    a.thr_n = thr_n;
    a.nonnegA = nonnegA;
    a.nonnegE = nonnegE;
    a.sumsq = sumsq;
    a.zero_slots = zero_slots;
    a.maxslot = maxslot;
    a.slab = pl.slab;
    a.kchunk = pl.kchunk_o;
    a.nz = (int)pl.nsplit_o;
    a.hankel_K = hankel_K;
    a.hg = hg;
    a.ablate = [] { const char* e = dev_get(DEV_FUSED_ABLATE); return e ? atoi(e) : 0; }();
    const bool hk = hankel_y != nullptr;
    const int nz8 = (a.nz + 7) / 8 * 8;
#define FZ_LAUNCH(KERN, NCL, RM, GRID)                                                                              \
    do {                                                                                                            \
        const size_t lds = (size_t)FzLds<NCL, RM>::TOTAL * sizeof(double);                                          \
        static bool attr_set = false;   /* (per instantiation; idempotent) */                                        \
        if (!attr_set) {                                                                                            \
            TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&KERN), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr_set = true;                                                                                        \
        }                                                                                                           \
        hipLaunchKernelGGL(KERN, dim3((unsigned)(GRID)), dim3(FZ_THREADS), lds, h->stream, a);                      \
    } while (0)
    if (N == 256) {
        if (r <= 8) {
            if (hk) FZ_LAUNCH((k_fused_zgram_256<8, true>), 256, 8, a.nz);
            else FZ_LAUNCH((k_fused_zgram_256<8, false>), 256, 8, a.nz);
        } else {
            if (hk) FZ_LAUNCH((k_fused_zgram_256<16, true>), 256, 16, a.nz);
            else FZ_LAUNCH((k_fused_zgram_256<16, false>), 256, 16, a.nz);
        }
    } else {
        if (r <= 8) {
            if (hk) FZ_LAUNCH((k_fused_zgram_512<8, true>), 384, 8, 4 * nz8);
            else FZ_LAUNCH((k_fused_zgram_512<8, false>), 384, 8, 4 * nz8);
        } else {
            if (hk) FZ_LAUNCH((k_fused_zgram_512<16, true>), 384, 16, 4 * nz8);
            else FZ_LAUNCH((k_fused_zgram_512<16, false>), 384, 16, 4 * nz8);
        }
    }
#undef FZ_LAUNCH
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

}  // namespace tlsq
