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
// N = 256: every workgroup owns all 136 tiles of the lower triangle (16 x 16 tiles of 16 x 16).  N = 512: two workgroups per row
// chunk, one per diagonal 256-column block (columns 0..255 / 256..511: each stores its Y, Z, R columns and owns the 136 tiles of
// its diagonal block); the off-diagonal 256 x 256 block of the Gram matrix is accumulated afterwards from the stored Z_{k+1} by
// k_gram_kc's off-diagonal body on a four-tile list (gemm.hip, gram_offdiag_*) and one kernel sums both slab sets in a fixed
// order (fused_zgram_finish).  (A first build gave the off-diagonal block to two more "rectangular" workgroups per chunk that
// swept 384 columns a second time for their operands: fp64 MFMA and vector instructions of one SIMD do not overlap on gfx950, so
// the redundant sweeps cost more matrix time than the fusion saved - dropped in round 5.)  The partial Gram matrices go to
// split-K slabs in the layout of gemm.hip's k_gram_kc and are summed in a fixed order by its k_slab_reduce (deterministic).
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
constexpr int FZ_NC = 256;                    // columns of a workgroup
constexpr int FZ_J = FZ_NC / 32;              // columns per sweep thread and stage
constexpr int FZ_MW = 8, FZ_SW = 4;           // MFMA waves, sweep waves
constexpr int FZ_THREADS = 64 * (FZ_MW + FZ_SW);
constexpr int FZ_RS_HK = 8, FZ_RS_D = 8;      // ring slots of a sweep thread with an implicit / explicit D
constexpr int FZ_DS = 4;                      // ring slots of D (explicit panels)
constexpr int FZ_TLO = 4;                     // rows of a T stage a sweep thread keeps in registers (the rest is read per column)
constexpr int FZ_NT = 17;                     // accumulator tiles per MFMA wave (136 = lower triangle of 16 x 16 tiles)

struct FusedArgs {
    const double* D;      // the panel D (ld), or the series y of an implicit Hankel D (HK: D[i, j] = y[i + j], i < hankel_K)
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
    double s_div;         // FIRST: the dual norm Y_1 = D / s_div is formed with (inv_mu = 1 / mu_1, thr_n = lambda / mu_1 then)
    int nonnegA, nonnegE;
    double* sumsq;        // 72 doubles (k_zsweep): [0, 64) partial sums of ||R_k||_F^2, [64 + maxslot] max |R_k[i, j]|
    double* zero_slots;
    int maxslot;
    double* slab;         // nz slabs of N x N (ld N)
    int64_t kchunk;       // rows per chunk (a multiple of 16)
    int nz;
    int64_t hankel_K;
    int ablate;           // development (FUSED_ABLATE): 1 no MFMAs, 2 no global stores
};

// one barrier of the workgroup that waits for this wave's LDS traffic only: __syncthreads() would also drain the global
// loads the sweep waves keep in flight across the stage boundary
__device__ __forceinline__ void fz_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Panel accesses of the sweep waves: buffer instructions with a uniform base (SGPR descriptor, rebuilt per column by scalar
// adds) and the thread's constant 32-bit byte offset - no vector instruction goes into an address (a flat 64-bit pointer per
// column and array costs two, and the compiler keeps dozens of them alive across the stage).  16 bytes, non-temporal.
typedef unsigned int fz_u4 __attribute__((ext_vector_type(4)));
// `left` = elements from ubase to the end of the array: the descriptor's range check turns whatever a prefetch reads beyond
// the panel (the stage after the last one) into zeros instead of a fault, so the loads need no branch
__device__ __forceinline__ unsigned fz_range(int64_t left) {
    // min(8 left, 0xFFFFFFF0), zero for left <= 0 - on the two 32-bit halves: the scalar unit has no 64-bit compare, and a
    // 64-bit compare on the vector unit is paid in matrix time
    // (the high halves go through an empty asm: otherwise the optimiser fuses the tests into 64-bit compares again)
    const uint64_t b = (uint64_t)left << 3;
    const uint32_t lo = (uint32_t)b;
    uint32_t hi = (uint32_t)(b >> 32), hi_left = (uint32_t)((uint64_t)left >> 32);
    asm volatile("" : "+s"(hi), "+s"(hi_left));
    hi = __builtin_amdgcn_readfirstlane(hi);
    hi_left = __builtin_amdgcn_readfirstlane(hi_left);
    if ((int32_t)hi_left < 0 || (hi_left | (uint32_t)left) == 0u) return 0u;
    return (hi != 0u || lo > 0xFFFFFFF0u) ? 0xFFFFFFF0u : lo;
}
// (the base is wave-uniform by construction; readfirstlane says so to the compiler, which otherwise wraps the instruction in a
//  waterfall loop whenever its uniformity analysis loses track - a no-op on values that already sit in scalar registers)
__device__ __forceinline__ double* fz_uniform(const double* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<double*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ d2 fz_ld(const double* ubase, int64_t left, unsigned voff) {
    const unsigned range = __builtin_amdgcn_readfirstlane(fz_range(left));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(fz_uniform(ubase), 0, range, 0x00020000);
    return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 2));
}
typedef unsigned int fz_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double fz_ld1(const double* ubase, int64_t left, unsigned voff) {   // one double, default cache policy
    const unsigned range = __builtin_amdgcn_readfirstlane(fz_range(left));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(fz_uniform(ubase), 0, range, 0x00020000);
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0));
}
// stores are issued unconditionally as well (a store under a branch makes the compiler's vmcnt bookkeeping assume the shorter
// path at every join and drain half of the ring): `left` = 0 switches a store off, a lane offset of 0xFFFFFFFF drops that lane
__device__ __forceinline__ void fz_st(double* ubase, int64_t left, unsigned voff, d2 v) {
    const unsigned range = __builtin_amdgcn_readfirstlane(fz_range(left));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(fz_uniform(ubase), 0, range, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fz_u4, v), r, voff, 0, 2);
}

// LDS (doubles): two stage buffers [column][16 k, swizzled], Vs [column][RMAX + 2], two T stages [i][16 rows], two windows
// of the Hankel series (16 + 255 values, padded), reductions
template <int RMAX>
struct FzLds {
    static constexpr int ZS = FZ_NC * FZ_R;
    static constexpr int VSP = RMAX + 2;
    static constexpr int VS = FZ_NC * VSP;
    static constexpr int TS = RMAX * FZ_R;
    static constexpr int WS = 288;
    static constexpr int RED = 16;
    static constexpr int TOTAL = 2 * ZS + VS + 2 * TS + 3 * WS + RED;   // (third window: zeros)
};

// One workgroup: rows [kbeg, kend) x columns [gc0, gc0 + 256) of the panels; Cz = its slab of partial Gram entries.
//   RMAX: compiled length of the factor product (r <= RMAX); HK: implicit Hankel D (one channel, lag 1);
//   NN: the nonnegA / nonnegE projections are compiled in; RS: ring slots (columns in flight per sweep thread: RS - 1);
//   FIRST: iteration 1 instead (k_first_shrink: Y_1 = D / s, A = 0, Z_1 - src/robustPCA.jl:181, :188-192 - nothing but D is read);
//   GR: the Gram matrix accumulated is that of the residual R_k (what the convergence test :225 takes the norm of) instead
//   of Z_{k+1}'s - the rows of R_k go to the LDS stage, R_k itself need not be stored
template <int RMAX, bool HK, bool NN, int RS, bool FIRST, bool GR>
__device__ __forceinline__ void fused_body(const FusedArgs& P, int64_t kbeg, int64_t kend, int gc0, double* __restrict__ Cz,
                                           double* __restrict__ smem) {
    using L = FzLds<RMAX>;
    constexpr int J = FZ_J, PF = RS - 1;
    static_assert(J % RS == 0, "ring slots are static");
    double* const Zs = smem;
    double* const sVs = smem + 2 * L::ZS;
    double* const sT = sVs + L::VS;
    double* const sW = sT + 2 * L::TS;
    double* const sZero = sW + 2 * L::WS;   // a window of zeros: what the rows below the last Hankel row read
    double* const sRed = sZero + L::WS;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int nst = (int)((kend - kbeg + FZ_R - 1) / FZ_R);
    const int64_t M = P.M, ld = P.ld;
    const int r = P.r;

    // ---- Vs -> LDS (all waves), zero beyond r -------------------------------------------------------------------------
    if (!FIRST)
    for (int e = tid; e < FZ_NC * RMAX; e += FZ_THREADS) {
        const int c = e % FZ_NC, i = e / FZ_NC;
        sVs[c * L::VSP + i] = i < r ? P.Vs[(size_t)(gc0 + c) + (size_t)i * P.N] : 0.0;
    }

    for (int e = tid; e < L::WS; e += FZ_THREADS) sZero[e] = 0.0;

    if (wave < FZ_MW) {
        // =================================== MFMA waves ===================================
        const int fr = lane & 15, fk = lane >> 4;
        const int swz = 2 * ((fr >> 1) & 7);
        int lq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) lq[q] = fr * FZ_R + ((4 * q + fk) ^ swz);
        d4 acc[FZ_NT];
#pragma unroll
        for (int t = 0; t < FZ_NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
        // tile of position t: rows (A operand) from fragment a(t), columns (B operand) from fragment b(t):
        // wave w owns tile rows 15 - w (columns 0..15 - w: positions 0..15 - w) and w (columns 0..w: the other w + 1 positions)
        const int hi_row = 15 - wave, lo_row = wave, nhi = 16 - wave;
        fz_barrier();   // B0: Vs, T stage 0
        fz_barrier();   // B1: stage 0 in Zs[0]
        for (int s = 0; s < nst; ++s) {
            const double* zb = Zs + (s & 1) * L::ZS;
            if (!(P.ablate & 1))
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double* zq = zb + lq[q];
                const double fa_hi = zq[hi_row * 256];
                const double fa_lo = zq[lo_row * 256];
#pragma unroll
                for (int t = 0; t < FZ_NT; ++t) {
                    const bool hi = t < 9 || t < nhi;
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
        const int64_t ldc = P.N;
#pragma unroll
        for (int t = 0; t < FZ_NT; ++t) {
            const bool hi = t < 9 || t < nhi;
            const int a = hi ? hi_row : lo_row, b = hi ? t : t - nhi;
            const int64_t j = gc0 + 16 * b + fr;
            const int64_t i0 = gc0 + 16 * a + fk;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) Cz[j + (i0 + 4 * rr) * ldc] = acc[t][rr];
        }
    } else {
        // =================================== sweep waves ===================================
        // (vector instructions and fp64 MFMAs of one SIMD do not overlap on gfx950 - tools/ubench/coexec_f64.hip - so every
        //  instruction here is paid in matrix time: addresses are uniform base + a constant 32-bit thread offset, the
        //  threshold is the two-instruction clamp form, and the sweep wave gets the pipe whenever it is ready)
        __builtin_amdgcn_s_setprio(3);
        const int st = tid - 64 * FZ_MW;
        const int p = st & 7, cg = st >> 3;                   // row pair of the stage, first column
        const int swc = 2 * ((cg >> 1) & 7);                  // (columns cg + 32 j: the swizzle does not depend on j)
        const int zoff = cg * FZ_R + ((2 * p) ^ swc);         // + 32 j * 16 per column
        const int ti = st >> 4, tr = st & 15;                 // element of a T stage this thread fetches
        const unsigned toff = (unsigned)(((int64_t)cg * ld + 2 * p) * 8);   // byte offset of the thread's row pair in column cg
        const int woff = 2 * p + cg;                          // Hankel window: D[row0 + 2p + q, gc0 + cg + 32 j] = w[woff + q + 32 j]
        const double mu = P.mu, inv_mu = P.inv_mu, inv_mu_n = P.inv_mu_n, thr_n = P.thr_n, nthr_n = -P.thr_n;
        const bool nonnegA = NN && P.nonnegA != 0, nonnegE = NN && P.nonnegE != 0;
        double* const Rp = P.R;
        double* const Rbase = Rp ? Rp : P.Zout;                  // (a switched-off store still needs a valid descriptor base)
        const int64_t stleft = (P.ablate & 2) ? 0 : M * (int64_t)P.N;
        double ss = 0.0, rmax = 0.0;
        const int64_t colstep = 32 * ld;                      // elements between a thread's consecutive columns

        // The register ring: Y and Z of the next RS - 1 columns, D (explicit panels) of the next DS - 1 - what a sweep thread has
        // in flight is what hides the HBM latency (one sweep wave per SIMD: nobody else issues loads)
        struct Col {
            d2 y, z;
        };
        constexpr int DS = HK ? 1 : FZ_DS, DPF = DS - 1;
        static_assert(J % DS == 0 && RS % DS == 0, "ring slots are static");
        Col ring[RS];
        d2 dring[DS];
        // element offset (uniform) of column j of the stage that starts at row r0
        auto ubase = [&](int j, int64_t r0) -> int64_t { return (int64_t)gc0 * ld + r0 + (int64_t)j * colstep; };
        // loads of column j of the stage starting at row r0: unconditional (the ring must stay in registers - a load under a
        // branch sends its slot to scratch memory); a stage that reaches beyond the chunk or the panel reads rows nobody uses
        const int64_t ntot = M * (int64_t)P.N;
        // (vo: the thread's byte offset, or 0xFFFFFFFF for a row pair beyond the chunk - the range check then returns zeros, and
        //  zeros in give zeros out: no mask anywhere downstream)
        auto issue = [&](int j, int64_t r0, unsigned vo) -> Col {
            Col c;
            const int64_t u = ubase(j, r0);
            c.y = fz_ld(P.Yin + u, ntot - u, vo);
            c.z = fz_ld(P.Zin + u, ntot - u, vo);
            return c;
        };
        auto issue_d = [&](int j, int64_t r0, unsigned vo) -> d2 {
            const int64_t u = ubase(j, r0);
            return fz_ld(P.D + u, ntot - u, vo);
        };
        // element (row r0 + tr, column ti) of T: zero beyond column r - 1 through the range check (rows beyond the panel only
        // occur in a stage whose rows are masked)
        const unsigned tvoff = (unsigned)(((int64_t)ti * M + tr) * 8);
        auto t_fetch = [&](int64_t r0) -> double {
            return fz_ld1(P.Tm + r0, M * (int64_t)r - r0, (ti < RMAX && r0 + tr < kend) ? tvoff : 0xFFFFFFF8u);
        };
        // Hankel window of the stage at r0: w[i] = y[r0 + gc0 + i], i < 16 + 255 (thread st: i = st and, st < 16, 256 + st);
        // the series has hankel_K + N - 1 samples
        auto w_fetch = [&](int64_t r0, double& w0, double& w1) {
            const int64_t base = r0 + gc0, left = P.hankel_K + P.N - 1 - base;
            w0 = fz_ld1(P.D + base, left, (unsigned)st * 8u);
            w1 = fz_ld1(P.D + base, left, st < 16 ? (unsigned)(256 + st) * 8u : 0xFFFFFFF8u);
        };
        auto stage_full = [&](int64_t r0) { return r0 + FZ_R <= kend && (!HK || r0 + FZ_R <= P.hankel_K); };

        // stage s of this chunk: reads T stage (and window) buffer s & 1, writes Zs[s & 1]; T / window of stage s + 1 go to the
        // other buffers.  The columns are worked through strictly one after the other (sched_barrier): left to itself the
        // scheduler interleaves all of a stage's columns and spills half of them.  (One copy of this body in the kernel:
        // the ring is ~100 registers of loop state.)
        auto sweep_stage = [&](int s) {
            // (the stage index is made opaque to the optimiser: otherwise the uniform base of every (array, column) pair - 40
            //  pointers - is hoisted out of the stage loop and the scalar registers spill into vector lanes; readfirstlane tells
            //  the uniformity analysis that what comes out of the asm is still a scalar)
            int so = s;
            asm volatile("" : "+s"(so));
            so = __builtin_amdgcn_readfirstlane(so);
            const int64_t r0 = kbeg + (int64_t)so * FZ_R;
            const bool full = stage_full(r0);
            const bool more = s + 1 < nst;
            double tnext = 0.0, wn0 = 0.0, wn1 = 0.0;
            if (!FIRST && st < L::TS && more) tnext = t_fetch(r0 + FZ_R);
            if (HK && more) w_fetch(r0 + FZ_R, wn0, wn1);
            // rows of T: the first columns stay in registers for the stage, the rest is read again per panel column
            // (the register file is what limits this kernel: 168 per wave)
            constexpr int TLO = RMAX < FZ_TLO ? RMAX : FZ_TLO;
            const double* tb = sT + (s & 1) * L::TS + 2 * p;
            const double* wb = sW + (s & 1) * L::WS + woff;
            d2 t[TLO];
#pragma unroll
            for (int i = 0; i < TLO; ++i) t[i] = FIRST ? d2{0.0, 0.0} : *reinterpret_cast<const d2*>(tb + i * FZ_R);
            double* zs = Zs + (s & 1) * L::ZS + zoff;
            // rows beyond the chunk (the panel's last rows, in the last stage of the last chunk): loaded as zeros, not stored;
            // rows below the last Hankel row (zero pad rows of D): they read the window of zeros
            const unsigned vo_c = (full || r0 + 2 * p < kend) ? toff : 0xFFFFFFFFu;
            const unsigned vo_n = (r0 + FZ_R + 2 * p < kend) ? toff : 0xFFFFFFFFu;
            const double* wb0 = wb;
            const double* wb1 = wb + 1;
            if (HK && !full) {
                if (!(r0 + 2 * p < P.hankel_K)) wb0 = sZero;
                if (!(r0 + 2 * p + 1 < P.hankel_K)) wb1 = sZero;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
            for (int g = 0; g < J / RS; ++g) {
#pragma unroll
                for (int jj = 0; jj < RS; ++jj) {
                    const int j = RS * g + jj;
                    if constexpr (!FIRST) {
                        const int jr = (j + PF) % J;
                        ring[(jj + PF) % RS] = issue(jr, j + PF >= J ? r0 + FZ_R : r0, j + PF >= J ? vo_n : vo_c);
                    }
                    if constexpr (!HK) {
                        const int jr = (j + DPF) % J;
                        dring[(jj + DPF) % DS] = issue_d(jr, j + DPF >= J ? r0 + FZ_R : r0, j + DPF >= J ? vo_n : vo_c);
                    }
                    const Col c = FIRST ? Col{d2{0.0, 0.0}, d2{0.0, 0.0}} : ring[jj];
                    d2 cd;
                    if constexpr (HK) {
                        // D[row, gc0 + cg + 32 j] = y[row + gc0 + cg + 32 j] from the stage's window
                        cd[0] = wb0[32 * j];
                        cd[1] = wb1[32 * j];
                    } else {
                        cd = dring[jj % DS];
                    }
                    d2 rr = d2{0.0, 0.0}, yn, zn;
                    if constexpr (FIRST) {
                        // k_first_shrink's statements (A = 0): the same clamp form of the threshold as below
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const double y1 = cd[q] / P.s_div;                // Y ./= dual_norm                        :181
                            yn[q] = y1;
                            const double tt = inv_mu * y1;
                            const double x = cd[q] + tt;                      // A = 0                                  :188
                            double ee = x - __builtin_fmin(__builtin_fmax(x, nthr_n), thr_n);
                            if (nonnegE) ee = fz_pos(ee);
                            zn[q] = (cd[q] - ee) + tt;                        //                                        :192
                        }
                    } else {
                    const double* vs = sVs + (cg + 32 * j) * L::VSP;
                    double a0 = 0.0, a1 = 0.0;
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
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        double a = q == 0 ? a0 : a1;
                        if (nonnegA) a = fz_pos(a);                       // A .= max.(A,0)                          :217-219
                        const double w = c.z[q] - a;
                        const double res = w - inv_mu * c.y[q];          // R_k = D - A - E                         :221
                        ss = __builtin_fma(res, res, ss);                 // (a bound, never a result)
                        rmax = __builtin_fmax(rmax, __builtin_fabs(res)); // (a NaN never wins: the bound stays a bound)
                        rr[q] = res;
                        const double y1 = mu * w;                         // Y_{k+1} = Y + mu R                      :222
                        yn[q] = y1;
                        const double tt = inv_mu_n * y1;                  // next iteration, mu_{k+1}                :188
                        // soft_th(x, e) = max(x - e, 0) + min(x + e, 0) (src/robustPCA.jl:1) = x - clamp(x, -e, e) for e >= 0:
                        // the same single rounding in all three branches, the same +0 inside the band, NaN / Inf carried by
                        // the subtraction (the launcher sends negative or non-finite thresholds to k_zsweep)
                        const double x = (cd[q] - a) + tt;
                        double ee = x - __builtin_fmin(__builtin_fmax(x, nthr_n), thr_n);
                        if (nonnegE) ee = fz_pos(ee);                     //                                         :189-191
                        zn[q] = (cd[q] - ee) + tt;                       //                                         :192
                    }
                    }
                    *reinterpret_cast<d2*>(zs + 32 * j * FZ_R) = GR ? rr : zn;
                    {
                        const int64_t u = ubase(j, r0);
                        fz_st(Rbase + u, Rp ? stleft - u : 0, vo_c, rr);
                        fz_st(P.Yout + u, stleft - u, vo_c, yn);
                        fz_st(P.Zout + u, stleft - u, vo_c, zn);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!FIRST && st < L::TS && more) sT[((s + 1) & 1) * L::TS + ti * FZ_R + tr] = tnext;
            if (HK && more) {
                double* wnx = sW + ((s + 1) & 1) * L::WS;
                wnx[st] = wn0;
                if (st < 16) wnx[256 + st] = wn1;
            }
        };

        // prologue: T stage 0, window 0, first loads
        if (!FIRST && st < L::TS) sT[ti * FZ_R + tr] = t_fetch(kbeg);
        if (HK) {
            double w0, w1;
            w_fetch(kbeg, w0, w1);
            sW[st] = w0;
            if (st < 16) sW[256 + st] = w1;
        }
        const unsigned vo_0 = (kbeg + 2 * p < kend) ? toff : 0xFFFFFFFFu;
        if constexpr (!FIRST) {
#pragma unroll
            for (int j = 0; j < PF; ++j) ring[j] = issue(j, kbeg, vo_0);
        }
        if constexpr (!HK) {
#pragma unroll
            for (int j = 0; j < DPF; ++j) dring[j] = issue_d(j, kbeg, vo_0);
        }
        fz_barrier();   // B0
        for (int s = 0; s <= nst; ++s) {   // (stage s is swept while the MFMA waves work on stage s - 1)
            if (s < nst) sweep_stage(s);
            fz_barrier();
        }
        // ||R_k||_F^2 and max |R_k| of this workgroup's columns (bounds for the convergence test, never results)
        if (P.sumsq) {
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
        if (P.sumsq && st == 0) {
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

// blockIdx.x = chunk * (N / 256) + column block
template <int RMAX, bool HK, bool NN, int RS, bool FIRST, bool GR>
__global__ __launch_bounds__(FZ_THREADS) void k_fused_zgram(const FusedArgs P) {
    extern __shared__ __attribute__((aligned(16))) double fz_smem[];
    if (P.zero_slots && blockIdx.x == 0 && threadIdx.x < 72) P.zero_slots[threadIdx.x] = 0.0;
    const int ncb = P.N / FZ_NC;
    const int z = blockIdx.x / ncb, cb = blockIdx.x % ncb;
    const int64_t kbeg = (int64_t)z * P.kchunk;
    if (z >= P.nz || kbeg >= P.M) return;
    const int64_t kend = kbeg + P.kchunk < P.M ? kbeg + P.kchunk : P.M;
    fused_body<RMAX, HK, NN, RS, FIRST, GR>(P, kbeg, kend, FZ_NC * cb, P.slab + (size_t)z * P.N * P.N, fz_smem);
}

inline bool fz_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Can the fused kernel serve this sweep?  fp64 panels of 256 or 512 columns, contiguous (ld = M), even M, rank <= 16,
// 16-byte aligned panels, a thread's 32-bit byte offset must reach its 31st column, a proper threshold; an implicit Hankel
// D only with one channel and lag 1.  Below ~16 stages per workgroup the separate kernels are the better schedule.
bool fused_zgram_ok(int64_t M, int64_t N, int64_t r, const void* D, const void* Yin, const void* Yout, const void* Zin,
                    const void* Zout, const void* R, bool hankel, double thr_n, HankelGeom hg) {
    if (dev_is(DEV_NO_FUSED_ZGRAM, '1')) return false;
    // N = 512: the kernel covers the two diagonal 256-column blocks of the Gram matrix (two workgroups per row chunk), the caller
    // adds the off-diagonal block from the stored panel (gemm.hip, gram_offdiag_*): worth it from ~50000 rows (measured
    // sweep + Gram 389 -> ~350 us at 50000 rows, 793 -> ~690 at 100000, 1534 -> ~1240 at 200000)
    if (N != FZ_NC && !(N == 2 * FZ_NC && !dev_is(DEV_FUSED_ZGRAM_N512, '0'))) return false;
    if (r < 0 || r > 16 || (M & 1)) return false;
    if (!(thr_n >= 0.0) || !std::isfinite(thr_n)) return false;
    if (hankel && (hg.lag != 1 || hg.Dch != 1)) return false;
    if ((31 * M + 16) * 8 >= ((int64_t)1 << 32)) return false;
    // a stage touches 16 rows of every column: with a leading dimension that is a multiple of a large power of two all those
    // 128-byte pieces fall into the same few memory channels (measured: 65536 x 512 602 us against 510 us for the two kernels,
    // while 50000 and 100000 rows gain 4 % and 13 %) - such panels keep the separate kernels
    if (M % 2048 == 0 && !dev_is(DEV_FUSED_ZGRAM_N512, '2')) return false;
    const int64_t min_rows = [] { const char* e = dev_get(DEV_FUSED_ZGRAM_MINROWS); return e ? atol(e) : 0L; }();
    // one workgroup per CU whatever M is, each with its own 272 KB of partial sums: below ~400k rows the slabs (134 MB, a sixth
    // of a panel there) and their reduction cost what the fusion saves (measured: 131072 rows 395 us against 402 us for the
    // two kernels, 1e6 rows 1845 against 2822)
    if (M < (min_rows > 0 ? min_rows : (N == FZ_NC ? 400000 : 65536))) return false;
    if (!(hankel || fz_aligned16(D)) || !fz_aligned16(Yin) || !fz_aligned16(Yout) || !fz_aligned16(Zin) || !fz_aligned16(Zout) ||
        !fz_aligned16(R))
        return false;
    return true;
}

// the shape part of the test alone (what a rank of a row-sharded group can tell the others before the loop: solver.hip)
bool fused_zgram_shape_ok(int64_t M, int64_t N, bool hankel) {
    return fused_zgram_ok(M, N, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, hankel, 0.0, HankelGeom());
}

// K split and slabs of the fused kernel, as a GramPlan whose reduction is gram_reduce (gemm.hip)
int fused_zgram_plan(Handle* h, int64_t M, int64_t N, GramPlan* pl) {
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device);
    if (ncu < 8) ncu = 8;
    const int per = (int)(N / FZ_NC);            // workgroups per row chunk
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
    if (N != FZ_NC) {   // (the off-diagonal block's slabs and tile list exist from here on as well)
        GramPlan pl2;
        TLSQ_TRY(gram_offdiag_plan(h, N, M, &pl2));
    }
    return TLSQ_OK;
}

// What follows the kernel: N = 256 - the fixed-order sum of its slabs; N = 512 - the off-diagonal block of the Gram matrix from
// the panel the kernel has written (Zout: device, M x N, ld M), then the sum of both slab sets.  G: N x N, ld N, both triangles.
int fused_zgram_finish(Handle* h, const GramPlan& pl, const double* Zout, int64_t M, int64_t N, double* G) {
    if (N == FZ_NC) return gram_reduce(h, h->stream, pl, G, N);
    GramPlan pl2;
    TLSQ_TRY(gram_offdiag_plan(h, N, M, &pl2));
    TLSQ_TRY(gram_offdiag_launch(h, h->stream, pl2, Zout, M, M));
    return gram_reduce2(h, h->stream, pl, pl2, G, N);
}

namespace {
template <int RM, bool HKF, bool NNF, int RSF, bool FIRSTF, bool GRF = false>
int fz_launch(Handle* h, const FusedArgs& a, unsigned grid) {
    const size_t lds = (size_t)FzLds<RM>::TOTAL * sizeof(double);
    // the attribute belongs to the function object of the CURRENT device: set on every launch, as gram16.hip / cholesky.hip /
    // jacobi.hip do (a process-wide "already set" flag left ranks 1.. of a tlsq_create_multi group without it)
    TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_zgram<RM, HKF, NNF, RSF, FIRSTF, GRF>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_fused_zgram<RM, HKF, NNF, RSF, FIRSTF, GRF>), dim3(grid), dim3(FZ_THREADS), lds, h->stream, a);
    TLSQ_HIP(h, hipGetLastError());
    if (a.nz > 0) ++h->kern_fused_zgram;
    return TLSQ_OK;
}
}  // namespace

// the dynamic-LDS attribute of every instantiation rpca_core may launch (a launch with zero workgroups each: nothing runs)
int fused_zgram_warm(Handle* h) {
    if (h->fused_warm_done) return TLSQ_OK;   // (per handle = per device: the attributes belong to the device's function objects)
    FusedArgs a = {};
    a.nz = 0;
    a.N = FZ_NC;
    a.kchunk = FZ_R;
#define FZ_WARM(RM, HKF, NNF, RSF)                                   \
    TLSQ_TRY((fz_launch<RM, HKF, NNF, RSF, false, false>(h, a, 1))); \
    TLSQ_TRY((fz_launch<RM, HKF, NNF, RSF, false, true>(h, a, 1)))
    FZ_WARM(8, true, false, FZ_RS_HK);
    FZ_WARM(16, true, false, FZ_RS_HK);
    FZ_WARM(8, false, false, FZ_RS_D);
    FZ_WARM(16, false, false, FZ_RS_D);
#undef FZ_WARM
    TLSQ_TRY((fz_launch<8, true, false, FZ_RS_HK, true>(h, a, 1)));
    TLSQ_TRY((fz_launch<8, false, false, FZ_RS_D, true>(h, a, 1)));
    h->fused_warm_done = true;
    return TLSQ_OK;
}

// first = true: iteration 1 (launch_first_shrink's arguments: Y_1 = D / s_div to Yout, Z_1 to Zout, inv_mu = 1 / mu_1,
// thr_n = lambda / mu_1; Tm, Vs, Yin, Zin, R, mu, inv_mu_n, nonnegA, sumsq unused); gram_of_r: the slabs receive R_k' R_k
// instead of Z_{k+1}' Z_{k+1} and R_k is not stored
int launch_fused_zgram(Handle* h, const GramPlan& pl, const double* D, const double* Tm, const double* Vs, const double* Yin,
                       double* Yout, const double* Zin, double* Zout, double* R, int64_t M, int64_t N, int64_t r, double mu,
                       double inv_mu, int nonnegA, double inv_mu_n, double thr_n, int nonnegE, double* sumsq, double* zero_slots,
                       const double* hankel_y, int64_t hankel_K, int maxslot, bool first, double s_div, bool gram_of_r) {
    if (!sumsq || maxslot > 7) maxslot = -1;
    FusedArgs a;
    a.D = hankel_y ? hankel_y : D;
    a.Tm = Tm;
    a.Vs = Vs;
    a.Yin = first ? Yout : Yin;
    a.Yout = Yout;
    a.Zin = first ? Zout : Zin;
    a.Zout = Zout;
    a.R = (first || gram_of_r) ? nullptr : R;
    a.M = M;
    a.ld = M;
    a.N = (int)N;
    a.r = first ? 0 : (int)r;
    a.mu = mu;
    a.inv_mu = inv_mu;
    a.inv_mu_n = inv_mu_n;
    a.thr_n = thr_n;
    a.s_div = s_div;
    a.nonnegA = nonnegA;
    a.nonnegE = nonnegE;
    a.sumsq = first ? nullptr : sumsq;
    a.zero_slots = zero_slots;
    a.maxslot = maxslot;
    a.slab = pl.slab;
    a.kchunk = pl.kchunk_o;
    a.nz = (int)pl.nsplit_o;
    a.hankel_K = hankel_K;
    a.ablate = [] { const char* e = dev_get(DEV_FUSED_ABLATE); return e ? atoi(e) : 0; }();
    const bool hk = hankel_y != nullptr, nn = nonnegA || nonnegE;
    const unsigned grid = (unsigned)(a.nz * (N / FZ_NC));
    if (first) {
        if (hk) return nonnegE ? fz_launch<8, true, true, FZ_RS_HK, true>(h, a, grid) : fz_launch<8, true, false, FZ_RS_HK, true>(h, a, grid);
        return nonnegE ? fz_launch<8, false, true, FZ_RS_D, true>(h, a, grid) : fz_launch<8, false, false, FZ_RS_D, true>(h, a, grid);
    }
#define FZ_PICK2(RM, GRF)                                                                     \
    do {                                                                                      \
        if (hk && nn) return fz_launch<RM, true, true, FZ_RS_HK, false, GRF>(h, a, grid);     \
        else if (hk) return fz_launch<RM, true, false, FZ_RS_HK, false, GRF>(h, a, grid);     \
        else if (nn) return fz_launch<RM, false, true, FZ_RS_D, false, GRF>(h, a, grid);      \
        else return fz_launch<RM, false, false, FZ_RS_D, false, GRF>(h, a, grid);             \
    } while (0)
#define FZ_PICK(RM)                     \
    do {                                \
        if (gram_of_r) FZ_PICK2(RM, true); \
        else FZ_PICK2(RM, false);       \
    } while (0)
    if (r <= 8) FZ_PICK(8);
    else FZ_PICK(16);
#undef FZ_PICK
#undef FZ_PICK2
}

}  // namespace tlsq
