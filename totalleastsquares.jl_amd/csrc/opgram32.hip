// The large-mode operator product  Y = Z'(Z X)  for fp32 panels on the fp32 MFMA, second form (round 5) - what the randomized
// hook's block power method (svdstep.hip, src/robustPCA.jl:195-197) spends its time in at BASELINE config 5 (65536 x 4096, blocks
// of 74 columns: 43 GFLOP per half, 0.27 ms at the fp32 MFMA peak, 0.2 ms of HBM).  The first form (gemm.hip: k_tsmm_f32 reads
// the packed block straight from L2 for every 32 rows and splits the inner dimension over the waves of a workgroup;
// k_zt_f32mfma is the Gram kernel's 128 x 128 tile with 54 of its 128 second-operand columns empty) ran at 0.65 + 0.62 ms.
// Here, in both halves, the SHARED operand (the block) goes through LDS once per workgroup and chunk and the PRIVATE operand
// (the panel) goes from global memory straight into the MFMA's A fragments:
//
//   k_zx_f32   T32 (M x LW, fp32) = Z Wt         workgroup = 4 waves x 32 rows; per chunk of 32 panel columns the 32 x LW
//                                               slice of the packed block is staged in LDS (contiguous in memory: Wt is
//                                               [column][LW]); a lane loads TWO consecutive rows of one panel column per k-step
//                                               (the MFMA's row index is a label: tile t holds rows r0 + 2 i + t) - a wave's load
//                                               covers four 128-byte lines; 10 MFMAs per k-step and wave.
//   k_zty_f32  slab[z] (N x LW, fp64) = Z[rows z]' T32[rows z]
//                                               workgroup = 4 waves x 16 panel columns; per chunk of 32 rows the 32 x LW slice
//                                               of T32 is staged as [column][32 + 4]; a lane loads
//                                               8 consecutive rows of its panel column (2 x dwordx4) - the contraction index is a
//                                               label too: MFMA x of a chunk contracts over rows {8 fk + x}; 5 MFMAs per row
//                                               quad; the rows are split over gridDim / (N / 64) workgroups, slabs reduced in
//                                               fixed order (k_zty_reduce).  (The 36-float pitch was chosen for "lanes 0-15" as a
//                                               ds_read_b128 group; the real groups - see gram16.hip - leave two 2-way conflicts per group.)
// fp32 sums are folded into fp64 accumulators after every fourth chunk (128 terms), as in the first form: the result carries the
// fp32 rounding of X and T32 (6e-8) and ~1e-7 from the sums.  Shapes the fast form does not take (M % 128, N % 64, alignment)
// run the first form (also panels below 64 M entries, where its finer grid wins, and blocks of more than 80 columns).
#include "common.hpp"
#include "internal.hpp"

namespace tlsq {

namespace {
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int OG_KC = 32;            // inner indices per staged chunk (both kernels)
constexpr int OG_FOLD = 4;           // chunks per fp64 fold-in (128 terms per fp32 sum)
constexpr int OG_PM = OG_KC + 4;     // floats per staged T32 column: 36 fr mod 64 = 16 distinct multiples of 4 (b128 reads of a quarter wave)
constexpr int OG_RING = 3;           // chunks in flight: the panel fragments of chunk c + 3 are requested when chunk c has been consumed,
                                     // the staged operand of chunk c + 2 likewise (loads complete in order: waiting for the staged
                                     // operand of the next chunk leaves the panel loads two chunks = ~2 us of MFMA work to arrive)

template <int NCT>
__global__ __launch_bounds__(256, 2) void k_zx_f32(const float* __restrict__ Z, int64_t ldz, const float* __restrict__ Wt,
                                                   float* __restrict__ T, int64_t ldt, int N) {
    constexpr int LW = 16 * NCT;
    constexpr int LWP = (NCT == 4) ? 80 : ((NCT == 2 || NCT == 6) ? LW + 16 : LW);   // staged pitch: fk * LWP mod 64 distinct (conflict-free b32 reads)
    constexpr int CH = OG_KC * LW;          // floats per chunk of the packed block (contiguous in memory)
    constexpr int CHP = OG_KC * LWP;        // ... staged
    constexpr int NV = CH / 4;              // float4 per chunk (128 NCT)
    constexpr int SL = (NV + 255) / 256;    // float4 slots per thread
    constexpr int KS = OG_KC / 4;           // k-steps per chunk
    __shared__ __attribute__((aligned(16))) float sW[OG_RING * CHP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * 128 + 32 * w;
    const float* za = Z + r0 + 2 * fr + (int64_t)fk * ldz;
    f4 acc[2][NCT];
    d4 acc64[2][NCT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            acc[t][c] = f4{0.f, 0.f, 0.f, 0.f};
            acc64[t][c] = d4{0.0, 0.0, 0.0, 0.0};
        }
    const int nch = N / OG_KC;
    f4 wreg[SL];
    f2 a[OG_RING][KS];
    auto load_w = [&](int ch) {   // (ch clamped by the caller: every load is unconditional)
        const f4* src = reinterpret_cast<const f4*>(Wt + (size_t)ch * CH);
#pragma unroll
        for (int i = 0; i < SL; ++i) {
            int e = tid + 256 * i;
            if (SL * 256 != NV && e >= NV) e = tid;   // (a partial last slot repeats slot 0: no branch around a load)
            wreg[i] = src[e];
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < SL; ++i) {
            int e = tid + 256 * i;
            if (SL * 256 != NV && e >= NV) e = tid;   // (the same value to the same place as slot 0)
            *reinterpret_cast<f4*>(sW + buf * CHP + (e / (LW / 4)) * LWP + 4 * (e % (LW / 4))) = wreg[i];
        }
    };
    auto load_a = [&](f2* av, int ch) {
        const float* p = za + (int64_t)ch * OG_KC * ldz;
#pragma unroll
        for (int s = 0; s < KS; ++s) av[s] = *reinterpret_cast<const f2*>(p + (int64_t)(4 * s) * ldz);
    };
    auto fold = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc64[t][c][q] += (double)acc[t][c][q];
                acc[t][c] = f4{0.f, 0.f, 0.f, 0.f};
            }
    };
    const int last = nch - 1;
    auto clampc = [&](int ch) { return ch < last ? ch : last; };
    // chunk `ch` sits in ring slot u (fragments a[u], staged operand sW[u]); afterwards the slot's fragments are re-requested for
    // chunk ch + 3 and the staged operand of chunk ch + 1 - in registers since the previous step - goes to slot u + 1
    auto step = [&](auto uc, int ch) {
        constexpr int u = decltype(uc)::value;
        const float* sb = sW + u * CHP + fk * LWP + fr;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float b[NCT];
#pragma unroll
            for (int c = 0; c < NCT; ++c) b[c] = sb[(4 * s) * LWP + 16 * c];
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s][0], b[c], acc[0][c], 0, 0, 0);
                acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s][1], b[c], acc[1][c], 0, 0, 0);
            }
        }
        if ((ch % OG_FOLD) == OG_FOLD - 1) fold();
        __builtin_amdgcn_sched_barrier(0);   // (keeps the steps apart: merged, their fragments do not fit the register file)
        store_w((u + 1) % OG_RING);       // (slot u + 1 held chunk ch - 2: every wave passed the barrier of step ch - 1 after reading it)
        __syncthreads();
        load_w(clampc(ch + 2));
        load_a(a[u], clampc(ch + OG_RING));
        __builtin_amdgcn_sched_barrier(0);
    };
    load_w(0);
    load_a(a[0], 0);
    load_a(a[1], clampc(1));
    load_a(a[2], clampc(2));
    store_w(0);
    __syncthreads();
    load_w(clampc(1));
    std::integral_constant<int, 0> U0;
    std::integral_constant<int, 1> U1;
    std::integral_constant<int, 2> U2;
    int ch = 0;
    for (; ch + OG_RING <= nch; ch += OG_RING) {
        step(U0, ch);
        step(U1, ch + 1);
        step(U2, ch + 2);
    }
    if (ch < nch) step(U0, ch);
    if (ch + 1 < nch) step(U1, ch + 1);
    fold();
    // v_mfma_f32_16x16x4_f32: register q of lane (fr, fk) is D[i = 4 fk + q][j = fr]; tile t, row label i = panel row r0 + 2 i + t
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f2 v;
            v[0] = (float)acc64[0][c][q];
            v[1] = (float)acc64[1][c][q];
            *reinterpret_cast<f2*>(T + r0 + 2 * (4 * fk + q) + (int64_t)(16 * c + fr) * ldt) = v;
        }
}

constexpr int ZTY_WAVES = 4;                    // 16 panel columns each
constexpr int ZTY_COLS = 16 * ZTY_WAVES;        // panel columns per workgroup
template <int NCT>
__global__ __launch_bounds__(64 * ZTY_WAVES, 3) void k_zty_f32(const float* __restrict__ Z, int64_t ldz, const float* __restrict__ T32,
                                                              int64_t ldt, double* __restrict__ slab, int64_t N, int64_t K,
                                                              int64_t kchunk, int64_t slab_stride, int ntiles, int nsplit) {
    constexpr int NT = 64 * ZTY_WAVES;
    constexpr int LW = 16 * NCT;
    constexpr int CH = LW * OG_PM;            // floats per staged chunk
    constexpr int NV = LW * (OG_KC / 4);      // float4 per chunk
    constexpr int SL = (NV + NT - 1) / NT;
    constexpr int QA = OG_KC / 16;            // float4 of the panel per lane and chunk
    __shared__ __attribute__((aligned(16))) float sT[OG_RING * CH];
    const int64_t nwork = (int64_t)ntiles * nsplit;
    const int64_t cpx = (nwork + 7) / 8;
    const int64_t item = (int64_t)(blockIdx.x % 8) * cpx + (int64_t)(blockIdx.x / 8);   // XCD x takes a contiguous run of items
    if (item >= nwork) return;
    const int z = (int)(item / ntiles), ti = (int)(item % ntiles);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int64_t kbeg = (int64_t)z * kchunk;
    const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    const int nch = (int)((kend - kbeg) / OG_KC);   // (>= 1: the host makes every split non-empty)
    const int64_t n0 = (int64_t)ti * ZTY_COLS + 16 * w;
    const bool active = n0 < N;               // (N a multiple of 16: a wave's 16 columns are all inside or all outside)
    const float* za = Z + (active ? n0 + fr : 0) * ldz + kbeg + 4 * QA * fk;
    f4 acc[NCT];
    d4 acc64[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        acc[c] = f4{0.f, 0.f, 0.f, 0.f};
        acc64[c] = d4{0.0, 0.0, 0.0, 0.0};
    }
    f4 treg[SL];
    f4 a[OG_RING][QA];
    auto load_t = [&](int ch) {
#pragma unroll
        for (int i = 0; i < SL; ++i) {
            int e = tid + NT * i;
            if (SL * NT != NV && e >= NV) e = tid;   // (a partial last slot repeats slot 0: no branch around a load)
            const int j = e / (OG_KC / 4), q = e % (OG_KC / 4);
            treg[i] = *reinterpret_cast<const f4*>(T32 + kbeg + (int64_t)ch * OG_KC + 4 * q + (int64_t)j * ldt);
        }
    };
    auto store_t = [&](int buf) {
#pragma unroll
        for (int i = 0; i < SL; ++i) {
            int e = tid + NT * i;
            if (SL * NT != NV && e >= NV) e = tid;
            const int j = e / (OG_KC / 4), q = e % (OG_KC / 4);
            *reinterpret_cast<f4*>(sT + buf * CH + j * OG_PM + 4 * q) = treg[i];
        }
    };
    auto load_a = [&](f4* av, int ch) {
        const f4* p = reinterpret_cast<const f4*>(za + (int64_t)ch * OG_KC);
#pragma unroll
        for (int q = 0; q < QA; ++q) av[q] = p[q];
    };
    auto fold = [&]() {
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc64[c][q] += (double)acc[c][q];
            acc[c] = f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    const int last = nch - 1;
    auto clampc = [&](int ch) { return ch < last ? ch : last; };
    auto step = [&](auto uc, int ch) {
        constexpr int u = decltype(uc)::value;
        // lane (fr, fk) holds rows 4 QA fk .. + 4 QA - 1 of the chunk: MFMA (q, x) contracts over rows {4 QA fk + 4 q + x}
        const float* sb = sT + u * CH + fr * OG_PM + 4 * QA * fk;
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            f4 b[NCT];
#pragma unroll
            for (int c = 0; c < NCT; ++c) b[c] = *reinterpret_cast<const f4*>(sb + (16 * c) * OG_PM + 4 * q);
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int c = 0; c < NCT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][q][x], b[c][x], acc[c], 0, 0, 0);
        }
        if ((ch % OG_FOLD) == OG_FOLD - 1) fold();
        __builtin_amdgcn_sched_barrier(0);
        store_t((u + 1) % OG_RING);
        __syncthreads();
        load_t(clampc(ch + 2));
        load_a(a[u], clampc(ch + OG_RING));
        __builtin_amdgcn_sched_barrier(0);
    };
    load_t(0);
    load_a(a[0], 0);
    load_a(a[1], clampc(1));
    load_a(a[2], clampc(2));
    store_t(0);
    __syncthreads();
    load_t(clampc(1));
    std::integral_constant<int, 0> U0;
    std::integral_constant<int, 1> U1;
    std::integral_constant<int, 2> U2;
    int ch = 0;
    for (; ch + OG_RING <= nch; ch += OG_RING) {
        step(U0, ch);
        step(U1, ch + 1);
        step(U2, ch + 2);
    }
    if (ch < nch) step(U0, ch);
    if (ch + 1 < nch) step(U1, ch + 1);
    fold();
    if (!active) return;
    // register q of lane (fr, fk): D[i = 4 fk + q][j = fr] = Y[n0 + 4 fk + q][16 c + fr]
    double* out = slab + (int64_t)z * slab_stride;
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) out[(n0 + 4 * fk + q) * LW + 16 * c + fr] = acc64[c][q];
}

// Y[i + j ldy] = sum over the row splits of slab[z][j + i pitch]  (i < N, j < p), fixed order
__global__ __launch_bounds__(256) void k_zty_reduce(const double* __restrict__ slab, int64_t slab_stride, int nsplit, int pitch,
                                                    double* __restrict__ Y, int64_t ldy, int64_t N, int p) {
    const int64_t total = N * p;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e % N;
        const int j = (int)(e / N);
        double sacc = 0.0;
        for (int z = 0; z < nsplit; ++z) sacc += slab[(int64_t)z * slab_stride + j + i * pitch];
        Y[i + (int64_t)j * ldy] = sacc;
    }
}
// Wt[k lw + j] = (float) W[k + j ldw]  (j < r, else 0): the packed second operand of k_zx_f32
__global__ __launch_bounds__(256) void k_pack_wt32(const double* __restrict__ W, int64_t ldw, int K, int r, int lw,
                                                   float* __restrict__ Wt) {
    const int total = K * lw;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int j = e % lw, k = e / lw;
        Wt[e] = j < r ? (float)W[(size_t)k + (size_t)j * ldw] : 0.f;
    }
}
// out (n x lw fp32, ld n) = (float) V (n x r fp64, ld ldv), columns r.. zero
__global__ __launch_bounds__(256) void k_cols_to_f32(const double* __restrict__ V, int64_t ldv, int64_t n, int r, int lw,
                                                     float* __restrict__ out) {
    const int64_t total = n * lw;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e % n;
        const int j = (int)(e / n);
        out[e] = j < r ? (float)V[i + (int64_t)j * ldv] : 0.f;
    }
}
// Tm (M x r fp64, ld M) = (double) T32 (M x lw fp32, ld ldt)
__global__ __launch_bounds__(256) void k_cols_to_f64(const float* __restrict__ T32, int64_t ldt, int64_t M, int r,
                                                     double* __restrict__ Tm) {
    const int64_t total = M * r;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e % M, j = e / M;
        Tm[e] = (double)T32[i + j * ldt];
    }
}
// T32 (M x lw fp32, ld M; columns r.. zero) and Tm (M x r fp64) = ZQ (M x p of the lwq stored columns, fp32, ld M) * C (p x r):
// C[k][j] = S[k + order[sel[j]] p] * g[j] - the factor Z X[:, sel] diag(g) of the rebuild from the hook's own product Z Q
// (X = Q S).  One thread per row, C in LDS as [k][j] (broadcast reads), r <= 80 accumulators in registers.
template <int RMAX>
__global__ __launch_bounds__(256) void k_zq_times_c(const float* __restrict__ ZQ, int64_t M, int p, const double* __restrict__ S,
                                                    const int32_t* __restrict__ cols, const double* __restrict__ g, int r, int lw,
                                                    float* __restrict__ T32, double* __restrict__ Tm) {
    extern __shared__ float sC[];   // p x RMAX
    for (int e = threadIdx.x; e < p * RMAX; e += 256) {
        const int k = e / RMAX, j = e % RMAX;
        sC[e] = j < r ? (float)(S[(size_t)k + (size_t)cols[j] * p] * g[j]) : 0.f;
    }
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= M) return;
    float acc[RMAX];
#pragma unroll
    for (int j = 0; j < RMAX; ++j) acc[j] = 0.f;
    for (int k = 0; k < p; ++k) {
        const float a = ZQ[row + (int64_t)k * M];
        const float* c = sC + k * RMAX;
#pragma unroll
        for (int j = 0; j < RMAX; ++j) acc[j] += a * c[j];
    }
#pragma unroll
    for (int j = 0; j < RMAX; ++j) {
        if (j < lw) T32[row + (int64_t)j * M] = acc[j];
        if (j < r) Tm[row + (int64_t)j * M] = (double)acc[j];
    }
}
}   // namespace

bool op_gram_f32_fast_ok(const float* Z, int64_t ldz, int64_t M, int64_t N, int64_t p) {
    return p > 0 && p <= 80 && (M % 128) == 0 && (N % 64) == 0 && (ldz % 4) == 0 && (reinterpret_cast<uintptr_t>(Z) % 16) == 0 &&
           M * N >= ((int64_t)1 << 26);   // (below ~64 M entries the first form's finer grid wins: 8192 x 1024, 0.064 against 0.092 ms)
}

// T32 (M x 16 nct, ld M) = Z Wt;  slab / Y = Z' T32   (the caller has packed X into Wt [N][16 nct] fp32 and owns the buffers)
int op_gram_f32_fast(Handle* h, const float* Z, int64_t ldz, int64_t M, int64_t N, const float* wt, float* t32, double* Y,
                     int64_t ldy, int64_t p) {
    const int nct = (int)((p + 15) / 16), lw = 16 * nct;
    {
        const dim3 grid((unsigned)(M / 128));
#define ZX_LAUNCH(NC) hipLaunchKernelGGL((k_zx_f32<NC>), grid, dim3(256), 0, h->stream, Z, ldz, wt, t32, M, (int)N)
        switch (nct) {
            case 1: ZX_LAUNCH(1); break;
            case 2: ZX_LAUNCH(2); break;
            case 3: ZX_LAUNCH(3); break;
            case 4: ZX_LAUNCH(4); break;
            default: ZX_LAUNCH(5); break;
        }
#undef ZX_LAUNCH
    }
    TLSQ_HIP(h, hipGetLastError());
    const int64_t ntiles = (N + ZTY_COLS - 1) / ZTY_COLS;
    int64_t nsplit = std::max<int64_t>(1, (1024 + ntiles - 1) / ntiles);   // four workgroups per CU
    nsplit = std::min<int64_t>(nsplit, std::max<int64_t>(1, M / (4 * OG_KC)));
    int64_t kchunk = (M + nsplit - 1) / nsplit;
    kchunk = (kchunk + OG_KC - 1) / OG_KC * OG_KC;
    nsplit = (M + kchunk - 1) / kchunk;
    const int64_t slab_stride = N * lw;
    void* slab;
    TLSQ_TRY(ws_get(h, WS_SLAB, (size_t)(nsplit * slab_stride) * 8, &slab));
    const int64_t nwork = ntiles * nsplit, cpx = (nwork + 7) / 8;
#define ZTY_LAUNCH(NC)                                                                                                         \
    hipLaunchKernelGGL((k_zty_f32<NC>), dim3((unsigned)(8 * cpx)), dim3(64 * ZTY_WAVES), 0, h->stream, Z, ldz, (const float*)t32, M, \
                       (double*)slab, N, M, kchunk, slab_stride, (int)ntiles, (int)nsplit)
    switch (nct) {
        case 1: ZTY_LAUNCH(1); break;
        case 2: ZTY_LAUNCH(2); break;
        case 3: ZTY_LAUNCH(3); break;
        case 4: ZTY_LAUNCH(4); break;
        default: ZTY_LAUNCH(5); break;
    }
#undef ZTY_LAUNCH
    hipLaunchKernelGGL(k_zty_reduce, dim3((unsigned)std::min<int64_t>((N * p + 255) / 256, 2048)), dim3(256), 0, h->stream,
                       (const double*)slab, slab_stride, (int)nsplit, lw, Y, ldy, N, (int)p);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}


// The factors of A_k for the wide sweep (sweeps.hip, k_zsweep_wide) on an fp32 panel, ranks 33..80:
//   T32  (M x lw fp32, ld M) = Z Vg   on the fp32 MFMA (k_zx_f32: 0.43 ms at 65536 x 4096 against 1.11 ms of the widening
//                                     fp64 product), lw = 64 for r <= 64, else 80
//   Vs32 (N x lw fp32, ld N) = Vs
//   Tm   (M x r fp64)        = T32 widened: what every other consumer of the factors reads (final A, final E)
bool wide_factors_ok(const float* Z, int64_t ldz, int64_t M, int64_t N, int64_t r) {
    return r > 32 && r <= 80 && op_gram_f32_fast_ok(Z, ldz, M, N, r <= 64 ? 64 : 80);
}

int wide_factors_f32(Handle* h, const float* Z, int64_t ldz, int64_t M, int64_t N, const double* Vg, const double* Vs,
                     int64_t r, double* Tm, const float** T32_out, const float** Vs32_out, int* lw_out) {
    const int lw = r <= 64 ? 64 : 80;
    void *wt, *t32, *vs32;
    TLSQ_TRY(ws_get(h, WS_OPW, (size_t)N * lw * 8, &wt));
    TLSQ_TRY(ws_get(h, WS_T32, (size_t)M * lw * 4, &t32));
    TLSQ_TRY(ws_get(h, WS_VS32, (size_t)N * lw * 4, &vs32));
    hipLaunchKernelGGL(k_cols_to_f32, dim3((unsigned)std::min<int64_t>((N * lw + 255) / 256, 1024)), dim3(256), 0, h->stream, Vs, N,
                       N, (int)r, lw, (float*)vs32);
    if (h->absmax_panel == (const void*)Z && ldz == M && !dev_is(DEV_OPGRAM_H3, '0')) {
        // (the sweep that wrote the panel left its maximum: the product on the fp16 MFMA, operands split in registers - opgram16.hip)
        void* sc;
        TLSQ_TRY(ws_get(h, WS_H16S, 64, &sc));
        const unsigned int* zmax = reinterpret_cast<const unsigned int*>(reinterpret_cast<char*>(sc) + 40);
        TLSQ_TRY(tsmm_f32_h3(h, Z, ldz, M, N, Vg, N, r, lw, (float*)t32, zmax, nullptr));
    } else {
        hipLaunchKernelGGL(k_pack_wt32, dim3((unsigned)std::min<int64_t>((N * lw + 255) / 256, 1024)), dim3(256), 0, h->stream, Vg, N,
                           (int)N, (int)r, lw, (float*)wt);
        const dim3 grid((unsigned)(M / 128));
        if (lw == 64) hipLaunchKernelGGL((k_zx_f32<4>), grid, dim3(256), 0, h->stream, Z, ldz, (const float*)wt, (float*)t32, M, (int)N);
        else hipLaunchKernelGGL((k_zx_f32<5>), grid, dim3(256), 0, h->stream, Z, ldz, (const float*)wt, (float*)t32, M, (int)N);
    }
    if (Tm)
        hipLaunchKernelGGL(k_cols_to_f64, dim3((unsigned)std::min<int64_t>((M * r + 255) / 256, 4096)), dim3(256), 0, h->stream,
                           (const float*)t32, M, M, (int)r, Tm);
    TLSQ_HIP(h, hipGetLastError());
    *T32_out = (const float*)t32;
    *Vs32_out = (const float*)vs32;
    *lw_out = lw;
    return TLSQ_OK;
}

// The same factors when the hook's Rayleigh-Ritz product Z Q is still there (SubspaceState::hook_zq): T32 = (Z Q) S[:, cols] diag(g)
// - 21 MB read instead of a pass over the panel (0.04 ms against 0.37 at 65536 x 4096, 74 -> 64 columns).  cols_dev: r int32,
// g_dev: r doubles (device).
int wide_factors_from_zq(Handle* h, const float* ZQ, int64_t M, int64_t N, int64_t p, const double* S_dev, const int32_t* cols_dev,
                         const double* g_dev, const double* Vs, int64_t r, double* Tm, const float** T32_out,
                         const float** Vs32_out, int* lw_out) {
    const int lw = r <= 64 ? 64 : 80;
    void *t32, *vs32;
    TLSQ_TRY(ws_get(h, WS_T32, (size_t)M * lw * 4, &t32));
    TLSQ_TRY(ws_get(h, WS_VS32, (size_t)N * lw * 4, &vs32));
    hipLaunchKernelGGL(k_cols_to_f32, dim3((unsigned)std::min<int64_t>((N * lw + 255) / 256, 1024)), dim3(256), 0, h->stream, Vs, N,
                       N, (int)r, lw, (float*)vs32);
    const dim3 grid((unsigned)((M + 255) / 256));
    if (lw == 64)
        hipLaunchKernelGGL((k_zq_times_c<64>), grid, dim3(256), (size_t)p * 64 * 4, h->stream, ZQ, M, (int)p, S_dev, cols_dev, g_dev,
                           (int)r, lw, (float*)t32, Tm);
    else
        hipLaunchKernelGGL((k_zq_times_c<80>), grid, dim3(256), (size_t)p * 80 * 4, h->stream, ZQ, M, (int)p, S_dev, cols_dev, g_dev,
                           (int)r, lw, (float*)t32, Tm);
    TLSQ_HIP(h, hipGetLastError());
    *T32_out = (const float*)t32;
    *Vs32_out = (const float*)vs32;
    *lw_out = lw;
    return TLSQ_OK;
}

}   // namespace tlsq
