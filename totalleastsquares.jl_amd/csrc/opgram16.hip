// The operator product  Y = Z'(Z X)  of the randomized hook (svdstep.hip; src/robustPCA.jl:195-197) for fp32 panels on the fp16
// MFMA, third form (round 5).  The second form (opgram32.hip) runs both halves at the practical ceiling of the fp32 MFMA (0.43 +
// 0.45 ms at 65536 x 4096, 74 columns: 107 TFLOP/s); the panel is 1.07 GB, i.e. 0.2 ms per half at the HBM rate.  Here every
// operand is split on the fly into two fp16 numbers after a power-of-two scaling - x c = h + l + d, |d| <= 2^-22 |x c| (see
// gram16.hip) - and every product runs as  h h + h l + l h  on v_mfma_f32_16x16x32_f16 (2.5 PFLOP/s: three of them cost a fifth
// of the fp32 instruction's time for the same tile):
//   * the panel is the PRIVATE operand of a wave in both halves (opgram32.hip): its fragments are loaded as fp32 exactly as
//     before and converted in registers - 8 (k_zx_h) / 5 (k_zty_h) vector instructions per entry pair, once per entry;
//   * the block X is split by the pack kernel into two pre-swizzled fp16 planes (a stage is a straight copy into LDS);
//   * T32 = Z X stays an fp32 panel (its consumers - the hook's factor, the sweep - read it); the staging threads of k_zty_h
//     split it on the way into LDS.
// The scales: max |Z| is left by the sweep that wrote the panel (Handle::absmax_panel, sweeps.hip) - without it this form is
// not used (a pass for the maximum would cost what it saves); max |X| by the pack kernel; max |T32| by k_zx_h's epilogue.
// fp32 partial sums are folded into fp64 every 128 terms as in the other forms.  Error against float64: see
// tests/test_gpu_tsqr.py (the products carry 22 bits of every operand: ~1e-7 of ||Z||^2 ||x||, the fp32 forms 2e-9).
#include "common.hpp"
#include "internal.hpp"

namespace tlsq {

namespace {
typedef _Float16 qh8 __attribute__((ext_vector_type(8)));
typedef _Float16 qh4 __attribute__((ext_vector_type(4)));
typedef _Float16 qh2 __attribute__((ext_vector_type(2)));
typedef float qf4 __attribute__((ext_vector_type(4)));
typedef float qf2 __attribute__((ext_vector_type(2)));
typedef double qd4 __attribute__((ext_vector_type(4)));
typedef unsigned int qu4 __attribute__((ext_vector_type(4)));
typedef unsigned int qu2 __attribute__((ext_vector_type(2)));

constexpr int QK = 32;        // inner indices per chunk = one MFMA step
constexpr int QFOLD = 4;      // chunks per fp64 fold-in
constexpr int QRING = 3;      // chunks of panel fragments in flight (see opgram32.hip)

// slot of row group g (8 halfs) of staged column c: conflict-free ds_read_b128 for the real lane groups (gram16.hip)
__device__ __forceinline__ int q_sw(int c) { return (-(c >> 2)) & 3; }

// c = 2^(target - e) for max = 1.f x 2^e (bits of a non-negative float); 1 for zero / non-finite maxima
__device__ __forceinline__ float q_scale(unsigned int maxbits, int target) {
    int e = (int)(maxbits >> 23) - 127;
    if (maxbits == 0 || maxbits >= 0x7F800000u) e = target;
    int se = target - e;
    se = se > 120 ? 120 : (se < -100 ? -100 : se);
    return __uint_as_float((unsigned int)(se + 127) << 23);
}
__device__ __forceinline__ void q_split(float x, _Float16& hv, _Float16& lv) {
    hv = (_Float16)x;
    lv = (_Float16)(x - (float)hv);
}

// max |X| of the block (N x p fp64, ld ldx) as float bits
__global__ __launch_bounds__(256) void k_xmax_bits(const double* __restrict__ X, int64_t ldx, int64_t N, int p,
                                                   unsigned int* __restrict__ out) {
    const int64_t total = N * p;
    unsigned int m = 0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const float v = (float)X[(e % N) + (e / N) * ldx];
        const unsigned int a = __float_as_uint(v) & 0x7FFFFFFFu;
        m = a > m ? a : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned int o = (unsigned int)__shfl_xor((int)m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// The two fp16 planes of the block, chunk by chunk of 32 rows of X: plane[(chunk lw + j) 32 + 8 (g ^ sw(j)) + r] = split(X[32 chunk +
// 8 g + r, j] cX), columns p.. zero.  Wh and Wl: N x lw halfs each.
__global__ __launch_bounds__(256) void k_pack_x16(const double* __restrict__ X, int64_t ldx, int64_t N, int p, int lw,
                                                  const unsigned int* __restrict__ xmax, _Float16* __restrict__ Wh,
                                                  _Float16* __restrict__ Wl) {
    const float c = q_scale(*xmax, 13);
    const int64_t total = N * lw;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t k = e % N;
        const int j = (int)(e / N);
        const float x = j < p ? (float)X[k + (int64_t)j * ldx] * c : 0.f;
        _Float16 hv, lv;
        q_split(x, hv, lv);
        const int64_t chunk = k >> 5;
        const int kk = (int)(k & 31), g = kk >> 3, r = kk & 7;
        const int64_t o = (chunk * lw + j) * 32 + 8 * (g ^ q_sw(j)) + r;
        Wh[o] = hv;
        Wl[o] = lv;
    }
}

// T32 (M x LW fp32, ld ldt) = Z (M x N fp32, ld ldz) X;  tmax: max |T32| as float bits (zeroed by the caller)
template <int NCT>
__global__ __launch_bounds__(256, 2) void k_zx_h(const float* __restrict__ Z, int64_t ldz, const _Float16* __restrict__ Wh,
                                                 const _Float16* __restrict__ Wl, float* __restrict__ T, int64_t ldt, int N,
                                                 const unsigned int* __restrict__ zmax, const unsigned int* __restrict__ xmax,
                                                 unsigned int* __restrict__ tmax) {
    constexpr int LW = 16 * NCT;
    constexpr int CH = LW * 32;              // halfs per plane and chunk (contiguous in memory, pre-swizzled)
    constexpr int NV = 2 * CH / 8;           // 16-byte pieces per chunk (both planes)
    constexpr int SL = (NV + 255) / 256;
    __shared__ __attribute__((aligned(16))) _Float16 sW[QRING * 2 * CH];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, kg = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * 128 + 32 * w;
    const float cz = q_scale(*zmax, 13), cx = q_scale(*xmax, 13);
    const double inv = 1.0 / ((double)cz * (double)cx);
    const float* za = Z + r0 + 2 * fr + (int64_t)(8 * kg) * ldz;
    qf4 acc[2][NCT];
    qd4 acc64[2][NCT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            acc[t][c] = qf4{0.f, 0.f, 0.f, 0.f};
            acc64[t][c] = qd4{0.0, 0.0, 0.0, 0.0};
        }
    const int nch = N / QK;
    qu4 wreg[SL];
    qf2 a[QRING][8];
    auto load_w = [&](int ch) {
#pragma unroll
        for (int i = 0; i < SL; ++i) {
            int e = tid + 256 * i;
            if (SL * 256 != NV && e >= NV) e = tid;
            const _Float16* src = e < CH / 8 ? Wh + (size_t)ch * CH + 8 * e : Wl + (size_t)ch * CH + 8 * (e - CH / 8);
            wreg[i] = *reinterpret_cast<const qu4*>(src);
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < SL; ++i) {
            int e = tid + 256 * i;
            if (SL * 256 != NV && e >= NV) e = tid;
            *reinterpret_cast<qu4*>(sW + buf * 2 * CH + 8 * e) = wreg[i];
        }
    };
    auto load_a = [&](qf2* av, int ch) {
        const float* p = za + (int64_t)ch * QK * ldz;
#pragma unroll
        for (int q = 0; q < 8; ++q) av[q] = *reinterpret_cast<const qf2*>(p + (int64_t)q * ldz);
    };
    auto fold = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc64[t][c][q] += (double)acc[t][c][q];
                acc[t][c] = qf4{0.f, 0.f, 0.f, 0.f};
            }
    };
    const int last = nch - 1;
    auto clampc = [&](int ch) { return ch < last ? ch : last; };
    const int ob = fr * 32 + 8 * (kg ^ q_sw(fr));
    auto step = [&](auto uc, int ch) {
        constexpr int u = decltype(uc)::value;
        // this lane's eight panel entries of the chunk, two rows (tiles t = 0, 1), split in registers
        qh8 ah[2], al[2];
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                _Float16 hv, lv;
                q_split(a[u][q][t] * cz, hv, lv);
                ah[t][q] = hv;
                al[t][q] = lv;
            }
        const _Float16* sh = sW + u * 2 * CH + ob;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const qh8 bh = *reinterpret_cast<const qh8*>(sh + (16 * c) * 32);
            const qh8 bl = *reinterpret_cast<const qh8*>(sh + CH + (16 * c) * 32);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh, acc[t][c], 0, 0, 0);
                acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl, acc[t][c], 0, 0, 0);
                acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, acc[t][c], 0, 0, 0);
            }
        }
        if ((ch % QFOLD) == QFOLD - 1) fold();
        __builtin_amdgcn_sched_barrier(0);
        store_w((u + 1) % QRING);
        __syncthreads();
        load_w(clampc(ch + 2));
        load_a(a[u], clampc(ch + QRING));
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
    for (; ch + QRING <= nch; ch += QRING) {
        step(U0, ch);
        step(U1, ch + 1);
        step(U2, ch + 2);
    }
    if (ch < nch) step(U0, ch);
    if (ch + 1 < nch) step(U1, ch + 1);
    fold();
    // register q of lane (fr, kg): D[i = 4 kg + q][j = fr]; tile t, row label i = panel row r0 + 2 i + t
    unsigned int tm = 0;
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            qf2 v;
            v[0] = (float)(acc64[0][c][q] * inv);
            v[1] = (float)(acc64[1][c][q] * inv);
            *reinterpret_cast<qf2*>(T + r0 + 2 * (4 * kg + q) + (int64_t)(16 * c + fr) * ldt) = v;
            const unsigned int b0 = __float_as_uint(v[0]) & 0x7FFFFFFFu, b1 = __float_as_uint(v[1]) & 0x7FFFFFFFu;
            tm = b0 > tm ? b0 : tm;
            tm = b1 > tm ? b1 : tm;
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned int o = (unsigned int)__shfl_xor((int)tm, off, 64);
        tm = o > tm ? o : tm;
    }
    if (lane == 0 && tm) atomicMax(tmax, tm);
}

constexpr int QZ_WAVES = 4;
constexpr int QZ_COLS = 16 * QZ_WAVES;
constexpr int QZK = 64;       // rows per stage of k_zty_h: two MFMA steps per barrier (one step per barrier: 0.30 ms, barrier-paced)
constexpr int QZRING = 2;     // stages of panel fragments in flight
// slab[z] (N x LW fp64, [n LW + j]) = Z[rows z]' T32[rows z]
template <int NCT>
__global__ __launch_bounds__(64 * QZ_WAVES, 2) void k_zty_h(const float* __restrict__ Z, int64_t ldz, const float* __restrict__ T32,
                                                            int64_t ldt, double* __restrict__ slab, int64_t N, int64_t K,
                                                            int64_t kchunk, int64_t slab_stride, int ntiles, int nsplit,
                                                            const unsigned int* __restrict__ zmax,
                                                            const unsigned int* __restrict__ tmax) {
    constexpr int NT = 64 * QZ_WAVES;
    constexpr int LW = 16 * NCT;
    constexpr int CH = LW * 32;               // halfs per plane and MFMA step
    constexpr int SB = 4 * CH;                // halfs per stage buffer: step 0 (hi, lo), step 1 (hi, lo)
    constexpr int NV = LW * 16;               // float4 of T32 per stage
    constexpr int SL = (NV + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) _Float16 sT[2 * SB];
    const int64_t nwork = (int64_t)ntiles * nsplit;
    const int64_t cpx = (nwork + 7) / 8;
    const int64_t item = (int64_t)(blockIdx.x % 8) * cpx + (int64_t)(blockIdx.x / 8);
    if (item >= nwork) return;
    const int z = (int)(item / ntiles), ti = (int)(item % ntiles);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, kg = lane >> 4;
    const int64_t kbeg = (int64_t)z * kchunk;
    const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    const int nst = (int)((kend - kbeg) / QZK);
    const int64_t n0 = (int64_t)ti * QZ_COLS + 16 * w;
    const bool active = n0 < N;
    const float cz = q_scale(*zmax, 13), ct = q_scale(*tmax, 13);
    const double inv = 1.0 / ((double)cz * (double)ct);
    const float* za = Z + (active ? n0 + fr : 0) * ldz + kbeg + 8 * kg;
    qf4 acc[NCT];
    qd4 acc64[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        acc[c] = qf4{0.f, 0.f, 0.f, 0.f};
        acc64[c] = qd4{0.0, 0.0, 0.0, 0.0};
    }
    qf4 treg[SL];
    qf4 a[QZRING + 1][4];
    auto load_t = [&](int st) {
#pragma unroll
        for (int i = 0; i < SL; ++i) {
            int e = tid + NT * i;
            if (SL * NT != NV && e >= NV) e = tid;
            const int j = e >> 4, q = e & 15;
            treg[i] = *reinterpret_cast<const qf4*>(T32 + kbeg + (int64_t)st * QZK + 4 * q + (int64_t)j * ldt);
        }
    };
    // (rows 4 q .. 4 q + 3 of column j, q < 16: MFMA step q / 8, row group (q % 8) / 2, offset 4 (q & 1); both planes)
    auto store_t = [&](int buf) {
#pragma unroll
        for (int i = 0; i < SL; ++i) {
            int e = tid + NT * i;
            if (SL * NT != NV && e >= NV) e = tid;
            const int j = e >> 4, q = e & 15;
            qh4 hh, ll;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                _Float16 hv, lv;
                q_split(treg[i][k] * ct, hv, lv);
                hh[k] = hv;
                ll[k] = lv;
            }
            _Float16* d = sT + buf * SB + (q >> 3) * 2 * CH + j * 32 + 8 * (((q & 7) >> 1) ^ q_sw(j)) + 4 * (q & 1);
            *reinterpret_cast<qh4*>(d) = hh;
            *reinterpret_cast<qh4*>(d + CH) = ll;
        }
    };
    auto load_a = [&](qf4* av, int st) {
        const float* p = za + (int64_t)st * QZK;
        av[0] = *reinterpret_cast<const qf4*>(p);
        av[1] = *reinterpret_cast<const qf4*>(p + 4);
        av[2] = *reinterpret_cast<const qf4*>(p + 32);
        av[3] = *reinterpret_cast<const qf4*>(p + 36);
    };
    auto fold = [&]() {
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc64[c][q] += (double)acc[c][q];
            acc[c] = qf4{0.f, 0.f, 0.f, 0.f};
        }
    };
    const int last = nst - 1;
    auto clamps = [&](int st) { return st < last ? st : last; };
    const int ob = fr * 32 + 8 * (kg ^ q_sw(fr));
    // stage st: planes in LDS buffer st & 1; the fragments of ring slot u (= st % 3) were requested three stages ago
    auto stage = [&](auto uc, auto bc, int st) {
        constexpr int u = decltype(uc)::value;
        constexpr int bsel = decltype(bc)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qh8 ah, al;   // rows 32 ks + 8 kg .. + 7 of the stage, this lane's panel column
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                _Float16 hv, lv;
                q_split(a[u][2 * ks + (q >> 2)][q & 3] * cz, hv, lv);
                ah[q] = hv;
                al[q] = lv;
            }
            const _Float16* sh = sT + bsel * SB + ks * 2 * CH + ob;
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const qh8 bh = *reinterpret_cast<const qh8*>(sh + (16 * c) * 32);
                const qh8 bl = *reinterpret_cast<const qh8*>(sh + CH + (16 * c) * 32);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[c], 0, 0, 0);
            }
        }
        if ((st & 1) == 1) fold();                 // every 128 rows
        __builtin_amdgcn_sched_barrier(0);
        store_t(bsel ^ 1);
        __syncthreads();
        load_t(clamps(st + 2));
        load_a(a[u], clamps(st + 3));
        __builtin_amdgcn_sched_barrier(0);
    };
    load_t(0);
    load_a(a[0], 0);
    load_a(a[1], clamps(1));
    load_a(a[2], clamps(2));
    store_t(0);
    __syncthreads();
    load_t(clamps(1));
    std::integral_constant<int, 0> I0;
    std::integral_constant<int, 1> I1;
    std::integral_constant<int, 2> I2;
    int st = 0;
    for (; st + 6 <= nst; st += 6) {               // (ring of three fragment slots x two LDS buffers: period six)
        stage(I0, I0, st);
        stage(I1, I1, st + 1);
        stage(I2, I0, st + 2);
        stage(I0, I1, st + 3);
        stage(I1, I0, st + 4);
        stage(I2, I1, st + 5);
    }
    if (st < nst) stage(I0, I0, st);
    if (st + 1 < nst) stage(I1, I1, st + 1);
    if (st + 2 < nst) stage(I2, I0, st + 2);
    if (st + 3 < nst) stage(I0, I1, st + 3);
    if (st + 4 < nst) stage(I1, I0, st + 4);
    fold();
    if (!active) return;
    double* out = slab + (int64_t)z * slab_stride;
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) out[(n0 + 4 * kg + q) * LW + 16 * c + fr] = acc64[c][q] * inv;
}

__global__ __launch_bounds__(256) void k_zty_h_reduce(const double* __restrict__ slab, int64_t slab_stride, int nsplit, int pitch,
                                                      double* __restrict__ Y, int64_t ldy, int64_t N, int p) {
    const int64_t total = N * p;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e % N;
        const int j = (int)(e / N);
        double sacc = 0.0;
        for (int zz = 0; zz < nsplit; ++zz) sacc += slab[(int64_t)zz * slab_stride + j + i * pitch];
        Y[i + (int64_t)j * ldy] = sacc;
    }
}
}   // namespace

// Y (N x p fp64, ld ldy) = Z'(Z X) with the panel's maximum known (zmax_bits: device word, float bits of max |Z|); T32 (M x 16
// ceil(p / 16), ld M) = Z X is left in t32 as in the other forms.  Shapes: as op_gram_f32_fast_ok.
// first half alone: T32 (M x lw fp32, ld M; lw a multiple of 16 >= p, columns p.. zero) = Z X - the factor product of the rebuild
// for ranks above 32 (opgram32.hip, wide_factors_f32) when the panel's maximum is known.  tmax_out (optional): device word that
// receives max |T32| as float bits.
int tsmm_f32_h3(Handle* h, const float* Z, int64_t ldz, int64_t M, int64_t N, const double* X, int64_t ldx, int64_t p, int lw,
                float* t32, const unsigned int* zmax_bits, unsigned int** tmax_out) {
    const int nct = lw / 16;
    void *wh, *wl, *sc;
    TLSQ_TRY(ws_get(h, WS_OPW, (size_t)N * lw * 8, &wh));
    wl = reinterpret_cast<_Float16*>(wh) + (size_t)N * lw;
    TLSQ_TRY(ws_get(h, WS_OPSC, 64, &sc));
    unsigned int* xmax = reinterpret_cast<unsigned int*>(sc);
    unsigned int* tmax = xmax + 2;
    TLSQ_HIP(h, hipMemsetAsync(sc, 0, 16, h->stream));
    hipLaunchKernelGGL(k_xmax_bits, dim3(256), dim3(256), 0, h->stream, X, ldx, N, (int)p, xmax);
    hipLaunchKernelGGL(k_pack_x16, dim3((unsigned)std::min<int64_t>((N * lw + 255) / 256, 1024)), dim3(256), 0, h->stream, X, ldx, N,
                       (int)p, lw, (const unsigned int*)xmax, (_Float16*)wh, (_Float16*)wl);
    const dim3 grid((unsigned)(M / 128));
#define ZXH(NC)                                                                                                               \
    hipLaunchKernelGGL((k_zx_h<NC>), grid, dim3(256), 0, h->stream, Z, ldz, (const _Float16*)wh, (const _Float16*)wl, t32, M, (int)N, \
                       zmax_bits, (const unsigned int*)xmax, tmax)
    switch (nct) {
        case 1: ZXH(1); break;
        case 2: ZXH(2); break;
        case 3: ZXH(3); break;
        case 4: ZXH(4); break;
        default: ZXH(5); break;
    }
#undef ZXH
    TLSQ_HIP(h, hipGetLastError());
    ++h->kern_zx_h;
    if (tmax_out) *tmax_out = tmax;
    return TLSQ_OK;
}

int op_gram_f32_h3(Handle* h, const float* Z, int64_t ldz, int64_t M, int64_t N, const double* X, int64_t ldx, float* t32, double* Y,
                   int64_t ldy, int64_t p, const unsigned int* zmax_bits) {
    const int nct = (int)((p + 15) / 16), lw = 16 * nct;
    unsigned int* tmax = nullptr;
    TLSQ_TRY(tsmm_f32_h3(h, Z, ldz, M, N, X, ldx, p, lw, t32, zmax_bits, &tmax));
    const int64_t ntiles = (N + QZ_COLS - 1) / QZ_COLS;
    int64_t nsplit = std::max<int64_t>(1, (1024 + ntiles - 1) / ntiles);
    nsplit = std::min<int64_t>(nsplit, std::max<int64_t>(1, M / (4 * QZK)));
    int64_t kchunk = (M + nsplit - 1) / nsplit;
    kchunk = (kchunk + QZK - 1) / QZK * QZK;
    nsplit = (M + kchunk - 1) / kchunk;
    const int64_t slab_stride = N * lw;
    void* slab;
    TLSQ_TRY(ws_get(h, WS_SLAB, (size_t)(nsplit * slab_stride) * 8, &slab));
    const int64_t nwork = ntiles * nsplit, cpx = (nwork + 7) / 8;
#define ZTYH(NC)                                                                                                                  \
    hipLaunchKernelGGL((k_zty_h<NC>), dim3((unsigned)(8 * cpx)), dim3(64 * QZ_WAVES), 0, h->stream, Z, ldz, (const float*)t32, M,     \
                       (double*)slab, N, M, kchunk, slab_stride, (int)ntiles, (int)nsplit, zmax_bits, (const unsigned int*)tmax)
    switch (nct) {
        case 1: ZTYH(1); break;
        case 2: ZTYH(2); break;
        case 3: ZTYH(3); break;
        case 4: ZTYH(4); break;
        default: ZTYH(5); break;
    }
#undef ZTYH
    ++h->kern_zty_h;
    hipLaunchKernelGGL(k_zty_h_reduce, dim3((unsigned)std::min<int64_t>((N * p + 255) / 256, 2048)), dim3(256), 0, h->stream,
                       (const double*)slab, slab_stride, (int)nsplit, lw, Y, ldy, N, (int)p);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

}   // namespace tlsq
