// G = Z'Z for fp32 panels of more than 2048 columns on the fp16 MFMA at fp32 accuracy (round 5) - the gesdd work of
// src/robustPCA.jl:194 and the opnorm of :177 / :225 on the small side, for BASELINE config 5 (65536 x 4096 fp32: 1.1 TFLOP
// per Gram matrix, 10.2 ms on the fp32 MFMA = 0.70 of its 157 TFLOP/s, three quarters of an exact-mode iteration).
//
// Every entry z of the panel is split into two fp16 numbers after an exact power-of-two scaling c (max |z| c in [2^13, 2^14)):
//     z c = h + l + d,   h = fp16(z c),  l = fp16(z c - h),  |d| <= 2^-23 |z c|        (both subtractions are exact in fp32)
// - 22 significant bits in the pair, against 24 in the fp32 number - and
//     sum_k z_ki z_kj  ~  c^-2 sum_k (h_ki h_kj + h_ki l_kj + l_ki h_kj)
// with the three products on v_mfma_f32_16x16x32_f16 (2.5 PFLOP/s dense: three of them cost 0.19 of the fp32 MFMA's time for
// the same tile).  The l l term that is left out is 2^-22 of the product with a random sign; the fp32 partial sums of the MFMA are
// folded into fp64 accumulators every 64 rows exactly as in k_gram_f32mfma, so the result carries the same ~2e-8 sigma_max^2.
//
//   k_absmax_bits   max |z| of the panel as a bit pattern (one atomicMax per workgroup)
//   k_split_f16     the two fp16 planes H, L (same column-major layout as Z) and the scale
//   k_gram_h3       lower-triangle 128 x 128 tiles x row splits -> fp64 slabs: 8 waves, wave tile 64 x 32, stages of 32 rows (one
//                   MFMA step), both planes of both panels through LDS (64-byte columns, row groups swizzled: conflict-free reads),
//                   global loads four stages ahead in registers (a stage is 24 MFMAs = 0.2 us: the HBM latency spans several)
//   k_h3_reduce     G = c^-2 sum of the slabs, one writer per entry pair (exactly symmetric)
#include <hip/hip_fp16.h>

#include "common.hpp"
#include "internal.hpp"

namespace tlsq {

namespace {
typedef _Float16 gh8 __attribute__((ext_vector_type(8)));
typedef _Float16 gh4 __attribute__((ext_vector_type(4)));
typedef float gf4 __attribute__((ext_vector_type(4)));
typedef double gd4 __attribute__((ext_vector_type(4)));
typedef unsigned int gu4 __attribute__((ext_vector_type(4)));

constexpr int HT = 128;               // tile of G
constexpr int HK = 32;                // rows per stage = one 16 x 16 x 32 step
constexpr int HP = 32;                // halfs per staged column (64 bytes: the stage's four 16-byte row groups, swizzled - h3_sw)
constexpr int HPANEL = HT * HP;       // halfs per plane of a panel (10 KB)
constexpr int HBUF = 4 * HPANEL;      // A hi, A lo, B hi, B lo (40 KB)
constexpr int HRING = 4;              // stages of global loads in flight (registers)
// (stages per fp64 fold-in: template parameter of the kernel - 2, 4 or 8 stages = 64, 128, 256 rows; GRAM_H3_FOLD, default 4)

__global__ __launch_bounds__(256) void k_absmax_bits(const float* __restrict__ Z, int64_t n, unsigned int* __restrict__ out) {
    const int64_t nv = n / 4;
    unsigned int m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        const gu4 v = reinterpret_cast<const gu4*>(Z)[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned int a = v[c] & 0x7FFFFFFFu;
            m = a > m ? a : m;
        }
    }
    for (int64_t i = nv * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned int a = __float_as_uint(Z[i]) & 0x7FFFFFFFu;
        m = a > m ? a : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned int o = (unsigned int)__shfl_xor((int)m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// scale[0] = c (a power of two), scale[1] = c^-2 as a double; NaN / Inf in the panel: c = 1 (they propagate through the planes)
__global__ __launch_bounds__(256) void k_split_f16(const float* __restrict__ Z, int64_t n, const unsigned int* __restrict__ maxbits,
                                                   _Float16* __restrict__ H, _Float16* __restrict__ L, double* __restrict__ scale) {
    const unsigned int mb = *maxbits;
    int e = (int)(mb >> 23) - 127;            // max = 1.f x 2^e  (denormal max: e = -127, the scale saturates below)
    if (mb == 0 || mb >= 0x7F800000u) e = 13;
    int se = 13 - e;                          // c = 2^se: max c in [2^13, 2^14)
    se = se > 120 ? 120 : (se < -100 ? -100 : se);
    const float c = __uint_as_float((unsigned int)(se + 127) << 23);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scale[0] = (double)c;
        scale[1] = 1.0 / ((double)c * (double)c);
    }
    const int64_t nv = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        const gf4 v = __builtin_nontemporal_load(reinterpret_cast<const gf4*>(Z) + i);
        gh4 hh, ll;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x = v[k] * c;
            const _Float16 hv = (_Float16)x;
            hh[k] = hv;
            ll[k] = (_Float16)(x - (float)hv);
        }
        reinterpret_cast<gh4*>(H)[i] = hh;
        reinterpret_cast<gh4*>(L)[i] = ll;
    }
}

// ds_read_b128 serves a wave in four groups of 16 lanes - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 - with
// the bank = (byte address / 4) mod 64: a group holds all 16 columns of a fragment, eight of them with row group g and eight with
// g + 1.  With an unpadded 64-byte column pitch the four columns c, c + 4, c + 8, c + 12 share a 16-bank quarter; storing row group
// g of column c in slot g ^ sw(c), sw = (0, 3, 2, 1)[(c >> 2) & 3], gives them four different slots in every group (checked by
// enumeration for all four groups) - a plain 80-byte pitch left three two-way conflicts per group.
__device__ __forceinline__ int h3_sw(int c) { return (-(c >> 2)) & 3; }

template <int HFOLD>
__global__ __launch_bounds__(512, 2) void k_gram_h3(const _Float16* __restrict__ H, const _Float16* __restrict__ L, int64_t ld,
                                                    double* __restrict__ slab, int64_t N, int64_t K, int64_t kchunk,
                                                    int64_t slab_stride, int ntiles, int nsplit, const int32_t* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) _Float16 hsm[];   // 2 x HBUF
    const int64_t nwork = (int64_t)ntiles * nsplit;
    const int64_t cpx = (nwork + 7) / 8;
    const int64_t item = (int64_t)(blockIdx.x % 8) * cpx + (int64_t)(blockIdx.x / 8);   // XCD x takes a contiguous run of items
    if (item >= nwork) return;
    const int z = (int)(item / ntiles);
    const int t = (int)(item % ntiles);
    const int ti = order[2 * t], tj = order[2 * t + 1];   // blocked order (gram_tile_table): 8 x 4 tiles share 12 panels in L2
    const int64_t kbeg = (int64_t)z * kchunk;
    const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    const int nst = (int)((kend - kbeg) / HK);
    const int64_t i0 = (int64_t)ti * HT, j0 = (int64_t)tj * HT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wj = w & 3, wi = w >> 2;
    const int fr = lane & 15, kg = lane >> 4;
    // staging: thread -> (column tid / 4, rows 8 (tid % 4) .. + 7) of each of the four planes
    const int scol = tid >> 2, spart = tid & 3;
    const int64_t ga = (i0 + scol) * ld + kbeg + 8 * spart;
    const int64_t gb = (j0 + scol) * ld + kbeg + 8 * spart;
    const int so = scol * HP + 8 * (spart ^ h3_sw(scol));
    gu4 ring[HRING][4];
    auto gload = [&](gu4* r, int s) {
        const int64_t o = (int64_t)s * HK;
        r[0] = *reinterpret_cast<const gu4*>(H + ga + o);
        r[1] = *reinterpret_cast<const gu4*>(L + ga + o);
        r[2] = *reinterpret_cast<const gu4*>(H + gb + o);
        r[3] = *reinterpret_cast<const gu4*>(L + gb + o);
    };
    auto sstore = [&](const gu4* r, int buf) {
        _Float16* b = hsm + buf * HBUF + so;
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<gu4*>(b + p * HPANEL) = r[p];
    };
    gf4 acc[4][2];
    gd4 acc64[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            acc[a][b] = gf4{0.f, 0.f, 0.f, 0.f};
            acc64[a][b] = gd4{0.0, 0.0, 0.0, 0.0};
        }
    auto fold = [&]() {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc64[a][b][q] += (double)acc[a][b][q];
                acc[a][b] = gf4{0.f, 0.f, 0.f, 0.f};
            }
    };
    const int last = nst - 1;
    auto clamps = [&](int s) { return s < last ? s : last; };
    const int oa = (wi * 64 + fr) * HP + 8 * (kg ^ h3_sw(fr)), ob = (wj * 32 + fr) * HP + 8 * (kg ^ h3_sw(fr));   // (tile bases are multiples of 16)
    // Stage s: its planes are in LDS buffer s & 1 (stored during stage s - 1).  The waves of a workgroup meet at a barrier every
    // stage, i.e. they run in step: fragment reads must overlap the MFMAs inside each wave, not across waves.  First third: the
    // hi x hi products on fragments fetched during the previous stage, while the lo fragments of this stage are read and the
    // registers of ring slot (s + 1) % 4 - stage s + 1, requested during stage s - 3 - go to the other buffer; barrier; the other
    // two thirds (hi x lo, lo x hi) while the hi fragments of stage s + 1 are read from that buffer.  Ring slot s % 4 then
    // requests stage s + 4.
    gh8 ah[2][4], bh[2][2];
    auto read_hi = [&](gh8* a4, gh8* b2, int buf) {
        const _Float16* base = hsm + buf * HBUF;
#pragma unroll
        for (int a = 0; a < 4; ++a) a4[a] = *reinterpret_cast<const gh8*>(base + oa + a * 16 * HP);
#pragma unroll
        for (int b = 0; b < 2; ++b) b2[b] = *reinterpret_cast<const gh8*>(base + 2 * HPANEL + ob + b * 16 * HP);
    };
    auto stage = [&](auto uc, int s) {
        constexpr int u = decltype(uc)::value;
        constexpr int c = u & 1;
        const _Float16* base = hsm + (u & 1) * HBUF;
        gh8 al[4], bl[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) bl[b] = *reinterpret_cast<const gh8*>(base + 3 * HPANEL + ob + b * 16 * HP);
#pragma unroll
        for (int a = 0; a < 4; ++a) al[a] = *reinterpret_cast<const gh8*>(base + HPANEL + oa + a * 16 * HP);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c][a], bh[c][b], acc[a][b], 0, 0, 0);
        sstore(ring[(u + 1) % HRING], (u + 1) & 1);
        __syncthreads();
        read_hi(ah[c ^ 1], bh[c ^ 1], (u + 1) & 1);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c][a], bl[b], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[a], bh[c][b], acc[a][b], 0, 0, 0);
#if !defined(H3_ABLATE) || H3_ABLATE != 3
        if ((s % HFOLD) == HFOLD - 1) fold();
#endif
#if !defined(H3_ABLATE) || H3_ABLATE != 1
        gload(ring[u], clamps(s + HRING));
#endif
        __builtin_amdgcn_sched_barrier(0);
    };
    if (nst > 0) {
        gload(ring[0], 0);
        gload(ring[1], clamps(1));
        gload(ring[2], clamps(2));
        gload(ring[3], clamps(3));
        sstore(ring[0], 0);
        __syncthreads();
        read_hi(ah[0], bh[0], 0);
        std::integral_constant<int, 0> U0;
        std::integral_constant<int, 1> U1;
        std::integral_constant<int, 2> U2;
        std::integral_constant<int, 3> U3;
        int s = 0;
        for (; s + HRING <= nst; s += HRING) {
            stage(U0, s);
            stage(U1, s + 1);
            stage(U2, s + 2);
            stage(U3, s + 3);
        }
        if (s < nst) stage(U0, s);
        if (s + 1 < nst) stage(U1, s + 1);
        if (s + 2 < nst) stage(U2, s + 2);
        fold();
    }
    // register q of lane (fr, kg): D[i = 4 kg + q][j = fr] - row i of the A tile (column of Z i0 + ...), column j of the B tile
    double* __restrict__ Cz = slab + (int64_t)z * slab_stride;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int64_t j = j0 + wj * 32 + b * 16 + fr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t i = i0 + wi * 64 + a * 16 + 4 * kg + q;
                Cz[j + i * N] = acc64[a][b][q];
            }
        }
}

// G[j + i ldg] = G[i + j ldg] = scale[1] * sum_z slab[z][j + i N]  for j <= i: one writer per entry pair, fixed order
__global__ __launch_bounds__(256) void k_h3_reduce(const double* __restrict__ slab, int64_t slab_stride, int nsplit,
                                                   const double* __restrict__ scale, double* __restrict__ G, int64_t ldg, int N) {
    const double sc = scale[1];
    const int64_t total = (int64_t)N * N, stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        const int j = (int)(e % N), i = (int)(e / N);
        if (j > i) continue;
        double s = 0.0;
        for (int zz = 0; zz < nsplit; ++zz) s += slab[(int64_t)zz * slab_stride + j + (int64_t)i * N];
        s *= sc;
        G[j + (int64_t)i * ldg] = s;
        G[i + (int64_t)j * ldg] = s;
    }
}
}   // namespace

bool gram_h3_ok(const float* Z, int64_t ld, int64_t N, int64_t K) {
    return N >= 1024 && (N % HT) == 0 && (K % HK) == 0 && K >= 4096 && ld == K && (reinterpret_cast<uintptr_t>(Z) % 16) == 0 &&
           N <= 16384;
}

int gram_h3(Handle* h, const float* Z, int64_t ld, double* G, int64_t ldg, int64_t N, int64_t K) {
    const int64_t n = N * K;
    void *hp, *lp, *sc;
    TLSQ_TRY(ws_get(h, WS_H16, (size_t)n * 2, &hp));
    TLSQ_TRY(ws_get(h, WS_L16, (size_t)n * 2, &lp));
    TLSQ_TRY(ws_get(h, WS_H16S, 64, &sc));
    unsigned int* maxbits = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(sc) + 32);
    TLSQ_HIP(h, hipMemsetAsync(maxbits, 0, 4, h->stream));
    hipLaunchKernelGGL(k_absmax_bits, dim3(2048), dim3(256), 0, h->stream, Z, n, maxbits);
    hipLaunchKernelGGL(k_split_f16, dim3(4096), dim3(256), 0, h->stream, Z, n, (const unsigned int*)maxbits, (_Float16*)hp,
                       (_Float16*)lp, (double*)sc);
    TLSQ_HIP(h, hipGetLastError());
    const int64_t nti = N / HT, ntiles = nti * (nti + 1) / 2;
    // row splits: ~2 rounds of 512 work items (two workgroups per CU), at least 32 stages each, slabs below 2 GB
    int64_t nsplit = std::max<int64_t>(1, (1024 + ntiles - 1) / ntiles);
    nsplit = std::min<int64_t>(nsplit, std::max<int64_t>(1, K / (32 * HK)));
    nsplit = std::min<int64_t>(nsplit, std::max<int64_t>(1, (int64_t)(((size_t)2 << 30) / ((size_t)N * N * 8))));
    int64_t kchunk = (K + nsplit - 1) / nsplit;
    kchunk = (kchunk + HK * 8 - 1) / (HK * 8) * (HK * 8);
    nsplit = (K + kchunk - 1) / kchunk;
    const int64_t slab_stride = N * N;
    void* slab;
    TLSQ_TRY(ws_get(h, WS_SLAB, (size_t)(nsplit * slab_stride) * 8, &slab));
    const int32_t* tab = nullptr;
    TLSQ_TRY(gram_tile_table(h, nti, &tab));
    const int64_t nwork = ntiles * nsplit, cpx = (nwork + 7) / 8;
    const size_t lds = (size_t)2 * HBUF * 2;
    const int fold = [] { const char* e = dev_get(DEV_GRAM_H3_FOLD); const int v = e ? atoi(e) : 4; return v == 2 || v == 8 ? v : 4; }();
#define H3_LAUNCH(F)                                                                                                             \
    do {                                                                                                                         \
        TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_gram_h3<F>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lds));                                                                              \
        hipLaunchKernelGGL(k_gram_h3<F>, dim3((unsigned)(8 * cpx)), dim3(512), lds, h->stream, (const _Float16*)hp,              \
                           (const _Float16*)lp, ld, (double*)slab, N, K, kchunk, slab_stride, (int)ntiles, (int)nsplit, tab);    \
    } while (0)
    if (fold == 2) H3_LAUNCH(2);
    else if (fold == 8) H3_LAUNCH(8);
    else H3_LAUNCH(4);
#undef H3_LAUNCH
    hipLaunchKernelGGL(k_h3_reduce, dim3(2048), dim3(256), 0, h->stream, (const double*)slab, slab_stride, (int)nsplit,
                       (const double*)sc, G, ldg, (int)N);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

}   // namespace tlsq
