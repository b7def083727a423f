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
//   k_gram_h3       lower-triangle 128 x 128 tiles x row splits -> fp64 slabs: 8 waves, wave tile 64 x 32, stages of 64 rows (two
//                   MFMA steps; a staged column is one 128-byte line), both planes of both panels through two LDS buffers filled by
//                   global_load_lds_dwordx4 (no staging registers; row groups swizzled: conflict-free 16-byte fragment reads),
//                   every fragment set requested two groups of 8 MFMAs before its first use, one barrier per stage
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
constexpr int HK = 64;                // rows per stage = two 16 x 16 x 32 steps: a column of a stage is one 128-byte line
constexpr int HP = 64;                // halfs per staged column (128 bytes: the stage's eight 16-byte row groups, swizzled - h3_sw)
constexpr int HPANEL = HT * HP;       // halfs per plane of a panel (16 KB)
constexpr int HBUF = 4 * HPANEL;      // A hi, A lo, B hi, B lo (64 KB)
// (fp64 fold-in: template parameter of the kernel - every stage or every 2 or 4 stages = 64, 128, 256 rows; GRAM_H3_FOLD, default 128)

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
// g + 1 (g even).  With a 128-byte column pitch even and odd columns own one half of the banks each; storing row group g of column c
// in slot g ^ sw(c), sw = (c >> 1) & 7, gives the eight columns of a half eight different slots in every group (the two kinds
// differ in bit 0 only, and sw separates {0, 1, 6, 7} from {2, 3, 4, 5}; checked by enumeration) - a padded pitch left two-way
// conflicts on three columns of every group.  The stores (eight lanes per column) stay conflict-free under any such swizzle.
__device__ __forceinline__ int h3_sw(int c) { return (c >> 1) & 7; }

template <int HFOLD>
// (one workgroup per CU: 128 KB of stage buffers; the (512, 2) bound of round 5 only capped the registers - same time)
__global__ __launch_bounds__(512, 1) void k_gram_h3(const _Float16* __restrict__ H, const _Float16* __restrict__ L, int64_t ld,
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
    // Staging: global -> LDS loads that bypass the registers (global_load_lds_dwordx4: a wave instruction fills 1 KB of LDS, lane l
    // at base + 16 l) - wave w brings columns 8 w .. 8 w + 7 (+ 64) of each plane, lane = (column l / 8, slot l % 8), and the row
    // group swizzle is applied on the global side: slot q of column c receives row group q ^ sw(c).  A wave's instruction covers
    // eight whole 128-byte lines; no staging registers, no ds_write_b128.
    const int dcol = w * 8 + (lane >> 3), dslot = lane & 7;
    const int64_t da0 = (i0 + dcol) * ld + kbeg + 8 * (dslot ^ h3_sw(dcol));
    const int64_t da1 = (i0 + dcol + 64) * ld + kbeg + 8 * (dslot ^ h3_sw(dcol + 64));
    const int64_t db0 = (j0 + dcol) * ld + kbeg + 8 * (dslot ^ h3_sw(dcol));
    const int64_t db1 = (j0 + dcol + 64) * ld + kbeg + 8 * (dslot ^ h3_sw(dcol + 64));
    auto dload = [&](int buf, int s) {
        const int64_t o = (int64_t)s * HK;
        _Float16* b = hsm + buf * HBUF + (w * 8) * HP;
        __builtin_amdgcn_global_load_lds(H + da0 + o, b, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(H + da1 + o, b + 64 * HP, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(L + da0 + o, b + HPANEL, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(L + da1 + o, b + HPANEL + 64 * HP, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(H + db0 + o, b + 2 * HPANEL, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(H + db1 + o, b + 2 * HPANEL + 64 * HP, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(L + db0 + o, b + 3 * HPANEL, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(L + db1 + o, b + 3 * HPANEL + 64 * HP, 16, 0, 0);
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
    // fragment of k-step ks (row groups 4 ks + kg) of the 16 columns at `col0` of a plane: column col0 + fr, slot (4 ks + kg) ^ sw
    const int ca = (wi * 64 + fr) * HP, cb = (wj * 32 + fr) * HP;
    const int swf = h3_sw(fr);                     // (tile bases are multiples of 16: sw(col) = sw(fr))
    const int g0 = 8 * (kg ^ swf), g1 = 8 * ((4 + kg) ^ swf);
    auto read_a = [&](gh8* a4, const _Float16* plane, int g) {
#pragma unroll
        for (int a = 0; a < 4; ++a) a4[a] = *reinterpret_cast<const gh8*>(plane + ca + a * 16 * HP + g);
    };
    auto read_b = [&](gh8* b2, const _Float16* plane, int g) {
#pragma unroll
        for (int b = 0; b < 2; ++b) b2[b] = *reinterpret_cast<const gh8*>(plane + cb + b * 16 * HP + g);
    };
    auto mm = [&](const gh8* a4, const gh8* b2) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a4[a], b2[b], acc[a][b], 0, 0, 0);
    };
    // Stage s (64 rows, two MFMA steps, six groups of 8 MFMAs): its planes are in LDS buffer s & 1.  The waves of a workgroup meet at
    // a barrier every stage, i.e. they run in step and request their fragments in bursts (8 waves x 6 KB): a group of 8 MFMAs is
    // 128 cycles, shorter than such a burst takes - so every fragment set is requested TWO groups before its first use (four
    // sets of 24 registers; with one group of distance the MFMA pipe was busy 35 % of the time).  Before the fifth group: this
    // wave's part of stage s + 1 has landed (vmcnt), behind the barrier everybody's has, and nobody reads buffer s & 1 any more
    // (the last fragments taken from it are in registers) - stage s + 2 is requested into it, and the first two fragment sets of
    // stage s + 1 from the other buffer.
    gh8 ah0[4], bh0[2], al0[4], bl0[2], ah1[4], bh1[2], al1[4], bl1[2];
    auto stage = [&](auto uc, int s) {
        constexpr int u = decltype(uc)::value;
        const _Float16* base = hsm + u * HBUF;
        const _Float16* nbase = hsm + (u ^ 1) * HBUF;
        read_a(ah1, base, g1);
        read_b(bh1, base + 2 * HPANEL, g1);
        mm(ah0, bh0);                              // hi x hi, step 0
        read_a(al1, base + HPANEL, g1);
        read_b(bl1, base + 3 * HPANEL, g1);
        mm(ah0, bl0);                              // hi x lo, step 0
        mm(al0, bh0);                              // lo x hi, step 0
        mm(ah1, bh1);                              // hi x hi, step 1
        __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0) (lgkmcnt / expcnt untouched)
        __syncthreads();
#if !defined(H3_ABLATE) || H3_ABLATE != 1
        dload(u, clamps(s + 2));
#endif
        read_a(ah0, nbase, g0);
        read_b(bh0, nbase + 2 * HPANEL, g0);
        read_a(al0, nbase + HPANEL, g0);
        read_b(bl0, nbase + 3 * HPANEL, g0);
        mm(ah1, bl1);                              // hi x lo, step 1
        mm(al1, bh1);                              // lo x hi, step 1
#if !defined(H3_ABLATE) || H3_ABLATE != 3
        if ((s % HFOLD) == HFOLD - 1) fold();
#endif
        __builtin_amdgcn_sched_barrier(0);
    };
    if (nst > 0) {
        dload(0, 0);
        dload(1, clamps(1));
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        read_a(ah0, hsm, g0);
        read_b(bh0, hsm + 2 * HPANEL, g0);
        read_a(al0, hsm + HPANEL, g0);
        read_b(bl0, hsm + 3 * HPANEL, g0);
        std::integral_constant<int, 0> U0;
        std::integral_constant<int, 1> U1;
        int s = 0;
        for (; s + 2 <= nst; s += 2) {
            stage(U0, s);
            stage(U1, s + 1);
        }
        if (s < nst) stage(U0, s);
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

int absmax_bits_f32(Handle* h, const float* Z, int64_t n, unsigned int* out_bits) {
    TLSQ_HIP(h, hipMemsetAsync(out_bits, 0, 4, h->stream));
    hipLaunchKernelGGL(k_absmax_bits, dim3(2048), dim3(256), 0, h->stream, Z, n, out_bits);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

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
    if (h->absmax_panel == (const void*)Z) {
        // (the sweep that wrote this panel left its max |z| at + 40: no pass of our own - 0.25 ms at 65536 x 4096)
        maxbits = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(sc) + 40);
    } else {
        TLSQ_HIP(h, hipMemsetAsync(maxbits, 0, 4, h->stream));
        hipLaunchKernelGGL(k_absmax_bits, dim3(2048), dim3(256), 0, h->stream, Z, n, maxbits);
    }

    hipLaunchKernelGGL(k_split_f16, dim3(4096), dim3(256), 0, h->stream, Z, n, (const unsigned int*)maxbits, (_Float16*)hp,
                       (_Float16*)lp, (double*)sc);
    TLSQ_HIP(h, hipGetLastError());
    const int64_t nti = N / HT, ntiles = nti * (nti + 1) / 2;
    // row splits: ~4 rounds of 256 work items (one workgroup per CU), at least 32 stages each, slabs below 2 GB
    int64_t nsplit = std::max<int64_t>(1, (1024 + ntiles - 1) / ntiles);
    nsplit = std::min<int64_t>(nsplit, std::max<int64_t>(1, K / (16 * HK)));
    nsplit = std::min<int64_t>(nsplit, std::max<int64_t>(1, (int64_t)(((size_t)2 << 30) / ((size_t)N * N * 8))));
    int64_t kchunk = (K + nsplit - 1) / nsplit;
    kchunk = (kchunk + HK * 4 - 1) / (HK * 4) * (HK * 4);
    nsplit = (K + kchunk - 1) / kchunk;
    const int64_t slab_stride = N * N;
    void* slab;
    TLSQ_TRY(ws_get(h, WS_SLAB, (size_t)(nsplit * slab_stride) * 8, &slab));
    const int32_t* tab = nullptr;
    TLSQ_TRY(gram_tile_table(h, nti, &tab));
    const int64_t nwork = ntiles * nsplit, cpx = (nwork + 7) / 8;
    const size_t lds = (size_t)2 * HBUF * 2;
    const int fold = [] { const char* e = dev_get(DEV_GRAM_H3_FOLD); const int v = e ? atoi(e) : 2; return v == 1 || v == 4 ? v : 2; }();   // stages of 64 rows
#define H3_LAUNCH(F)                                                                                                             \
    do {                                                                                                                         \
        TLSQ_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_gram_h3<F>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lds));                                                                              \
        hipLaunchKernelGGL(k_gram_h3<F>, dim3((unsigned)(8 * cpx)), dim3(512), lds, h->stream, (const _Float16*)hp,              \
                           (const _Float16*)lp, ld, (double*)slab, N, K, kchunk, slab_stride, (int)ntiles, (int)nsplit, tab);    \
    } while (0)
    if (fold == 1) H3_LAUNCH(1);
    else if (fold == 4) H3_LAUNCH(4);
    else H3_LAUNCH(2);
#undef H3_LAUNCH
    ++h->kern_gram_h3;
    hipLaunchKernelGGL(k_h3_reduce, dim3(2048), dim3(256), 0, h->stream, (const double*)slab, slab_stride, (int)nsplit,
                       (const double*)sc, G, ldg, (int)N);
    TLSQ_HIP(h, hipGetLastError());
    return TLSQ_OK;
}

}   // namespace tlsq
