// micro-benchmark (development tool): fp64 MFMA Gram G = Z'Z for a tall column-major Z (M x N, N a multiple of 128) -
// candidate workgroup shapes / software pipelines for the product kernel in csrc/gemm.hip, timed with HIP events and
// with in-kernel cycle counters.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench/build/syrk_f64 tools/ubench/syrk_f64.hip
//   tools/ubench/build/syrk_f64 [M N reps]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);  \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

// work item -> (z, ti, tj), lower triangle of 128 x 128 tiles, XCD-aware (blocks b, b+8, .. share an XCD)
__device__ __forceinline__ bool work_item(int ntiles, int nsplit, int& z, int& ti, int& tj) {
    const long nwork = (long)ntiles * nsplit;
    const long cpx = (nwork + 7) / 8;
    const long item = (long)(blockIdx.x % 8) * cpx + (long)(blockIdx.x / 8);
    if (item >= nwork) return false;
    z = (int)(item / ntiles);
    const int t = (int)(item % ntiles);
    ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    tj = t - ti * (ti + 1) / 2;
    return true;
}

// ---- variant A: NW waves, wave tile (16*NA) x (16*NB), K stage TK, classic "compute, then store the next panel, barrier"
// ---- variant B: same shapes, software-pipelined: next panel stored mid-stage, barrier before the last k-step,
//                 first fragments of the next stage fetched across the stage boundary
template <int NW, int NA, int NB, int TK, bool PIPE, bool B128, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void k_syrk(const double* __restrict__ Z, long ld, double* __restrict__ slab,
                                                  long slab_stride, int N, long K, long kchunk, int nti, int nsplit,
                                                  unsigned long long* __restrict__ cyc) {
    constexpr int LDK = TK + 2;
    constexpr int PANEL = 128 * LDK;
    constexpr int NT = NW * 64;
    constexpr int WJN = 128 / (16 * NB);           // waves along j
    constexpr int NQ = TK / 4;                     // k-steps per stage
    constexpr int SLOTS = 128 * TK / 2 / NT;       // d2 loads per thread and panel
    static_assert((128 / (16 * NA)) * WJN == NW, "wave grid");
    extern __shared__ __attribute__((aligned(16))) double smem[];   // A[2], B[2]
    int z, ti, tj;
    if (!work_item(nti * (nti + 1) / 2, nsplit, z, ti, tj)) return;
    const long kbeg = (long)z * kchunk;
    const long kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    const int nstage = (int)((kend - kbeg) / TK);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wj = w % WJN, wi = w / WJN;
    const int fr = lane & 15, fk = lane >> 4;
    const long i0 = (long)ti * 128, j0 = (long)tj * 128;

    d4 acc[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};

    // staging map: slot e = tid + NT*s -> column r = e / (TK/2), k pair = e % (TK/2)
    d2 ra[SLOTS], rb[SLOTS];
    auto gload = [&](long k0) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int e = tid + NT * s;
            const int r = e / (TK / 2), kp = e % (TK / 2);
            ra[s] = *reinterpret_cast<const d2*>(Z + k0 + 2 * kp + (i0 + r) * ld);
            rb[s] = *reinterpret_cast<const d2*>(Z + k0 + 2 * kp + (j0 + r) * ld);
        }
    };
    auto sstore = [&](int buf) {
        double* sa = smem + buf * PANEL;
        double* sb = smem + (2 + buf) * PANEL;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int e = tid + NT * s;
            const int r = e / (TK / 2), kp = e % (TK / 2);
            *reinterpret_cast<d2*>(sa + r * LDK + 2 * kp) = ra[s];
            *reinterpret_cast<d2*>(sb + r * LDK + 2 * kp) = rb[s];
        }
    };
    // fragments of k-step q of buffer `buf`.  B128: k-step pairs (2u, 2u+1) share one 16-byte read per tile - lane (fr, fk)
    // holds k = 8u + 2 fk + {0, 1}, element m feeds MFMA 2u + m (the same permutation of k on both operands)
    double fa[2][NA], fb[2][NB];
    d2 ga[2][NA], gb[2][NB];
    auto frag = [&](int buf, int q, int slot) {
        const double* sa = smem + buf * PANEL + (wi * 16 * NA + fr) * LDK;
        const double* sb = smem + (2 + buf) * PANEL + (wj * 16 * NB + fr) * LDK;
        if (B128) {
#pragma unroll
            for (int a = 0; a < NA; ++a) ga[slot][a] = *reinterpret_cast<const d2*>(sa + a * 16 * LDK + 8 * q + 2 * fk);
#pragma unroll
            for (int b = 0; b < NB; ++b) gb[slot][b] = *reinterpret_cast<const d2*>(sb + b * 16 * LDK + 8 * q + 2 * fk);
        } else {
#pragma unroll
            for (int a = 0; a < NA; ++a) fa[slot][a] = sa[a * 16 * LDK + 4 * q + fk];
#pragma unroll
            for (int b = 0; b < NB; ++b) fb[slot][b] = sb[b * 16 * LDK + 4 * q + fk];
        }
    };
    auto mma = [&](int slot) {
        if (B128) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[slot][a][m], gb[slot][b][m], acc[a][b], 0, 0, 0);
        } else {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][a], fb[slot][b], acc[a][b], 0, 0, 0);
        }
    };
    constexpr int NS = B128 ? NQ / 2 : NQ;   // fragment steps per stage

    unsigned long long t0 = 0;
    if (nstage > 0) {
        gload(kbeg);
        sstore(0);
    }
    __syncthreads();
    if (PIPE) {
        if (nstage > 1) gload(kbeg + TK);
        frag(0, 0, 0);
        t0 = __builtin_amdgcn_s_memtime();
        for (int s = 0; s < nstage; ++s) {
            const int cur = s & 1;
            const bool more = s + 1 < nstage;
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                if (q == NS - 2 && more && ABL < 2) sstore(cur ^ 1);   // (NS >= 2)
                if (q == NS - 1) {
                    if (more) {
                        if (ABL < 3) __syncthreads();
                        if (ABL < 4) frag(cur ^ 1, 0, (q + 1) & 1);
                        if (s + 2 < nstage && ABL < 1) gload(kbeg + (long)(s + 2) * TK);
                    }
                } else {
                    if (ABL < 4) frag(cur, q + 1, (q + 1) & 1);
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the fragment reads of the next step ahead of this step's MFMAs
                mma(q & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        t0 = __builtin_amdgcn_s_memtime();
        for (int s = 0; s < nstage; ++s) {
            const int cur = s & 1;
            const bool more = s + 1 < nstage;
            if (more) gload(kbeg + (long)(s + 1) * TK);
            frag(cur, 0, 0);
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                if (q + 1 < NS) frag(cur, q + 1, (q + 1) & 1);
                mma(q & 1);
            }
            if (more) sstore(cur ^ 1);
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (cyc && tid == 0 && blockIdx.x == 17) {
        cyc[0] = t1 - t0;
        cyc[1] = (unsigned long long)nstage;
    }
    double* __restrict__ Cz = slab + (long)z * slab_stride;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const long j = j0 + wj * 16 * NB + b * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long i = i0 + wi * 16 * NA + a * 16 + fk + 4 * r;
                Cz[j + i * (long)N] = acc[a][b][r];
            }
        }
}


// ---- variant C: 8 waves, wave tile 64 x 32, 16-byte fragment reads, software-pipelined as B, and every LDS / global
// ---- access of a step is pinned between two MFMAs of that step (the wave issues in order: a burst of eight ds_write or
// ---- global_load instructions ahead of the MFMAs leaves the matrix pipe idle on both waves of a SIMD at once)
template <int TK, int ILV>
__global__ __launch_bounds__(512) void k_syrk_c(const double* __restrict__ Z, long ld, double* __restrict__ slab,
                                                long slab_stride, int N, long K, long kchunk, int nti, int nsplit,
                                                unsigned long long* __restrict__ cyc) {
    constexpr int LDK = TK + 2;
    constexpr int PANEL = 128 * LDK;
    constexpr int SL = TK / 8;    // 16-byte slots per thread and panel
    constexpr int NS = TK / 8;    // fragment steps (two k-steps = 16 MFMAs each) per stage
    extern __shared__ __attribute__((aligned(16))) double smem[];   // A[2], B[2]
    int z, ti, tj;
    if (!work_item(nti * (nti + 1) / 2, nsplit, z, ti, tj)) return;
    const long kbeg = (long)z * kchunk;
    const long kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    const int nstage = (int)((kend - kbeg) / TK);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wj = w & 3, wi = w >> 2;
    const int fr = lane & 15, fk = lane >> 4;
    const long i0 = (long)ti * 128, j0 = (long)tj * 128;

    d4 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
    const double* pa[SL];
    const double* pb[SL];
    int so[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        const int e = tid + 512 * s;
        const int r = e / (TK / 2), kp = e % (TK / 2);
        pa[s] = Z + kbeg + 2 * kp + (i0 + r) * ld;
        pb[s] = Z + kbeg + 2 * kp + (j0 + r) * ld;
        so[s] = r * LDK + 2 * kp;
    }
    const int oa = (wi * 64 + fr) * LDK + 2 * fk, ob = (wj * 32 + fr) * LDK + 2 * fk;
    d2 ra[SL], rb[SL], ga[2][4], gb[2][2];

#define C_FR(buf, q, slot, i)                                                                                   \
    do {                                                                                                        \
        if ((i) < 4) ga[slot][(i) & 3] = *reinterpret_cast<const d2*>(smem + (buf) * PANEL + oa + ((i) & 3) * 16 * LDK + 8 * (q)); \
        else gb[slot][(i) & 1] = *reinterpret_cast<const d2*>(smem + (2 + (buf)) * PANEL + ob + ((i) & 1) * 16 * LDK + 8 * (q)); \
    } while (0)
#define C_SW(buf, i)                                                                                    \
    do {                                                                                                \
        if ((i) < SL) *reinterpret_cast<d2*>(smem + (buf) * PANEL + so[(i) % SL]) = ra[(i) % SL];        \
        else *reinterpret_cast<d2*>(smem + (2 + (buf)) * PANEL + so[(i) % SL]) = rb[(i) % SL];           \
    } while (0)
#define C_GL(i, koff)                                                                  \
    do {                                                                               \
        if ((i) < SL) ra[(i) % SL] = *reinterpret_cast<const d2*>(pa[(i) % SL] + (koff)); \
        else rb[(i) % SL] = *reinterpret_cast<const d2*>(pb[(i) % SL] + (koff));          \
    } while (0)
#define C_MF(slot, t)                                                                                              \
    acc[((t) >> 1) & 3][(t) & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[slot][((t) >> 1) & 3][(t) >> 3],        \
                                                                        gb[slot][(t) & 1][(t) >> 3], acc[((t) >> 1) & 3][(t) & 1], 0, 0, 0)
    // KIND 0: fragments of step q+1 (same buffer); 1: + panel store into the other buffer; 2: fragments of step 0 of the
    // other buffer + global loads of the stage after the next; 3: MFMAs only
    auto step = [&](auto kind, int cur, int q, long koff) {
        constexpr int KIND = decltype(kind)::value;
        const int slot = q & 1, nslot = slot ^ 1;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            C_MF(slot, t);
            if (ILV) __builtin_amdgcn_sched_barrier(0);
            const int u = ILV == 2 ? t : t;   // item position
            if (KIND != 3 && u < 6) {
                if (KIND == 2) C_FR(cur ^ 1, 0, nslot, u);
                else C_FR(cur, q + 1, nslot, u);
            } else if (KIND == 1 && u - 6 < 2 * SL) {
                C_SW(cur ^ 1, u - 6);
            } else if (KIND == 2 && u - 6 < 2 * SL) {
                C_GL(u - 6, koff);
            }
            if (ILV) __builtin_amdgcn_sched_barrier(0);
        }
    };
    std::integral_constant<int, 0> K0;
    std::integral_constant<int, 1> K1;
    std::integral_constant<int, 2> K2;
    std::integral_constant<int, 3> K3;

    if (nstage > 0) {
#pragma unroll
        for (int i = 0; i < 2 * SL; ++i) C_GL(i, 0L);
#pragma unroll
        for (int i = 0; i < 2 * SL; ++i) C_SW(0, i);
    }
    __syncthreads();
    if (nstage > 1) {
#pragma unroll
        for (int i = 0; i < 2 * SL; ++i) C_GL(i, (long)TK);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) C_FR(0, 0, 0, i);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s + 1 < nstage; ++s) {
        const int cur = s & 1;
        const long koff = (long)(s + 2 < nstage ? s + 2 : nstage - 1) * TK;
#pragma unroll
        for (int q = 0; q + 2 < NS; ++q) step(K0, cur, q, 0L);
        step(K1, cur, NS - 2, 0L);
        __syncthreads();
        step(K2, cur, NS - 1, koff);
    }
    if (nstage > 0) {
        const int cur = (nstage - 1) & 1;
#pragma unroll
        for (int q = 0; q + 1 < NS; ++q) step(K0, cur, q, 0L);
        step(K3, cur, NS - 1, 0L);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (cyc && tid == 0 && blockIdx.x == 17) {
        cyc[0] = t1 - t0;
        cyc[1] = (unsigned long long)nstage;
    }
    double* __restrict__ Cz = slab + (long)z * slab_stride;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const long j = j0 + wj * 32 + b * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long i = i0 + wi * 64 + a * 16 + fk + 4 * r;
                Cz[j + i * (long)N] = acc[a][b][r];
            }
        }
}

__global__ __launch_bounds__(256) void k_reduce(const double* __restrict__ slab, long slab_stride, int nsplit,
                                                double* __restrict__ C, int N) {
    const long total = (long)N * N;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long j = e % N, i = e / N;
        if (j / 128 > i / 128) continue;
        double s = 0.0;
        for (int zz = 0; zz < nsplit; ++zz) s += slab[(long)zz * slab_stride + j + i * N];
        C[j + i * N] = s;
        C[i + j * N] = s;
    }
}

struct Ctx {
    double *Z, *slab, *G, *Gref;
    unsigned long long* cyc;
    long M;
    int N, reps;
};

typedef void (*kern_t)(const double*, long, double*, long, int, long, long, int, int, unsigned long long*);
void run_k(const Ctx& c, const char* name, int wgs_target, kern_t kern, int NW, int NA, int NB, int TK);
template <int NW, int NA, int NB, int TK, bool PIPE, bool B128, int ABL = 0>
void run(const Ctx& c, const char* name, int wgs_target) {
    run_k(c, name, wgs_target, k_syrk<NW, NA, NB, TK, PIPE, B128, ABL>, NW, NA, NB, TK);
}
void run_k(const Ctx& c, const char* name, int wgs_target, kern_t kern, int NW, int NA, int NB, int TK) {
    const int nti = c.N / 128, ntiles = nti * (nti + 1) / 2;
    int nsplit = wgs_target / ntiles;
    if (nsplit < 1) nsplit = 1;
    long kchunk = (c.M + nsplit - 1) / nsplit;
    kchunk = (kchunk + TK - 1) / TK * TK;
    nsplit = (int)((c.M + kchunk - 1) / kchunk);
    if (c.M % TK) {
        printf("%s: M must be a multiple of %d here\n", name, TK);
        return;
    }
    const long nwork = (long)ntiles * nsplit, cpx = (nwork + 7) / 8;
    const size_t lds = (size_t)4 * 128 * (TK + 2) * 8;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&e2));
    auto launch = [&] {
        hipLaunchKernelGGL(kern, dim3((unsigned)(8 * cpx)), dim3(NW * 64), lds, 0, c.Z, c.M, c.slab, (long)c.N * c.N, c.N, c.M,
                           kchunk, nti, nsplit, c.cyc);
    };
    auto reduce = [&] {
        hipLaunchKernelGGL(k_reduce, dim3(1024), dim3(256), 0, 0, c.slab, (long)c.N * c.N, nsplit, c.G, c.N);
    };
    for (int i = 0; i < 300; ++i) {   // clocks up
        launch();
        reduce();
    }
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms_k = 0, ms_all = 0;
    CK(hipEventRecord(e0));
    for (int i = 0; i < c.reps; ++i) launch();
    CK(hipEventRecord(e1));
    for (int i = 0; i < c.reps; ++i) {
        launch();
        reduce();
    }
    CK(hipEventRecord(e2));
    CK(hipEventSynchronize(e2));
    CK(hipEventElapsedTime(&ms_k, e0, e1));
    CK(hipEventElapsedTime(&ms_all, e1, e2));
    unsigned long long cy[2];
    CK(hipMemcpy(cy, c.cyc, 16, hipMemcpyDeviceToHost));
    // check against the reference Gram
    std::vector<double> g((size_t)c.N * c.N), gr((size_t)c.N * c.N);
    CK(hipMemcpy(g.data(), c.G, g.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gr.data(), c.Gref, g.size() * 8, hipMemcpyDeviceToHost));
    double err = 0, nrm = 0;
    for (size_t i = 0; i < g.size(); ++i) {
        err = fmax(err, fabs(g[i] - gr[i]));
        nrm = fmax(nrm, fabs(gr[i]));
    }
    const double us_k = ms_k * 1e3 / c.reps, us_all = ms_all * 1e3 / c.reps;
    const double flop = 2.0 * c.M * 128.0 * 128.0 * ntiles;
    printf("%-44s wgs=%4ld nsplit=%3d lds=%6zu: kernel %7.1f us (%5.1f TF)  +reduce %7.1f us | loop %.0f cyc/stage (%.0f%% of MFMA-bound) err %.1e\n",
           name, nwork, nsplit, lds, us_k, flop / us_k / 1e6, us_all, (double)cy[0] / (double)cy[1],
           100.0 * (NA * NB * (TK / 4) * 64.0 * (NW / 4)) / ((double)cy[0] / (double)cy[1]), err / nrm);
}

__global__ void k_fill(double* Z, long n) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        unsigned long long x = (unsigned long long)e * 0x9E3779B97F4A7C15ull + 12345;
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 32;
        Z[e] = (double)(x & 0xFFFFF) / 524288.0 - 1.0;
    }
}

int main(int argc, char** argv) {
    Ctx c;
    c.M = argc > 1 ? atol(argv[1]) : 20000;
    c.N = argc > 2 ? atoi(argv[2]) : 512;
    c.reps = argc > 3 ? atoi(argv[3]) : 50;
    c.M = c.M / 32 * 32;
    CK(hipMalloc(&c.Z, (size_t)c.M * c.N * 8));
    CK(hipMalloc(&c.slab, (size_t)1 << 30));
    CK(hipMalloc(&c.G, (size_t)c.N * c.N * 8));
    CK(hipMalloc(&c.Gref, (size_t)c.N * c.N * 8));
    CK(hipMalloc(&c.cyc, 64));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, c.Z, c.M * c.N);
    CK(hipDeviceSynchronize());
    printf("Gram %ld x %d fp64\n", c.M, c.N);
    // reference = the round-2 product shape (4 waves, 64x64, TK 16, classic), whatever it computes first
    {
        Ctx r = c;
        r.G = c.Gref;
        r.reps = 2;
        run<4, 4, 4, 16, false, false>(r, "(reference pass)", 256);
    }
    run<8, 4, 2, 32, true, true, 1>(c, "B  8w 64x32 tk32 b128 ABL1 no gload", 256);
    run<8, 4, 2, 32, true, true, 2>(c, "B  8w 64x32 tk32 b128 ABL2 no gload/sstore", 256);
    run<8, 4, 2, 32, true, true, 3>(c, "B  8w 64x32 tk32 b128 ABL3 no barrier either", 256);
    run<8, 4, 2, 32, true, true, 4>(c, "B  8w 64x32 tk32 b128 ABL4 MFMA only", 256);
    run<4, 4, 4, 32, true, true, 4>(c, "B  4w 64x64 tk32 b128 ABL4 MFMA only", 256);
    run<4, 4, 4, 32, true, true, 3>(c, "B  4w 64x64 tk32 b128 ABL3 frag+MFMA", 256);
    run_k(c, "C  8w 64x32 tk32 b128 interleaved", 256, k_syrk_c<32, 1>, 8, 4, 2, 32);
    run_k(c, "C  8w 64x32 tk32 b128 (compiler order)", 256, k_syrk_c<32, 0>, 8, 4, 2, 32);
    run_k(c, "C  8w 64x32 tk16 b128 interleaved", 256, k_syrk_c<16, 1>, 8, 4, 2, 16);
    for (int wg : {256}) {
        run<4, 4, 4, 16, false, false>(c, "A  4w 64x64 tk16 classic", wg);
        run<4, 4, 4, 16, true, false>(c, "B  4w 64x64 tk16 pipelined", wg);
        run<4, 4, 4, 32, true, false>(c, "B  4w 64x64 tk32 pipelined", wg);
        run<8, 4, 2, 16, false, false>(c, "A  8w 64x32 tk16 classic", wg);
        run<8, 4, 2, 16, true, false>(c, "B  8w 64x32 tk16 pipelined", wg);
        run<8, 4, 2, 32, true, false>(c, "B  8w 64x32 tk32 pipelined", wg);
        run<8, 4, 2, 32, false, false>(c, "A  8w 64x32 tk32 classic", wg);
        run<8, 4, 2, 16, true, true>(c, "B  8w 64x32 tk16 pipelined b128", wg);
        run<8, 4, 2, 32, true, true>(c, "B  8w 64x32 tk32 pipelined b128", wg);
        run<4, 4, 4, 32, true, true>(c, "B  4w 64x64 tk32 pipelined b128", wg);
    }
    return 0;
}
