// micro-benchmark (development tool): do fp64 vector instructions execute beside fp64 MFMAs on one SIMD?
// One workgroup per CU; waves [0, nm) per SIMD run a dependent-free MFMA stream, the others a VALU stream of the given kind.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/build/coexec_f64 tools/ubench/coexec_f64.hip && tools/ubench/build/coexec_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// mode of the non-MFMA waves: 0 idle (exit), 1 v_fma_f64 x8 independent chains, 2 v_fma_f32 x8 chains, 3 f64 single dependent chain,
// 4 f64 add/mul/cmp mix
template <int KIND>
__global__ __launch_bounds__(768) void k(int nmfma_waves, int iters_m, int iters_v, double* out) {
    const int wave = threadIdx.x >> 6;
    if (wave < nmfma_waves) {
        d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
        for (int i = 0; i < iters_m; ++i) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
        out[blockIdx.x * 1024 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else {
        if (KIND == 0) return;
        if (KIND == 1 || KIND == 3 || KIND == 4) {
            double x[8];
            for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 1e-3 + j;
            const double m = 1.0000001, c = 1e-9;
            for (int i = 0; i < iters_v; ++i) {
                if (KIND == 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = __builtin_fma(x[j], m, c);
                } else if (KIND == 3) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[0] = __builtin_fma(x[0], m, c);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { x[j] = x[j] * m; x[j] = x[j] > 5.0 ? x[j] - 1.0 : x[j] + c; }
                }
            }
            double s = 0;
            for (int j = 0; j < 8; ++j) s += x[j];
            out[blockIdx.x * 1024 + threadIdx.x] = s;
        } else {
            float x[8];
            for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 1e-3f + j;
            const float m = 1.0000001f, c = 1e-9f;
            for (int i = 0; i < iters_v; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], m, c);
            }
            float s = 0;
            for (int j = 0; j < 8; ++j) s += x[j];
            out[blockIdx.x * 1024 + threadIdx.x] = s;
        }
    }
}

template <int KIND>
float run(int threads, int nm, int im, int iv, double* out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, nm, im, iv, out);
    CK(hipEventRecord(e0));
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, nm, im, iv, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5 * 1000;
}

int main() {
    double* out; CK(hipMalloc(&out, 256 * 1024 * 8));
    const int im = 4000;   // 16000 MFMAs per wave
    // MFMA alone: 4 waves (1/SIMD), 8 waves (2/SIMD)
    printf("MFMA alone  4 waves x %d MFMA: %.1f us   (64 cyc each at 2.4 GHz: %.1f us)\n", 4 * im, run<0>(256, 4, im, 0, out), 4 * im * 64 / 2400.0);
    printf("MFMA alone  8 waves (2/SIMD, half each): %.1f us\n", run<0>(512, 8, im / 2, 0, out));
    // VALU alone: 4 waves, 8 independent chains; iv chosen so that it takes about as long as the MFMA part
    const int iv = 12000;
    printf("v_fma_f64 x8 indep alone (4 waves, %d FMA each): %.1f us\n", 8 * iv, run<1>(256, 0, 0, iv, out));
    printf("v_fma_f32 x8 indep alone: %.1f us\n", run<2>(256, 0, 0, iv, out));
    printf("v_fma_f64 dependent chain alone: %.1f us\n", run<3>(256, 0, 0, iv / 4, out));
    printf("f64 mul/cmp/add mix alone: %.1f us\n", run<4>(256, 0, 0, iv / 3, out));
    // together: 8 MFMA waves + 4 VALU waves
    printf("8 MFMA waves + 4 waves v_fma_f64 indep: %.1f us\n", run<1>(768, 8, im / 2, iv, out));
    printf("8 MFMA waves + 4 waves v_fma_f32 indep: %.1f us\n", run<2>(768, 8, im / 2, iv, out));
    printf("8 MFMA waves + 4 waves f64 dependent chain: %.1f us\n", run<3>(768, 8, im / 2, iv / 4, out));
    printf("8 MFMA waves + 4 waves f64 mix: %.1f us\n", run<4>(768, 8, im / 2, iv / 3, out));
    printf("4 MFMA waves + 4 waves v_fma_f64 indep (512 thr): %.1f us\n", run<1>(512, 4, im, iv, out));
    return 0;
}
