// micro-benchmark: issue cost of the fp64 MFMA forms and of v_fma_f64 on gfx950 (development tool)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, unsigned long long* cyc) {
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double f[16];
    for (int i = 0; i < 16; ++i) f[i] = i + a;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i][0], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) f[i] = __builtin_fma(f[i], b, a);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int waves = 1; waves <= 8; waves *= 2) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int grid : {1, 256}) {
                if (waves * 64 > 256 && true) { /* more than 4 waves per block not allowed by launch bounds */ }
                int threads = waves * 64 > 256 ? 256 : waves * 64;
                int g = grid * (waves > 4 ? 2 : 1);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(g), dim3(threads), 0, 0, out, iters, cyc);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(g), dim3(threads), 0, 0, out, iters, cyc);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(g), dim3(threads), 0, 0, out, iters, cyc);
                hipDeviceSynchronize();
                unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
                const int per = mode == 2 ? 16 : 8;
                const char* nm = mode == 0 ? "mfma_f64_16x16x4" : mode == 1 ? "mfma_f64_4x4x4  " : "v_fma_f64       ";
                printf("%s waves/block=%d blocks=%d: %.1f cycles per instruction per wave\n", nm, threads / 64, g,
                       (double)c / (iters * per));
            }
        }
    }
    return 0;
}
