// check of the reciprocal-square-root sequence of jacobi.hip (v_rsq_f64 + two Newton steps) against 1 / sqrt
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__device__ __forceinline__ double jr_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * __builtin_fma(-hx * y, y, 1.5);
    y = y * __builtin_fma(-hx * y, y, 1.5);
    return y;
}
__global__ void k(const double* x, double* y, double* y0, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { y[i] = jr_rsqrt(x[i]); y0[i] = __builtin_amdgcn_rsq(x[i]); }
}
int main() {
    const int n = 12;
    double hx[n] = {4.0, 0.5, 1.0, 0.75, 1e-9, 1e-18, 1e9, 3.7e-5, 1e-30, 1e30, 2.0, 1e-300};
    double *dx, *dy, *dy0, hy[n], hy0[n];
    hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8); hipMalloc(&dy0, n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy, dy0, n);
    hipMemcpy(hy, dy, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hy0, dy0, n * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("x=%.3e  rsq=%.17g  refined=%.17g  exact=%.17g  rel=%.2e\n", hx[i], hy0[i], hy[i], 1.0 / sqrt(hx[i]), fabs(hy[i] * sqrt(hx[i]) - 1.0));
    return 0;
}
