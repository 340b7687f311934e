// Accuracy of v_rcp_f64 on gfx950, raw and after one Newton step, against IEEE division.
// hipcc --offload-arch=gfx950 -O3 -o rcp_accuracy rcp_accuracy.hip && ./rcp_accuracy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double lo, double hi, long n, double* out) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double e0 = 0, e1 = 0, e2 = 0;
    for (long i = gid; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        const double u = (double)(z >> 11) * 0x1p-53;
        const double x = lo * exp2(u * log2(hi / lo));
        const double num = 1.0 + (double)((z >> 3) & 0xfffff) * 0x1p-20;
        const double ex = num / x;
        double r = __builtin_amdgcn_rcp(x);
        e0 = fmax(e0, fabs(num * r - ex) / ex);
        const double e = __builtin_fma(-x, r, 1.0);
        r = __builtin_fma(r, e, r);
        const double q = num * r;
        e1 = fmax(e1, fabs(q - ex) / ex);
        const double rem = __builtin_fma(-x, q, num);
        e2 = fmax(e2, fabs(__builtin_fma(rem, r, q) - ex) / ex);
    }
    for (int o = 32; o; o >>= 1) { e0 = fmax(e0, __shfl_down(e0, o)); e1 = fmax(e1, __shfl_down(e1, o)); e2 = fmax(e2, __shfl_down(e2, o)); }
    if ((threadIdx.x & 63) == 0) {
        atomicMax((unsigned long long*)&out[0], (unsigned long long)__double_as_longlong(e0));
        atomicMax((unsigned long long*)&out[1], (unsigned long long)__double_as_longlong(e1));
        atomicMax((unsigned long long*)&out[2], (unsigned long long)__double_as_longlong(e2));
    }
}
int main() {
    double* d; hipMalloc(&d, 24);
    const double ranges[][2] = {{1e-3, 1e3}, {1e-30, 1e-20}, {1e20, 1e30}, {1.0, 2.0}};
    for (auto& r : ranges) {
        hipMemset(d, 0, 24);
        hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, r[0], r[1], 1L << 28, d);
        double h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("x in [%g,%g]: max rel err  raw rcp %.3e (2^%.1f)  +1 Newton %.3e (%.2f ulp)  +residual %.3e (%.2f ulp)\n", r[0], r[1],
               h[0], log2(h[0]), h[1], h[1] / 0x1p-53 / 2, h[2], h[2] / 0x1p-53 / 2);
    }
    return 0;
}
