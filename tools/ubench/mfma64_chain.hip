// How fast does ONE wavefront per SIMD issue v_mfma_f64_16x16x4_f64 when the accumulators form chains of
// distance D (D independent accumulators used round-robin)?  Prints cycles per MFMA (s_memtime) for D = 1, 2, 4, 8.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/mfma64_chain tools/ubench/mfma64_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int D, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void k(double* out, unsigned long long* ticks, int iters) {
    d4 acc[D];
    for (int i = 0; i < D; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) acc[u % D] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u % D], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < D; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int D, int WAVES> void run(const char* name) {
    double* out; unsigned long long* t;
    hipMalloc(&out, 256 * WAVES * 64 * 8); hipMalloc(&t, 256 * 8);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<D, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, t, iters);
    hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i];
    m /= 256;
    printf("%s: chain distance %d, %d wavefronts per CU: %.1f ticks per MFMA per wavefront\n", name, D, WAVES, m / (iters * 32.0));
    hipFree(out); hipFree(t);
}
int main() {
    run<1, 4>("f64"); run<2, 4>("f64"); run<4, 4>("f64"); run<8, 4>("f64");
    run<2, 8>("f64"); run<4, 8>("f64");
    return 0;
}
