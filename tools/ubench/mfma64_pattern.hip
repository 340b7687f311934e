// MFMA issue rate of k_fused_wide64's two products with everything else taken away (operands in registers, no loads):
//   D pattern : 2 accumulators alternate, the B operand (V) changes with every MFMA, the A operand every second
//   V' pattern: a new pair of accumulator tiles every 4 MFMAs (2 TPW tiles in all), B operand from 8 registers
// One wavefront per SIMD.  Prints s_memtime ticks per MFMA.
// Build: hipcc -w --offload-arch=gfx950 -O3 [-mllvm -amdgpu-mfma-vgpr-form=1] -o tools/ubench/bin/mfma64_pattern tools/ubench/mfma64_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int TPW = 9;
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(double* out, unsigned long long* ticks, int iters, const double* in) {
    d4 Vin[2][TPW], Vn[2][TPW], hf[2];
    for (int f = 0; f < 2; ++f)
        for (int t = 0; t < TPW; ++t) {
            Vin[f][t] = d4{in[threadIdx.x], in[threadIdx.x + 1], in[threadIdx.x + 2], in[threadIdx.x + 3 + t]};
            Vn[f][t] = d4{0, 0, 0, 0};
        }
    hf[0] = Vin[0][0]; hf[1] = Vin[1][1];
    double fr[4] = {in[0], in[1], in[2], in[3]};
    d4 da[2] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int l = 0; l < 2 * TPW; ++l) {
                const int kk = l >> 1, h = l & 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = j & 1, x = j >> 1;
                    da[f] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[(l + x) & 3], Vin[f][kk][2 * h + x], da[f], 0, 0, 0);
                }
            }
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int l = 0; l < 2 * TPW; ++l) {
                const int kk = l >> 1, h = l & 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = j & 1, x = j >> 1;
                    Vn[f][kk] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[(l + x) & 3], hf[f][2 * h + x], Vn[f][kk], 0, 0, 0);
                }
            }
        }
        if (MODE == 2) { hf[0] = da[0]; hf[1] = da[1]; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = da[0][0] + da[1][1];
    for (int f = 0; f < 2; ++f)
        for (int t = 0; t < TPW; ++t) s += Vn[f][t][0] + Vn[f][t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int per_iter) {
    double *out, *in; unsigned long long* t;
    (void)hipMalloc(&out, 256 * 256 * 8); (void)hipMalloc(&t, 256 * 8); (void)hipMalloc(&in, 4096);
    (void)hipMemset(in, 0, 4096);
    const int iters = 500;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, t, iters, in);
    (void)hipDeviceSynchronize();
    unsigned long long h[256]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i];
    m /= 256;
    printf("%s: %.1f ticks per MFMA\n", name, m / ((double)iters * per_iter));
}
int main() {
    run<0>("D pattern", 4 * 2 * TPW);
    run<1>("V' pattern", 4 * 2 * TPW);
    run<2>("both", 8 * 2 * TPW);
    return 0;
}
