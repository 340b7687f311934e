// Microbenchmark: do f64 MFMA and f64 VALU share an execution pipe on gfx950?
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_f64 mfma_valu_f64.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

// mode: per-wave role selected by wave index within the workgroup
//  roles: 0 = MFMA loop, 1 = f64 FMA loop, 2 = f64 rcp loop, 3 = f32 FMA loop
__global__ __launch_bounds__(512) void k(int role_even, int role_odd, int iters, double* out) {
    const int w = threadIdx.x >> 6;
    // waves w and w+4 share a SIMD (wave -> SIMD is cyclic over 4): first half = "even", second = "odd"
    const int role = (w < 4) ? role_even : role_odd;
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.0000001, c = 0.5;
    f64x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    float fa = a, fb = 1.0000001f, fc = 0.5f;
    double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    float y0 = fa, y1 = fa + 1, y2 = fa + 2, y3 = fa + 3;
    if (role == 0) {
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
        }
    } else if (role == 1) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                x0 = __builtin_fma(x0, b, c); x1 = __builtin_fma(x1, b, c);
                x2 = __builtin_fma(x2, b, c); x3 = __builtin_fma(x3, b, c);
            }
        }
    } else if (role == 2) {
        for (int i = 0; i < iters; ++i) {
            x0 = __builtin_amdgcn_rcp(x0) + c; x1 = __builtin_amdgcn_rcp(x1) + c;
            x2 = __builtin_amdgcn_rcp(x2) + c; x3 = __builtin_amdgcn_rcp(x3) + c;
        }
    } else if (role == 3) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                y0 = __builtin_fmaf(y0, fb, fc); y1 = __builtin_fmaf(y1, fb, fc);
                y2 = __builtin_fmaf(y2, fb, fc); y3 = __builtin_fmaf(y3, fb, fc);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[1] + x0 + x1 + x2 + x3 + y0 + y1 + y2 + y3;
}

static float run(int re, int ro, int iters, double* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, re, ro, iters, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, re, ro, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    double* out; hipMalloc(&out, 256 * 512 * 8);
    const int it = 20000;
    const char* names[] = {"mfma_f64", "fma_f64 x16", "rcp_f64 x4(+add)", "fma_f32 x16", "idle"};
    int combos[][2] = {{0, 4}, {1, 4}, {2, 4}, {3, 4}, {0, 0}, {0, 1}, {0, 2}, {0, 3}, {1, 1}};
    for (auto& c : combos) {
        float ms = run(c[0], c[1], it, out);
        printf("%-18s | %-18s : %8.3f ms   (%.1f cycles/iter at 2.4 GHz)\n", names[c[0]], names[c[1]], ms,
               ms * 1e-3 * 2.4e9 / it);
    }
    return 0;
}
