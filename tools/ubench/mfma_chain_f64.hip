// Microbenchmark: issue interval vs dependent-accumulate latency of v_mfma_f64_16x16x4_f64, and the cost of
// f64 VALU work next to it, with ONE wavefront per SIMD (the situation of k_fused_all's sweep).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_chain_f64 mfma_chain_f64.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
#define SB __builtin_amdgcn_sched_barrier(0)

template <int CHAINS>
__device__ __forceinline__ void mfma_chains(double a, double b, f64x4 (&acc)[8], int n) {
    for (int i = 0; i < n; i += CHAINS) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = MF(a, b, acc[c]);
    }
}

__global__ __launch_bounds__(256) void k(int test, int reps, long long* cyc, double* out) {
    const int lane = threadIdx.x & 63;
    double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    f64x4 acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = f64x4{0, 0, 0, 0};
    double x0 = a, x1 = b, x2 = a + 1, x3 = b + 1;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        switch (test) {
            case 0: mfma_chains<1>(a, b, acc, 64); break;
            case 1: mfma_chains<2>(a, b, acc, 64); break;
            case 2: mfma_chains<4>(a, b, acc, 64); break;
            case 3: mfma_chains<8>(a, b, acc, 64); break;
            case 4:      // the unit: a 7-chain, then 8 MFMAs in two chains; 8 units
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    f64x4 d = {0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < 7; ++s) d = MF(a, b, d);
                    acc[2][u & 3] += d[0];          // keep d alive cheaply (1 VALU)
                    SB;
#pragma unroll
                    for (int s = 0; s < 4; ++s) { acc[0] = MF(a, b, acc[0]); acc[1] = MF(b, a, acc[1]); }
                    SB;
                }
                break;
            case 5:      // two 7-chains interleaved (D of two tiles at once), then 16 MFMAs in two chains; 4 pairs
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    f64x4 d = {0, 0, 0, 0}, e = {0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < 7; ++s) { d = MF(a, b, d); e = MF(b, a, e); }
                    acc[2][u & 3] += d[0] + e[1];
                    SB;
#pragma unroll
                    for (int s = 0; s < 8; ++s) { acc[0] = MF(a, b, acc[0]); acc[1] = MF(b, a, acc[1]); }
                    SB;
                }
                break;
            case 6:      // 32 dependent f64 multiplies
#pragma unroll
                for (int i = 0; i < 32; ++i) x0 = x0 * b;
                break;
            case 7:      // 32 f64 multiplies in 4 independent chains
#pragma unroll
                for (int i = 0; i < 8; ++i) { x0 = x0 * b; x1 = x1 * b; x2 = x2 * b; x3 = x3 * b; }
                break;
            case 8:      // 8 dependent v_rcp_f64
#pragma unroll
                for (int i = 0; i < 8; ++i) x0 = __builtin_amdgcn_rcp(x0);
                break;
            case 11:     // 32 dependent v_fma_f64
#pragma unroll
                for (int i = 0; i < 32; ++i) x0 = __builtin_fma(x0, b, a);
                break;
            case 12:     // 32 v_fma_f64 in 4 chains
#pragma unroll
                for (int i = 0; i < 8; ++i) { x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); x2 = __builtin_fma(x2, b, a); x3 = __builtin_fma(x3, b, a); }
                break;
            case 13:     // 32 v_add_f64 in 4 chains
#pragma unroll
                for (int i = 0; i < 8; ++i) { x0 = x0 + b; x1 = x1 + b; x2 = x2 + b; x3 = x3 + b; }
                break;
            case 14:     // 8 v_rcp_f64 in 4 chains
#pragma unroll
                for (int i = 0; i < 2; ++i) { x0 = __builtin_amdgcn_rcp(x0); x1 = __builtin_amdgcn_rcp(x1); x2 = __builtin_amdgcn_rcp(x2); x3 = __builtin_amdgcn_rcp(x3); }
                break;
            case 15:     // 32 products as v_fma_f64 x, x, b, 0 (inline asm), 4 chains
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("v_fma_f64 %0, %0, %1, 0" : "+v"(x0) : "v"(b));
                    asm volatile("v_fma_f64 %0, %0, %1, 0" : "+v"(x1) : "v"(b));
                    asm volatile("v_fma_f64 %0, %0, %1, 0" : "+v"(x2) : "v"(b));
                    asm volatile("v_fma_f64 %0, %0, %1, 0" : "+v"(x3) : "v"(b));
                }
                break;
            case 16:     // 32 v_mul_f32 in 4 chains (reference)
            {
                float f0 = (float)x0, f1 = (float)x1, f2 = (float)x2, f3 = (float)x3, fb = (float)b;
#pragma unroll
                for (int i = 0; i < 8; ++i) { f0 *= fb; f1 *= fb; f2 *= fb; f3 *= fb; }
                x0 = f0; x1 = f1; x2 = f2; x3 = f3;
                break;
            }
            case 17:     // 32 integer v_max_u32 / v_sub in 4 chains (the range test's kind)
            {
                unsigned u0 = (unsigned)lane, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, ub = (unsigned)reps;
#pragma unroll
                for (int i = 0; i < 8; ++i) { u0 = max(u0 - ub, u1); u1 = max(u1 - ub, u2); u2 = max(u2 - ub, u3); u3 = max(u3 - ub, u0); }
                x0 += u0; x1 += u1; x2 += u2; x3 += u3;
                break;
            }
            case 9:      // case 4 with a 24-operation f64 clump (4 chains x 6) between the D chain and the V' MFMAs
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    f64x4 d = {0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < 7; ++s) d = MF(a, b, d);
                    SB;
                    double y0 = d[0], y1 = d[1], y2 = d[2], y3 = d[3];
#pragma unroll
                    for (int i = 0; i < 6; ++i) { y0 = y0 * b; y1 = y1 * b; y2 = y2 * b; y3 = y3 * b; }
                    SB;
                    acc[0] = MF(y0, a, acc[0]); acc[1] = MF(y1, a, acc[1]);
                    acc[0] = MF(y2, a, acc[0]); acc[1] = MF(y3, a, acc[1]);
#pragma unroll
                    for (int s = 0; s < 2; ++s) { acc[0] = MF(a, b, acc[0]); acc[1] = MF(b, a, acc[1]); }
                    SB;
                }
                break;
            case 10:     // case 9 software-pipelined: the clump works on the PREVIOUS unit's d
            {
                f64x4 dp = {1, 1, 1, 1};
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    f64x4 d = {0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < 7; ++s) d = MF(a, b, d);
                    SB;
                    double y0 = dp[0], y1 = dp[1], y2 = dp[2], y3 = dp[3];
#pragma unroll
                    for (int i = 0; i < 6; ++i) { y0 = y0 * b; y1 = y1 * b; y2 = y2 * b; y3 = y3 * b; }
                    SB;
                    acc[0] = MF(y0, a, acc[0]); acc[1] = MF(y1, a, acc[1]);
                    acc[0] = MF(y2, a, acc[0]); acc[1] = MF(y3, a, acc[1]);
#pragma unroll
                    for (int s = 0; s < 2; ++s) { acc[0] = MF(a, b, acc[0]); acc[1] = MF(b, a, acc[1]); }
                    SB;
                    dp = d;
                }
                acc[3] = dp;
                break;
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    double s = x0 + x1 + x2 + x3;
    for (int c = 0; c < 8; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    long long* cyc; double* out;
    hipMalloc(&cyc, 256 * 8); hipMalloc(&out, 256 * 256 * 8);
    const char* names[] = {"64 MFMA, 1 chain", "64 MFMA, 2 chains", "64 MFMA, 4 chains", "64 MFMA, 8 chains",
                           "8 x (7-chain + 8 MFMA in 2 chains) = 120 MFMA", "4 x (two 7-chains interleaved + 16 in 2 chains) = 120 MFMA",
                           "32 dependent v_mul_f64", "32 v_mul_f64 in 4 chains", "8 dependent v_rcp_f64",
                           "case 4 + 24 f64 multiplies between D and V' (reads d at once)",
                           "same, software-pipelined (clump reads the previous d)",
                           "32 dependent v_fma_f64", "32 v_fma_f64 in 4 chains", "32 v_add_f64 in 4 chains", "8 v_rcp_f64 in 4 chains",
                           "32 v_fma_f64 x,x,b,0 (asm) in 4 chains", "32 v_mul_f32 in 4 chains (+8 cvt)", "64 int sub/max in 4 chains"};
    const int reps = 200;
    for (int t = 0; t <= 17; ++t) {
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, t, reps, cyc, out);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, t, reps, cyc, out);
        hipDeviceSynchronize();
        long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256.0 * reps;
        printf("%-75s : %9.1f cycles per repetition\n", names[t], m);
    }
    return 0;
}
