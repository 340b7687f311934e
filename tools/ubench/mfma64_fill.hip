// What does each companion of k_fused_wide64's MFMAs cost a wavefront that is alone on its SIMD?  The D product's
// stream (two accumulators alternating, asm MFMAs, B operands in AGPRs) with, per group of four MFMAs:
//   0: nothing   1: + s_nop 1 before each MFMA   2: + one ds_read_b128 and its wait   3: + one LDS-DMA (M0 save/restore)
//   4: + s_waitcnt vmcnt(30)      (cumulative).  Prints s_memtime ticks per MFMA.
// Build: hipcc -w --offload-arch=gfx950 -O3 -o tools/ubench/bin/mfma64_fill tools/ubench/mfma64_fill.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int TPW = 9;
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(double* out, unsigned long long* ticks, int iters, const double* in, const char* stream, unsigned wrap) {
    extern __shared__ __attribute__((aligned(16))) char ring[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    d4 Vin[2][TPW];
    for (int f = 0; f < 2; ++f)
        for (int t = 0; t < TPW; ++t) Vin[f][t] = d4{in[lane], in[lane + 1], in[lane + 2], in[lane + 3 + t]};
    d4 da[2] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
    char* ring_w = ring + w * 32768;
    const unsigned ring_s = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring_w);
    const char* g = stream + (size_t)(blockIdx.x * 4 + w) * (1 << 20);
    unsigned pi = 0;
    d2 fc = d2{in[0], in[1]}, f1 = fc;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int l = 0; l < 2 * TPW; ++l) {
            const int kk = l >> 1, h = l & 1;
            d2 fn = fc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = j & 1, x = j >> 1;
                if (MODE >= 1)
                    asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(da[f]) : "v"(fc[x]), "a"(Vin[f][kk][2 * h + x]));
                else
                    asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(da[f]) : "v"(fc[x]), "a"(Vin[f][kk][2 * h + x]));
                __builtin_amdgcn_sched_barrier(0);
                if (j == 0 && MODE >= 2 && MODE != 6) {
                    if (MODE == 4) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
                    fn = *reinterpret_cast<const d2*>(ring_w + ((pi + 1) & 31) * 1024 + lane * 16);
                }
                if (j == 1 && (MODE == 3 || MODE == 4 || MODE == 6)) {
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(lane * 16), "s"(ring_s + (pi & 31) * 1024), "s"(g) : "memory");
                    g += 1024;
                    if (((pi + 1) & (wrap - 1)) == 0) g -= (size_t)wrap << 10;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 5) { fc = f1; f1 = fn; } else fc = fn;
            ++pi;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n\ts_nop 3\n\ts_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = da[0][0] + da[1][1] + fc[0];
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, unsigned wrap = 1024) {
    double *out, *in; unsigned long long* t; char* stream;
    (void)hipMalloc(&out, 256 * 256 * 8); (void)hipMalloc(&t, 256 * 8); (void)hipMalloc(&in, 4096); (void)hipMalloc(&stream, (size_t)1024 << 20);
    (void)hipMemset(in, 0, 4096); (void)hipMemset(stream, 0, (size_t)1024 << 20);
    const int iters = 300;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 131072, 0, out, t, iters, in, stream, wrap);
    (void)hipDeviceSynchronize();
    unsigned long long h[256]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i];
    m /= 256;
    printf("%s: %.1f ticks per MFMA\n", name, m / ((double)iters * 8 * TPW));
    (void)hipFree(out); (void)hipFree(t); (void)hipFree(in); (void)hipFree(stream);
}
int main() {
    run<0>("MFMAs alone"); run<1>("+ s_nop 1"); run<2>("+ ds_read_b128"); run<3>("+ LDS-DMA (1 MiB per wavefront: HBM)"); run<3>("+ LDS-DMA (32 KiB per wavefront: L2)", 32);
    run<4>("+ vmcnt wait (L2)", 32); run<5>("ds_read two positions ahead, no DMA"); run<6>("DMA (L2) but no ds_read", 32);
    return 0;
}
