// Does the register file a fragment lands in matter?  v_mfma_f64_16x16x4_f64, one wavefront per SIMD, one ds_read_b128
// (or one buffer_load_dwordx4) per four MFMAs; accumulators / fragments in the VGPR (v) or accumulator (a) half.
// Build: hipcc -w --offload-arch=gfx950 -O3 -o tools/ubench/bin/mfma64_ports tools/ubench/mfma64_ports.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int TPW = 9;
// ACC: 0 = accumulators in VGPRs, 1 = in AGPRs.  FR: 0 = fragment via ds_read into VGPRs, 1 = ds_read into AGPRs,
// 2 = buffer_load into VGPRs, 3 = no load at all, 4 = two ds_read_b64 into VGPRs
template <int ACC, int FR, int BIG = 0>
__global__ __launch_bounds__(256, 1) void k(double* out, unsigned long long* ticks, int iters, const double* in, const char* stream) {
    extern __shared__ __attribute__((aligned(16))) char ring_dyn[];
    char (*ring)[8192] = reinterpret_cast<char (*)[8192]>(ring_dyn + (BIG ? BIG * 1024 : 0));
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    d4 Vin[2][TPW];
    for (int f = 0; f < 2; ++f)
        for (int t = 0; t < TPW; ++t) Vin[f][t] = d4{in[lane], in[lane + 1], in[lane + 2], in[lane + 3 + t]};
    d4 da[2] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
    const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&ring[w][0] + lane * 16;
    const unsigned long long gp = (unsigned long long)(stream + w * 8192);
    const u4 rs = u4{(unsigned)gp, (unsigned)(gp >> 32) & 0xffffu, 8192u, 0x00020000u};
    d2 fc = d2{in[0], in[1]}, fn = fc;
    unsigned pi = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int l = 0; l < 2 * TPW; ++l) {
            const int kk = l >> 1, h = l & 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = j & 1, x = j >> 1;
                if (ACC == 0 && FR != 1) asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(da[f]) : "v"(fc[x]), "a"(Vin[f][kk][2 * h + x]));
                if (ACC == 0 && FR == 1) asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(da[f]) : "a"(fc[x]), "a"(Vin[f][kk][2 * h + x]));
                if (ACC == 1 && FR != 1) asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(da[f]) : "v"(fc[x]), "v"(Vin[f][kk][2 * h + x]));
                if (ACC == 1 && FR == 1) asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(da[f]) : "a"(fc[x]), "v"(Vin[f][kk][2 * h + x]));
                if (j == 0) {
                    const unsigned addr = la + ((pi + 1) & 7) * 1024;
                    if (FR == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(fn) : "v"(addr) : "memory");
                    if (FR == 1) asm volatile("ds_read_b128 %0, %1" : "=a"(fn) : "v"(addr) : "memory");
                    if (FR == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(fn) : "v"(lane * 16 + ((pi + 1) & 7) * 1024), "s"(rs) : "memory");
                    if (FR == 4) asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8" : "=v"(fn[0]), "=v"(fn[1]) : "v"(addr) : "memory");
                }
                if (j == 3) {
                    if (FR == 0 || FR == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fn)::"memory");
                    if (FR == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+a"(fn)::"memory");
                    if (FR == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(fn)::"memory");
                }
            }
            fc = fn;
            ++pi;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = da[0][0] + da[1][1] + fc[0];
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int ACC, int FR, int BIG = 0> void run(const char* name) {
    double *out, *in; unsigned long long* t; char* stream;
    (void)hipMalloc(&out, 256 * 256 * 8); (void)hipMalloc(&t, 256 * 8); (void)hipMalloc(&in, 4096); (void)hipMalloc(&stream, 65536);
    (void)hipMemset(in, 0, 4096); (void)hipMemset(stream, 0, 65536);
    const int iters = 300;
    for (int rep = 0; rep < 2; ++rep) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<ACC, FR, BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); hipLaunchKernelGGL((k<ACC, FR, BIG>), dim3(256), dim3(256), (BIG + 32) * 1024, 0, out, t, iters, in, stream); }
    (void)hipDeviceSynchronize();
    unsigned long long h[256]; (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i];
    m /= 256;
    printf("%s: %.1f ticks per MFMA\n", name, m / ((double)iters * 8 * TPW));
    (void)hipFree(out); (void)hipFree(t); (void)hipFree(in); (void)hipFree(stream);
}
int main() {
    run<0, 3>("acc v, no fragment load          ");
    run<0, 0>("acc v, ds_read_b128 -> v         ");
    run<0, 0, 60>("   ... ring at LDS offset 60 KiB  ");
    run<0, 0, 100>("   ... ring at LDS offset 100 KiB ");
    run<0, 0, 120>("   ... ring at LDS offset 120 KiB ");
    run<0, 4>("acc v, 2 x ds_read_b64 -> v      ");
    run<0, 1>("acc v, ds_read_b128 -> a         ");
    run<1, 3>("acc a, no fragment load          ");
    run<1, 0>("acc a, ds_read_b128 -> v         ");
    run<1, 1>("acc a, ds_read_b128 -> a         ");
    run<0, 2>("acc v, buffer_load_dwordx4 -> v  ");
    run<1, 2>("acc a, buffer_load_dwordx4 -> v  ");
    return 0;
}
