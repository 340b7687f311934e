// Microbenchmark (round 4): where may the update's f64 VALU chain stand relative to the 15 f64 MFMAs of a
// k_fused_all unit?  One unit = D = A_j^T V (7-chain), the quotient chain (range test, batch inversion with one
// v_rcp_f64 + Newton step, 8 products: 9 dependent levels), V' += A_j H'_j (8 MFMAs in two 4-chains).
// Dictionary fragments are constants in registers: this measures issue behaviour only.
//
//   variant 0: the MFMAs alone
//           1: clumped as the kernel of round 3 issues it: D(k) | chain(k) | V'(k)
//           2: software-pipelined by one unit: chain(k) dealt out between the MFMAs of D(k+1), then V'(k)
//           3: three stages: slot k = V'(k-1) and D(k+1) with chain(k) dealt out one level per MFMA gap
//           4: clumped, two tiles' chains interleaved level by level (the round-2 pair batch)
//           5: variant 1 with every v_mul_f64 written as v_fma_f64 x, y, 0
//           6: variant 3 with v_fma_f64 products
//           7: the chains alone (no MFMAs), one tile after the other
//           8: the chains alone, two tiles interleaved level by level
//           9: slot k = D(k+1) and V'(k-1) MFMAs ALTERNATING (no MFMA follows one it depends on), chain(k) one level per gap
//          10: the MFMA order of 9 alone
//          11: slot k = D(k+1) | chain(k) clumped (reads the d finished a slot ago: no wait states) | V'(k)
//          12: variant 1 with the reciprocal seeded in float32 (v_cvt_f32_f64, v_rcp_f32, v_cvt_f64_f32 + one Newton step)
//          13: variant 11 with the float32 seed
// Each variant runs with 1 and 2 wavefronts per SIMD (256 / 512 threads per workgroup, one workgroup per CU); KT = 4
// tiles per wavefront also with 3 per SIMD (768 threads).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench/bin/unit_sched tools/ubench/unit_sched.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
#define SB __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ unsigned hi_word(double x) { return (unsigned)(__double_as_longlong(x) >> 32); }

template <bool FMA>
__device__ __forceinline__ double mul(double x, double y) {
    return FMA ? __builtin_fma(x, y, 0.0) : x * y;
}

// the chain by levels; state of one tile's chain in flight
struct Chain {
    double a, b, ab, R, e, Ra, Rb, r0, r1, r2, r3, q0, q1, q2, q3;
    unsigned worst;
};
template <bool FMA, bool SEED32 = false>
__device__ __forceinline__ void level(int L, Chain& c, const f64x4& d, const f64x4& p, double (&h)[4], unsigned lo) {
    switch (L) {
        case 0:
            c.worst = max(max(hi_word(d[0]) - lo, hi_word(d[1]) - lo), max(hi_word(d[2]) - lo, hi_word(d[3]) - lo));
            c.a = mul<FMA>(d[0], d[1]);
            c.b = mul<FMA>(d[2], d[3]);
            break;
        case 1: c.ab = mul<FMA>(c.a, c.b); break;
        case 2: c.R = SEED32 ? (double)__builtin_amdgcn_rcpf((float)c.ab) : __builtin_amdgcn_rcp(c.ab); break;
        case 3: c.e = __builtin_fma(-c.ab, c.R, 1.0); break;
        case 4: c.R = __builtin_fma(c.R, c.e, c.R); break;
        case 5: c.Ra = mul<FMA>(c.R, c.b); c.Rb = mul<FMA>(c.R, c.a); break;
        case 6:
            c.r0 = mul<FMA>(c.Ra, d[1]); c.r1 = mul<FMA>(c.Ra, d[0]);
            c.r2 = mul<FMA>(c.Rb, d[3]); c.r3 = mul<FMA>(c.Rb, d[2]);
            break;
        case 7:
            c.q0 = mul<FMA>(p[0], c.r0); c.q1 = mul<FMA>(p[1], c.r1);
            c.q2 = mul<FMA>(p[2], c.r2); c.q3 = mul<FMA>(p[3], c.r3);
            break;
        case 8:
            h[0] = mul<FMA>(h[0], c.q0); h[1] = mul<FMA>(h[1], c.q1);
            h[2] = mul<FMA>(h[2], c.q2); h[3] = mul<FMA>(h[3], c.q3);
            break;
    }
}

template <int VARIANT, int KT, int THREADS>
__global__ __launch_bounds__(THREADS) void k(int reps, long long* cyc, double* out, unsigned* flag) {
    constexpr bool FMA = VARIANT == 5 || VARIANT == 6;
    const int lane = threadIdx.x & 63;
    double h[KT][4];
    f64x4 p[KT];
    double a1[7], a2[2][4], v[7];
    // values that keep h near 1: d ~ 7 * a1 * v + l1 ~ 1, p ~ 1
    for (int s = 0; s < 7; ++s) { a1[s] = 0.25 + 1e-3 * lane + 1e-4 * s; v[s] = 0.5 + 1e-3 * s; }
    for (int u = 0; u < 2; ++u)
        for (int r = 0; r < 4; ++r) a2[u][r] = 0.1 + 1e-3 * (u * 4 + r) + 1e-4 * lane;
    for (int kk = 0; kk < KT; ++kk)
        for (int r = 0; r < 4; ++r) { h[kk][r] = 1.0 + 1e-3 * (kk + r); p[kk][r] = 14.0 + 0.01 * r; }
    f64x4 dinit = {1e-3, 1e-3, 1e-3, 1e-3};
    {   // (a start value in registers, as the kernel's l1 + eps: SrcC of the first MFMA of a D chain)
        double x0 = dinit[0], x1 = dinit[1], x2 = dinit[2], x3 = dinit[3];
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        dinit = f64x4{x0, x1, x2, x3};
    }
    for (int kk = 0; kk < KT; ++kk)
        for (int r = 0; r < 4; ++r) { double x = p[kk][r]; asm volatile("" : "+v"(x)); p[kk][r] = x; }
    const unsigned lo = 0x30500000u;
    unsigned bad = 0;
    f64x4 vn[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r_ = 0; r_ < reps; ++r_) {
        // keep the loop-invariant operands opaque so that nothing is hoisted or folded
#pragma unroll
        for (int s = 0; s < 7; ++s) asm volatile("" : "+v"(v[s]));
        vn[0] = f64x4{0, 0, 0, 0};
        vn[1] = f64x4{0, 0, 0, 0};
        auto dmf = [&](f64x4& d, int s) {
            if (s == 0) asm volatile("" : "+v"(a1[0]));
            d = MF(a1[s], v[s], d);
        };
        auto vmf = [&](int kk, int i) {      // i-th of the 8 V' MFMAs of tile kk
            const int u = i & 1, r = i >> 1;
            vn[u] = MF(a2[u][r], h[kk][r], vn[u]);
        };
        if (VARIANT == 0) {
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                f64x4 d = dinit;
#pragma unroll
                for (int s = 0; s < 7; ++s) dmf(d, s);
                SB;
                h[kk][0] += d[0] * 1e-30;       // (keeps d alive: one VALU)
#pragma unroll
                for (int i = 0; i < 8; ++i) vmf(kk, i);
                SB;
            }
        } else if (VARIANT == 1 || VARIANT == 5) {
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                f64x4 d = dinit;
#pragma unroll
                for (int s = 0; s < 7; ++s) dmf(d, s);
                SB;
                Chain c;
#pragma unroll
                for (int L = 0; L < 9; ++L) { level<FMA>(L, c, d, p[kk], h[kk], lo); SB; }
                bad |= c.worst;
#pragma unroll
                for (int i = 0; i < 8; ++i) vmf(kk, i);
                SB;
            }
        } else if (VARIANT == 2) {
            f64x4 d = dinit;
#pragma unroll
            for (int s = 0; s < 7; ++s) dmf(d, s);
            SB;
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                f64x4 dn = dinit;
                Chain c;
#pragma unroll
                for (int s = 0; s < 7; ++s) {
                    if (kk + 1 < KT) { dmf(dn, s); SB; }
                    level<FMA>(s, c, d, p[kk], h[kk], lo);
                    SB;
                }
                level<FMA>(7, c, d, p[kk], h[kk], lo); SB;
                level<FMA>(8, c, d, p[kk], h[kk], lo); SB;
                bad |= c.worst;
#pragma unroll
                for (int i = 0; i < 8; ++i) vmf(kk, i);
                SB;
                d = dn;
            }
        } else if (VARIANT == 3 || VARIANT == 6) {
            // slot kk: D(kk+1) [7 MFMAs], then V'(kk-1) [8 MFMAs]; chain(kk) one level per gap from the first gap
            // on (d(kk) was finished 8 MFMAs ago: no wait states; h'(kk-1) left the chain in the previous slot)
            f64x4 d = dinit;
#pragma unroll
            for (int s = 0; s < 7; ++s) dmf(d, s);
            SB;
#pragma unroll
            for (int kk = 0; kk <= KT; ++kk) {
                f64x4 dn = dinit;
                Chain c;
                int L = 0;
#pragma unroll
                for (int s = 0; s < 7; ++s) {
                    if (kk + 1 < KT) { dmf(dn, s); SB; }
                    if (kk < KT && L < 9) { level<FMA>(L, c, d, p[kk < KT ? kk : 0], h[kk < KT ? kk : 0], lo); ++L; SB; }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (kk > 0) { vmf(kk - 1, i); SB; }
                    if (kk < KT && L < 9) { level<FMA>(L, c, d, p[kk < KT ? kk : 0], h[kk < KT ? kk : 0], lo); ++L; SB; }
                }
                if (kk < KT) bad |= c.worst;
                d = dn;
            }
        } else if (VARIANT == 4) {
#pragma unroll
            for (int kk = 0; kk < KT; kk += 2) {
                f64x4 da = dinit, db = dinit;
#pragma unroll
                for (int s = 0; s < 7; ++s) dmf(da, s);
#pragma unroll
                for (int s = 0; s < 7; ++s) dmf(db, s);
                SB;
                Chain ca, cb;
#pragma unroll
                for (int L = 0; L < 9; ++L) {
                    level<FMA>(L, ca, da, p[kk], h[kk], lo);
                    level<FMA>(L, cb, db, p[kk + 1], h[kk + 1], lo);
                    SB;
                }
                bad |= ca.worst | cb.worst;
#pragma unroll
                for (int i = 0; i < 8; ++i) vmf(kk, i);
#pragma unroll
                for (int i = 0; i < 8; ++i) vmf(kk + 1, i);
                SB;
            }
        } else if (VARIANT == 9 || VARIANT == 10) {
            f64x4 d = dinit;
#pragma unroll
            for (int s = 0; s < 7; ++s) dmf(d, s);
            SB;
#pragma unroll
            for (int kk = 0; kk <= KT; ++kk) {
                f64x4 dn = dinit;
                Chain c;
                int L = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (i < 7 && kk + 1 < KT) { dmf(dn, i); SB; }
                    if (VARIANT == 9 && kk < KT && L < 9) { level<FMA>(L, c, d, p[kk < KT ? kk : 0], h[kk < KT ? kk : 0], lo); ++L; SB; }
                    if (kk > 0) { vmf(kk - 1, i); SB; }
                    if (VARIANT == 9 && kk < KT && L < 9) { level<FMA>(L, c, d, p[kk < KT ? kk : 0], h[kk < KT ? kk : 0], lo); ++L; SB; }
                }
                if (VARIANT == 9 && kk < KT) bad |= c.worst;
                if (VARIANT == 10 && kk < KT) h[kk][0] += d[0] * 1e-30;
                d = dn;
            }
        } else if (VARIANT == 11 || VARIANT == 13) {
            f64x4 d = dinit;
#pragma unroll
            for (int s = 0; s < 7; ++s) dmf(d, s);
            SB;
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                f64x4 dn = dinit;
                if (kk + 1 < KT) {
#pragma unroll
                    for (int s = 0; s < 7; ++s) dmf(dn, s);
                }
                SB;
                Chain c;
#pragma unroll
                for (int L = 0; L < 9; ++L) { level<FMA, VARIANT == 13>(L, c, d, p[kk], h[kk], lo); SB; }
                bad |= c.worst;
#pragma unroll
                for (int i = 0; i < 8; ++i) vmf(kk, i);
                SB;
                d = dn;
            }
        } else if (VARIANT == 12) {
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                f64x4 d = dinit;
#pragma unroll
                for (int s = 0; s < 7; ++s) dmf(d, s);
                SB;
                Chain c;
#pragma unroll
                for (int L = 0; L < 9; ++L) { level<FMA, true>(L, c, d, p[kk], h[kk], lo); SB; }
                bad |= c.worst;
#pragma unroll
                for (int i = 0; i < 8; ++i) vmf(kk, i);
                SB;
            }
        } else if (VARIANT == 7) {
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                f64x4 d = dinit;
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r] += v[r] * h[kk][r];
                SB;
                Chain c;
#pragma unroll
                for (int L = 0; L < 9; ++L) { level<FMA>(L, c, d, p[kk], h[kk], lo); SB; }
                bad |= c.worst;
            }
        } else if (VARIANT == 8) {
#pragma unroll
            for (int kk = 0; kk < KT; kk += 2) {
                f64x4 da = dinit, db = dinit;
#pragma unroll
                for (int r = 0; r < 4; ++r) { da[r] += v[r] * h[kk][r]; db[r] += v[r] * h[kk + 1][r]; }
                SB;
                Chain ca, cb;
#pragma unroll
                for (int L = 0; L < 9; ++L) {
                    level<FMA>(L, ca, da, p[kk], h[kk], lo);
                    level<FMA>(L, cb, db, p[kk + 1], h[kk + 1], lo);
                    SB;
                }
                bad |= ca.worst | cb.worst;
            }
        }
        // renormalise so that the values stay finite whatever the variant computes (4 VALU per sweep and tile)
#pragma unroll
        for (int kk = 0; kk < KT; ++kk)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[kk][r] = h[kk][r] > 2.0 ? 1.0 : (h[kk][r] < 0.5 ? 1.0 : h[kk][r]);
        v[0] += vn[0][0] * 1e-300 + vn[1][1] * 1e-300;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 12 + (threadIdx.x >> 6)] = t1 - t0;
    double s = v[0];
    for (int kk = 0; kk < KT; ++kk) s += h[kk][0] + h[kk][1] + h[kk][2] + h[kk][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (bad == 0xFFFFFFFFu) *flag = 1;
}

template <int VARIANT, int KT, int THREADS>
static void run(const char* name, int reps, long long* cyc, double* out, unsigned* flag) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<VARIANT, KT, THREADS>), dim3(256), dim3(THREADS), 0, 0, reps, cyc, out, flag);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<VARIANT, KT, THREADS>), dim3(256), dim3(THREADS), 0, 0, reps, cyc, out, flag);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> hc(256 * 12);
    hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost);
    const int threads = THREADS;
    const int waves = threads / 64;
    std::vector<double> per;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < waves; ++w) per.push_back((double)hc[b * 12 + w] / ((double)reps * KT));
    std::sort(per.begin(), per.end());
    const double med = per[per.size() / 2];
    // per SIMD and unit: a SIMD holds waves / 4 wavefronts, each doing KT units per repetition
    const double per_simd_unit = med / (waves / 4);
    printf("%-52s KT=%d %d waves/SIMD: %8.1f ticks per unit and wavefront, %8.1f per unit and SIMD  (%.3f ms; %.1f cycles/unit/SIMD by wall at 2.4 GHz)\n",
           name, KT, waves / 4, med, per_simd_unit, ms, ms * 1e-3 * 2.4e9 / ((double)reps * KT * (waves / 4)));
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    long long* cyc; double* out; unsigned* flag;
    hipMalloc(&cyc, 256 * 12 * 8); hipMalloc(&out, 256 * 768 * 8); hipMalloc(&flag, 4);
#define RUN3(V, NAME)                                    \
    run<V, 8, 256>(NAME, reps, cyc, out, flag);          \
    run<V, 8, 512>(NAME, reps, cyc, out, flag);          \
    run<V, 4, 256>(NAME, reps, cyc, out, flag);          \
    run<V, 4, 512>(NAME, reps, cyc, out, flag);          \
    run<V, 4, 768>(NAME, reps, cyc, out, flag);
    RUN3(0, "0 MFMAs alone (15 per unit)")
    RUN3(1, "1 clumped: D | chain | V'")
    RUN3(2, "2 pipelined: chain(k) inside D(k+1), then V'(k)")
    RUN3(3, "3 three stages: chain(k) inside V'(k-1) + D(k+1)")
    RUN3(4, "4 clumped pairs, chains interleaved by level")
    RUN3(5, "5 clumped, products as v_fma_f64")
    RUN3(6, "6 three stages, products as v_fma_f64")
    RUN3(7, "7 chains alone, one after the other")
    RUN3(8, "8 chains alone, pairs interleaved by level")
    RUN3(9, "9 D(k+1)/V'(k-1) alternating, chain(k) in the gaps")
    RUN3(10, "10 the MFMA order of 9 alone")
    RUN3(11, "11 D(k+1) | chain(k) clumped | V'(k)")
    RUN3(12, "12 clumped, f32-seeded reciprocal")
    RUN3(13, "13 variant 11 + f32 seed")
    return 0;
}
