// Dense contractions of the exemplar-NMF path on the gfx950 matrix cores.
//
//   k_gemm_nt      C = L R^T on zero-padded, row-major workspace operands (both k-contiguous),
//                  LDS-staged k-slabs (double buffered), one 16x16x4 MFMA per (tile, k-step).
//                  Instantiated for float64 (v_mfma_f64_16x16x4_f64) and float32
//                  (v_mfma_f32_16x16x4_f32 - exact f32, no reduced-precision path on gfx950).
//                  With MU=true the epilogue is the multiplicative update itself, so the
//                  denominator tile never leaves the accumulator registers:
//                     GRAM      H' = mu(H, P, H G^T)     sklearn _nmf.py:554,620-629
//                     FACTORED  H' = mu(H, P, V A_t^T)   with V = H A_m^T from the plain form
//   k_gemm_strided bounds-checked, arbitrary strides, for caller-owned memory (evc_synthesize:
//                  04_align_n_nmf.py:391 np.matmul(H.T, B)).
//
// Everything in the workspace is stored frames-as-rows ("t-major"): rows of L are frames,
// rows of R are exemplars (or bins), so C[t][n] rows are contiguous in n and a 16-lane
// group of the MFMA result writes one 128-byte (f64) segment.
#include "evc_internal.h"
#include <type_traits>

#include <stdlib.h>

namespace evc {

constexpr int KS = 16;  // k-slab depth staged per barrier
constexpr int NT_EPI_KL = 100, NT_EPI_STORE = 101;      // epilogue bodies besides the four eps modes

template <typename T, int E> struct VecOf { typedef T type __attribute__((ext_vector_type(E))); };

template <typename T, int BM, int BN, int WM, int WN, bool MU>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void k_gemm_nt(
    const T* __restrict__ L, int ldl, const T* __restrict__ R, int ldr, T* __restrict__ C, int ldc,
    int Kd, MuEpilogue<T> ep, long slab, int jv) {
    // (An XCD-aware renumbering of the blocks, as in k_gemm2, was measured here and dropped: no change at C3, where
    // it cut each XCD's share of the dictionary to one eighth, and 18 % slower on Griffin-Lim's 84-tile products.)
    if (ep.gate && *ep.gate == 0) return;      // (uniform: every utterance has stopped)
    const unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    // split-K: block z of gridDim.z owns the k-slabs [z n / Z, (z + 1) n / Z) of the n = Kd / KS (the split need not
    // divide them) and writes its partial product to slab z
    const int s_lo = (int)((long)bz * (Kd / KS) / gridDim.z), s_hi = (int)((long)(bz + 1) * (Kd / KS) / gridDim.z);
    L += (long)s_lo * KS;
    R += (long)s_lo * KS;
    C += (long)bz * slab;
    constexpr int NWN = BN / WN;
    constexpr int NTHR = (BM / WM) * NWN * 64;
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int EL = BM * KS / NTHR;   // elements of the L slab each thread stages
    constexpr int ER = BN * KS / NTHR;
    static_assert(EL >= 1 && ER >= 1 && KS % EL == 0 && KS % ER == 0, "staging shape");
    typedef typename Mma<T>::acc_t acc_t;
    typedef typename VecOf<T, EL>::type vecL;
    typedef typename VecOf<T, ER>::type vecR;

    __shared__ T sL[2][KS][BM + 1];
    __shared__ T sR[2][KS][BN + 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w / NWN, wn = w % NWN;
    const int i16 = lane & 15, q = lane >> 4;
    const long bi = (long)by * BM, bj = (long)bx * BN;
    (void)jv;   // (k_gemm2 skips the zero rows of R from jv on.  Here both forms of the skip - a run-time bound in the
                // unrolled loops, and a slab body specialised per group count - made the kernel 5-8 % SLOWER at C3.)

    if (MU) {
        // a block whose frames all belong to stopped utterances only carries H over
        int any = 0;
        if (tid < BM) {
            int u = ep.frame_utt[bi + tid];
            any = (u >= 0) && (ep.active[u] != 0);
        }
        if (!__syncthreads_or(any)) {
            if (C != ep.Hin) {
                for (int e = tid; e < BM * BN; e += NTHR) {
                    long r = bi + e / BN, c = bj + e % BN;
                    C[r * ldc + c] = ep.Hin[r * ep.ldh + c];
                }
            }
            return;
        }
    }

    const int lrow = tid / (KS / EL), lk = (tid % (KS / EL)) * EL;
    const int rrow = tid / (KS / ER), rk = (tid % (KS / ER)) * ER;
    const T* gL = L + (bi + lrow) * ldl + lk;
    const T* gR = R + (bj + rrow) * ldr + rk;

    acc_t acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b) acc[a][b] = acc_t{0, 0, 0, 0};

    vecL vl = *reinterpret_cast<const vecL*>(gL);
    vecR vr = *reinterpret_cast<const vecR*>(gR);
#pragma unroll
    for (int e = 0; e < EL; ++e) sL[0][lk + e][lrow] = vl[e];
#pragma unroll
    for (int e = 0; e < ER; ++e) sR[0][rk + e][rrow] = vr[e];
    __syncthreads();

    const int nslab = s_hi - s_lo;
    for (int sl = 0; sl < nslab; ++sl) {
        const int buf = sl & 1;
        const bool more = sl + 1 < nslab;
        if (more) {  // next slab's global loads fly while this slab feeds the matrix cores
            vl = *reinterpret_cast<const vecL*>(gL + (long)(sl + 1) * KS);
            vr = *reinterpret_cast<const vecR*>(gR + (long)(sl + 1) * KS);
        }
        // fragments of k-step st + 1 are requested before the MFMAs of k-step st issue (two register sets): the
        // LDS latency then runs beside the matrix pipe instead of in front of every group of MFMAs
        T fa[2][MI], fb[2][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[0][mi] = sL[buf][q][wm * WM + 16 * mi + i16];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fb[0][ni] = sR[buf][q][wn * WN + 16 * ni + i16];
#pragma unroll
        for (int st = 0; st < KS / 4; ++st) {
            if (st + 1 < KS / 4) {
                const int kk = 4 * (st + 1) + q;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) fa[(st + 1) & 1][mi] = sL[buf][kk][wm * WM + 16 * mi + i16];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) fb[(st + 1) & 1][ni] = sR[buf][kk][wn * WN + 16 * ni + i16];
            }
            __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks the reads behind the MFMAs)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = Mma<T>::mma(fa[st & 1][mi], fb[st & 1][ni], acc[mi][ni]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
#pragma unroll
            for (int e = 0; e < EL; ++e) sL[buf ^ 1][lk + e][lrow] = vl[e];
#pragma unroll
            for (int e = 0; e < ER; ++e) sR[buf ^ 1][rk + e][rrow] = vr[e];
        }
        __syncthreads();
    }

    // Epilogue.  The guard mode is a template argument of the body (one uniform switch in front of it instead of one
    // per element), a stopped frame is a select, the zero padding of the last exemplar block a wave-uniform case -
    // as in k_gemm2: straight-line code in which the divisions of a tile overlap.
    auto epilogue_body = [&](auto mode_tag, auto edge_tag) {
        constexpr int MODE = decltype(mode_tag)::value;          // eps mode; NT_EPI_KL; NT_EPI_STORE
        constexpr bool edge = decltype(edge_tag)::value;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = bi + wm * WM + 16 * mi + Mma<T>::row(lane, r);
                bool live = true;
                if (MU) {
                    const int u = ep.frame_utt[row];
                    live = (u >= 0) && (ep.active[u] != 0);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const long col = bj + wn * WN + 16 * ni + i16;
                    if (MODE == NT_EPI_STORE) {
                        C[row * ldc + col] = acc[mi][ni][r];
                    } else {
                        const T h = ep.Hin[row * ep.ldh + col];
                        T out;
                        if (MODE == NT_EPI_KL) {
                            out = h * acc[mi][ni][r];
                        } else {
                            const T p = ep.P[row * ep.ldh + col];
                            out = mu_update<T>(h, p, acc[mi][ni][r], MODE, ep.eps, ep.l1);
                        }
                        if (edge && col >= ep.N) out = T(0);     // keep the zero padding exact (0/0 modes)
                        C[row * ldc + col] = live ? out : h;
                    }
                }
            }
        }
    };
    auto epilogue = [&](auto mode_tag) {
        if (MU && bj + BN > ep.N) epilogue_body(mode_tag, std::true_type{});
        else epilogue_body(mode_tag, std::false_type{});
    };
    if (!MU) {
        epilogue(std::integral_constant<int, NT_EPI_STORE>{});
    } else if (ep.kl) {
        epilogue(std::integral_constant<int, NT_EPI_KL>{});
    } else {
        switch (ep.eps_mode) {
            case EVC_EPS_ADD: epilogue(std::integral_constant<int, EVC_EPS_ADD>{}); break;
            case EVC_EPS_ZERO_REPLACE: epilogue(std::integral_constant<int, EVC_EPS_ZERO_REPLACE>{}); break;
            case EVC_EPS_CLAMP: epilogue(std::integral_constant<int, EVC_EPS_CLAMP>{}); break;
            default: epilogue(std::integral_constant<int, EVC_EPS_NONE>{}); break;
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, bool MU>
static hipError_t launch_nt(const T* L, int ldl, const T* R, int ldr, T* C, int ldc, int I, int J,
                            int Kd, const MuEpilogue<T>& ep, hipStream_t s, int splits = 1, long slab = 0, int jv = 0) {
    dim3 grid(J / BN, I / BM, splits), block((BM / WM) * (BN / WN) * 64);
    hipLaunchKernelGGL((k_gemm_nt<T, BM, BN, WM, WN, MU>), grid, block, 0, s, L, ldl, R, ldr, C, ldc,
                       Kd, ep, slab, jv > 0 ? jv : J);
    return hipGetLastError();
}

// C = sum_z part[z]   (fixed order: bitwise reproducible)
template <typename T>
__global__ __launch_bounds__(256) void k_sum_slabs(const T* __restrict__ part, long slab, int splits, long n,
                                                   T* __restrict__ C, const int* gate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || (gate && *gate == 0)) return;
    T acc = part[i];
    for (int z = 1; z < splits; ++z) acc += part[z * slab + i];
    C[i] = acc;
}

template <typename T>
hipError_t sum_slabs(const T* part, long slab, int splits, T* C, hipStream_t s, const int* gate) {
    hipLaunchKernelGGL((k_sum_slabs<T>), dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, s, part, slab, splits,
                       slab, C, gate);
    return hipGetLastError();
}

// Which generation of the contraction kernel serves a call: k_gemm2 for float32 (77 against 65 Tflop/s on the
// STFT flow); float64 stays on k_gemm_nt, which k_gemm2 does not beat (its 64-cycle MFMAs hide what k_gemm2
// removes).  Diagnostic builds (-DEVC_DIAG_GEMM_V1 / -DEVC_DIAG_GEMM2_F64) force one or the other for A/B timing;
// the shipped library reads nothing from the environment.
template <typename T> static bool use_gemm2() {
#if defined(EVC_DIAG_GEMM_V1)
    return false;
#elif defined(EVC_DIAG_GEMM2_F64)
    return true;
#else
    return sizeof(T) == 4;
#endif
}

template <typename T>
hipError_t gemm_nt(const T* L, int ldl, const T* R, int ldr, T* C, int ldc, int I, int J, int Kd,
                   hipStream_t s, T* scratch, size_t scratch_elems, int* splits_out, int j_valid, const int* gate) {
    if (splits_out) *splits_out = 0;
    if (I <= 0 || J <= 0) return hipSuccess;
    if (use_gemm2<T>() && gemm2_ok<T>(L, ldl, R, ldr, C, ldc, I, J, Kd)) {
        int dev = 0, cus = 0;
        if (scratch && (hipGetDevice(&dev) != hipSuccess ||
                        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess))
            cus = 0;
        return gemm2<T>(L, ldl, R, ldr, C, ldc, I, J, Kd, s, scratch, scratch_elems, splits_out, cus, j_valid, gate);
    }
    if (I % 64 || J % 64 || Kd % KS || Kd <= 0) return hipErrorInvalidValue;
    MuEpilogue<T> ep{};
    ep.gate = gate;
    // few output tiles (one utterance): 64x64 tiles put 2-4x more workgroups on the 256 CUs, and a long
    // contraction is additionally split over blockIdx.z into slabs of partial products (summed in order)
    const long blocks = (long)(I / 64) * (J / 64);
    if ((long)((I + 127) / 128) * (J / 64) < 256) {
        int splits = 1;
        const long slab = (long)I * ldc;
        if (scratch && ldc == J) {
            // The split (<= 8, not necessarily a divisor of the k-slabs) that is expected to finish first: a CU works
            // on up to four of these workgroups at once, one wavefront of each per SIMD, so a launch lasts about
            // (most workgroups on one CU) x (one workgroup's work ~ 1 / split), a little longer when few workgroups
            // share a CU (their latencies are less well covered: 0.6 / 0.8 / 0.9 / 0.95 of the matrix rate for 1..4).
            // C3's V = H Am^T (99 tiles): 7 ranges put at most 3 workgroups on a CU where 8 put 4 on some.  Every
            // workgroup keeps a worthwhile chunk (512 deep for long contractions, 64 for short ones such as the
            // 400-point DFTs of Griffin-Lim, whose 84 tiles would otherwise run 25 slabs each on a third of the CUs).
            int dev = 0, cus = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
                cus = 256;
            static const double eff[5] = {1.0, 0.6, 0.8, 0.9, 0.95};
            const int nslabs = Kd / KS, chunk = Kd >= 1024 ? 512 : 64;
            double best = 1e30;
            for (int sp = 1; sp <= 8 && Kd >= 1024; ++sp) {
                if (blocks * sp > 4L * cus || (sp > 1 && (nslabs / sp) * KS < chunk) || (size_t)sp * slab > scratch_elems) continue;
                const long per_cu = (blocks * sp + cus - 1) / cus;
                const double cost = (double)per_cu / sp / eff[per_cu];
                if (cost < best - 1e-9) { best = cost; splits = sp; }
            }
            // short contractions (measured on Griffin-Lim's: 12.7 ms per 300 iterations with the rule above against
            // 11.2): the largest split that divides the k-slabs evenly
            for (int sp = 2; sp <= 8 && Kd < 1024; ++sp)
                if (nslabs % sp == 0 && blocks * sp <= 1024 && Kd / sp >= chunk && (size_t)sp * slab <= scratch_elems)
                    splits = sp;
        }
        if (splits == 1) return launch_nt<T, 64, 64, 32, 32, false>(L, ldl, R, ldr, C, ldc, I, J, Kd, ep, s, 1, 0, j_valid);
        hipError_t e = launch_nt<T, 64, 64, 32, 32, false>(L, ldl, R, ldr, scratch, ldc, I, J, Kd, ep, s, splits, slab, j_valid);
        if (e != hipSuccess) return e;
        if (splits_out) {      // the caller's next kernel sums the slabs itself (scratch + z * I * ldc, z < splits)
            *splits_out = splits;
            return hipSuccess;
        }
        hipLaunchKernelGGL((k_sum_slabs<T>), dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, s, scratch, slab, splits,
                           slab, C, gate);
        return hipGetLastError();
    }
    if (I % 128) {      // short batches are padded to 64 frames only: 64-row blocks
        if (J % 128 == 0) return launch_nt<T, 64, 128, 32, 32, false>(L, ldl, R, ldr, C, ldc, I, J, Kd, ep, s, 1, 0, j_valid);
        return launch_nt<T, 64, 64, 32, 32, false>(L, ldl, R, ldr, C, ldc, I, J, Kd, ep, s, 1, 0, j_valid);
    }
    if (J % 128 == 0) return launch_nt<T, 128, 128, 64, 32, false>(L, ldl, R, ldr, C, ldc, I, J, Kd, ep, s, 1, 0, j_valid);
    return launch_nt<T, 128, 64, 32, 32, false>(L, ldl, R, ldr, C, ldc, I, J, Kd, ep, s, 1, 0, j_valid);
}

template <typename T>
hipError_t gemm_nt_mu(const T* L, int ldl, const T* R, int ldr, T* Hout, int I, int J, int Kd,
                      const MuEpilogue<T>& ep, hipStream_t s) {
    if (I <= 0 || J <= 0) return hipSuccess;
    if (use_gemm2<T>() && J % 128 == 0 && gemm2_ok<T>(L, ldl, R, ldr, Hout, ep.ldh, I, J, Kd) &&
        gemm2_ok<T>(ep.Hin, ep.ldh, ep.Hin, ep.ldh, ep.kl ? ep.Hin : ep.P, ep.ldh, I, J, Kd))
        return gemm2_mu<T>(L, ldl, R, ldr, Hout, I, J, Kd, ep, s);
    if (I % 64 || J % 128 || Kd % KS || Kd <= 0) return hipErrorInvalidValue;
    // Tile quantisation for one or two utterances: 128x128 tiles run in rounds of 256 (one per CU), 64x128
    // tiles in rounds of 512 (two per CU, half the work each; a trailing all-padding row tile leaves at
    // once).  C3 (768 x 8192): 384 full tiles = 2 rounds against 768 half tiles = 2 half rounds.
    const long b128 = (long)((I + 127) / 128) * (J / 128), b64 = (long)(I / 64) * (J / 128);
    const long t128 = 2 * ((b128 + 255) / 256), t64 = (b64 + 511) / 512;     // in half-tile rounds
    if (I % 128 || (I <= 2048 && t64 < t128))
        return launch_nt<T, 64, 128, 32, 32, true>(L, ldl, R, ldr, Hout, ep.ldh, I, J, Kd, ep, s);
    return launch_nt<T, 128, 128, 64, 32, true>(L, ldl, R, ldr, Hout, ep.ldh, I, J, Kd, ep, s);
}

// ------------------------------------------------------------------------------------------
// general strides, bounds checked (caller memory)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_gemm_strided(const T* __restrict__ L, long lsi, long lsk,
                                                      const T* __restrict__ R, long rsj, long rsk,
                                                      T* __restrict__ C, long csi, long csj, int I,
                                                      int J, int Kd) {
    typedef typename Mma<T>::acc_t acc_t;
    __shared__ T sL[KS][64 + 1];
    __shared__ T sR[KS][64 + 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1, i16 = lane & 15, q = lane >> 4;
    const long bi = (long)blockIdx.y * 64, bj = (long)blockIdx.x * 64;
    acc_t acc[2][2];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) acc[a][b] = acc_t{0, 0, 0, 0};

    // element (row, k) of the slab each thread stages: along k when k is the contiguous direction
    int lrow[4], lk[4], rrow[4], rk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (lsk == 1) { lk[e] = tid & 15; lrow[e] = (tid >> 4) + 16 * e; } else { lrow[e] = tid & 63; lk[e] = (tid >> 6) + 4 * e; }
        if (rsk == 1) { rk[e] = tid & 15; rrow[e] = (tid >> 4) + 16 * e; } else { rrow[e] = tid & 63; rk[e] = (tid >> 6) + 4 * e; }
    }
    T pl[4], pr[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long gi = bi + lrow[e], gk = k0 + lk[e];
            pl[e] = (gi < I && gk < Kd) ? L[gi * lsi + gk * lsk] : T(0);
            const long gj = bj + rrow[e], gk2 = k0 + rk[e];
            pr[e] = (gj < J && gk2 < Kd) ? R[gj * rsj + gk2 * rsk] : T(0);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < Kd; k0 += KS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sL[lk[e]][lrow[e]] = pl[e];
            sR[rk[e]][rrow[e]] = pr[e];
        }
        __syncthreads();
        if (k0 + KS < Kd) fetch(k0 + KS);      // the next slab's loads fly while this one is multiplied
#pragma unroll
        for (int st = 0; st < KS / 4; ++st) {
            const int kk = 4 * st + q;
            T a[2], b[2];
            for (int mi = 0; mi < 2; ++mi) a[mi] = sL[kk][wm * 32 + 16 * mi + i16];
            for (int ni = 0; ni < 2; ++ni) b[ni] = sR[kk][wn * 32 + 16 * ni + i16];
            for (int mi = 0; mi < 2; ++mi)
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = Mma<T>::mma(a[mi], b[ni], acc[mi][ni]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = bi + wm * 32 + 16 * mi + Mma<T>::row(lane, r);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const long col = bj + wn * 32 + 16 * ni + i16;
                if (row < I && col < J) C[row * csi + col * csj] = acc[mi][ni][r];
            }
        }
}

template <typename T>
hipError_t gemm_strided(const T* L, long lsi, long lsk, const T* R, long rsj, long rsk, T* C,
                        long csi, long csj, int I, int J, int Kd, hipStream_t s) {
    if (I <= 0 || J <= 0) return hipSuccess;
    dim3 grid((J + 63) / 64, (I + 63) / 64), block(256);
    hipLaunchKernelGGL((k_gemm_strided<T>), grid, block, 0, s, L, lsi, lsk, R, rsj, rsk, C, csi, csj,
                       I, J, Kd);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// synthesis Y = B H for a handful of bins (Mb <= 64) and a long contraction (N exemplars): tall and
// skinny, bound by the single pass over H.  One workgroup per 16 frames; its 8 wavefronts split N, each
// accumulating 16 x 16 output tiles straight from global memory, then a fixed-order LDS reduction.
// Caller strides.
//   H(t, n) = H[t hst + n hsn],  B(n, mb) = B[n bsn + mb bsm],  Y(t, mb) = Y[t yst + mb ysm]
// ------------------------------------------------------------------------------------------
constexpr int SYN_WAVES = 8;
constexpr int SYN_MAX_MB = 64;

template <typename T, int MT>
__global__ __launch_bounds__(SYN_WAVES * 64) void k_synth_skinny(const T* __restrict__ H, long hst, long hsn,
                                                                const T* __restrict__ B, long bsn, long bsm,
                                                                T* __restrict__ Y, long yst, long ysm, int T_,
                                                                int Mb, int N) {
    typedef typename Mma<T>::acc_t acc_t;
    __shared__ T red[SYN_WAVES][MT][4][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i16 = lane & 15, q = lane >> 4;
    const long t = (long)blockIdx.x * 16 + i16;
    acc_t acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = acc_t{0, 0, 0, 0};
    const int tiles = (N + 15) / 16;
    const int per = (tiles + SYN_WAVES - 1) / SYN_WAVES;
    const int j1 = min(tiles, (w + 1) * per);
    const bool t_ok = t < T_;
    // four exemplar tiles per round: their loads are all issued before the first MFMA (the kernel is a single
    // pass over H, so the only thing to hide is load latency)
    constexpr int UNR = 4;
    for (int j = w * per; j < j1; j += UNR) {
        T hv[UNR][4], bv[UNR][MT][4];
#pragma unroll
        for (int v = 0; v < UNR; ++v) {
            const long n0 = 16L * (j + v) + q;        // k-step s <-> exemplar n0 + 4 s (the 4 lane groups of a
                                                      // row read 4 consecutive exemplars per instruction)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const long n = n0 + 4 * s;
                const bool n_ok = (j + v < j1) && (n < N);
                // out-of-range lanes read element 0 and discard it: an unconditional load keeps the round's
                // loads in flight together (a predicated one becomes a branch with its own wait)
                const T hval = H[(t_ok && n_ok) ? t * hst + n * hsn : 0];
                hv[v][s] = (t_ok && n_ok) ? hval : T(0);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int mb = 16 * m + i16;
                    const bool ok = n_ok && mb < Mb;
                    const T bval = B[ok ? n * bsn + mb * bsm : 0];
                    bv[v][m][s] = ok ? bval : T(0);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < UNR; ++v)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = Mma<T>::mma(hv[v][s], bv[v][m][s], acc[m]);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[w][m][r][lane] = acc[m][r];
    __syncthreads();
    for (int e = tid; e < MT * 256; e += SYN_WAVES * 64) {
        const int m = e >> 8, r = (e >> 6) & 3, l = e & 63;
        T v = T(0);
#pragma unroll
        for (int ww = 0; ww < SYN_WAVES; ++ww) v += red[ww][m][r][l];
        const long tr = (long)blockIdx.x * 16 + Mma<T>::row(l, r);     // accumulator row = frame
        const int mb = 16 * m + (l & 15);
        if (tr < T_ && mb < Mb) Y[tr * yst + mb * ysm] = v;
    }
}

template <typename T>
hipError_t synth_skinny(const T* H, long hst, long hsn, const T* B, long bsn, long bsm, T* Y, long yst,
                        long ysm, int T_, int Mb, int N, hipStream_t s) {
    if (T_ <= 0 || Mb <= 0) return hipSuccess;
    if (Mb > SYN_MAX_MB) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((T_ + 15) / 16)), block(SYN_WAVES * 64);
    switch ((Mb + 15) / 16) {
        case 1: hipLaunchKernelGGL((k_synth_skinny<T, 1>), grid, block, 0, s, H, hst, hsn, B, bsn, bsm, Y, yst, ysm, T_, Mb, N); break;
        case 2: hipLaunchKernelGGL((k_synth_skinny<T, 2>), grid, block, 0, s, H, hst, hsn, B, bsn, bsm, Y, yst, ysm, T_, Mb, N); break;
        case 3: hipLaunchKernelGGL((k_synth_skinny<T, 3>), grid, block, 0, s, H, hst, hsn, B, bsn, bsm, Y, yst, ysm, T_, Mb, N); break;
        default: hipLaunchKernelGGL((k_synth_skinny<T, 4>), grid, block, 0, s, H, hst, hsn, B, bsn, bsm, Y, yst, ysm, T_, Mb, N); break;
    }
    return hipGetLastError();
}

#define EVC_INST(T)                                                                                  \
    template hipError_t gemm_nt<T>(const T*, int, const T*, int, T*, int, int, int, int, hipStream_t, T*, size_t, int*, int, const int*); \
    template hipError_t gemm_nt_mu<T>(const T*, int, const T*, int, T*, int, int, int,                \
                                      const MuEpilogue<T>&, hipStream_t);                            \
    template hipError_t sum_slabs<T>(const T*, long, int, T*, hipStream_t, const int*);                          \
    template hipError_t gemm_strided<T>(const T*, long, long, const T*, long, long, T*, long, long,  \
                                        int, int, int, hipStream_t);                                  \
    template hipError_t synth_skinny<T>(const T*, long, long, const T*, long, long, T*, long, long,  \
                                        int, int, int, hipStream_t);
EVC_INST(double)
EVC_INST(float)

}  // namespace evc
