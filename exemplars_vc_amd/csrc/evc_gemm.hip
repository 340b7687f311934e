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

namespace evc {

constexpr int KS = 16;  // k-slab depth staged per barrier

template <typename T, int E> struct VecOf { typedef T type __attribute__((ext_vector_type(E))); };

template <typename T, int BM, int BN, int WM, int WN, bool MU>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void k_gemm_nt(
    const T* __restrict__ L, int ldl, const T* __restrict__ R, int ldr, T* __restrict__ C, int ldc,
    int Kd, MuEpilogue<T> ep, long slab) {
    // split-K: blockIdx.z owns k in [z Kd, (z+1) Kd) and writes its partial product to slab z
    L += (long)blockIdx.z * Kd;
    R += (long)blockIdx.z * Kd;
    C += (long)blockIdx.z * slab;
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
    const int lane = tid & 63, w = tid >> 6;
    const int wm = w / NWN, wn = w % NWN;
    const int i16 = lane & 15, q = lane >> 4;
    const long bi = (long)blockIdx.y * BM, bj = (long)blockIdx.x * BN;

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

    const int nslab = Kd / KS;
    for (int sl = 0; sl < nslab; ++sl) {
        const int buf = sl & 1;
        const bool more = sl + 1 < nslab;
        if (more) {  // next slab's global loads fly while this slab feeds the matrix cores
            vl = *reinterpret_cast<const vecL*>(gL + (long)(sl + 1) * KS);
            vr = *reinterpret_cast<const vecR*>(gR + (long)(sl + 1) * KS);
        }
#pragma unroll
        for (int st = 0; st < KS / 4; ++st) {
            const int kk = 4 * st + q;
            T a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = sL[buf][kk][wm * WM + 16 * mi + i16];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = sR[buf][kk][wn * WN + 16 * ni + i16];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = Mma<T>::mma(a[mi], b[ni], acc[mi][ni]);
        }
        if (more) {
#pragma unroll
            for (int e = 0; e < EL; ++e) sL[buf ^ 1][lk + e][lrow] = vl[e];
#pragma unroll
            for (int e = 0; e < ER; ++e) sR[buf ^ 1][rk + e][rrow] = vr[e];
        }
        __syncthreads();
    }

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
                if (MU) {
                    const T h = ep.Hin[row * ep.ldh + col];
                    T out = h;
                    if (live) {
                        if (ep.kl) {
                            out = h * acc[mi][ni][r];
                        } else {
                            const T p = ep.P[row * ep.ldh + col];
                            out = mu_update<T>(h, p, acc[mi][ni][r], ep.eps_mode, ep.eps, ep.l1);
                        }
                        if (col >= ep.N) out = T(0);   // keep the zero padding exact (0/0 modes)
                    }
                    C[row * ldc + col] = out;
                } else {
                    C[row * ldc + col] = acc[mi][ni][r];
                }
            }
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, bool MU>
static hipError_t launch_nt(const T* L, int ldl, const T* R, int ldr, T* C, int ldc, int I, int J,
                            int Kd, const MuEpilogue<T>& ep, hipStream_t s, int splits = 1, long slab = 0) {
    dim3 grid(J / BN, I / BM, splits), block((BM / WM) * (BN / WN) * 64);
    hipLaunchKernelGGL((k_gemm_nt<T, BM, BN, WM, WN, MU>), grid, block, 0, s, L, ldl, R, ldr, C, ldc,
                       Kd / splits, ep, slab);
    return hipGetLastError();
}

// C = sum_z part[z]   (fixed order: bitwise reproducible)
template <typename T>
__global__ __launch_bounds__(256) void k_sum_slabs(const T* __restrict__ part, long slab, int splits, long n,
                                                   T* __restrict__ C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    T acc = part[i];
    for (int z = 1; z < splits; ++z) acc += part[z * slab + i];
    C[i] = acc;
}

template <typename T>
hipError_t gemm_nt(const T* L, int ldl, const T* R, int ldr, T* C, int ldc, int I, int J, int Kd,
                   hipStream_t s, T* scratch, size_t scratch_elems) {
    if (I <= 0 || J <= 0) return hipSuccess;
    if (I % 128 || J % 64 || Kd % KS || Kd <= 0) return hipErrorInvalidValue;
    MuEpilogue<T> ep{};
    // few output tiles (one utterance): 64x64 tiles put 2-4x more workgroups on the 256 CUs, and a long
    // contraction is additionally split over blockIdx.z into slabs of partial products (summed in order)
    const long blocks = (long)(I / 64) * (J / 64);
    if ((long)(I / 128) * (J / 64) < 256) {
        int splits = 1;
        const long slab = (long)I * ldc;
        if (scratch && ldc == J) {
            while (splits < 8 && blocks * splits * 2 <= 512 && (Kd / (splits * 2)) % KS == 0 && Kd / (splits * 2) >= 512 &&
                   (size_t)(splits * 2) * slab <= scratch_elems)
                splits *= 2;
        }
        if (splits == 1) return launch_nt<T, 64, 64, 32, 32, false>(L, ldl, R, ldr, C, ldc, I, J, Kd, ep, s);
        hipError_t e = launch_nt<T, 64, 64, 32, 32, false>(L, ldl, R, ldr, scratch, ldc, I, J, Kd, ep, s, splits, slab);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_sum_slabs<T>), dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, s, scratch, slab, splits,
                           slab, C);
        return hipGetLastError();
    }
    if (J % 128 == 0) return launch_nt<T, 128, 128, 64, 32, false>(L, ldl, R, ldr, C, ldc, I, J, Kd, ep, s);
    return launch_nt<T, 128, 64, 32, 32, false>(L, ldl, R, ldr, C, ldc, I, J, Kd, ep, s);
}

template <typename T>
hipError_t gemm_nt_mu(const T* L, int ldl, const T* R, int ldr, T* Hout, int I, int J, int Kd,
                      const MuEpilogue<T>& ep, hipStream_t s) {
    if (I <= 0 || J <= 0) return hipSuccess;
    if (I % 128 || J % 128 || Kd % KS || Kd <= 0) return hipErrorInvalidValue;
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

    for (int k0 = 0; k0 < Kd; k0 += KS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int row, k;
            if (lsk == 1) { k = tid & 15; row = (tid >> 4) + 16 * e; } else { row = tid & 63; k = (tid >> 6) + 4 * e; }
            const long gi = bi + row, gk = k0 + k;
            sL[k][row] = (gi < I && gk < Kd) ? L[gi * lsi + gk * lsk] : T(0);
            if (rsk == 1) { k = tid & 15; row = (tid >> 4) + 16 * e; } else { row = tid & 63; k = (tid >> 6) + 4 * e; }
            const long gj = bj + row, gk2 = k0 + k;
            sR[k][row] = (gj < J && gk2 < Kd) ? R[gj * rsj + gk2 * rsk] : T(0);
        }
        __syncthreads();
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

#define EVC_INST(T)                                                                                  \
    template hipError_t gemm_nt<T>(const T*, int, const T*, int, T*, int, int, int, int, hipStream_t, T*, size_t); \
    template hipError_t gemm_nt_mu<T>(const T*, int, const T*, int, T*, int, int, int,                \
                                      const MuEpilogue<T>&, hipStream_t);                            \
    template hipError_t gemm_strided<T>(const T*, long, long, const T*, long, long, T*, long, long,  \
                                        int, int, int, hipStream_t);
EVC_INST(double)
EVC_INST(float)

}  // namespace evc
