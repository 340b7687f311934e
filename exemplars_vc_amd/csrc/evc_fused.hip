// Fused, persistent FACTORED multiplicative update for short dictionaries (M <= 32 bins,
// float64): the headline kernel of the C2 / C5 configurations (M = 25).
//
//   per iteration, per frame column h (N values):   v = A h            (M values)
//                                                    d = A^T v, p = A^T x
//                                                    h <- mu(h, p, d)
//
// One workgroup (8 wavefronts) owns 16*C frames for ALL iterations of a launch, so no
// inter-workgroup communication exists.  Each wavefront sweeps its share of the 16-exemplar
// tiles of the dictionary; for a tile it issues v_mfma_f64_16x16x4_f64 for
//     D = A_j^T V      (MSTEPS k-steps over the bins)
//     P = A_j^T X      (MSTEPS k-steps; recomputed, so P never exists in memory)
//     V' += A_j H'_j   (4 k-steps over the tile's 16 exemplars, per 16-bin tile)
// with the update in between, entirely in registers:
//   * the D/P accumulator layout (lane: frame = lane&15, exemplars 4(lane>>4)+r) is also the
//     B-operand layout of the V' product, so the updated tile feeds the next MFMA directly;
//   * the V' accumulator layout (lane: frame, bins (lane>>4)+4r) is the B-operand layout of the
//     next iteration's D product.
// Only V' crosses wavefronts: 8 partial sums are combined through LDS in a fixed order
// (bitwise reproducible).  H streams through HBM / Infinity Cache once per iteration (one read,
// one write) in a tile-packed layout whose every access is a full 1 KiB wave transaction; the
// dictionary fragments (two pre-packed orders, ~2 MB at N = 4096) are re-read from L2.
//
// Reference arithmetic: sklearn _nmf.py:526-556,612-631 / pymf nmf.py:66-70 re-associated as
// A^T (A H) (SURVEY.md section 7 "Which algebra").
#include "evc_internal.h"

namespace evc {

typedef double f64x2 __attribute__((ext_vector_type(2)));

__host__ __device__ inline int fused_msteps(int M) { return M <= 16 ? (M + 3) / 4 : 4 + (M - 16 + 3) / 4; }
// bin handled by k-step s for lane group q
__device__ __forceinline__ int bin_of(int s, int q) { return 16 * (s >> 2) + q + 4 * (s & 3); }

// ------------------------------------------------------------------------------------------
// packing (runs once per call)
// ------------------------------------------------------------------------------------------
// A1p[j][s][l] = A[bin_of(s, l>>4)][16 j + 4 (l&3) + ((l&15)>>2)]   (A-operand of D and P)
// A2p[j][u][r][l] = A[16 u + (l&15)][16 j + 4 (l>>4) + r]           (A-operand of V')
__global__ void k_pack_dict(const double* __restrict__ At, int ldA, int NT, int msteps, int mtiles,
                            double* __restrict__ A1p, double* __restrict__ A2p) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n1 = (long)NT * msteps * 64, n2 = (long)NT * mtiles * 4 * 64;
    if (gid < n1) {
        const int l = gid & 63, s = (gid >> 6) % msteps;
        const long j = (gid >> 6) / msteps;
        const int i = l & 15;
        const long n = 16 * j + 4 * (i & 3) + (i >> 2);
        A1p[gid] = At[n * ldA + bin_of(s, l >> 4)];
    } else if (gid < n1 + n2) {
        const long g = gid - n1;
        const int l = g & 63, r = (g >> 6) & 3, u = (g >> 8) % mtiles;
        const long j = (g >> 8) / mtiles;
        const long n = 16 * j + 4 * (l >> 4) + r;
        A2p[g] = At[n * ldA + 16 * u + (l & 15)];
    }
}

// Xp[tt][s][l] = X[bin_of(s, l>>4)][16 tt + (l&15)]                  (B-operand of P)
__global__ void k_pack_frames(const double* __restrict__ Xt, int ldx, long TTp, int msteps,
                              double* __restrict__ Xp) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= TTp * msteps * 64) return;
    const int l = gid & 63, s = (gid >> 6) % msteps;
    const long tt = (gid >> 6) / msteps;
    Xp[gid] = Xt[(16 * tt + (l & 15)) * ldx + bin_of(s, l >> 4)];
}

// Hp[tt][j][p][l] (2 doubles) = H[16 j + 4 (l>>4) + 2p + {0,1}][16 tt + (l&15)]
template <bool PACK>
__global__ void k_pack_h(double* __restrict__ Ht, int ldh, long TTp, int NT, f64x2* __restrict__ Hp) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= TTp * NT * 128) return;
    const int l = gid & 63, p = (gid >> 6) & 1;
    const long tile = gid >> 7;
    const long j = tile % NT, tt = tile / NT;
    double* src = Ht + (16 * tt + (l & 15)) * ldh + 16 * j + 4 * (l >> 4) + 2 * p;
    if (PACK) Hp[gid] = *reinterpret_cast<const f64x2*>(src);
    else *reinterpret_cast<f64x2*>(src) = Hp[gid];
}

// ------------------------------------------------------------------------------------------
// the persistent kernel
// ------------------------------------------------------------------------------------------
struct FusedArgs {
    const double* A1p;
    const double* A2p;
    const double* Xp;
    f64x2* Hp;
    double* Vp;              // [TTp][8][64] V in B-operand order, carried between launches
    double* err2;            // [T] per-frame squared residual (written when write_err)
    const int* frame_utt;
    const int* active;
    int NT, TT, N, T_;
    int iters;               // updates performed by this launch
    int first;               // 1: V is computed from H by a pre-pass, 0: V is loaded from Vp
    int write_err;
    int eps_mode;
    double eps, l1;
};

constexpr int FW = 8;  // wavefronts per workgroup

template <int MSTEPS, int C>
__global__ __launch_bounds__(FW * 64) void k_fused_mu(FusedArgs a) {
    constexpr int MT = MSTEPS > 4 ? 2 : 1;
    extern __shared__ double red[];      // [FW][C][MT][4][64]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = lane >> 4;
    const long tt0 = (long)blockIdx.x * C;

    // which of my frames take part in this launch
    bool live[C];
    int any = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const long t = 16 * (tt0 + c) + (lane & 15);
        int u = -1;
        if (tt0 + c < a.TT && t < a.T_) u = a.frame_utt[t];
        live[c] = (u >= 0) && (a.active[u] != 0);
        any |= live[c];
    }
    if (!__syncthreads_or(any)) return;

    double xf[C][MSTEPS], vf[C][MSTEPS];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int s = 0; s < MSTEPS; ++s) {
            const bool in = tt0 + c < a.TT;
            const long o = ((tt0 + c) * MSTEPS + s) * 64 + lane;
            xf[c][s] = in ? a.Xp[o] : 0.0;
            vf[c][s] = (in && !a.first) ? a.Vp[((tt0 + c) * 8 + s) * 64 + lane] : 0.0;
        }

    // combine the 8 wavefronts' partial V' (fixed order) into the B-operand registers
    auto reduce_v = [&](f64x4 (&vn)[C][MT]) {
        __syncthreads();                 // previous readers of `red` are done
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int u = 0; u < MT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[(((w * C + c) * MT + u) * 4 + r) * 64 + lane] = vn[c][u][r];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int s = 0; s < MSTEPS; ++s) {
                double acc = 0.0;
#pragma unroll
                for (int ww = 0; ww < FW; ++ww)
                    acc += red[(((ww * C + c) * MT + (s >> 2)) * 4 + (s & 3)) * 64 + lane];
                vf[c][s] = acc;
            }
    };

    if (a.first) {      // V = A H for the incoming activations
        f64x4 vn[C][MT];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int u = 0; u < MT; ++u) vn[c][u] = f64x4{0, 0, 0, 0};
        for (int j = w; j < a.NT; j += FW) {
            double a2[MT][4];
#pragma unroll
            for (int u = 0; u < MT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) a2[u][r] = a.A2p[(((long)j * MT + u) * 4 + r) * 64 + lane];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (tt0 + c >= a.TT) continue;
                const long hb = ((tt0 + c) * a.NT + j) * 128 + lane;
                const f64x2 h01 = a.Hp[hb], h23 = a.Hp[hb + 64];
                const double h[4] = {h01[0], h01[1], h23[0], h23[1]};
#pragma unroll
                for (int u = 0; u < MT; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) vn[c][u] = Mma<double>::mma(a2[u][r], h[r], vn[c][u]);
            }
        }
        reduce_v(vn);
    }

    for (int it = 0; it < a.iters; ++it) {
        f64x4 vn[C][MT];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int u = 0; u < MT; ++u) vn[c][u] = f64x4{0, 0, 0, 0};

        for (int j = w; j < a.NT; j += FW) {
            double a1[MSTEPS], a2[MT][4];
#pragma unroll
            for (int s = 0; s < MSTEPS; ++s) a1[s] = a.A1p[((long)j * MSTEPS + s) * 64 + lane];
#pragma unroll
            for (int u = 0; u < MT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) a2[u][r] = a.A2p[(((long)j * MT + u) * 4 + r) * 64 + lane];
            const int n0 = 16 * j + 4 * q;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (tt0 + c >= a.TT) continue;     // uniform
                const long hb = ((tt0 + c) * a.NT + j) * 128 + lane;
                const f64x2 h01 = a.Hp[hb], h23 = a.Hp[hb + 64];
                f64x4 d = {0, 0, 0, 0}, p = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < MSTEPS; ++s) {
                    d = Mma<double>::mma(a1[s], vf[c][s], d);
                    p = Mma<double>::mma(a1[s], xf[c][s], p);
                }
                double h[4] = {h01[0], h01[1], h23[0], h23[1]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double hn = mu_update<double>(h[r], p[r], d[r], a.eps_mode, a.eps, a.l1);
                    hn = (n0 + r < a.N) ? hn : 0.0;          // exemplar padding stays exactly 0
                    h[r] = live[c] ? hn : h[r];              // stopped utterances are frozen
                }
                a.Hp[hb] = f64x2{h[0], h[1]};
                a.Hp[hb + 64] = f64x2{h[2], h[3]};
#pragma unroll
                for (int u = 0; u < MT; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) vn[c][u] = Mma<double>::mma(a2[u][r], h[r], vn[c][u]);
            }
        }
        reduce_v(vn);
    }

    if (w == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (tt0 + c >= a.TT) continue;
            double e = 0.0;
#pragma unroll
            for (int s = 0; s < MSTEPS; ++s) {
                a.Vp[((tt0 + c) * 8 + s) * 64 + lane] = vf[c][s];
                const double df = xf[c][s] - vf[c][s];
                e += df * df;
            }
            if (a.write_err) {           // sum over the 4 lane groups holding one frame's bins
                e += __shfl_xor(e, 16, 64);
                e += __shfl_xor(e, 32, 64);
                const long t = 16 * (tt0 + c) + lane;
                if (lane < 16 && t < a.T_) a.err2[t] = e;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
bool fused_supported(int M, int N, int T_, int dtype) {
    return dtype == EVC_F64 && M >= 1 && M <= 32 && N >= 1 && T_ >= 1;
}

FusedLayout fused_layout(int M, int N, int T_) {
    FusedLayout f;
    f.msteps = fused_msteps(M);
    f.mtiles = M > 16 ? 2 : 1;
    f.NT = (N + 15) / 16;
    f.TT = (T_ + 15) / 16;
    f.TTp = round_up(f.TT, 4);
    f.a1 = (size_t)f.NT * f.msteps * 64;
    f.a2 = (size_t)f.NT * f.mtiles * 4 * 64;
    f.xp = (size_t)f.TTp * f.msteps * 64;
    f.hp = (size_t)f.TTp * f.NT * 256;
    f.vp = (size_t)f.TTp * 8 * 64;
    return f;
}

hipError_t fused_pack(const FusedLayout& f, const FusedBuffers& b, const double* At, int ldA,
                      const double* Xt, int ldx, double* Ht, int ldh, hipStream_t s) {
    {
        const long n = (long)f.a1 + (long)f.a2;
        hipLaunchKernelGGL(k_pack_dict, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, At, ldA, f.NT,
                           f.msteps, f.mtiles, b.A1p, b.A2p);
    }
    hipLaunchKernelGGL(k_pack_frames, dim3((unsigned)((f.xp + 255) / 256)), dim3(256), 0, s, Xt, ldx,
                       (long)f.TTp, f.msteps, b.Xp);
    const long nh = (long)f.TTp * f.NT * 128;
    hipLaunchKernelGGL((k_pack_h<true>), dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, s, Ht, ldh,
                       (long)f.TTp, f.NT, reinterpret_cast<f64x2*>(b.Hp));
    return hipGetLastError();
}

hipError_t fused_unpack(const FusedLayout& f, const FusedBuffers& b, double* Ht, int ldh, hipStream_t s) {
    const long nh = (long)f.TTp * f.NT * 128;
    hipLaunchKernelGGL((k_pack_h<false>), dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, s, Ht, ldh,
                       (long)f.TTp, f.NT, reinterpret_cast<f64x2*>(b.Hp));
    return hipGetLastError();
}

template <int MSTEPS, int C>
static hipError_t launch_fused(const FusedArgs& a, hipStream_t s) {
    constexpr int MT = MSTEPS > 4 ? 2 : 1;
    const size_t lds = (size_t)FW * C * MT * 4 * 64 * sizeof(double);
    const unsigned grid = (unsigned)((a.TT + C - 1) / C);
    if (lds > 48 * 1024) {   // per-launch, so that no mutable global state is kept
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_mu<MSTEPS, C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_fused_mu<MSTEPS, C>), dim3(grid), dim3(FW * 64), lds, s, a);
    return hipGetLastError();
}

template <int C>
static hipError_t dispatch_msteps(int msteps, const FusedArgs& a, hipStream_t s) {
    switch (msteps) {
        case 1: return launch_fused<1, C>(a, s);
        case 2: return launch_fused<2, C>(a, s);
        case 3: return launch_fused<3, C>(a, s);
        case 4: return launch_fused<4, C>(a, s);
        case 5: return launch_fused<5, C>(a, s);
        case 6: return launch_fused<6, C>(a, s);
        case 7: return launch_fused<7, C>(a, s);
        case 8: return launch_fused<8, C>(a, s);
        default: return hipErrorInvalidValue;
    }
}

int fused_pick_c(int T_) {
    const int TT = (T_ + 15) / 16;
    if (TT >= 1024) return 2;     // >= 512 workgroups of 32 frames: amortise the dictionary reads
    return 1;                     // few frames: as many workgroups as possible
}

hipError_t fused_iterate(const FusedLayout& f, const FusedBuffers& b, const UttState& u, int N, int T_,
                         int iters, int first, int write_err, double* err2, int eps_mode, double eps,
                         double l1, int c_override, hipStream_t s) {
    FusedArgs a;
    a.A1p = b.A1p; a.A2p = b.A2p; a.Xp = b.Xp; a.Hp = reinterpret_cast<f64x2*>(b.Hp); a.Vp = b.Vp;
    a.err2 = err2; a.frame_utt = u.frame_utt; a.active = u.active;
    a.NT = f.NT; a.TT = f.TT; a.N = N; a.T_ = T_;
    a.iters = iters; a.first = first; a.write_err = write_err;
    a.eps_mode = eps_mode; a.eps = eps; a.l1 = l1;
    const int C = c_override > 0 ? c_override : fused_pick_c(T_);
    switch (C) {
        case 1: return dispatch_msteps<1>(f.msteps, a, s);
        case 2: return dispatch_msteps<2>(f.msteps, a, s);
        case 4: return dispatch_msteps<4>(f.msteps, a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace evc
