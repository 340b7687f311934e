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
#include "evc_fused_common.h"

namespace evc {

// ------------------------------------------------------------------------------------------
// packing (runs once per call)
// ------------------------------------------------------------------------------------------
// Dictionary fragments, two consecutive k-steps per lane side by side (one 16-byte load per pair):
// A1p[j][s/2][l][s&1]      = A[bin_of(s, l>>4)][16 j + 4 (l&3) + ((l&15)>>2)]   (A-operand of D and P;
//                            s padded to an even count, the pad is zero)
// A2p[j][u][r/2][l][r&1]   = A[16 u + (l&15)][16 j + 4 (l>>4) + r]              (A-operand of V')
// At has n_rows rows (zero beyond the true N); exemplar slots from n_rows on - the tile count may be padded past
// the array, see fused_layout - are written as zeros, never read.
__global__ void k_pack_dict(const double* __restrict__ At, int ldA, int n_rows, int NT, int msteps, int mtiles,
                            double* __restrict__ A1p, double* __restrict__ A2p, int ones_bin) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int msp = (msteps + 1) & ~1;
    const long n1 = (long)NT * msp * 64, n2 = (long)NT * mtiles * 4 * 64;
    if (gid < n1) {
        const int e = gid & 1, l = (gid >> 1) & 63, s = 2 * (int)((gid >> 7) % (msp / 2)) + e;
        const long j = (gid >> 7) / (msp / 2);
        const int i = l & 15;
        const long n = 16 * j + 4 * (i & 3) + (i >> 2);
        const int bin = bin_of(s, l >> 4);
        if (A1p) A1p[gid] = (s < msteps && bin == ones_bin) ? 1.0 : ((s < msteps && n < n_rows) ? At[n * ldA + bin] : 0.0);
    } else if (gid < n1 + n2) {
        const long g = gid - n1;
        const int e = g & 1, l = (g >> 1) & 63, r = 2 * (int)((g >> 7) & 1) + e, u = (g >> 8) % mtiles;
        const long j = (g >> 8) / mtiles;
        const long n = 16 * j + 4 * (l >> 4) + r;
        if (A2p) A2p[g] = n < n_rows ? At[n * ldA + 16 * u + (l & 15)] : 0.0;
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

// Hp[tt][j][p][l] (2 doubles) = H[16 j + 4 (l>>4) + 2p + {0,1}][16 tt + (l&15)]: the activation tiles,
// in accumulator order.
// Hp <-> the caller's H (either layout, any strides): one wavefront per 16x16 tile, staged through
// a wave-private LDS tile so that the caller-side accesses are whole 128-byte rows.  Out-of-range
// frames / exemplars read as 0 and are not written.
//   TO_PACKED: caller -> Hp (import of a given H0);  else: Hp -> caller (export of the result)
template <bool TO_PACKED>
__global__ __launch_bounds__(256) void k_hp_io(double* __restrict__ H, long ldh, int frame_major, int T_,
                                                int N, f64x2* __restrict__ Hp, long n_tiles, int NT) {
    __shared__ double buf[4][16][17];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long tile = (long)blockIdx.x * 4 + w;
    if (tile >= n_tiles) return;
    const long tt = tile / NT, j = tile % NT;
    const long t0 = 16 * tt, n0 = 16 * j;
    double (*b)[17] = buf[w];                    // b[row][col]: row = caller's outer index within the tile
    const int pt = lane & 15, pq = lane >> 4;    // packed order: frame pt, exemplars 4 pq + r
    const int rr = lane >> 2, rc = (lane & 3) * 4;   // row order: row rr, columns rc .. rc+3
    if (TO_PACKED) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long t = frame_major ? t0 + rr : t0 + rc + e, n = frame_major ? n0 + rc + e : n0 + rr;
            double v = 0.0;
            if (t < T_ && n < N) v = frame_major ? H[t * ldh + n] : H[n * ldh + t];
            b[rr][rc + e] = v;
        }
        double h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = frame_major ? b[pt][4 * pq + r] : b[4 * pq + r][pt];
        Hp[(tile * 2 + 0) * 64 + lane] = f64x2{h[0], h[1]};
        Hp[(tile * 2 + 1) * 64 + lane] = f64x2{h[2], h[3]};
    } else {
        const f64x2 h01 = Hp[(tile * 2 + 0) * 64 + lane], h23 = Hp[(tile * 2 + 1) * 64 + lane];
        const double h[4] = {h01[0], h01[1], h23[0], h23[1]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (frame_major) b[pt][4 * pq + r] = h[r]; else b[4 * pq + r][pt] = h[r];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long t = frame_major ? t0 + rr : t0 + rc + e, n = frame_major ? n0 + rc + e : n0 + rr;
            if (t < T_ && n < N) {
                if (frame_major) H[t * ldh + n] = b[rr][rc + e]; else H[n * ldh + t] = b[rr][rc + e];
            }
        }
    }
}

// Hp <- per-utterance constant (EVC_INIT_SKLEARN / CONST), zero in the padding
__global__ __launch_bounds__(256) void k_fill_hp(f64x2* __restrict__ Hp, long n_tiles, int NT, int N, int T_,
                                                  UttState u) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n_tiles * 128) return;
    const int l = gid & 63, p = (gid >> 6) & 1;
    const long tile = gid >> 7;
    const long tt = tile / NT, j = tile % NT;
    const long t = 16 * tt + (l & 15);
    const long n = 16 * j + 4 * (l >> 4) + 2 * p;
    double v = 0.0;
    if (t < T_) { const int id = u.frame_utt[t]; if (id >= 0) v = u.h0[id]; }
    Hp[gid] = f64x2{n < N ? v : 0.0, n + 1 < N ? v : 0.0};
}

// Y (caller layout) <- Yp[tt][s][l] = Y[bin_of(s, l>>4)][16 tt + (l&15)]   (accumulator order of B H)
__global__ __launch_bounds__(256) void k_unpack_y(const double* __restrict__ Yp, long n_elems, int msteps, int Mb,
                                                   int T_, double* __restrict__ Y, long ldy, int frame_major) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n_elems) return;
    const int l = gid & 63, s = (gid >> 6) % msteps;
    const long tt = (gid >> 6) / msteps;
    const long t = 16 * tt + (l & 15);
    const int mb = bin_of(s, l >> 4);
    if (t >= T_ || mb >= Mb) return;
    const double v = Yp[(tt * 8 + s) * 64 + l];
    if (frame_major) Y[t * ldy + mb] = v; else Y[(long)mb * ldy + t] = v;
}

// Registers of one 16-exemplar dictionary tile (both operand orders) and of the C activation
// tiles that go with it.
template <int MSTEPS, int C> struct TileRegs {
    static constexpr int MT = MSTEPS > 4 ? 2 : 1;
    double a1[MSTEPS];
    double a2[MT][4];
    f64x2 h01[C], h23[C];
};

constexpr int NW = 8;   // wavefronts per workgroup

template <int MSTEPS, int C>
__global__ __launch_bounds__(NW * 64) void k_fused_mu(FusedArgs a) {
    constexpr int MT = MSTEPS > 4 ? 2 : 1;
    constexpr int MSP = (MSTEPS + 1) & ~1;       // k-steps padded to pairs in A1p
    constexpr int E = C * MT * 4 * 64;           // doubles in one V (accumulator order)
    extern __shared__ double lds[];
    double* red = lds;                           // [NW][E]   partial V' of every wavefront
    double* vL = lds + NW * E;                   // [C][MT*4][64]  V, B-operand order
    double* xL = vL + E;                         // [C][MT*4][64]  X, B-operand order
    double* rL = xL + E;                         // [C][MT*4][64]  X / max(V, eps) (KL numerator operand)
    const bool kl = a.loss == EVC_LOSS_KL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar loop and addresses
    const int q = lane >> 4;
    const long tt0 = (long)blockIdx.x * C;
    const double* __restrict__ A1p = a.A1p;
    const double* __restrict__ A2p = a.A2p;
    f64x2* __restrict__ Hp = a.Hp;
    const int NT = a.NT;

    // which of my frames take part in this launch
    bool live[C];
    int any = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const long t = 16 * (tt0 + c) + (lane & 15);
        int u = -1;
        if (tt0 + c < a.TT && t < a.T_) u = a.frame_utt[t];
        live[c] = (u >= 0) && (a.force_live || a.active[u] != 0);
        any |= live[c];
    }
    if (!__syncthreads_or(any)) return;
    if (a.skip_all_live) {    // C == 1 here: same frame <-> workgroup mapping as k_fused_res
        const long t = 16 * tt0 + (lane & 15);
        if (__syncthreads_and((t >= a.T_) || live[0])) return;
    }

    // stage X (and the carried V) in LDS
    for (int e = tid; e < E; e += NW * 64) {
        const int c = e / (MT * 256), s = (e >> 6) % (MT * 4), l = e & 63;
        const bool in = (tt0 + c < a.TT) && s < MSTEPS;
        const double x = in ? a.Xp[((tt0 + c) * MSTEPS + s) * 64 + l] : 0.0;
        const double v = (in && !a.first) ? a.Vp[((tt0 + c) * 8 + s) * 64 + l] : 0.0;
        xL[e] = x;
        vL[e] = v;
        rL[e] = x / (v < a.eps ? a.eps : v);
    }
    __syncthreads();

    typedef TileRegs<MSTEPS, C> Regs;
    auto load_tile = [&](Regs& R, int j, bool with_a1) {
        if (with_a1) {
#pragma unroll
            for (int s = 0; s < MSTEPS; s += 2) {
                const f64x2 v = reinterpret_cast<const f64x2*>(A1p)[((long)j * (MSP / 2) + (s >> 1)) * 64 + lane];
                R.a1[s] = v[0];
                if (s + 1 < MSTEPS) R.a1[s + 1] = v[1];
            }
        }
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f64x2 v = reinterpret_cast<const f64x2*>(A2p)[(((long)j * MT + u) * 2 + (r >> 1)) * 64 + lane];
                R.a2[u][r] = v[0];
                R.a2[u][r + 1] = v[1];
            }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (tt0 + c >= a.TT) { R.h01[c] = f64x2{0, 0}; R.h23[c] = f64x2{0, 0}; continue; }
            const long hb = ((tt0 + c) * NT + j) * 128 + lane;
            R.h01[c] = Hp[hb];
            R.h23[c] = Hp[hb + 64];
        }
    };

    // V' partials -> LDS -> each wavefront sums a slice over the NW partials in fixed order -> vL
    auto reduce_v = [&](f64x4 (&vn)[C][MT]) {
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int u = 0; u < MT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[w * E + ((c * MT + u) * 4 + r) * 64 + lane] = vn[c][u][r];
        __syncthreads();
        for (int e = w * 64 + lane; e < E; e += NW * 64) {
            double acc = 0.0;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) acc += red[ww * E + e];
            vL[e] = acc;
            if (kl) rL[e] = xL[e] / (acc < a.eps ? a.eps : acc);     // sklearn _nmf.py:572-576
        }
        __syncthreads();
    };

    if (a.first) {      // V = A H for the incoming activations
        f64x4 vn[C][MT];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int u = 0; u < MT; ++u) vn[c][u] = f64x4{0, 0, 0, 0};
        for (int j = w; j < NT; j += NW) {
            Regs R;
            load_tile(R, j, false);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double h[4] = {R.h01[c][0], R.h01[c][1], R.h23[c][0], R.h23[c][1]};
#pragma unroll
                for (int u = 0; u < MT; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) vn[c][u] = Mma<double>::mma(R.a2[u][r], h[r], vn[c][u]);
            }
        }
        reduce_v(vn);
    }

    // frames that are frozen / padded, or a ragged last exemplar tile, need per-element selects;
    // everything else runs the select-free sweep
    bool all_live = true;
#pragma unroll
    for (int c = 0; c < C; ++c) all_live = all_live && live[c];
    // (whole all-padding exemplar tiles exist once fused_layout pads NT to rounds of 8 wavefronts: under the
    // guarded modes 0 * 0 / guard keeps them at zero, the unguarded mode would turn them - and through V'
    // every frame of the workgroup - into NaN)
    const bool masked = !__syncthreads_and(all_live) || (a.N & 15) != 0 ||
                        (a.eps_mode == EVC_EPS_NONE && 16L * NT > a.N);
    const double eps = a.eps;

    auto sweep = [&](auto mul_first_tag, auto masked_tag, auto kl_tag) {
        constexpr bool MUL_FIRST = decltype(mul_first_tag)::value;
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr bool KL = decltype(kl_tag)::value;
        const int mode = a.eps_mode;
        // l1 (sklearn _nmf.py:615-617) and pymf's +eps (nmf.py:68) ride in the accumulator's start value
        const double d0 = a.l1 + (mode == EVC_EPS_ADD ? a.eps : 0.0);
        const f64x4 dinit = {d0, d0, d0, d0};
        const unsigned lo = fast_lo(mode, eps);
        for (int it = 0; it < a.iters; ++it) {
            f64x4 vn[C][MT];
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int u = 0; u < MT; ++u) vn[c][u] = f64x4{0, 0, 0, 0};
            for (int j = w; j < NT; j += NW) {
                Regs R;
                load_tile(R, j, true);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (MASKED && tt0 + c >= a.TT) continue;     // uniform
                    f64x4 d = dinit, p = {0, 0, 0, 0};
                    if (KL) {      // p <- (A_j / colsum)^T (X / max(V, eps)): the complete KL factor
#pragma unroll
                        for (int s = 0; s < MSTEPS; ++s)
                            p = Mma<double>::mma(R.a1[s], rL[(c * MT * 4 + s) * 64 + lane], p);
                    } else {
#pragma unroll
                        for (int s = 0; s < MSTEPS; ++s) {
                            d = Mma<double>::mma(R.a1[s], vL[(c * MT * 4 + s) * 64 + lane], d);
                            p = Mma<double>::mma(R.a1[s], xL[(c * MT * 4 + s) * 64 + lane], p);
                        }
                    }
                    HTile h;
                    h[0] = R.h01[c][0]; h[1] = R.h01[c][1]; h[2] = R.h23[c][0]; h[3] = R.h23[c][1];
                    if (MASKED) {
                        HTile hn = h;
                        if (KL) { for (int r = 0; r < 4; ++r) hn[r] *= p[r]; }
                        else if (a.exact_div) mu_tile<MUL_FIRST, true>(hn, p, d, mode, eps, lo);
                        else mu_tile<MUL_FIRST>(hn, p, d, mode, eps, lo);
                        const int n0 = 16 * j + 4 * q;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double v = (n0 + r < a.N) ? hn[r] : 0.0;   // exemplar padding stays 0
                            h[r] = live[c] ? v : h[r];                      // stopped utterances are frozen
                        }
                    } else if (KL) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[r] *= p[r];
                    } else if (a.exact_div) {
                        mu_tile<MUL_FIRST, true>(h, p, d, mode, eps, lo);
                    } else {
                        mu_tile<MUL_FIRST>(h, p, d, mode, eps, lo);
                    }
                    const long hb = ((tt0 + c) * NT + j) * 128 + lane;
                    Hp[hb] = f64x2{h[0], h[1]};
                    Hp[hb + 64] = f64x2{h[2], h[3]};
#pragma unroll
                    for (int u = 0; u < MT; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) vn[c][u] = Mma<double>::mma(R.a2[u][r], h[r], vn[c][u]);
                }
            }
            reduce_v(vn);
        }
    };
    const bool mul_first = a.eps_mode == EVC_EPS_ADD || a.eps_mode == EVC_EPS_NONE;
    if (kl) {
        if (masked) sweep(std::false_type{}, std::true_type{}, std::true_type{});
        else sweep(std::false_type{}, std::false_type{}, std::true_type{});
    } else if (mul_first) {
        if (masked) sweep(std::true_type{}, std::true_type{}, std::false_type{});
        else sweep(std::true_type{}, std::false_type{}, std::false_type{});
    } else {
        if (masked) sweep(std::false_type{}, std::true_type{}, std::false_type{});
        else sweep(std::false_type{}, std::false_type{}, std::false_type{});
    }

    // carry V to the next launch; per-frame squared residual of the final activations
    for (int e = tid; e < E; e += NW * 64) {
        const int c = e / (MT * 256), s = (e >> 6) % (MT * 4), l = e & 63;
        if (tt0 + c < a.TT && s < MSTEPS) a.Vp[((tt0 + c) * 8 + s) * 64 + l] = vL[e];
    }
    if (a.write_err && w < C && tt0 + w < a.TT) {
        const int c = w;
        double e = 0.0;
#pragma unroll
        for (int s = 0; s < MSTEPS; ++s) {
            const double x = xL[(c * MT * 4 + s) * 64 + lane], v = vL[(c * MT * 4 + s) * 64 + lane];
            e += kl ? kl_terms(x, v, a.eps) : (x - v) * (x - v);
        }
        e += __shfl_xor(e, 16, 64);      // the 4 lane groups hold one frame's bins
        e += __shfl_xor(e, 32, 64);
        const long t = 16 * (tt0 + c) + lane;
        if (lane < 16 && t < a.T_) a.err2[t] = e;
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
    f.M = M;
    // from 1024 exemplars on (k_fused_res: 8 wavefronts x >= 8 tiles) the tile count is padded to whole
    // rounds of the 8 wavefronts; padding exemplars have zero dictionary columns and zero activations, and
    // 0 * 0 / guard keeps them at zero under every guarded eps mode
    f.NT = N >= 1024 ? round_up(N, 128) / 16 : (N + 15) / 16;
    // k_fused_all wants whole members of 32 tiles (512 exemplars): pad to that when it costs <= 12.5 % (any N from
    // 4 members on; the padding exemplars cost a proportional share of one member's sweep)
    const int nt32 = round_up(f.NT, 32);
    if (nt32 * 100L <= f.NT * 112L + f.NT / 2) f.NT = nt32;
    f.TT = (T_ + 15) / 16;
    f.TTp = round_up(f.TT, 4);
    f.a1 = (size_t)f.NT * ((f.msteps + 1) & ~1) * 64;
    f.a2 = (size_t)f.NT * f.mtiles * 4 * 64;
    f.xp = (size_t)f.TTp * f.msteps * 64;
    f.hp = (size_t)f.TTp * f.NT * 256;
    f.vp = (size_t)f.TTp * 8 * 64;
    return f;
}

hipError_t fused_pack_dict(const FusedLayout& f, double* A1p, double* A2p, const double* At, int ldA, int n_rows,
                           hipStream_t s, int ones_bin) {
    const long n = (long)f.a1 + (long)f.a2;
    hipLaunchKernelGGL(k_pack_dict, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, At, ldA, n_rows, f.NT, f.msteps,
                       f.mtiles, A1p, A2p, ones_bin);
    return hipGetLastError();
}

hipError_t fused_pack_frames(const FusedLayout& f, double* Xp, const double* Xt, int ldx, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_frames, dim3((unsigned)((f.xp + 255) / 256)), dim3(256), 0, s, Xt, ldx,
                       (long)f.TTp, f.msteps, Xp);
    return hipGetLastError();
}

// H: the caller's activation matrix (frame_major: H[t*ldh+n], else H[n*ldh+t])
hipError_t fused_import_h(const FusedLayout& f, double* Hp, const double* H, long ldh, int frame_major, int T_,
                          int N, hipStream_t s) {
    const long tiles = (long)f.TTp * f.NT;
    hipLaunchKernelGGL((k_hp_io<true>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s,
                       const_cast<double*>(H), ldh, frame_major, T_, N, reinterpret_cast<f64x2*>(Hp), tiles, f.NT);
    return hipGetLastError();
}

hipError_t fused_export_h(const FusedLayout& f, const double* Hp, double* H, long ldh, int frame_major, int T_,
                          int N, hipStream_t s) {
    const long tiles = (long)f.TT * f.NT;
    hipLaunchKernelGGL((k_hp_io<false>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, H, ldh, frame_major,
                       T_, N, reinterpret_cast<f64x2*>(const_cast<double*>(Hp)), tiles, f.NT);
    return hipGetLastError();
}

// one wavefront per bin: lanes stride the exemplars, then a fixed shuffle tree
__global__ __launch_bounds__(64) void k_rowsum(const double* __restrict__ At, int ldA, int M, int N,
                                               double* __restrict__ rsum) {
    const int m = blockIdx.x, lane = threadIdx.x;
    double acc = 0.0;
    if (m < M)
        for (int n = lane; n < N; n += 64) acc += At[(long)n * ldA + m];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) rsum[m] = acc;
}

hipError_t fused_rowsum(const double* At, int ldA, int M, int N, double* rsum, hipStream_t s) {
    hipLaunchKernelGGL(k_rowsum, dim3(32), dim3(64), 0, s, At, ldA, M, N, rsum);
    return hipGetLastError();
}

hipError_t fused_fill_h(const FusedLayout& f, double* Hp, int N, int T_, const UttState& u, hipStream_t s) {
    const long tiles = (long)f.TTp * f.NT;
    hipLaunchKernelGGL(k_fill_hp, dim3((unsigned)((tiles * 128 + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<f64x2*>(Hp), tiles, f.NT, N, T_, u);
    return hipGetLastError();
}

template <int MSTEPS, int C>
static hipError_t launch_fused(const FusedArgs& a, hipStream_t s) {
    constexpr int MT = MSTEPS > 4 ? 2 : 1;
    constexpr int E = C * MT * 4 * 64;
    const size_t lds = (size_t)(NW + 3) * E * sizeof(double);
    const unsigned grid = (unsigned)((a.TT + C - 1) / C);
    if (lds > 48 * 1024) {   // per-launch, so that no mutable global state is kept
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_mu<MSTEPS, C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_fused_mu<MSTEPS, C>), dim3(grid), dim3(NW * 64), lds, s, a);
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

// frame tiles (of 16) per workgroup of the general kernel: 2 amortise the dictionary reads when
// there are plenty of workgroups, 1 otherwise (and always next to k_fused_res)
static hipError_t launch_general(const FusedLayout& f, const FusedArgs& a, int c_req, hipStream_t s) {
    const int C = a.skip_all_live ? 1 : (c_req > 0 ? c_req : (f.TT >= 1024 ? 2 : 1));
    switch (C) {
        case 1: return dispatch_msteps<1>(f.msteps, a, s);
        case 2: return dispatch_msteps<2>(f.msteps, a, s);
        default: return hipErrorInvalidValue;
    }
}

// c_req: 0 = automatic (k_fused_res where it applies), 1 / 2 = force the general kernel with that many
// frame tiles per workgroup (tests, A/B timing).
// all_live_known: no stopping rule is in force, so every utterance is active for the whole call.
// exact_div: correctly rounded quotients (general kernel only; see exact_div() in evc_fused_common.h).
hipError_t fused_iterate(const FusedLayout& f, const FusedBuffers& b, const UttState& u, int N, int T_,
                         int iters, int first, int write_err, double* err2, int eps_mode, double eps,
                         double l1, int c_req, int all_live_known, int loss, int exact_div, hipStream_t s) {
    FusedArgs a;
    a.A1p = b.A1p; a.A2p = b.A2p; a.Xp = b.Xp; a.Hp = reinterpret_cast<f64x2*>(b.Hp); a.Vp = b.Vp;
    a.err2 = err2; a.frame_utt = u.frame_utt; a.active = u.active;
    a.NT = f.NT; a.TT = f.TT; a.N = N; a.T_ = T_;
    a.iters = iters; a.first = first; a.write_err = write_err; a.skip_all_live = 0; a.force_live = 0;
    a.loss = loss; a.exact_div = exact_div;
    a.eps_mode = eps_mode; a.eps = eps; a.l1 = l1;
    a.coop_c = 1; a.coop_buf = nullptr; a.coop_cnt = nullptr; a.coop_abort = nullptr; a.groups = 0;
    a.init_const = 0; a.h0 = nullptr; a.rsum = nullptr; a.Hx = nullptr; a.ldhx = 0; a.hx_frame_major = 0;
    a.M = f.M; a.spare_q = -1; a.stagger_cycles = 0;
    const bool xy = c_req == 0 && b.xy_c >= 2 && b.coop_buf && b.coop_cnt &&
                    fused_xy_members(f.NT, N, eps_mode, exact_div, loss) == b.xy_c;
    const bool all_res = xy || (c_req == 0 && b.all_c >= 1 && b.coop_buf && b.coop_cnt &&
                                fused_all_members(f.NT, N, eps_mode, exact_div, loss) == b.all_c);
    const bool resident = all_res || (c_req == 0 && fused_res_supported(N, eps_mode, exact_div));
    if (!resident) return launch_general(f, a, c_req, s);
    if (first && all_res && b.init_const && iters > 0) {
        // k_fused_all starts from the utterances' constants itself (every utterance is active at the first launch)
        a.init_const = 1; a.h0 = u.h0; a.rsum = b.rsum;
    } else if (first) {          // V = A H (and the residual at init) by the general kernel's pre-pass
        FusedArgs p = a;
        p.iters = 0;
        p.write_err = (iters == 0) ? write_err : 0;
        hipError_t e = launch_general(f, p, 1, s);
        if (e != hipSuccess) return e;
        a.first = 0;
    }
    if (iters == 0) return hipSuccess;
    hipError_t e;
    if (all_res) {
        a.coop_c = xy ? b.xy_c : b.all_c; a.coop_buf = b.coop_buf; a.coop_cnt = b.coop_cnt;
        a.coop_abort = b.coop_cnt + COOP_MAX_TILES;
        if (all_live_known) { a.Hx = b.Hx; a.ldhx = b.ldhx; a.hx_frame_major = b.hx_frame_major; }
        e = xy ? fused_xy_launch(f.msteps, a, b.n_cus, s) : fused_all_launch(f.msteps, a, b.n_cus, s);
        a.coop_c = 1;            // the general kernel behind it takes no part in any exchange
        a.first = 0; a.init_const = 0; a.Hx = nullptr;
    } else {
        if (b.coop_c > 1 && b.coop_buf && b.coop_cnt) {
            a.coop_c = b.coop_c; a.coop_buf = b.coop_buf; a.coop_cnt = b.coop_cnt;
            a.coop_abort = b.coop_cnt + COOP_MAX_TILES;
        }
        e = fused_res_launch(f.msteps, a, s);
    }
    if (e != hipSuccess || all_live_known) return e;
    a.skip_all_live = 1;         // workgroups holding frames of stopped utterances
    return launch_general(f, a, 1, s);
}

// Y = B H straight from the packed activations: the V pre-pass with B's fragments in place of A's
// (HBM-bound on Hp), then a small unpack into the caller's layout.  fb.A2p must hold B's fragments.
hipError_t fused_synthesize(const FusedLayout& fB, const double* B2p, const double* Hp, double* Yp,
                            const UttState& u, int N, int T_, int Mb, double* Y, long ldy, int frame_major,
                            hipStream_t s) {
    FusedArgs a;
    a.A1p = B2p; a.A2p = B2p; a.Xp = Yp;        // A1p / Xp are not used by a pure pre-pass
    a.Hp = reinterpret_cast<f64x2*>(const_cast<double*>(Hp)); a.Vp = Yp;
    a.err2 = nullptr; a.frame_utt = u.frame_utt; a.active = u.active;
    a.NT = fB.NT; a.TT = fB.TT; a.N = N; a.T_ = T_;
    a.iters = 0; a.first = 1; a.write_err = 0; a.skip_all_live = 0; a.force_live = 1; a.loss = EVC_LOSS_FROBENIUS; a.exact_div = 0;
    a.Hx = nullptr; a.ldhx = 0; a.hx_frame_major = 0;
    a.init_const = 0; a.h0 = nullptr; a.rsum = nullptr; a.M = fB.M; a.spare_q = -1; a.stagger_cycles = 0;
    a.coop_c = 1; a.coop_buf = nullptr; a.coop_cnt = nullptr; a.coop_abort = nullptr; a.groups = 0;
    a.eps_mode = EVC_EPS_ADD; a.eps = 0; a.l1 = 0;
    hipError_t e = dispatch_msteps<1>(fB.msteps, a, s);
    if (e != hipSuccess) return e;
    const long n = (long)fB.TT * fB.msteps * 64;
    hipLaunchKernelGGL(k_unpack_y, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Yp, n, fB.msteps, Mb, T_,
                       Y, ldy, frame_major);
    return hipGetLastError();
}

}  // namespace evc
