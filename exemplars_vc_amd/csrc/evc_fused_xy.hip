// k_fused_xy (round 4): the all-resident fused FACTORED kernel (float64, M <= 32) with the exchange of the partial V'
// hidden behind matrix work of the SAME wavefronts, and two wavefronts per SIMD multiplying side by side.
//
// k_fused_all (evc_fused_all.hip) gives a frame tile to C members of 512 exemplars and alternates two members per CU:
// one sweeps while the other exchanges.  Measured (profiles/r04_unit_sched.md): a unit of 15 f64 MFMAs
// and its 33 vector instructions takes ~1260 cycles on a SIMD that one wavefront has to itself and ~1185 per unit when
// two wavefronts share the SIMD (the one's vector work and wait states run beside the other's MFMAs; placing the
// vector work between the MFMAs of ONE wavefront is slower than leaving it in one piece); a C2 step of k_fused_all is
// 1560 cycles per unit - a step lasts as long as the longer of sweep and exchange plus a workgroup barrier, and the
// SIMD belongs to one wavefront at a time.  Its one-member form (C1: no exchange, both halves sweeping together) runs
// at 1300.
//
// Here a workgroup is ONE member (4 wavefronts, one per SIMD; two workgroups per CU, free-running: no barrier couples
// them) of a group that owns TWO frame tiles, X and Y (32 frames).  A wavefront holds 4 exemplar tiles of X and the same
// 4 of Y - activations and numerators, 128 registers as before - so a member is 16 tiles = 256 exemplars and a group has
// NT / 16 members (16 at C2, 64 at C5).  The wavefronts sweep X and Y in turn, and the three phases of an exchange
// (reduce-scatter + all-gather as in k_fused_all: P1 publish the member's partial, P2 sum my slice of all partials and
// publish it, P3 fetch the summed slices) stand at fixed places of the OTHER tile's sweep:
//
//     X units 0-1 | P2(Y) | X units 2-3 | P1(X)  P3(Y) | Y units 0-1 | P2(X) | Y units 2-3 | P1(Y)  P3(X) | ...
//
// Every hand-off (P1 -> P2, P2 -> P3) has two units of sweep (~2 us with the other workgroup of the CU sweeping too)
// to travel; a wavefront that nevertheless finds its words missing polls (bounded), and meanwhile the SIMD belongs to
// the other workgroup's wavefront.  What a workgroup barrier did in k_fused_all (closing a step) is gone: the only
// barriers are among the 4 wavefronts of a member, around the LDS arrays of a phase.
//
// Exchange protocol, visibility and the bounded waits are k_fused_all's (C < 0 branch: ragged slices, any member
// count 2 .. 128; epoch bit in the lowest mantissa bit; two buffers per tile alternate by exchange parity; a wait that
// runs out raises coop_abort, everybody leaves, the host redoes the solve without exchange).  A buffer is rewritten two
// exchanges later: a member reaches P1(e + 2) only behind P3(e + 1), i.e. after every peer published its slice of e + 1,
// which in the peer's program order follows its P2(e) and P3(e) - so nobody still reads what is overwritten.
//
// Requirements (fused_xy_members): guarded eps mode, fast quotients, Frobenius loss, NT a multiple of 16, 2 .. 128
// members, and two workgroups resident per CU (launch_bounds(256, 2); the launcher checks the occupancy).
#include "evc_fused_common.h"

namespace evc {

constexpr int XW = 4;                  // wavefronts per workgroup (= member)
constexpr int XKT = 4;                 // exemplar tiles per wavefront and frame tile
constexpr int XTILES = XW * XKT;       // exemplar tiles per member
constexpr int XTHREADS = XW * 64;
constexpr unsigned XY_POLL_LIMIT = 1u << 17;

#ifdef EVC_XY_TIMING    // diagnostic build only (tools/ubench/fused_xy_bench.hip); no stamp executes in the library
#define XY_STAMP(i)                                                                                   \
    do {                                                                                              \
        if (a.dbg && lane == 0 && w == 0 && it < 64)                                                  \
            a.dbg[((long)blockIdx.x * 64 + it) * 16 + (i)] = __builtin_amdgcn_s_memtime();           \
    } while (0)
#else
#define XY_STAMP(i)
#endif

// x + (x of lane ^ o) for o = 1, 2, 4, 8 as DPP moves (quad_perm, row_half_mirror, row_mirror: once the quads / halves of
// a row hold one value each, the mirrors exchange exactly the partner's), else through the permute network.  Every
// lane of a 2o-lane group ends with the same value: floating-point addition is commutative.
template <int CTRL>
__device__ __forceinline__ double xy_dpp_add(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return x + __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double xy_group_sum(double x, int L) {     // L: lanes per group, a power of two
    if (L > 1) x = xy_dpp_add<0xB1>(x);          // quad_perm [1,0,3,2]
    if (L > 2) x = xy_dpp_add<0x4E>(x);          // quad_perm [2,3,0,1]
    if (L > 4) x = xy_dpp_add<0x141>(x);         // row_half_mirror
    if (L > 8) x = xy_dpp_add<0x140>(x);         // row_mirror
    if (L > 16) x += __shfl_xor(x, 16, 64);
    if (L > 32) x += __shfl_xor(x, 32, 64);
    return x;
}

// LDS-only barrier among the 4 wavefronts of the workgroup: the fragment prefetches (vector memory) stay in flight
__device__ __forceinline__ void xy_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// DREG: the denominator's start value l1 (+ pymf's eps) lives in registers (8 per lane) as SrcC of the first MFMA of
// every D chain.  Otherwise (DREG = false; the host picks it whenever it can) SrcC is the literal 0: either the start
// value is 0 (sklearn without L1: C2), or it rides in a spare bin of the last k-step - M is not a multiple of 4, the
// packed dictionary holds ones in bin M (fused_pack_dict, ones_bin) and that bin of V's B-operand image is set to the
// start value (a.spare_q = M & 3: the lane group of the k-step that holds bin M).
template <int MSTEPS, bool DREG>
__global__ __launch_bounds__(XTHREADS, 2) void k_fused_xy(FusedArgs a) {
    constexpr int MT = MSTEPS > 4 ? 2 : 1;
    constexpr int E = MT * 4 * 64;               // stride of one V image (accumulator order)
    constexpr int NE = MSTEPS * 64;              // elements of V actually used
    constexpr int MSP = (MSTEPS + 1) & ~1;       // k-steps padded to pairs in A1p
    __shared__ double s_red[XW * E];             // partial V' of every wavefront (the tile just swept)
    __shared__ double s_v[2][E];                 // V of X / Y, B-operand order
    __shared__ double s_x[2][E];                 // X of X / Y, B-operand order
    __shared__ f64x2 s_p[XW][XKT][2][64];        // numerator tiles of Y (wavefront-private: 32 registers per lane freed)
    __shared__ int s_rs[2 + 4][XTHREADS];        // a thread's words of the reduce-scatter (read in the phases only)
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int CR = a.coop_c;                     // members per group
    const int member = blockIdx.x % CR;
    const int g = blockIdx.x / CR;
    const double* __restrict__ A1p = a.A1p;
    const double* __restrict__ A2p = a.A2p;
    f64x2* __restrict__ Hp = a.Hp;
    const int NT = a.NT;
    const long tile0 = (long)member * XTILES + w;        // this wavefront's tiles: tile0 + XW * k
    const unsigned ul = (unsigned)lane;
    long sw = 0;                                 // opaque zero (see k_fused_all): tile addresses stay scalar
#ifdef EVC_XY_NOFRAG
    bool nofrag_go = false;
#endif

    auto load_a1 = [&](double (&a1)[MSTEPS], int k) {
#ifdef EVC_XY_NOFRAG     // diagnostic (tools/ubench): no fragment traffic after the first tile (wrong results, timing only)
        if (sw != 0x7fffffff && nofrag_go) return;
#endif
        const f64x2* t = reinterpret_cast<const f64x2*>(A1p + (tile0 + sw + XW * k) * (MSP * 64));
#pragma unroll
        for (int s = 0; s < MSTEPS; s += 2) {
            if (s + 1 < MSTEPS) {
                const f64x2 v = (t + (s >> 1) * 64)[ul];
                a1[s] = v[0];
                a1[s + 1] = v[1];
            } else {
                a1[s] = reinterpret_cast<const double*>(&(t + (s >> 1) * 64)[ul])[0];
            }
        }
    };
    auto load_a2 = [&](double (&a2)[MT][4], int k) {
#ifdef EVC_XY_NOFRAG
        if (sw != 0x7fffffff && nofrag_go) return;
#endif
        const f64x2* t = reinterpret_cast<const f64x2*>(A2p + (tile0 + sw + XW * k) * (MT * 256));
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f64x2 v = (t + (u * 2 + (r >> 1)) * 64)[ul];
                a2[u][r] = v[0];
                a2[u][r + 1] = v[1];
            }
    };

    const int mode = a.eps_mode;
    const double eps = a.eps;
    const double d0 = a.l1 + (mode == EVC_EPS_ADD ? a.eps : 0.0);
    const f64x4 dinit = {d0, d0, d0, d0};
    const f64x4 zero4 = {0, 0, 0, 0};
    const unsigned lo = fast_lo(mode, eps);

    // reduce-scatter with ragged slices: what this thread publishes and fetches, the same in every exchange
    // (fewer elements than threads - M <= 12 - : several threads carry the same element, word for word)
    const int e0 = tid % NE, e1 = tid + XTHREADS;
    const bool has1 = e1 < NE;
    const int rs_es = (NE + CR - 1) / CR;                            // elements per slice
    int rs_len = NE - member * rs_es;                                // of which valid in mine
    rs_len = rs_len < 0 ? 0 : (rs_len > rs_es ? rs_es : rs_len);
    {
        const int j0 = e0 / rs_es, j1 = (has1 ? e1 : e0) / rs_es;
        s_rs[0][tid] = (j0 * CR + member) * rs_es + (e0 - j0 * rs_es);
        s_rs[1][tid] = (j1 * CR + member) * rs_es + ((has1 ? e1 : e0) - j1 * rs_es);
    }
    // P2 (summing my slice of the CR partials) in registers: L lanes side by side hold the members of one element (two
    // members per lane beyond 64), 64 / L elements per wave instruction, the slots of an exchange dealt round-robin to
    // the 4 wavefronts: at most 4 words per thread (2 at C2 and C5), no LDS staging and no barrier.  The slice region is
    // [m][ES] row-major at (member * CR) * ES.
    int rsL = 1;
    while (rsL < CR && rsL < 64) rsL <<= 1;
    const int rs_mpl = (CR + rsL - 1) / rsL;                         // members per lane (1 or 2)
    const int rs_epi = 64 / rsL;                                     // elements per wave instruction
    const int rs_slots = (rs_es + rs_epi - 1) / rs_epi;
    const int rs_f = ((rs_slots + XW - 1) / XW) * rs_mpl;            // words per thread and exchange (<= 4)
    {
        for (int f = 0; f < 4; ++f) {
            const int slot = (f / rs_mpl) * XW + w, m = (lane & (rsL - 1)) + (f % rs_mpl) * rsL;
            const int el = slot * rs_epi + lane / rsL;
            s_rs[2 + f][tid] = (f < rs_f && slot < rs_slots && el < rs_len && m < CR)
                                   ? (member * CR + m) * rs_es + el : -1;
        }
    }
    const size_t n_wg = (size_t)a.groups * CR;
    unsigned seq[2] = {0, 0};                    // exchanges completed for X / Y
    if (tid == 0) s_fail = 0;
    __syncthreads();

    for (long pair = g; 2 * pair < a.TT; pair += a.groups) {
        const long ttx = 2 * pair;
        // ---- which of the two tiles this launch processes (uniform over the group: it depends on the tile only)
        bool valid[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const long tt = ttx + c;
            const long t = 16 * tt + (lane & 15);
            int u = -1;
            if (tt < a.TT && t < a.T_) u = a.frame_utt[t];
            // padding frames count as live; a tile beyond the batch, or one holding frames of a stopped utterance
            // (left to the general kernel, skip_all_live), is not processed
            const bool live = tt < a.TT && ((t >= a.T_) || ((u >= 0) && (a.active[u] != 0)));
            valid[c] = __syncthreads_and(live) != 0;
        }
        if (!valid[0] && !valid[1]) continue;

        HTile h[2][XKT];
        f64x4 p[XKT];                            // numerators of X; Y's live in s_p
        double a1[MSTEPS], a2[MT][4];
        // ---- start values, X and V images, numerator tiles
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (!valid[c]) continue;
            const long tt = ttx + c;
            double* xL = s_x[c];
            double* vL = s_v[c];
            if (a.first && a.init_const) {
                // first launch from the utterances' constants: H = h0 (0 in the padding), V = A H = h0 rowsum(A)
                auto h0_of = [&](int fr) {
                    const long t = 16 * tt + fr;
                    const int u = t < a.T_ ? a.frame_utt[t] : -1;
                    return u >= 0 ? a.h0[u] : 0.0;
                };
                for (int e = tid; e < E; e += XTHREADS) {
                    const int s = e >> 6, l = e & 63;
                    const bool in = s < MSTEPS;
                    xL[e] = in ? a.Xp[(tt * MSTEPS + s) * 64 + l] : 0.0;
                    vL[e] = in ? a.rsum[bin_of(s, l >> 4)] * h0_of(l & 15) : 0.0;
                }
                const double hv = h0_of(lane & 15);
#pragma unroll
                for (int k = 0; k < XKT; ++k) {
                    const long n0 = 16 * (tile0 + XW * k) + 4 * (lane >> 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[c][k][r] = n0 + r < a.N ? hv : 0.0;
                }
            } else {
                for (int e = tid; e < E; e += XTHREADS) {
                    const int s = e >> 6, l = e & 63;
                    const bool in = s < MSTEPS;
                    xL[e] = in ? a.Xp[(tt * MSTEPS + s) * 64 + l] : 0.0;
                    vL[e] = in ? a.Vp[(tt * 8 + s) * 64 + l] : 0.0;
                }
#pragma unroll
                for (int k = 0; k < XKT; ++k) {
                    const f64x2* t = Hp + (tt * NT + tile0 + XW * k) * 128;
                    const f64x2 h01 = t[ul], h23 = t[ul + 64];
                    h[c][k][0] = h01[0]; h[c][k][1] = h01[1]; h[c][k][2] = h23[0]; h[c][k][3] = h23[1];
                }
            }
        }
        __syncthreads();
        load_a1(a1, 0);
#pragma unroll
        for (int k = 0; k < XKT; ++k) {          // numerator tiles of both frame tiles from one set of fragments
            double x[MSTEPS];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (!valid[c]) continue;
#pragma unroll
                for (int s = 0; s < MSTEPS; ++s) x[s] = s_x[c][s * 64 + lane];
                f64x4 acc = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < MSTEPS; ++s) acc = Mma<double>::mma(a1[s], x[s], acc);
                if (c == 0) {
                    p[k] = acc;
                } else {
                    s_p[w][k][0][lane] = f64x2{acc[0], acc[1]};
                    s_p[w][k][1][lane] = f64x2{acc[2], acc[3]};
                }
            }
            load_a1(a1, (k + 1) % XKT);
        }

        // ---- the pieces of the schedule
        double v[MSTEPS];
        f64x4 vn[MT];
        bool ok = true;
        // one unit: exemplar tile K of frame tile C.  K == 0 opens the sweep (V image -> registers), the last unit
        // leaves the wavefront's partial V' in s_red
        auto unit = [&](auto ctag, auto ktag) {
            constexpr int C = decltype(ctag)::value, K = decltype(ktag)::value;
            if (K == 0) {
                asm volatile("" : "+s"(sw));
#pragma unroll
                for (int s = 0; s < MSTEPS; ++s) v[s] = s_v[C][s * 64 + lane];
                if (!DREG && (lane >> 4) == a.spare_q) v[MSTEPS - 1] = d0;     // (spare_q = -1: no spare bin in use)
#pragma unroll
                for (int u = 0; u < MT; ++u) vn[u] = f64x4{0, 0, 0, 0};
            }
            __builtin_amdgcn_sched_barrier(0);
            load_a2(a2, K);
            f64x4 pk;
            if (C == 0) {
                pk = p[K];
            } else {
                const f64x2 pa = s_p[w][K][0][lane], pb = s_p[w][K][1][lane];
                pk = f64x4{pa[0], pa[1], pb[0], pb[1]};
            }
            f64x4 d = DREG ? dinit : zero4;
#pragma unroll
            for (int s = 0; s < MSTEPS; ++s) d = Mma<double>::mma(a1[s], v[s], d);
            load_a1(a1, (K + 1) % XKT);          // the next unit's (or the next sweep's first) fragments
            __builtin_amdgcn_sched_barrier(0);
            mu_tile_guarded(h[C][K], pk, d, mode, eps, lo);
#pragma unroll
            for (int u = 0; u < MT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) vn[u] = Mma<double>::mma(a2[u][r], h[C][K][r], vn[u]);
            if (K == XKT - 1) {
#pragma unroll
                for (int u = 0; u < MT; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (u * 4 + r < MSTEPS) s_red[w * E + (u * 4 + r) * 64 + lane] = vn[u][r];
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto xb1_of = [&](int c) {
            return reinterpret_cast<long long*>(a.coop_buf) +
                   (((size_t)(c * 2 + (seq[c] & 1)) * n_wg) + (size_t)g * CR) * ALL_RS_STRIDE;
        };
        auto xb2_of = [&](int c) {
            return reinterpret_cast<long long*>(a.coop_buf) + ALL_SLICE_OFFSET +
                   ((size_t)(c * 2 + (seq[c] & 1)) * a.groups + g) * 512;
        };
        // poll four words (one memory round trip for all) until each carries the epoch of this exchange
        auto fetch4 = [&](const long long* p0, const long long* p1, const long long* p2, const long long* p3,
                          long long tag, long long& b0, long long& b1, long long& b2, long long& b3) {
            unsigned polls = 0;
            for (;;) {
                b0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b1 = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b2 = __hip_atomic_load(p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b3 = __hip_atomic_load(p3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(!ok || (((b0 ^ tag) | (b1 ^ tag) | (b2 ^ tag) | (b3 ^ tag)) & 1) == 0)) break;
                if (++polls > XY_POLL_LIMIT ||
                    ((polls & 63) == 0 &&
                     __hip_atomic_load(a.coop_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
                    ok = false;
                __builtin_amdgcn_s_sleep(1);
            }
        };
        // The fetches are split: a phase's words are REQUESTED one unit before they are looked at (a blocking fetch is a
        // memory round trip of 0.8 - 1.5 us under load during which the wavefront multiplies nothing: four of them per
        // iteration were half of its time, profiles/r04_xy_notes.md); only words that then still carry the old epoch are
        // polled for.  fb0..3: the words in flight (P2: up to four of my slice, P3: two of the summed slices).
        long long fb0 = 0, fb1 = 0, fb2 = 0, fb3 = 0;
        // P1: the member's partial (sum over its 4 wavefronts, fixed order) goes out, laid out [slice][member][ES]
        // (behind a barrier that follows the last unit: every wavefront's partial is in s_red)
        auto p1_publish = [&](int c) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int ww = 0; ww < XW; ++ww) {
                s0 += s_red[ww * E + e0];
                s1 += s_red[ww * E + (has1 ? e1 : e0)];
            }
            long long* xb1 = xb1_of(c);
            const long long tag = (seq[c] >> 1) & 1;
            const int rs_pub0 = s_rs[0][tid], rs_pub1 = s_rs[1][tid];
            __hip_atomic_store(xb1 + rs_pub0, (__double_as_longlong(s0) & ~1LL) | tag, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            if (has1)
                __hip_atomic_store(xb1 + rs_pub1, (__double_as_longlong(s1) & ~1LL) | tag, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        };
        // P2: my slice of all CR partials, summed in registers: the rsL lanes of a group hold the members of one
        // element (a fixed tree: whoever reduces a slice, the published sum is what every member receives)
        auto p2_issue = [&](int c) {
            const long long* xb1 = xb1_of(c);
            const long long* own = xb1 + s_rs[0][tid];               // (threads without a word watch their own)
            const int i0 = s_rs[2][tid], i1 = s_rs[3][tid], i2 = s_rs[4][tid], i3 = s_rs[5][tid];
            fb0 = __hip_atomic_load(i0 >= 0 ? xb1 + i0 : own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (rs_f > 1) fb1 = __hip_atomic_load(i1 >= 0 ? xb1 + i1 : own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (rs_f > 2) {
                fb2 = __hip_atomic_load(i2 >= 0 ? xb1 + i2 : own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                fb3 = __hip_atomic_load(i3 >= 0 ? xb1 + i3 : own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto p2_finish = [&](int c) {
            long long* xb1 = xb1_of(c);
            long long* xb2 = xb2_of(c);
            const long long tag = (seq[c] >> 1) & 1;
            const int i0 = s_rs[2][tid], i1 = s_rs[3][tid], i2 = s_rs[4][tid], i3 = s_rs[5][tid];
            if (rs_f < 2) fb1 = fb0;
            if (rs_f < 3) { fb2 = fb0; fb3 = fb0; }
            if (!__all((((fb0 ^ tag) | (fb1 ^ tag) | (fb2 ^ tag) | (fb3 ^ tag)) & 1) == 0)) {
                const long long* own = xb1 + s_rs[0][tid];
                fetch4(i0 >= 0 ? xb1 + i0 : own, i1 >= 0 ? xb1 + i1 : own, i2 >= 0 ? xb1 + i2 : own,
                       i3 >= 0 ? xb1 + i3 : own, tag, fb0, fb1, fb2, fb3);
            }
            if (!ok) s_fail = 1;
            auto val = [&](long long b, int i) { return i >= 0 ? __longlong_as_double(b & ~1LL) : 0.0; };
            // words f and f + 1 of a slot pair are the two members of a lane when a lane holds two (rs_mpl == 2)
            const double v0 = val(fb0, i0), v1 = val(fb1, i1), v2 = val(fb2, i2), v3 = val(fb3, i3);
            double t0, t1, t2 = 0.0, t3 = 0.0;
            int n_t;
            if (rs_mpl == 2) { t0 = v0 + v1; t1 = v2 + v3; n_t = rs_f / 2; }
            else { t0 = v0; t1 = v1; t2 = v2; t3 = v3; n_t = rs_f; }
            t0 = xy_group_sum(t0, rsL);
            if (n_t > 1) t1 = xy_group_sum(t1, rsL);
            if (n_t > 2) { t2 = xy_group_sum(t2, rsL); t3 = xy_group_sum(t3, rsL); }
            if ((lane & (rsL - 1)) == 0) {
                const int el0 = w * rs_epi + lane / rsL;             // element of this lane group in slot w
                auto pub = [&](double t, int k) {                    // k-th slot of this wavefront
                    const int el = el0 + k * XW * rs_epi;
                    if (k < n_t && el < rs_len)
                        __hip_atomic_store(xb2 + member * rs_es + el, (__double_as_longlong(t) & ~1LL) | tag,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                };
                pub(t0, 0); pub(t1, 1); pub(t2, 2); pub(t3, 3);
            }
        };
        // P3: the summed slices (element e of V' is word e) -> the tile's V image (the caller's barrier follows)
        auto p3_issue = [&](int c) {
            const long long* xb2 = xb2_of(c);
            fb0 = __hip_atomic_load(xb2 + e0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fb1 = __hip_atomic_load(xb2 + (has1 ? e1 : e0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto p3_write = [&](int c) {
            const long long* xb2 = xb2_of(c);
            const long long tag = (seq[c] >> 1) & 1;
            if (!__all((((fb0 ^ tag) | (fb1 ^ tag)) & 1) == 0))
                fetch4(xb2 + e0, xb2 + (has1 ? e1 : e0), xb2 + e0, xb2 + e0, tag, fb0, fb1, fb2, fb3);
            s_v[c][e0] = __longlong_as_double(fb0 & ~1LL);
            if (has1) s_v[c][e1] = __longlong_as_double(fb1 & ~1LL);
            if (!ok) s_fail = 1;
            ++seq[c];
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        static_assert(XKT == 4, "the schedule below is written for four units per sweep");

        // One body serves pairs and lone tiles (the odd tile at the end of a batch, or a neighbour left to the general
        // kernel: nothing hides its exchange); a second copy of the loop for them cost 40 registers of spills in both.
        const bool vx = valid[0], vy = valid[1];
        bool failed = false;
        for (int it = 0; it < a.iters; ++it) {
#ifdef EVC_XY_NOFRAG
            nofrag_go = it > 0;
#endif
            const bool ey = vy && it > 0;        // Y's exchange of the previous iteration is in flight
            // ---- X sweeps; Y's exchange (opened by P1(Y) at the end of the previous body) proceeds
            XY_STAMP(0);
            if (vx) unit(I0{}, I0{});
            if (ey) p2_issue(1);
            XY_STAMP(1);
            if (vx) unit(I0{}, I1{});
            XY_STAMP(2);
            if (ey) p2_finish(1);
            XY_STAMP(3);
            if (vx) unit(I0{}, I2{});
            if (ey) p3_issue(1);
            if (vx) unit(I0{}, I3{});
            XY_STAMP(4);
            if (ey) p3_write(1);
            xy_barrier();                        // s_red and s_v[Y] are complete
            if (vx) p1_publish(0);
            XY_STAMP(5);
            if (s_fail) { failed = true; break; }
            // ---- Y sweeps; X's exchange proceeds
            if (vy) unit(I1{}, I0{});
            if (vx) p2_issue(0);
            if (vy) unit(I1{}, I1{});
            XY_STAMP(6);
            if (vx) p2_finish(0);
            XY_STAMP(7);
            if (vy) unit(I1{}, I2{});
            if (vx) p3_issue(0);
            if (vy) unit(I1{}, I3{});
            XY_STAMP(8);
            if (vx) p3_write(0);
            xy_barrier();
            if (vy) p1_publish(1);
            XY_STAMP(9);
            if (s_fail) { failed = true; break; }
        }
        if (!failed && vy && a.iters > 0) {      // Y's last exchange
            p2_issue(1);
            p2_finish(1);
            p3_issue(1);
            p3_write(1);
            xy_barrier();
            if (s_fail) failed = true;
        }
        if (failed) {                            // a peer never showed up: void the launch, let everybody leave
            if (tid == 0) __hip_atomic_store(a.coop_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }

        // ---- results of the pair
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (!valid[c]) continue;
            const long tt = ttx + c;
#pragma unroll
            for (int k = 0; k < XKT; ++k) {
                f64x2* t = Hp + (tt * NT + tile0 + XW * k) * 128;
                t[ul] = f64x2{h[c][k][0], h[c][k][1]};
                t[ul + 64] = f64x2{h[c][k][2], h[c][k][3]};
            }
            if (a.Hx) {                          // last launch: the caller's H as well (no separate export pass)
                const long t = 16 * tt + (lane & 15);
                if (t < a.T_) {
#pragma unroll
                    for (int k = 0; k < XKT; ++k) {
                        const long n0 = 16 * (tile0 + XW * k) + 4 * (lane >> 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n0 + r < a.N) {
                                if (a.hx_frame_major) a.Hx[t * a.ldhx + n0 + r] = h[c][k][r];
                                else a.Hx[(n0 + r) * a.ldhx + t] = h[c][k][r];
                            }
                    }
                }
            }
            // carry V to the next launch; per-frame squared residual of the final activations
            if (member == 0) {                   // every member holds the same V: one writes it
                for (int e = tid; e < NE; e += XTHREADS) a.Vp[(tt * 8 + (e >> 6)) * 64 + (e & 63)] = s_v[c][e];
                if (a.write_err && w == 0) {
                    double e = 0.0;
#pragma unroll
                    for (int s = 0; s < MSTEPS; ++s) {
                        const double x = s_x[c][s * 64 + lane], vv = s_v[c][s * 64 + lane];
                        e += (x - vv) * (x - vv);
                    }
                    e += __shfl_xor(e, 16, 64);  // the 4 lane groups hold one frame's bins
                    e += __shfl_xor(e, 32, 64);
                    const long t = 16 * tt + lane;
                    if (lane < 16 && t < a.T_) a.err2[t] = e;
                }
            }
        }
        __syncthreads();                         // s_x / s_v are rewritten for the next pair
    }
}

template <int MSTEPS, bool DREG>
static hipError_t launch_xy(FusedArgs a, int n_cus, hipStream_t s) {
    int occ = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_fused_xy<MSTEPS, DREG>, XTHREADS, 0);
    if (e != hipSuccess) return e;
    if (occ < 2) return hipErrorInvalidValue;
    const int cr = a.coop_c;
    if (cr < 2 || cr > ALL_MAX_MEMBERS) return hipErrorInvalidValue;
    long resident = 2L * n_cus;                  // two workgroups (members) per CU
    if (resident > ALL_MAX_WGS / 2) resident = ALL_MAX_WGS / 2;     // the exchange buffers hold [2 tiles][2 parities]
    int groups = (int)(resident / cr);
    const int want = (a.TT + 1) / 2;             // pairs of frame tiles
    if (groups > want) groups = want;
    if (groups < 1) return hipErrorInvalidValue;
    a.groups = groups;
    // stale words must not carry the epoch bit of the first two exchanges (0): fill with ones
    const size_t n_wg = (size_t)groups * cr;
    if (4 * n_wg * ALL_RS_STRIDE > (size_t)ALL_SLICE_OFFSET || 4L * groups * 512 > ALL_SLICE_ELEMS)
        return hipErrorInvalidValue;
    e = hipMemsetAsync(a.coop_buf, 0xFF, sizeof(double) * 4 * n_wg * ALL_RS_STRIDE, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.coop_buf + ALL_SLICE_OFFSET, 0xFF, sizeof(double) * 4 * (size_t)groups * 512, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_fused_xy<MSTEPS, DREG>), dim3((unsigned)n_wg), dim3(XTHREADS), 0, s, a);
    return hipGetLastError();
}

// members per pair of frame tiles k_fused_xy would use for this problem, 0 if it does not apply
int fused_xy_members(int NT, int N, int eps_mode, int exact_div, int loss) {
    if (eps_mode == EVC_EPS_NONE || exact_div || loss != EVC_LOSS_FROBENIUS) return 0;
    if (NT % XTILES) return 0;
    const int c = NT / XTILES;
    return (c >= 2 && c <= ALL_MAX_MEMBERS) ? c : 0;
}

hipError_t fused_xy_launch(int msteps, const FusedArgs& a_in, int n_cus, hipStream_t s) {
    if (a_in.NT % XTILES || !a_in.coop_buf || !a_in.coop_abort) return hipErrorInvalidValue;
    FusedArgs a = a_in;
    const double d0 = a.l1 + (a.eps_mode == EVC_EPS_ADD ? a.eps : 0.0);
    // the start value of the denominators: nothing to add, or through the spare bin (ones in the packed dictionary),
    // or - M a multiple of 4 - in registers
    bool dreg = false;
    if (d0 == 0.0 || a.M <= 0) {
        a.spare_q = -1;
        dreg = d0 != 0.0;
    } else if (a.M % 4) {
        a.spare_q = a.M & 3;
    } else {
        a.spare_q = -1;
        dreg = true;
    }
#define EVC_XY_CASE(MS) case MS: return dreg ? launch_xy<MS, true>(a, n_cus, s) : launch_xy<MS, false>(a, n_cus, s)
    switch (msteps) {
        EVC_XY_CASE(1); EVC_XY_CASE(2); EVC_XY_CASE(3); EVC_XY_CASE(4);
        EVC_XY_CASE(5); EVC_XY_CASE(6); EVC_XY_CASE(7); EVC_XY_CASE(8);
        default: return hipErrorInvalidValue;
    }
#undef EVC_XY_CASE
}

}  // namespace evc
