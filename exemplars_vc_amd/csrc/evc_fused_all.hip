// All-resident variant of the fused FACTORED kernel (float64, M <= 32): nothing streams.
//
// k_fused_res keeps half of a 16-frame block's activations in registers and recomputes the numerator
// tiles P = A_j^T X in every iteration (7 of the 22 MFMAs of a unit at M = 25), because one workgroup
// owns all N exemplars of its frames: at N = 4096 that is 512 KiB of H plus 512 KiB of P per 16 frames,
// more than a CU holds.  Here a frame tile is shared by C "members" (a group), each owning 32 exemplar
// tiles (512 exemplars): 4 wavefronts x 8 tiles, whose activations AND numerators stay in VGPRs (128 of
// the 256 registers of a wavefront) for all iterations of a frame tile.  A unit is then 15 MFMAs
// (D = A_j^T V: 7, V' += A_j H'_j: 8) instead of 20.5 on average, and the only memory traffic of the loop
// is the dictionary fragments (L2) and the exchange of the partial V' between the members of a group.
//
// The exchange costs a memory round trip per iteration (measured 3.5 us against a 3.8 us sweep), so it has
// to hide behind matrix work of ANOTHER frame tile.  Two free-running 4-wavefront workgroups per CU do not
// do that: coupled through their peers they fall into lock step - both sweep together (sharing the matrix
// pipes), then both wait together (measured: 10.9 us per iteration, 7.3 + 3.5).  So one workgroup of 8
// wavefronts holds TWO members, of two different groups: wavefronts 0-3 ("half" 0) and 4-7 (half 1; wave
// w and w + 4 share a SIMD).  They alternate by construction: in every step one half sweeps while the other
// exchanges, and a workgroup barrier closes the step, so a SIMD's matrix pipe always belongs to exactly one
// wavefront and the exchange of one frame tile always runs beside the sweep of the other.
//
//   step      0        1        2        3      ...   2K-1      2K
//   half 0  sweep 0   exch    sweep 1   exch    ...   exch       -
//   half 1    -      sweep 0   exch    sweep 1  ...  sweep K-1  exch
//
// The grid is persistent: one workgroup per CU, G = 2 floor(#CUs / C) groups walk the frame tiles g, g + G,
// ... (all members of a group walk the same list in lock step), so every member of every group is resident
// for the whole launch whatever the batch size.
//
// Exchange: after the sweep the half's wavefronts leave their partial V' in LDS; in the exchange step every
// thread sums its elements over the 4 wavefronts in fixed order, publishes them with agent-scope relaxed
// atomic stores (sc1: past the non-coherent per-XCD L2s) and fetches the same elements of its peers; the
// lowest mantissa bit of a word is the epoch of the buffer, so data and arrival travel in one word (each
// wavefront first watches one word per peer, so that the bulk fetch normally succeeds at once).  Every
// member sums the C words of an element in member order: all obtain the bitwise identical V'.  Two buffers
// alternate by exchange parity (a member overwrites a buffer two exchanges later, after it has read every
// peer's words of the exchange in between, which the peer published after reading this member's words of
// the exchange before).  Every wait is bounded; a member that gives up raises coop_abort, everybody leaves,
// and the host redoes the solve without inter-workgroup communication (evc_api.hip).
//
// Requirements (the host checks them, fused_all_members): guarded eps mode, fast quotients, Frobenius
// loss, NT a multiple of 32.  Frame tiles whose frames are not all live are left to the general kernel
// (skip_all_live), like in k_fused_res.
#include "evc_fused_common.h"

namespace evc {

constexpr int AW = 4;                  // wavefronts per half
constexpr int AKT = 8;                 // exemplar tiles per wavefront, all register-resident
constexpr int ATILES = AW * AKT;       // exemplar tiles per member
constexpr int ATHREADS = 2 * AW * 64;  // two halves per workgroup
constexpr unsigned ALL_POLL_LIMIT = 1u << 17;

#ifdef EVC_ALL_TIMING   // diagnostic build only (tools/ubench/fused_all_bench.hip); no stamp executes in the library
#define EVC_STAMP(i)                                                                                             \
    do {                                                                                                         \
        if (a.dbg && tt0 == g - half && lane == 0 && w == 0)                                                     \
            a.dbg[(((long)blockIdx.x * 2 + half) * (2 * a.iters + 1) + step) * 4 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define EVC_STAMP(i)
#endif

// dynamic LDS of k_fused_all<MSTEPS, .>: the D-product fragments of the member's tiles when they fit beside
// the exchange state (MSTEPS <= 7, i.e. M <= 28: 112 KiB + 42 KiB of the CU's 160 KiB)
template <int MSTEPS> struct AllLds {
    static constexpr size_t state_bytes = (size_t)(2 * AW + 4) * MSTEPS * 64 * sizeof(double) + 16;
    static constexpr size_t a1_full = (size_t)ATILES * MSTEPS * 64 * sizeof(double);
    static constexpr bool A1 = a1_full + state_bytes <= 160 * 1024;
    static constexpr size_t a1_bytes = A1 ? a1_full : 0;
    static constexpr size_t bytes = a1_bytes + state_bytes;
};

// DZERO: the denominator accumulators start at the inline constant 0 (l1 == 0 and no additive eps: the reference
// script's case) instead of at a value held in 8 VGPRs.
template <int MSTEPS, int C, bool DZERO>
__global__ __launch_bounds__(ATHREADS, 2) void k_fused_all(FusedArgs a) {
    constexpr int MT = MSTEPS > 4 ? 2 : 1;
    constexpr int NE = MSTEPS * 64;              // elements of V (accumulator order = B-operand order)
    constexpr int MSP = (MSTEPS + 1) & ~1;       // k-steps padded to pairs in A1p
    typedef AllLds<MSTEPS> L;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* const s_a1 = reinterpret_cast<double*>(smem);                 // [ATILES][MSTEPS * 64] (L::A1)
    double* const s_red = reinterpret_cast<double*>(smem + L::a1_bytes);  // [2][AW][NE] partial V' per wavefront
    double* const s_v = s_red + 2 * AW * NE;                              // [2][NE] V, B-operand order
    double* const s_sum = s_v + 2 * NE;                                   // [2][NE] the member's partial V'
    unsigned* const s_cnt = reinterpret_cast<unsigned*>(s_sum + 2 * NE);  // [2] wavefronts that have delivered
    int* const s_fail = reinterpret_cast<int*>(s_cnt + 2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = w8 >> 2, w = w8 & 3, th = tid & (AW * 64 - 1);
    const int member = blockIdx.x % C;
    const int g = 2 * (blockIdx.x / C) + half;           // the group this half is a member of
    double* const red = s_red + half * (AW * NE);
    double* const vL = s_v + half * NE;
    double* const sumL = s_sum + half * NE;
    const double* __restrict__ A1p = a.A1p;
    const double* __restrict__ A2p = a.A2p;
    f64x2* __restrict__ Hp = a.Hp;
    const int NT = a.NT;
    const long tile0 = (long)member * ATILES + w;       // this wavefront's tiles: tile0 + AW * k
    const unsigned ul = (unsigned)lane;
    // Tile base addresses are wave-uniform (SGPR base + unsigned per-lane offset).  `sw` is an opaque zero
    // refreshed once per sweep so that the 16 tile addresses are re-derived with scalar adds instead of being
    // hoisted out of the loops as per-lane 64-bit addresses (32 VGPRs).
    long sw = 0;

    // The D-product fragments of the member's 32 tiles live in LDS for the whole launch when they fit (both
    // halves of the workgroup are the same member of their groups, so they share them): with one wavefront per
    // SIMD sweeping, nobody covers an L2 round trip (measured: 1540 cycles per unit against 960 of MFMAs with
    // both fragment sets streamed from L2).  Per tile: MSTEPS/2 blocks of [lane][2 k-steps] + one [lane] block.
    if (L::A1) {
        constexpr int PAIRS = MSTEPS / 2, TS = MSTEPS * 64;
        const double* src = A1p + (long)member * ATILES * (MSP * 64);
        for (int i = tid; i < ATILES * TS; i += ATHREADS) {
            const int t = i / TS, r = i % TS;
            s_a1[i] = src[t * (MSP * 64) + (r < PAIRS * 128 ? r : PAIRS * 128 + 2 * (r - PAIRS * 128))];
        }
    }
    auto load_a1 = [&](double (&a1)[MSTEPS], int k) {
        if (L::A1) {
            const double* t = s_a1 + (w + AW * k) * (MSTEPS * 64);
#pragma unroll
            for (int s = 0; s + 1 < MSTEPS; s += 2) {
                const f64x2 v = reinterpret_cast<const f64x2*>(t)[(s >> 1) * 64 + lane];
                a1[s] = v[0];
                a1[s + 1] = v[1];
            }
            if (MSTEPS & 1) a1[MSTEPS - 1] = t[(MSTEPS / 2) * 128 + lane];
            return;
        }
        const f64x2* t = reinterpret_cast<const f64x2*>(A1p + (tile0 + sw + AW * k) * (MSP * 64));
#pragma unroll
        for (int s = 0; s < MSTEPS; s += 2) {
            if (s + 1 < MSTEPS) {
                const f64x2 v = t[(s >> 1) * 64 + ul];
                a1[s] = v[0];
                a1[s + 1] = v[1];
            } else {
                a1[s] = reinterpret_cast<const double*>(&t[(s >> 1) * 64 + ul])[0];
            }
        }
    };
    auto load_a2 = [&](double (&a2)[MT][4], int k) {
#ifdef EVC_DBG_NOA2      // diagnostic (tools/ubench): no fragment loads inside the sweep
        if (k > 0) return;
#endif
        const f64x2* t = reinterpret_cast<const f64x2*>(A2p + (tile0 + sw + AW * k) * (MT * 256));
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f64x2 v = t[(u * 2 + (r >> 1)) * 64 + ul];
                a2[u][r] = v[0];
                a2[u][r + 1] = v[1];
            }
    };

    const int mode = a.eps_mode;
    const double eps = a.eps;
    const double d0 = DZERO ? 0.0 : a.l1 + (mode == EVC_EPS_ADD ? a.eps : 0.0);
    const f64x4 dinit = {d0, d0, d0, d0};
    const unsigned lo = fast_lo(mode, eps);
    unsigned seq = 0;                            // exchanges done by this half's group so far
    if (tid == 0) { *s_fail = 0; s_cnt[0] = 0; s_cnt[1] = 0; }

    for (long tt0 = g - half; tt0 < a.TT; tt0 += a.groups) {     // half 0's tile decides (it has the lower index)
        const long tt = tt0 + half;
        bool valid;
        {
            const long t = 16 * tt + (lane & 15);
            int u = -1;
            if (tt < a.TT && t < a.T_) u = a.frame_utt[t];
            // padding frames count as live; a tile beyond the batch, or one holding frames of a stopped
            // utterance (left to the general kernel, skip_all_live), is not processed
            const bool live = tt < a.TT && ((t >= a.T_) || ((u >= 0) && (a.active[u] != 0)));
            const int v0 = __syncthreads_and(half == 0 ? live : true);
            const int v1 = __syncthreads_and(half == 1 ? live : true);
            valid = (half ? v1 : v0) != 0;
        }
        HTile h[AKT];
        f64x4 p[AKT];
        double a1[MSTEPS], a2[MT][4];
        if (valid) {
            for (int e = th; e < NE; e += AW * 64) vL[e] = a.Vp[(tt * 8 + (e >> 6)) * 64 + (e & 63)];
#pragma unroll
            for (int k = 0; k < AKT; ++k) {
                const f64x2* t = Hp + (tt * NT + tile0 + AW * k) * 128;
                const f64x2 h01 = t[ul], h23 = t[ul + 64];
                h[k][0] = h01[0]; h[k][1] = h01[1]; h[k][2] = h23[0]; h[k][3] = h23[1];
            }
        }
        __syncthreads();
        if (valid) {                             // numerator tiles, once per frame tile
            double x[MSTEPS];
#pragma unroll
            for (int s = 0; s < MSTEPS; ++s) x[s] = a.Xp[(tt * MSTEPS + s) * 64 + lane];
            load_a1(a1, 0);
#pragma unroll
            for (int k = 0; k < AKT; ++k) {
                f64x4 acc = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < MSTEPS; ++s) acc = Mma<double>::mma(a1[s], x[s], acc);
                load_a1(a1, (k + 1) % AKT);
                p[k] = acc;
            }
        }

        for (int step = 0; step <= 2 * a.iters; ++step) {
            const bool mine = (step & 1) == half;                 // this half's turn on the matrix pipes
            EVC_STAMP(0);
            if (valid && mine && step < 2 * a.iters) {
                // ---------------- sweep: h <- h p / (A_j^T V), V' += A_j h over the 8 resident tiles
                asm volatile("" : "+s"(sw));
                f64x4 vn[MT];
#pragma unroll
                for (int u = 0; u < MT; ++u) vn[u] = f64x4{0, 0, 0, 0};
                // One wavefront owns its SIMD during a sweep, so nothing covers its stalls, and on gfx950 an f64
                // MFMA and VALU instructions do not overlap.  Measured on this sweep (tools/ubench/fused_all_bench.hip;
                // the 15 MFMAs of a unit alone: 915 cycles): with the update arithmetic of ONE tile between D_k and
                // V'_k: +550..600 cycles, wherever its 27 instructions are placed (one clump, or level by level in
                // the gaps between the MFMAs): the quotient chain is 9 dependent f64 operations deep and a dependent
                // f64 VALU operation has ~28 cycles of latency (throughput ~6), which nothing else fills.  So the
                // sweep works on PAIRS of tiles: D_a, D_b (14 MFMAs), then the two tiles' update chains together
                // (independent: the latency slots of one are the issue slots of the other), then V'_a, V'_b.
                // D fragments come from LDS (s_a1), refilled pair of k-steps by pair of k-steps behind the MFMAs that
                // consume them; V' fragments are double-buffered and requested one pair (~2000 cycles) ahead.
                double v[MSTEPS];
#pragma unroll
                for (int s = 0; s < MSTEPS; ++s) v[s] = vL[s * 64 + lane];
                // d = A_k^T V with a1 = fragments of tile k; a1 is refilled with tile kn's behind each MFMA pair
                auto dchain = [&](f64x4& d, int kn) {
                    const double* t = s_a1 + (w + AW * kn) * (MSTEPS * 64);
#pragma unroll
                    for (int s = 0; s < MSTEPS; s += 2) {
                        d = Mma<double>::mma(a1[s], v[s], d);
                        if (s + 1 < MSTEPS) d = Mma<double>::mma(a1[s + 1], v[s + 1], d);
                        if (L::A1) {
                            if (s + 1 < MSTEPS) {
                                const f64x2 f = reinterpret_cast<const f64x2*>(t)[(s >> 1) * 64 + lane];
                                a1[s] = f[0];
                                a1[s + 1] = f[1];
                            } else {
                                a1[s] = t[(MSTEPS / 2) * 128 + lane];
                            }
                        }
                    }
                    if (!L::A1) load_a1(a1, kn);
                };
                auto vchain = [&](const double (&aa)[MT][4], const HTile& hh) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int u = 0; u < MT; ++u) vn[u] = Mma<double>::mma(aa[u][r], hh[r], vn[u]);
                };
                double a2b[MT][4];                   // second V'-fragment buffer (a2: even units, a2b: odd units)
                load_a2(a2, 0);
                load_a2(a2b, 1);
                static_assert(AKT % 2 == 0, "the sweep works on pairs of tiles");
#pragma unroll
                for (int k = 0; k < AKT; k += 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    f64x4 da = dinit, db = dinit;
                    dchain(da, k + 1);               // a1 holds tile k on entry (fetched behind the previous chain)
                    dchain(db, (k + 2) % AKT);       // (k = 6: tile 0, for the next sweep's first chain)
                    __builtin_amdgcn_sched_barrier(0);
                    double qa[4], qb[4];
                    const bool fab = mu_quot_fast2(p[k], da, p[k + 1], db, lo, qa, qb);
                    // (pinned: the optimiser would otherwise sink the quotients behind the rare-path branch)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { asm volatile("" : "+v"(qa[r])); asm volatile("" : "+v"(qb[r])); }
                    if (!__all(fab)) {
                        mu_quot_exact(p[k], da, mode, eps, qa);
                        mu_quot_exact(p[k + 1], db, mode, eps, qb);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { h[k][r] *= qa[r]; h[k + 1][r] *= qb[r]; }
                    __builtin_amdgcn_sched_barrier(0);
                    vchain(a2, h[k]);
                    if (k + 2 < AKT) load_a2(a2, k + 2);
                    vchain(a2b, h[k + 1]);
                    if (k + 2 < AKT) load_a2(a2b, k + 3);
                }
#pragma unroll
                for (int u = 0; u < MT; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (u * 4 + r < MSTEPS) red[w * NE + (u * 4 + r) * 64 + lane] = vn[u][r];
                // The wavefront that delivers last sums the four partials (fixed order) and publishes the member's
                // partial V' at once: the words travel to the peers while this half is still on its way to the
                // step barrier, so the exchange step that follows normally finds them all there.
                unsigned arrived = 0;
                if (lane == 0)
                    arrived = __hip_atomic_fetch_add(&s_cnt[half], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                arrived = __builtin_amdgcn_readfirstlane(arrived);
                if ((arrived & (AW - 1)) == AW - 1) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    long long* xb = reinterpret_cast<long long*>(a.coop_buf) +
                                    ((size_t)(seq & 1) * a.groups + g) * (size_t)(C * 512) + member * 512;
                    const long long tag = (seq >> 1) & 1;        // a buffer is reused every second exchange
#pragma unroll
                    for (int e = lane; e < NE; e += 64) {
                        double acc = 0.0;
#pragma unroll
                        for (int ww = 0; ww < AW; ++ww) acc += red[ww * NE + e];
                        sumL[e] = acc;
                        if (C > 1)
                            __hip_atomic_store(xb + e, (__double_as_longlong(acc) & ~1LL) | tag, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            } else if (valid && !mine && step > 0) {
                // ---------------- exchange: V' = sum over the members, no barrier inside
                const int e0 = th, e1 = th + AW * 64;
                const bool has1 = e1 < NE;
                double s0 = sumL[e0], s1 = sumL[has1 ? e1 : e0];
                if (C > 1) {
                    long long* xb = reinterpret_cast<long long*>(a.coop_buf) +
                                    ((size_t)(seq & 1) * a.groups + g) * (size_t)(C * 512);
                    const long long tag = (seq >> 1) & 1;
                    const long long m0 = (__double_as_longlong(s0) & ~1LL) | tag;
                    const long long m1 = (__double_as_longlong(s1) & ~1LL) | tag;
                    bool ok = true;
                    unsigned polls = 0;
                    {   // watch one word per peer (lane m <-> member m): C - 1 loads per poll and wavefront
                        const long long* sp = xb + (lane < C ? lane : 0) * 512 + (NE - 1);
                        for (;;) {
                            const long long b = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const bool ready = lane >= C || lane == member || (b & 1) == tag;
                            if (__all(ready)) break;
                            if (++polls > ALL_POLL_LIMIT ||
                                ((polls & 63) == 0 &&
                                 __hip_atomic_load(a.coop_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                                ok = false;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(2);
                        }
                    }
                    // every thread fetches its elements of all members; each word carries its own epoch bit, so
                    // a word that lags behind the watched one is simply fetched again
                    long long b0[C] = {}, b1[C] = {};
                    polls = 0;
                    while (ok) {
#pragma unroll
                        for (int m = 0; m < C; ++m)
                            b0[m] = __hip_atomic_load(xb + m * 512 + e0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int m = 0; m < C; ++m)     // (threads without a second element re-read their first)
                            b1[m] = __hip_atomic_load(xb + m * 512 + (has1 ? e1 : e0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        long long bad = 0;
#pragma unroll
                        for (int m = 0; m < C; ++m) {
                            b0[m] = (m == member) ? m0 : b0[m];
                            b1[m] = (m == member) ? m1 : b1[m];
                            bad |= (b0[m] ^ tag) | (b1[m] ^ tag);
                        }
                        if ((bad & 1) == 0) break;
                        if (++polls > ALL_POLL_LIMIT ||
                            ((polls & 63) == 0 &&
                             __hip_atomic_load(a.coop_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                            ok = false;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    s0 = 0.0; s1 = 0.0;
#pragma unroll
                    for (int m = 0; m < C; ++m) {                 // member order: identical on every member
                        s0 += __longlong_as_double(b0[m] & ~1LL);
                        s1 += __longlong_as_double(b1[m] & ~1LL);
                    }
                    ++seq;
                    if (!ok) *s_fail = 1;
                }
                vL[e0] = s0;
                if (has1) vL[e1] = s1;
            }
            EVC_STAMP(1);
            __syncthreads();
            EVC_STAMP(2);
            if (C > 1 && *s_fail) {               // a peer never showed up: void the launch, let everybody leave
                if (tid == 0) __hip_atomic_store(a.coop_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }

        if (valid) {
#pragma unroll
            for (int k = 0; k < AKT; ++k) {
                f64x2* t = Hp + (tt * NT + tile0 + AW * k) * 128;
                t[ul] = f64x2{h[k][0], h[k][1]};
                t[ul + 64] = f64x2{h[k][2], h[k][3]};
            }
            // carry V to the next launch; per-frame squared residual of the final activations
            if (member == 0) {                   // every member holds the same V: one writes it
                for (int e = th; e < NE; e += AW * 64) a.Vp[(tt * 8 + (e >> 6)) * 64 + (e & 63)] = vL[e];
                if (a.write_err && w == 0) {
                    double e = 0.0;
#pragma unroll
                    for (int s = 0; s < MSTEPS; ++s) {
                        const double x = a.Xp[(tt * MSTEPS + s) * 64 + lane], vv = vL[s * 64 + lane];
                        e += (x - vv) * (x - vv);
                    }
                    e += __shfl_xor(e, 16, 64);  // the 4 lane groups hold one frame's bins
                    e += __shfl_xor(e, 32, 64);
                    const long t = 16 * tt + lane;
                    if (lane < 16 && t < a.T_) a.err2[t] = e;
                }
            }
        }
        __syncthreads();                         // vL is rewritten for the next frame tile
    }
}

template <int MSTEPS, int C, bool DZERO>
static hipError_t launch_all_z(FusedArgs a, int n_cus, hipStream_t s) {
    const size_t lds = AllLds<MSTEPS>::bytes;
    hipError_t e = hipSuccess;
    if (lds > 64 * 1024) {   // per launch: no mutable global state is kept
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_all<MSTEPS, C, DZERO>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int occ = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_fused_all<MSTEPS, C, DZERO>, ATHREADS, lds);
    if (e != hipSuccess) return e;
    if (occ < 1) return hipErrorInvalidValue;
    long resident = n_cus;                       // one workgroup (two members) per CU
    if (2 * resident > ALL_MAX_WGS) resident = ALL_MAX_WGS / 2;
    int pairs = (int)(resident / C);             // pairs of groups
    const int want = (a.TT + 1) / 2;
    if (pairs > want) pairs = want;
    if (pairs < 1) return hipErrorInvalidValue;
    a.groups = 2 * pairs;
    if (C > 1) {
        // stale words must not carry the epoch bit of the first two exchanges (0): fill with ones
        e = hipMemsetAsync(a.coop_buf, 0xFF, sizeof(double) * 2 * (size_t)a.groups * C * 512, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_fused_all<MSTEPS, C, DZERO>), dim3((unsigned)(pairs * C)), dim3(ATHREADS), lds, s, a);
    return hipGetLastError();
}

template <int MSTEPS, int C>
static hipError_t launch_all(const FusedArgs& a, int n_cus, hipStream_t s) {
    const double d0 = a.l1 + (a.eps_mode == EVC_EPS_ADD ? a.eps : 0.0);
    return d0 == 0.0 ? launch_all_z<MSTEPS, C, true>(a, n_cus, s) : launch_all_z<MSTEPS, C, false>(a, n_cus, s);
}

template <int MSTEPS>
static hipError_t pick_c(const FusedArgs& a, int n_cus, hipStream_t s) {
    switch (a.NT / ATILES) {
        case 1: return launch_all<MSTEPS, 1>(a, n_cus, s);
        case 2: return launch_all<MSTEPS, 2>(a, n_cus, s);
        case 4: return launch_all<MSTEPS, 4>(a, n_cus, s);
        case 8: return launch_all<MSTEPS, 8>(a, n_cus, s);
        default: return hipErrorInvalidValue;
    }
}

// members per frame tile the all-resident kernel would use for this problem, 0 if it does not apply
int fused_all_members(int NT, int N, int eps_mode, int exact_div, int loss) {
    if (eps_mode == EVC_EPS_NONE || exact_div || loss != EVC_LOSS_FROBENIUS) return 0;
    if (NT % ATILES) return 0;
    const int c = NT / ATILES;
    return (c == 1 || c == 2 || c == 4 || c == 8) ? c : 0;
}

hipError_t fused_all_launch(int msteps, const FusedArgs& a, int n_cus, hipStream_t s) {
    if (a.NT % ATILES || !a.coop_buf || !a.coop_abort) return hipErrorInvalidValue;
    switch (msteps) {
        case 1: return pick_c<1>(a, n_cus, s);
        case 2: return pick_c<2>(a, n_cus, s);
        case 3: return pick_c<3>(a, n_cus, s);
        case 4: return pick_c<4>(a, n_cus, s);
        case 5: return pick_c<5>(a, n_cus, s);
        case 6: return pick_c<6>(a, n_cus, s);
        case 7: return pick_c<7>(a, n_cus, s);
        case 8: return pick_c<8>(a, n_cus, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace evc
