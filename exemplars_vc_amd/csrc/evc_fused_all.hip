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
// Other member counts (member count at run time; template C == 0: 16, 32, 64 - whole slices, the lean code of the
// common sizes; C == -1: 3, 5, 6, 7 and 9 .. 128 - ragged slices): fetching every peer's partial would cost
// (C - 1) x 3.5 KB per member and exchange, so the exchange becomes a reduce-scatter + all-gather (see the
// C <= 0 branches): two memory round trips, 7 KB fetched per member whatever C is.
// Measured at N = 16384 (C5): 5.8 us per step against 5.2 us at N = 4096, where the sweep is the longer half.
//
// Requirements (the host checks them, fused_all_members): guarded eps mode, fast quotients, Frobenius or KL loss, NT a
// multiple of 32 (fused_layout pads N to whole members when that costs <= 12.5 %), at most 128 members.  Frame
// tiles whose frames are not all live are left to the general kernel (skip_all_live), like in k_fused_res.
#include "evc_fused_common.h"

namespace evc {

constexpr int AW = 4;                  // wavefronts per half
constexpr int AKT = 8;                 // exemplar tiles per wavefront, all register-resident
constexpr int ATILES = AW * AKT;       // exemplar tiles per member
constexpr int ATHREADS = 2 * AW * 64;  // two halves per workgroup
constexpr unsigned ALL_POLL_LIMIT = 1u << 17;
#ifndef EVC_ALL_STAGGER_PER_ITER
#define EVC_ALL_STAGGER_PER_ITER 11000     // shader cycles of one iteration of one half (two steps) / 2 ... half a tile's duration = iters x one step
#endif
#ifndef EVC_ALL_EXCH_PRIO
#define EVC_ALL_EXCH_PRIO 3
#endif
#ifndef EVC_ALL_C1_ALTERNATE
#define EVC_ALL_C1_ALTERNATE 0
#endif

#ifdef EVC_ALL_TIMING   // diagnostic build only (tools/ubench/fused_all_bench.hip); no stamp executes in the library
#define EVC_STAMP(i)                                                                                             \
    do {                                                                                                         \
        if (a.dbg && tt0 == g - half && lane == 0 && w == 0)                                                     \
            a.dbg[(((long)blockIdx.x * 2 + half) * (2 * a.iters + 1) + step) * 4 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define EVC_STAMP(i)
#endif

// KL: the generalised Kullback-Leibler update (sklearn _nmf.py:557-606 with update_H=False): A1p then holds the
// dictionary divided by its column sums, the sweep's B operand is R = X / max(V, eps) instead of V, the update is
// h <- h * (A~_j^T R) - no numerator tiles, no division in the sweep - and R is formed where V is combined.
template <int MSTEPS, int C, bool KL>
__global__ __launch_bounds__(ATHREADS, 2) void k_fused_all(FusedArgs a) {
    constexpr int MT = MSTEPS > 4 ? 2 : 1;
    constexpr int E = MT * 4 * 64;               // stride of one V image (accumulator order)
    constexpr int NE = MSTEPS * 64;              // elements of V actually used
    constexpr int MSP = (MSTEPS + 1) & ~1;       // k-steps padded to pairs in A1p
    __shared__ double s_red[2][AW * E];          // partial V' of every wavefront, per half
    __shared__ double s_v[2][E];                 // V, B-operand order
    __shared__ double s_x[2][E];                 // X, B-operand order
    __shared__ double s_r[2][KL ? E : 1];        // KL: X / max(V, eps), B-operand order
    __shared__ double s_stage[2][C > 0 ? 1 : 768];   // reduce-scatter staging (more than 8 members only)
    __shared__ unsigned s_hb[2];                 // arrivals of a half's wavefronts at its LDS barrier
    __shared__ int s_fail;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = w8 >> 2, w = w8 & 3, th = tid & (AW * 64 - 1);
    const int CR = C > 0 ? C : a.coop_c;                     // members per group (C <= 0: at run time)
    const int member = blockIdx.x % CR;
    const int g = 2 * (blockIdx.x / CR) + half;          // the group this half is a member of
    double* const red = s_red[half];
    double* const vL = s_v[half];
    double* const xL = s_x[half];
    double* const rL = s_r[half];
    double* const stage = s_stage[half];
    unsigned hb = 0;                             // arrivals expected at this half's next LDS barrier
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

    // Round 4: the fragments come by buffer loads - the tile's byte offset rides in the instruction's scalar offset, the
    // lane's in a 32-bit register computed once, the k-step's in the immediate: no address arithmetic in the sweep.
    // (As global loads the lane term became a 64-bit shift-and-add per fragment group: 6 vector instructions per unit,
    // ~24 cycles beside the MFMAs; the compiler does not select the scalar-base form of global_load for it.)
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A1p), 0, NT * (MSP * 512), 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A2p), 0, NT * (MT * 2048), 0x00020000);
#ifdef EVC_ALL_NOFRAG
    bool nofrag_go = false;
#endif
    const unsigned ul16 = ul * 16u, ul8 = ul * 8u;
    auto load_a1 = [&](double (&a1)[MSTEPS], int k) {
#ifdef EVC_ALL_NOFRAG    // diagnostic (tools/ubench): no fragment traffic after the numerator pass (wrong results, timing only)
        if (nofrag_go) return;
#endif
        const int so = (int)(tile0 + sw + AW * k) * (MSP * 512);
#pragma unroll
        for (int s = 0; s < MSTEPS; s += 2) {
            if (s + 1 < MSTEPS) {
                const f64x2 v = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r1, ul16 + (s >> 1) * 1024, so, 0));
                a1[s] = v[0];
                a1[s + 1] = v[1];
            } else {        // (the odd k-step's half of the pair: the first double of the lane's 16 bytes)
                a1[s] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r1, ul16 + (s >> 1) * 1024, so, 0));
            }
        }
    };
    auto load_a2 = [&](double (&a2)[MT][4], int k) {
#ifdef EVC_ALL_NOFRAG
        if (nofrag_go) return;
#endif
        const int so = (int)(tile0 + sw + AW * k) * (MT * 2048);
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f64x2 v = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r2, ul16 + (u * 2 + (r >> 1)) * 1024, so, 0));
                a2[u][r] = v[0];
                a2[u][r + 1] = v[1];
            }
    };
    (void)ul8;

    const int mode = a.eps_mode;
    const double eps = a.eps;
    const double d0 = a.l1 + (mode == EVC_EPS_ADD ? a.eps : 0.0);
    const f64x4 dinit = {d0, d0, d0, d0};
    const unsigned lo = fast_lo(mode, eps);
    unsigned seq = 0;                            // exchanges done by this half's group so far
    // reduce-scatter exchange with ragged slices (C < 0): what this thread publishes and fetches, the same in every exchange
    int rs_es = 1, rs_len = 0, rs_pub0 = 0, rs_pub1 = 0, rs_src0 = -1, rs_src1 = -1, rs_src2 = -1;
    if (C < 0) {
        rs_es = (NE + CR - 1) / CR;                                  // elements per slice
        rs_len = NE - member * rs_es;                                // of which valid in mine
        rs_len = rs_len < 0 ? 0 : (rs_len > rs_es ? rs_es : rs_len);
        const int e0 = th, e1 = th + AW * 64;
        const int j0 = e0 / rs_es, j1 = (e1 < NE ? e1 : e0) / rs_es;
        rs_pub0 = (j0 * CR + member) * rs_es + (e0 - j0 * rs_es);
        rs_pub1 = (j1 * CR + member) * rs_es + ((e1 < NE ? e1 : e0) - j1 * rs_es);
        // my slice region is [m][ES] row-major at (member * CR) * ES: word idx = m * ES + el, valid while el < len
        auto src = [&](int idx) {
            if (idx >= CR * rs_es) return -1;
            const int el = idx % rs_es;
            return el < rs_len ? member * CR * rs_es + idx : -1;
        };
        rs_src0 = src(th);
        rs_src1 = src(th + AW * 64);
        rs_src2 = src(th + 2 * AW * 64);
    }
    if (tid == 0) s_fail = 0;
    if (tid < 2) s_hb[tid] = 0;

    // Short launches (the stop rule's 10-iteration pieces of a long batch): every group reaches the end of a frame tile
    // at the same moment, and the chip then does nothing but store and reload activations (2 x 66 MB per round at C2:
    // 21 us against 100 us of iterations, profiles/r04_default_call.md).  Every second pair of groups starts half a
    // tile's duration late, once per launch, so that only half of the CUs move activations at any time: the default
    // call's rate went from 0.845 to 0.866 of the rate without stop tests (same box, A/B; four or eight phases: the same -
    // a CU streams at ~25 GB/s however many others do, so what is left would have to overlap with the CU's own sweeps).
    if (a.stagger_cycles > 0 && ((blockIdx.x / CR) & 1)) {
        const long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < a.stagger_cycles) __builtin_amdgcn_s_sleep(32);
    }

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
        if (valid && a.first && a.init_const) {
            // first launch from the utterances' constants: H = h0 (0 in the padding), V = A H = h0 rowsum(A)
            auto h0_of = [&](int fr) {
                const long t = 16 * tt + fr;
                const int u = t < a.T_ ? a.frame_utt[t] : -1;
                return u >= 0 ? a.h0[u] : 0.0;
            };
            for (int e = th; e < E; e += AW * 64) {
                const int s = e >> 6, l = e & 63;
                const bool in = s < MSTEPS;
                xL[e] = in ? a.Xp[(tt * MSTEPS + s) * 64 + l] : 0.0;
                vL[e] = in ? a.rsum[bin_of(s, l >> 4)] * h0_of(l & 15) : 0.0;
                if (KL) rL[e] = xL[e] / (vL[e] < a.eps ? a.eps : vL[e]);
            }
            const double hv = h0_of(lane & 15);
#pragma unroll
            for (int k = 0; k < AKT; ++k) {
                const long n0 = 16 * (tile0 + AW * k) + 4 * (lane >> 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) h[k][r] = n0 + r < a.N ? hv : 0.0;
            }
        } else if (valid) {
            for (int e = th; e < E; e += AW * 64) {
                const int s = e >> 6, l = e & 63;
                const bool in = s < MSTEPS;
                xL[e] = in ? a.Xp[(tt * MSTEPS + s) * 64 + l] : 0.0;
                vL[e] = in ? a.Vp[(tt * 8 + s) * 64 + l] : 0.0;
                if (KL) rL[e] = xL[e] / (vL[e] < a.eps ? a.eps : vL[e]);       // sklearn _nmf.py:572-576
            }
#pragma unroll
            for (int k = 0; k < AKT; ++k) {
                const f64x2* t = Hp + (tt * NT + tile0 + AW * k) * 128;
                const f64x2 h01 = t[ul], h23 = t[ul + 64];
                h[k][0] = h01[0]; h[k][1] = h01[1]; h[k][2] = h23[0]; h[k][3] = h23[1];
            }
        }
        __syncthreads();
        if (valid && !KL) {                      // numerator tiles, once per frame tile
            double x[MSTEPS];
#pragma unroll
            for (int s = 0; s < MSTEPS; ++s) x[s] = xL[s * 64 + lane];
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

        if (valid && KL) load_a1(a1, 0);         // (the numerator pass, which otherwise leaves them, is skipped)
        for (int step = 0; step <= 2 * a.iters; ++step) {
#ifdef EVC_ALL_NOFRAG
            nofrag_go = step > 2;
#endif
            // this half's turn on the matrix pipes.  (One member per frame tile - N <= 512 - has no exchange to hide:
            // both halves then sweep in the same steps, two wavefronts per SIMD.)
            const bool mine = (C == 1 && !EVC_ALL_C1_ALTERNATE) ? (step & 1) == 0 : (step & 1) == half;
            EVC_STAMP(0);
            if (valid && mine && step < 2 * a.iters) {
                // ---------------- sweep: h <- h p / (A_j^T V), V' += A_j h over the 8 resident tiles
                asm volatile("" : "+s"(sw));
                double v[MSTEPS];
#pragma unroll
                for (int s = 0; s < MSTEPS; ++s) v[s] = KL ? rL[s * 64 + lane] : vL[s * 64 + lane];
                f64x4 vn[MT];
#pragma unroll
                for (int u = 0; u < MT; ++u) vn[u] = f64x4{0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < AKT; ++k) {
                    __builtin_amdgcn_sched_barrier(0);
                    load_a2(a2, k);
                    f64x4 d = KL ? f64x4{0, 0, 0, 0} : dinit;
#pragma unroll
                    for (int s = 0; s < MSTEPS; ++s) d = Mma<double>::mma(a1[s], v[s], d);
                    load_a1(a1, (k + 1) % AKT);      // the next unit's (or the next sweep's first) fragments
                    __builtin_amdgcn_sched_barrier(0);
                    if (KL) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[k][r] *= d[r];
                    } else {
                        mu_tile_guarded(h[k], p[k], d, mode, eps, lo);
                    }
#pragma unroll
                    for (int u = 0; u < MT; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) vn[u] = Mma<double>::mma(a2[u][r], h[k][r], vn[u]);
                }
#pragma unroll
                for (int u = 0; u < MT; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (u * 4 + r < MSTEPS) red[w * E + (u * 4 + r) * 64 + lane] = vn[u][r];
            } else if (valid && !mine && step > 0) {
                // ---------------- exchange: V' = sum over wavefronts and members, no barrier inside
                // (run-time member counts with M <= 12: fewer elements than threads - several threads carry one element, word
                // for word; a thread beyond NE used to poll a word of the summed slices that nobody writes)
                const int e0 = C <= 0 ? th % NE : th, e1 = th + AW * 64;
                const bool has1 = e1 < NE;
                // Round 4: the exchange is a chain of a few dozen dependent instructions (LDS, stores, polls) on a SIMD
                // whose other wavefront issues MFMAs and vector work back to back; at equal priority the older wavefront
                // wins the issue arbitration, so half 1's exchange (wavefronts 4-7) took 12.8 k cycles where half 0's
                // took 10.4 k (profiles/r04_fused_all_stamps_prio_and_11_tiles.txt).  The exchanging wavefront goes
                // first while it exchanges: its instructions are few, the sweep beside it does not notice.  Both
                // exchanges then take 10.9 k and a C2 step 11.8 k cycles instead of 12.6 k.
                __builtin_amdgcn_s_setprio(EVC_ALL_EXCH_PRIO);
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int ww = 0; ww < AW; ++ww) {
                    s0 += red[ww * E + e0];
                    s1 += red[ww * E + (has1 ? e1 : e0)];
                }
                if (C < 0) {
                    // ---- any member count without a template of its own (3, 5, 6, 7, 9 ... : N up to 32 x 512 x ...):
                    // reduce-scatter + all-gather.  Fetching every member's partial costs (C - 1) x 3.5 KB per member
                    // and exchange (108 KB at C = 32): far longer than the sweep it hides behind.  Instead member m sums
                    // slice m (ES = ceil(NE / C) elements; the last slices may be short or empty) of all C partials
                    // and publishes it; everybody then fetches the summed slices: 7 KB per member and exchange
                    // whatever C is, for a second memory round trip - which the alternation hides.
                    long long* xb1 = reinterpret_cast<long long*>(a.coop_buf) +
                                     ((size_t)(seq & 1) * a.groups + g) * (size_t)CR * ALL_RS_STRIDE;
                    long long* xb2 = reinterpret_cast<long long*>(a.coop_buf) + ALL_SLICE_OFFSET +
                                     ((size_t)(seq & 1) * a.groups + g) * 512;
                    const long long tag = (seq >> 1) & 1;
                    // partials: [slice j][member m][ES], so that what member j reduces is one contiguous run
                    __hip_atomic_store(xb1 + rs_pub0, (__double_as_longlong(s0) & ~1LL) | tag, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                    if (has1)
                        __hip_atomic_store(xb1 + rs_pub1, (__double_as_longlong(s1) & ~1LL) | tag, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    bool ok = true;
                    // poll three words (one memory round trip for all) until each carries the epoch of this exchange
                    auto fetch3 = [&](const long long* p0, const long long* p1, const long long* p2, long long& b0,
                                      long long& b1, long long& b2) {
                        unsigned polls = 0;
                        for (;;) {
                            b0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            b1 = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            b2 = __hip_atomic_load(p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (__all(!ok || (((b0 ^ tag) | (b1 ^ tag) | (b2 ^ tag)) & 1) == 0)) break;
                            if (++polls > ALL_POLL_LIMIT ||
                                ((polls & 63) == 0 &&
                                 __hip_atomic_load(a.coop_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
                                ok = false;
                            __builtin_amdgcn_s_sleep(1);
                        }
                    };
                    // reduce: my slice of all CR partials (one contiguous run: fully coalesced) goes through LDS; the
                    // half's 4 wavefronts meet at an LDS counter (a workgroup barrier would stop the other half's
                    // sweep); then 4 lanes per element sum the members q, q + 4, ... in order and combine (a fixed
                    // tree: every member obtains the bitwise identical V').  A thread's words are the same in every
                    // exchange (rs_src*, -1: none - it then re-reads a word it published itself).
                    {
                        long long b0, b1, b2;
                        const long long* own = xb1 + rs_pub0;
                        fetch3(rs_src0 >= 0 ? xb1 + rs_src0 : own, rs_src1 >= 0 ? xb1 + rs_src1 : own,
                               rs_src2 >= 0 ? xb1 + rs_src2 : own, b0, b1, b2);
                        if (rs_src0 >= 0) stage[th] = __longlong_as_double(b0 & ~1LL);
                        if (rs_src1 >= 0) stage[th + AW * 64] = __longlong_as_double(b1 & ~1LL);
                        if (rs_src2 >= 0) stage[th + 2 * AW * 64] = __longlong_as_double(b2 & ~1LL);
                        hb += AW;
                        if (lane == 0)
                            __hip_atomic_fetch_add(&s_hb[half], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        while ((int)(__hip_atomic_load(&s_hb[half], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - hb) < 0)
                            __builtin_amdgcn_s_sleep(1);
                        const int q = th & 3;
                        for (int el = th >> 2; el < rs_len; el += AW * 16) {
                            double v = 0.0;
                            for (int m = q; m < CR; m += 4) v += stage[m * rs_es + el];
                            v += __shfl_xor(v, 1, 64);
                            v += __shfl_xor(v, 2, 64);
                            if (q == 0)
                                __hip_atomic_store(xb2 + member * rs_es + el, (__double_as_longlong(v) & ~1LL) | tag,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    // gather: the summed slices (element e of V' is word e)
                    long long g0, g1, g2;
                    fetch3(xb2 + e0, xb2 + (has1 ? e1 : e0), xb2 + e0, g0, g1, g2);
                    s0 = __longlong_as_double(g0 & ~1LL);
                    s1 = __longlong_as_double(g1 & ~1LL);
                    ++seq;
                    if (!ok) s_fail = 1;
                } else if (C == 0) {
                    // ---- 16, 32 or 64 members (N = 8192, 16384, 32768): the same reduce-scatter with whole slices
                    // (CR divides NE) and two words per thread: the leaner code of the common sizes.  (original note:) reduce-scatter + all-gather.  Fetching every member's
                    // partial (C - 1 x 3.5 KB per member and exchange: 108 KB at C = 32) would make the exchange far
                    // longer than the sweep it hides behind.  Instead member m sums slice m (NE / C elements) of
                    // all C partials and publishes it; everybody then fetches the C summed slices: 7 KB per member
                    // and exchange whatever C is, for a second memory round trip - which the alternation hides.
                    const int ES = NE / CR;                          // elements per slice (CR divides 64)
                    long long* xb1 = reinterpret_cast<long long*>(a.coop_buf) +
                                     ((size_t)(seq & 1) * a.groups + g) * (size_t)CR * 512;
                    long long* xb2 = reinterpret_cast<long long*>(a.coop_buf) + ALL_SLICE_OFFSET +
                                     ((size_t)(seq & 1) * a.groups + g) * 512;
                    const long long tag = (seq >> 1) & 1;
                    // partials: [slice j][member m][ES], so that what member j reduces is one contiguous run
                    {
                        const int j0 = e0 / ES, j1 = (has1 ? e1 : e0) / ES;
                        __hip_atomic_store(xb1 + ((size_t)j0 * CR + member) * ES + (e0 - j0 * ES),
                                           (__double_as_longlong(s0) & ~1LL) | tag, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                        if (has1)
                            __hip_atomic_store(xb1 + ((size_t)j1 * CR + member) * ES + (e1 - j1 * ES),
                                               (__double_as_longlong(s1) & ~1LL) | tag, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                    }
                    bool ok = true;
                    // poll two words (one memory round trip for both) until each carries the epoch of this exchange
                    auto fetch2 = [&](const long long* p0, const long long* p1, long long& b0, long long& b1) {
                        unsigned polls = 0;
                        for (;;) {
                            b0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            b1 = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (__all(!ok || (((b0 ^ tag) | (b1 ^ tag)) & 1) == 0)) break;
                            if (++polls > ALL_POLL_LIMIT ||
                                ((polls & 63) == 0 &&
                                 __hip_atomic_load(a.coop_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
                                ok = false;
                            __builtin_amdgcn_s_sleep(1);
                        }
                    };
                    // reduce: my slice of all CR partials (NE words, contiguous: fully coalesced) goes through LDS;
                    // the half's 4 wavefronts meet at an LDS counter (a workgroup barrier would stop the other
                    // half's sweep); then 4 lanes per element sum CR / 4 members each in fixed order and combine
                    // (a fixed tree: every member obtains the bitwise identical V')
                    {
                        long long b0, b1;
                        const long long* mine_ = xb1 + (size_t)member * NE;
                        fetch2(mine_ + e0, mine_ + (has1 ? e1 : e0), b0, b1);
                        stage[e0] = __longlong_as_double(b0 & ~1LL);
                        if (has1) stage[e1] = __longlong_as_double(b1 & ~1LL);
                        hb += AW;
                        if (lane == 0)
                            __hip_atomic_fetch_add(&s_hb[half], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        while ((int)(__hip_atomic_load(&s_hb[half], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - hb) < 0)
                            __builtin_amdgcn_s_sleep(1);
                        // (4 lanes per element, each walking CR / 4 staged values.  Round 4 tried as many lanes as the half
                        // has - 16 per element at 32 members, a lane summing two values before a 4-step butterfly:
                        // slower, 1.63 against 1.53 ms per 60 iterations at C5 - the butterfly's permutes cost more than
                        // the LDS reads they replace)
                        if (th < 4 * ES) {
                            const int el = th >> 2, q = th & 3, per = CR >> 2;
                            double v = 0.0;
                            for (int i = 0; i < per; ++i) v += stage[(q * per + i) * ES + el];
                            v += __shfl_xor(v, 1, 64);
                            v += __shfl_xor(v, 2, 64);
                            if (q == 0)
                                __hip_atomic_store(xb2 + member * ES + el, (__double_as_longlong(v) & ~1LL) | tag,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    // gather: the C summed slices
                    long long g0, g1;
                    fetch2(xb2 + e0, xb2 + (has1 ? e1 : e0), g0, g1);
                    s0 = __longlong_as_double(g0 & ~1LL);
                    s1 = __longlong_as_double(g1 & ~1LL);
                    ++seq;
                    if (!ok) s_fail = 1;
                } else if (C > 1) {
                    long long* xb = reinterpret_cast<long long*>(a.coop_buf) +
                                    ((size_t)(seq & 1) * a.groups + g) * (size_t)(C * 512);
                    const long long tag = (seq >> 1) & 1;        // a buffer is reused every second exchange
                    const long long m0 = (__double_as_longlong(s0) & ~1LL) | tag;
                    const long long m1 = (__double_as_longlong(s1) & ~1LL) | tag;
                    __hip_atomic_store(xb + member * 512 + e0, m0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (has1) __hip_atomic_store(xb + member * 512 + e1, m1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                    long long b0[C > 0 ? C : 1] = {}, b1[C > 0 ? C : 1] = {};
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
                    if (!ok) s_fail = 1;
                }
                vL[e0] = s0;
                if (has1) vL[e1] = s1;
                if (KL) {
                    rL[e0] = xL[e0] / (s0 < a.eps ? a.eps : s0);
                    if (has1) rL[e1] = xL[e1] / (s1 < a.eps ? a.eps : s1);
                }
                __builtin_amdgcn_s_setprio(0);
            }
            EVC_STAMP(1);
            __syncthreads();
            EVC_STAMP(2);
            if (C != 1 && s_fail) {              // a peer never showed up: void the launch, let everybody leave
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
            if (a.Hx) {                          // last launch: the caller's H as well (no separate export pass)
                const long t = 16 * tt + (lane & 15);
                if (t < a.T_) {
#pragma unroll
                    for (int k = 0; k < AKT; ++k) {
                        const long n0 = 16 * (tile0 + AW * k) + 4 * (lane >> 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n0 + r < a.N) {
                                if (a.hx_frame_major) a.Hx[t * a.ldhx + n0 + r] = h[k][r];
                                else a.Hx[(n0 + r) * a.ldhx + t] = h[k][r];
                            }
                    }
                }
            }
            // carry V to the next launch; per-frame squared residual of the final activations
            if (member == 0) {                   // every member holds the same V: one writes it
                for (int e = th; e < NE; e += AW * 64) a.Vp[(tt * 8 + (e >> 6)) * 64 + (e & 63)] = vL[e];
                if (a.write_err && w == 0) {
                    double e = 0.0;
#pragma unroll
                    for (int s = 0; s < MSTEPS; ++s) {
                        const double x = xL[s * 64 + lane], vv = vL[s * 64 + lane];
                        e += KL ? kl_terms(x, vv, a.eps) : (x - vv) * (x - vv);
                    }
                    e += __shfl_xor(e, 16, 64);  // the 4 lane groups hold one frame's bins
                    e += __shfl_xor(e, 32, 64);
                    const long t = 16 * tt + lane;
                    if (lane < 16 && t < a.T_) a.err2[t] = e;
                }
            }
        }
        __syncthreads();                         // xL / vL are rewritten for the next frame tile
    }
}

template <int MSTEPS, int C, bool KL>
static hipError_t launch_all(FusedArgs a, int n_cus, hipStream_t s) {
    int occ = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_fused_all<MSTEPS, C, KL>, ATHREADS, 0);
    if (e != hipSuccess) return e;
    if (occ < 1) return hipErrorInvalidValue;
    long resident = n_cus;                       // one workgroup (two members) per CU
    if (2 * resident > ALL_MAX_WGS) resident = ALL_MAX_WGS / 2;
    const int cr = C > 0 ? C : a.coop_c;
    if (C == 0 && (cr < 8 || cr > 64 || (cr & (cr - 1)))) return hipErrorInvalidValue;
    if (C < 0 && (cr < 2 || cr > ALL_MAX_MEMBERS)) return hipErrorInvalidValue;
    int pairs = (int)(resident / cr);            // pairs of groups
    const int want = (a.TT + 1) / 2;
    if (pairs > want) pairs = want;
    if (pairs < 1) return hipErrorInvalidValue;
    a.groups = 2 * pairs;
    // stagger (see the kernel): launches of few iterations over many rounds of frame tiles
    a.stagger_cycles = (a.iters > 0 && a.iters <= 25 && a.TT >= 4L * a.groups && pairs >= 2) ? (long long)a.iters * EVC_ALL_STAGGER_PER_ITER : 0;
    if (C != 1) {
        // stale words must not carry the epoch bit of the first two exchanges (0): fill with ones
        e = hipMemsetAsync(a.coop_buf, 0xFF, sizeof(double) * 2 * (size_t)a.groups * cr * (C < 0 ? ALL_RS_STRIDE : 512), s);
        if (e != hipSuccess) return e;
    }
    if (C <= 0) {
        if (2L * a.groups * 512 > ALL_SLICE_ELEMS) return hipErrorInvalidValue;
        e = hipMemsetAsync(a.coop_buf + ALL_SLICE_OFFSET, 0xFF, sizeof(double) * 2 * (size_t)a.groups * 512, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_fused_all<MSTEPS, C, KL>), dim3((unsigned)(pairs * cr)), dim3(ATHREADS), 0, s, a);
    return hipGetLastError();
}

template <int MSTEPS, bool KL>
static hipError_t pick_c(const FusedArgs& a, int n_cus, hipStream_t s) {
    switch (a.NT / ATILES) {
        case 1: return launch_all<MSTEPS, 1, KL>(a, n_cus, s);
        case 2: return launch_all<MSTEPS, 2, KL>(a, n_cus, s);
        case 4: return launch_all<MSTEPS, 4, KL>(a, n_cus, s);
        // 8 members: the direct exchange (every member fetches all 8 partials: 296 GB of fabric traffic per C2 launch)
        // and the reduce-scatter (37 GB) run equally fast - 953 vs 958 k frames/s at C2, 537 vs 538 k for one
        // utterance - so the leaner one serves.  (Below 7 members its 4 lanes per element do not cover a slice.)
#ifdef EVC_ALL_DIRECT8   // diagnostic (tools/ubench): the direct all-to-all at 8 members (one hand-off, 8 x the fetched bytes)
        case 8: return launch_all<MSTEPS, 8, KL>(a, n_cus, s);
#else
        case 8:
#endif
        case 16:
        case 32:
        case 64: return launch_all<MSTEPS, 0, KL>(a, n_cus, s);    // run-time members, reduce-scatter, whole slices
        default: return launch_all<MSTEPS, -1, KL>(a, n_cus, s);  // any other count: ragged slices
    }
}

// members per frame tile the all-resident kernel would use for this problem, 0 if it does not apply
int fused_all_members(int NT, int N, int eps_mode, int exact_div, int loss) {
    // (either loss: the KL update needs the ZERO_REPLACE guard and no L1, which evc_nmf_solve checks)
    if (eps_mode == EVC_EPS_NONE || exact_div || (loss != EVC_LOSS_FROBENIUS && loss != EVC_LOSS_KL)) return 0;
    if (NT % ATILES) return 0;
    const int c = NT / ATILES;
    // The KL sweep has almost no VALU work (3.5 us), so with 2 .. 15 members a step is as long as its exchange and
    // k_fused_res's streaming form is as fast or faster (C2-sized batches, k frames/s, all-resident vs streamed:
    // N = 512: 19 800 vs 14 000; 2048: 2 234 vs 2 263; 4096: 930 vs 1 115; 8192: 458 vs 423; 16 384: 214 vs 138)
    if (loss == EVC_LOSS_KL && c >= 2 && c < 16) return 0;
    return (c >= 1 && c <= ALL_MAX_MEMBERS) ? c : 0;
}

hipError_t fused_all_launch(int msteps, const FusedArgs& a, int n_cus, hipStream_t s) {
    if (a.NT % ATILES || !a.coop_buf || !a.coop_abort) return hipErrorInvalidValue;
    switch (msteps) {
        case 1: return a.loss == EVC_LOSS_KL ? pick_c<1, true>(a, n_cus, s) : pick_c<1, false>(a, n_cus, s);
        case 2: return a.loss == EVC_LOSS_KL ? pick_c<2, true>(a, n_cus, s) : pick_c<2, false>(a, n_cus, s);
        case 3: return a.loss == EVC_LOSS_KL ? pick_c<3, true>(a, n_cus, s) : pick_c<3, false>(a, n_cus, s);
        case 4: return a.loss == EVC_LOSS_KL ? pick_c<4, true>(a, n_cus, s) : pick_c<4, false>(a, n_cus, s);
        case 5: return a.loss == EVC_LOSS_KL ? pick_c<5, true>(a, n_cus, s) : pick_c<5, false>(a, n_cus, s);
        case 6: return a.loss == EVC_LOSS_KL ? pick_c<6, true>(a, n_cus, s) : pick_c<6, false>(a, n_cus, s);
        case 7: return a.loss == EVC_LOSS_KL ? pick_c<7, true>(a, n_cus, s) : pick_c<7, false>(a, n_cus, s);
        case 8: return a.loss == EVC_LOSS_KL ? pick_c<8, true>(a, n_cus, s) : pick_c<8, false>(a, n_cus, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace evc
