// k_fused_wide: the fused FACTORED multiplicative update for wide spectra (32 < M <= 208 bins, float32) - the shapes
// the reference script really runs (|Re STFT|, M = 201: 04_align_n_nmf.py:315-326, config/config:12).
//
//   per iteration and frame:  D = A^T V (+ l1, + eps)      V = A H of the previous iteration
//                             H' = H (.) P (/) guard(D)     P = A^T X, formed once
//                             V' = A H'
//
// Who holds what.  A wavefront owns 16 frames.  Its V tile (M x 16: MT accumulator tiles, 52 registers at M = 201)
// and the V' it accumulates never leave its registers inside a task: the accumulator layout of V' IS the B-operand
// layout of the next D product (for the 16x16x4 f32 MFMA: lane = (q, frame), register r <-> row 4 q + r, which as a
// B operand is k = q of k-step r), and the accumulator layout of D is the layout of the H tile and the B-operand
// layout of V' += A_j H'_j.  No shuffle, no LDS round trip for V, D or H.  H and P stream through memory exactly
// once per iteration as whole 1 KiB tiles in that accumulator order; the denominator never exists in memory.
//
// The dictionary is read from L2 once per WORKGROUP and iteration: the W wavefronts of a workgroup are W different
// frame tiles sweeping the same exemplar blocks in lock step, and each block's two operand images (A-operand of
// D: 16 exemplars x M bins, A-operand of V': M bins x 16 exemplars; 2 x MT KiB) are copied global -> LDS by
// LDS-DMA (global_load_lds_dwordx4, no staging registers) one block ahead, into a double buffer; every fragment
// read is a conflict-free 1 KiB ds_read_b128 that feeds four k-steps.  One workgroup barrier per block.
//
// Filling 256 CUs with few frames.  An utterance is 43 frame tiles, 16 utterances are 688: far fewer than the
// 1024 SIMDs x 2.  So the exemplars of a frame group (W tiles) are split into c ranges, and the unit of work is a
// TASK (iteration, frame group, range): sweep the range's blocks, publish the partial V' (MT KiB per wavefront).
// Tasks are handed out by a ticket counter in a fixed global order (iteration-major), to however many workgroups
// are resident.  A task of iteration i+1 waits, on a per-group counter, for the c tasks of iteration i of its
// group - all of them hold EARLIER tickets, i.e. are running or done: no deadlock whatever the residency, no
// cooperative launch, no rounds, and load balance over any batch size (the tail is one task at the very end of
// the solve, not one per iteration).  Consumers sum the c partials in range order (bitwise reproducible).  With
// many ranges (one or two utterances: c = 23...) a reduce task per (iteration, group, slice) sums a slice of the
// c partials once, so that the traffic stays linear in c.
//
// Cross-workgroup visibility (guide: "Valid forms", first row of the measured table): every handed-off byte (V
// partials and sums, H, P) is stored with sc1 (write-through) 16-byte stores and loaded with sc1 16-byte loads;
// every storing wavefront drains its stores (s_waitcnt vmcnt(0)) before the workgroup barrier behind which ONE
// lane adds to the group's counter; the consumer's lane 0 polls the counter with sc1 loads and the other
// wavefronts load behind a barrier that lane then joins.  One workgroup per CU.
#include "evc_internal.h"

#include <stdlib.h>

namespace evc {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int WIDE_KL = 100;                   // update mode besides the four eps modes
constexpr long WIDE_SPIN_LIMIT = 1L << 25;     // polls of >= 0.1 us: seconds; only a wedged device gets there

struct WideArgs {
    const float* Aw;         // [NB][2][MT][64][4]  per exemplar block: D-operand image, then V'-operand image
    const float* Xw;         // [G W][MT][64][4]    frames in the V tile layout
    float* Hw;               // [G W][NB][64][4]    activations in accumulator order
    float* Pw;               // [G W][NB][64][4]    numerators A^T X
    float* Vpart;            // [2][G][c][W][MT][64][4]
    float* Vsum;             // [2][G][W][MT][64][4]   (reduce mode)
    unsigned* ticket;        // next task of this launch
    unsigned* done;          // [G] sweep tasks completed since the solve began
    unsigned* done_r;        // [G] reduce tasks completed
    int* abort;              // raised by a wait that ran out: everybody leaves, the export writes NaN
    const int* frame_utt;    // [>= T_]
    const int* active;       // [n_utt]
    const double* h0;        // [n_utt] constant start value per utterance
    int NB, TT, G, c, rmode;
    int it_begin, it_end;    // iterations of this launch; iteration 0 forms P and V = A H0
    int N, T_;
    int mode;                // EVC_EPS_* or WIDE_KL
    float eps, l1;
    int init_const;          // 1: iteration 0 fills H with the utterances' constants (else Hw holds the given H0)
    // Several stop checks per launch (round 4): the iterations snap_first, snap_first + snap_every, ... below
    // it_end - 1 are CHECK iterations inside the launch.  Their H' is stored twice - in place and into snapshot slot k -
    // and the sweep tasks of the following iteration, which sum the published V' anyway, leave the frames' squared
    // residuals in err2s[k].  The host evaluates the stop rules of those checks after the launch, in order, and copies
    // a snapshot back over the activations of an utterance that stopped there (it was iterated on to the end of the
    // launch).  snap_every == 0: off.
    int snap_every, snap_first;
    // 1: no ticket counter - workgroup b runs sweep task b (and reduce slice b) of every iteration.  For batches whose
    // sweep tasks of one iteration fit the CUs (G c <= CUs: up to ~5 utterances): with tickets drawn in order a workgroup
    // that finished early drew a reduce task whose group was still sweeping and sat on it (1 utterance: 30 us of a
    // reduce task's 36, 4 utterances: 97 of 106 - profiles/r04_wide_small_batches.md)
    int static_q;
    // 1 (static schedule with reduce slices only): no counters - every published float carries the hand-off's epoch bit
    // ((iteration >> 1) & 1; the slots alternate by iteration parity) in its lowest mantissa bit, readers poll the data
    // itself and clear the bit.  Takes the store drain, the counter and the poll of the counter out of both hand-offs of
    // an iteration; costs every partial sum at most one float32 ulp (as k_fused_all's exchange does in float64).
    int tagged;
    float* Hs;               // [slots][hs_stride]
    size_t hs_stride;
    double* err2s;           // [slots][err_stride]
    long err_stride;
};

__device__ __forceinline__ f32x4 ld_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16));
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, byte_off, 0, 16);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void glds16(const float* g, char* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
__device__ __forceinline__ unsigned ld_ctr(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Diagnostic builds only (tools/ubench/wide_bench.hip, -DEVC_WIDE_STAMP): thread 0 of a workgroup stamps the
// 100 MHz real-time counter at the phase boundaries of every task into a buffer of its own.
// -DEVC_WIDE_ABLATE=n (diagnostic, wrong results): what bounds a block step is found by taking one part out.
// 1: no H / P loads and no H store (the update runs on what is in the registers); 2: the block loop re-reads the
// first block's images (no LDS-DMA inside the loop); 3: both; 4: no update arithmetic (H' = H)
#ifndef EVC_WIDE_ABLATE
#define EVC_WIDE_ABLATE 0
#endif
#ifdef EVC_WIDE_STAMP
__device__ unsigned long long* evc_wide_dbg = nullptr;        // [tasks of the launch][8]
#define WSTAMP(k) do { if (tid == 0 && evc_wide_dbg) evc_wide_dbg[(size_t)tk * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define WNOTE(k, v) do { if (tid == 0 && evc_wide_dbg) evc_wide_dbg[(size_t)tk * 8 + (k)] = (unsigned long long)(v); } while (0)
#else
#define WSTAMP(k)
#define WNOTE(k, v)
#endif

// TG: the instance with tagged hand-offs (WideArgs.tagged).  As a run-time flag in one instance the extra code cost the
// other schedules 5 % (16 utterances of the STFT flow: 0.686 -> 0.646 of the peak over the whole call).
template <int MT, int W, bool TG>
__global__ __launch_bounds__(W * 64, W / 4) void k_fused_wide(WideArgs a) {
#ifdef EVC_WIDE_STAGGER          // diagnostic build: see "Stagger" below - measured slower, off
    constexpr bool STAGGER = W == 8;
#else
    constexpr bool STAGGER = false;
#endif
    constexpr int NSTAGE = 3;                             // LDS stages of block images
    constexpr int IMG = 2 * MT * 1024;                    // bytes of one block's two operand images
    constexpr int NPIECE = 2 * MT;                        // 1 KiB LDS-DMA pieces per block
    constexpr int PPW = (NPIECE + W - 1) / W;             // pieces per wavefront
    constexpr unsigned TILE_B = MT * 1024;                // bytes of one V tile
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [NSTAGE][IMG], then 16 bytes of control words
    volatile unsigned* s_ctl = reinterpret_cast<volatile unsigned*>(smem + NSTAGE * IMG);
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, i16 = lane & 15;
    const unsigned GC = (unsigned)(a.G * a.c);
    const unsigned per_it = a.rmode ? 2u * GC : GC;
    const unsigned total = per_it * (unsigned)(a.it_end - a.it_begin);
    const unsigned c = (unsigned)a.c;

    // The first ticket of a workgroup is its index (the counter starts at the grid size, wide_iterate): with first tickets
    // drawn from the counter, the workgroups that arrived first had drawn their SECOND ticket (requested at the top of a
    // task) before the last ones arrived, those started on tasks of the launch's second iteration and waited a whole task
    // for them, and a launch that starts in the middle of a solve did not recover from that in 30 iterations (19 % of
    // its time in dependency waits, tools/ubench/wide_bench.hip split mode, profiles/r04_wide_split_launch.md).
    unsigned nxt = 0;
    if (tid == 0) {
        nxt = blockIdx.x;
        s_ctl[0] = nxt;
        s_ctl[1] = 1u;
        s_ctl[2] = 0u;                         // tagged hand-offs: a bounded poll of the data ran out
    }
    __syncthreads();
    // tagged hand-offs (WideArgs.tagged): the epoch bit rides in the lowest mantissa bit of every float
    // (the bit that makes room for the tag is rounded away, half to even - every dropped bit is a tie: truncation shrank
    // every partial sum by half an ulp on average and doubled the float32 drift of a 150-iteration solve)
    auto tag_set = [](f32x4 v, unsigned tg) -> f32x4 {
        u32x4 b = __builtin_bit_cast(u32x4, v);
        b = ((b + (b & (b >> 1) & 1u)) & ~1u) | tg;
        return __builtin_bit_cast(f32x4, b);
    };
    auto tag_clear = [](f32x4 v) -> f32x4 {
        u32x4 b = __builtin_bit_cast(u32x4, v);
        b = b & ~1u;
        return __builtin_bit_cast(f32x4, b);
    };
    auto tag_bad = [](f32x4 v, unsigned tg) -> unsigned {      // != 0: some float of v does not carry epoch bit tg yet
        const u32x4 b = __builtin_bit_cast(u32x4, v);
        return ((b[0] ^ tg) | (b[1] ^ tg) | (b[2] ^ tg) | (b[3] ^ tg)) & 1u;
    };
    auto poll_gave_up = [&](long spins) -> bool {
        return spins > WIDE_SPIN_LIMIT ||
               ((spins & 63) == 0 && __hip_atomic_load(a.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
    };
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // thread 0 polls a counter until it reaches `need`; false (for every thread) when the solve was aborted
    auto wait_for = [&](const unsigned* ctr, unsigned need) -> bool {
        if (tid == 0) {
            unsigned ok = 1u;
            long spins = 0;
            while (ld_ctr(ctr) < need) {
                __builtin_amdgcn_s_sleep(2);
                ++spins;
                if ((spins & 63) == 0 && __hip_atomic_load(a.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    ok = 0u;
                    break;
                }
                if (spins > WIDE_SPIN_LIMIT) {
                    __hip_atomic_store(a.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0u;
                    break;
                }
            }
            s_ctl[1] = ok;
        }
        __syncthreads();
        return s_ctl[1] != 0u;
    };
    // this wavefront's share of block jb's images -> LDS stage st (asynchronous: counted in vmcnt)
    auto stage_block = [&](int jb, int st) {
        const float* src = a.Aw + (size_t)jb * (IMG / 4) + lane * 4;
        char* dst = smem + st * IMG;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int p = w + k * W;
            if (p < NPIECE) glds16(src + p * 256, dst + p * 1024);
        }
    };

    for (;;) {
        const unsigned tk = __builtin_amdgcn_readfirstlane(s_ctl[0]);
        if (tk >= total) break;
        __syncthreads();                       // everybody has read the ticket before thread 0 replaces it
        const unsigned itl = tk / per_it, rem = tk - itl * per_it;
        // The next ticket: static schedule - this workgroup's reduce slice, then its sweep of the next iteration.  Ticket
        // queue - drawn from the counter when this task is FINISHED (rounds 3 and early 4 requested it here, at the start,
        // to hide the atomic's latency: tickets were then bound to workgroups a whole task before they became free, and with
        // a few more tasks per iteration than CUs a launch fell into a slow pattern about one time in three - 8 utterances
        // of the STFT flow: 28.0 - 28.5 ms or 34.5 - 35.8 ms per 150 iterations; drawn late: 26.0 - 26.4 ms every time, and
        // every batch size gained, 16 utterances 50.4 -> 49.4 ms)
        if (tid == 0 && a.static_q) nxt = (a.rmode && rem < GC) ? tk + GC : (itl + 1) * per_it + blockIdx.x;
        const int it = a.it_begin + (int)itl;
        const bool reduce = rem >= GC;
        const unsigned idx = reduce ? rem - GC : rem;
        const int g = (int)(idx / c), e = (int)(idx - (unsigned)g * c);
        const unsigned par = (unsigned)(it & 1);
        WSTAMP(0);
        WNOTE(6, (reduce ? 1u : 0u) | ((unsigned)blockIdx.x << 8));
        WNOTE(7, ((unsigned long long)it << 32) | (unsigned)(g * 256 + e));

        if (reduce) {
            // ---- reduce task: slice e of the group's V' = sum over the c ranges, in range order ----
            if (!TG && !wait_for(a.done + g, c * (unsigned)(it + 1))) break;
            WSTAMP(1);
            const unsigned tg = ((unsigned)it >> 1) & 1u;         // (tagged) epoch bit of this iteration's partials and sums
            const unsigned U = W * MT * 64;                       // 16-byte units of a group's partial
            const unsigned lo = (unsigned)((unsigned long)e * U / c), hi = (unsigned)((unsigned long)(e + 1) * U / c);
            const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.Vpart + ((size_t)(par * a.G + g) * c) * U * 4, c * U * 16u);
            const __amdgpu_buffer_rsrc_t rout = make_rsrc(a.Vsum + (size_t)(par * a.G + g) * U * 4, U * 16u);
            // (sharing a unit's c partials between thread sets - one batch of loads in flight instead of two in a row - was
            // tried twice: with counters the wait for the slowest member stood in front of it either way (sum 5.8 -> 4.8 us,
            // wait 6.4 -> 7.6), with tagged hand-offs a step stayed at 43.0 us against 42.3; dropped)
            for (unsigned un = lo + tid; un < hi; un += W * 64) {
                // (tagged: units of frame tiles beyond the batch are never published - wavefronts without frames store
                // nothing - and never read: not waited for)
                if (TG && g * W + (int)(un / (MT * 64)) >= a.TT) continue;
                // 24 loads in flight at a time (each is a memory round trip), summed in range order
                f32x4 acc = f32x4{0, 0, 0, 0};
                for (unsigned m0 = 0; m0 < c; m0 += 24) {
                    f32x4 v[24];
                    for (long spins = 0;;) {
#pragma unroll
                        for (unsigned k = 0; k < 24; ++k) {
                            const unsigned m = m0 + k < c ? m0 + k : c - 1;
                            v[k] = ld_sc1(rin, (m * U + un) * 16u);
                        }
                        if (!TG) break;
                        unsigned bad = 0;
#pragma unroll
                        for (unsigned k = 0; k < 24; ++k) bad |= tag_bad(v[k], tg);
                        if (!bad) break;                          // (a partial that has not arrived is fetched again)
                        if (poll_gave_up(++spins)) {
                            s_ctl[2] = 1u;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
#pragma unroll
                    for (unsigned k = 0; k < 24; ++k) {
                        const f32x4 vk = TG ? tag_clear(v[k]) : v[k];
                        if (m0 + k < c) acc = (m0 + k) ? acc + vk : vk;
                    }
                }
                st_sc1(rout, un * 16u, TG ? tag_set(acc, tg) : acc);
            }
            WSTAMP(3);
            if (TG) {
                lds_barrier();                                    // (the stores stay in flight: readers poll the data)
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            WSTAMP(4);
            if (tid == 0) {
                if (!TG) __hip_atomic_fetch_add(a.done_r + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!a.static_q) nxt = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_ctl[0] = nxt;
            }
            if (TG) {
                lds_barrier();
                if (s_ctl[2]) {                                   // a partial never arrived: void the solve, everybody leaves
                    if (tid == 0) __hip_atomic_store(a.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            } else {
                __syncthreads();
            }
            WSTAMP(5);
            continue;
        }

        // ---- sweep task (iteration it, frame group g, exemplar range e) ----
        const int j0 = (int)((long)e * a.NB / a.c), j1 = (int)((long)(e + 1) * a.NB / a.c), nb = j1 - j0;
        const int ft = g * W + w;
        // check iterations inside the launch (see WideArgs): this iteration's H' also goes to snapshot slot snap_slot; the
        // V this task reads is the one a check iteration published: the range-0 task writes the residuals of slot err_slot
        int snap_slot = -1, err_slot = -1;
        if (a.snap_every > 0) {
            const int d0 = it - a.snap_first, d1 = d0 - 1;
            if (d0 >= 0 && it < a.it_end - 1 && d0 % a.snap_every == 0) snap_slot = d0 / a.snap_every;
            if (e == 0 && d1 >= 0 && it - 1 < a.it_end - 1 && d1 % a.snap_every == 0) err_slot = d1 / a.snap_every;
        }
        const bool on = ft < a.TT;                                // (wave-uniform) this wavefront has frames
        const bool kl = a.mode == WIDE_KL;
        if (nb > 0) stage_block(j0, 0);       // (the dictionary does not depend on anybody: its first block is on its way
                                              // while the group's counter is polled)
        // Is any frame of this group live (its utterance has not stopped)?  The same answer in every wavefront, without
        // a barrier; the flags do not change during a launch and the loads run beside the dependency wait.
        bool grp_live = false;
        if (it > 0) {
#pragma unroll
            for (int k = 0; k < (W * 16 + 63) / 64; ++k) {
                const int t = (g * W) * 16 + k * 64 + lane;
                if (k * 64 + lane < W * 16 && t < a.T_) {
                    const int ut = a.frame_utt[t];
                    if (ut >= 0 && a.active[ut] != 0) grp_live = true;
                }
            }
            grp_live = __ballot(grp_live ? 1 : 0) != 0;
        }
        if (!TG && it > 0 && !wait_for(a.rmode ? a.done_r + g : a.done + g, c * (unsigned)it)) break;
        WSTAMP(1);
        if (it > 0 && !grp_live) {
            // A frame group whose utterances have all stopped (or that is padding) does not sweep: its H stays, and so
            // does V' = A H - the task republishes the partial it published one iteration ago (the same bits).
            if (on) {
                const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.Vpart + ((((size_t)((par ^ 1u) * a.G + g) * c + e) * W + w)) * (TILE_B / 4), TILE_B);
                const __amdgpu_buffer_rsrc_t rdst = make_rsrc(a.Vpart + ((((size_t)(par * a.G + g) * c + e) * W + w)) * (TILE_B / 4), TILE_B);
                // (tagged: the partial of one iteration ago is this wavefront's own store - drained here - and goes out
                // again under this iteration's epoch bit.  First the sums of iteration it - 1 must be complete, as for a
                // sweep: that is what says every member has read the partials of it - 1, whose slot the NEXT iteration
                // overwrites - without it the members of a stopped group run ahead of each other's reduce slices.)
                if (TG) {
                    const __amdgpu_buffer_rsrc_t rq =
                        make_rsrc(a.Vsum + ((size_t)((par ^ 1u) * a.G + g) * W + w) * (TILE_B / 4), TILE_B);
                    const unsigned tq = ((unsigned)(it - 1) >> 1) & 1u;
                    for (long spins = 0;;) {
                        unsigned bad = 0;
#pragma unroll
                        for (int u = 0; u < MT; ++u) bad |= tag_bad(ld_sc1(rq, (u * 64 + lane) * 16u), tq);
                        if (__ballot(bad != 0) == 0) break;
                        if (poll_gave_up(++spins)) {
                            if (lane == 0) s_ctl[2] = 1u;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
#pragma unroll
                for (int u = 0; u < MT; ++u) {
                    const f32x4 v = ld_sc1(rsrc, (u * 64 + lane) * 16u);
                    st_sc1(rdst, (u * 64 + lane) * 16u, TG ? tag_set(v, ((unsigned)it >> 1) & 1u) : v);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the staged block of this task too)
            __syncthreads();
            if (tid == 0) {
                if (!TG) __hip_atomic_fetch_add(a.done + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!a.static_q) nxt = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_ctl[0] = nxt;
            }
            __syncthreads();
            if (TG && s_ctl[2]) {
                if (tid == 0) __hip_atomic_store(a.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            continue;
        }

        f32x4 Vin[MT], Vn[MT];
#pragma unroll
        for (int u = 0; u < MT; ++u) { Vin[u] = f32x4{0, 0, 0, 0}; Vn[u] = f32x4{0, 0, 0, 0}; }
        bool live = false;
        float h0v = 0.f;
        if (on) {
            const f32x4* xt = reinterpret_cast<const f32x4*>(a.Xw) + (size_t)ft * (MT * 64) + lane;
            if (it == 0) {
#pragma unroll
                for (int u = 0; u < MT; ++u) Vin[u] = xt[u * 64];
            } else if (a.rmode) {
                const __amdgpu_buffer_rsrc_t rv =
                    make_rsrc(a.Vsum + ((size_t)((par ^ 1u) * a.G + g) * W + w) * (TILE_B / 4), TILE_B);
                const unsigned tg = ((unsigned)(it - 1) >> 1) & 1u;       // (tagged) epoch bit of the sums of iteration it - 1
                for (long spins = 0;;) {
#pragma unroll
                    for (int u = 0; u < MT; ++u) Vin[u] = ld_sc1(rv, (u * 64 + lane) * 16u);
                    if (!TG) break;
                    unsigned bad = 0;
#pragma unroll
                    for (int u = 0; u < MT; ++u) bad |= tag_bad(Vin[u], tg);
                    if (__ballot(bad != 0) == 0) break;           // every slice of this tile has arrived
                    if (poll_gave_up(++spins)) {
                        if (lane == 0) s_ctl[2] = 1u;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (TG) {
#pragma unroll
                    for (int u = 0; u < MT; ++u) Vin[u] = tag_clear(Vin[u]);
                }
            } else {
                const __amdgpu_buffer_rsrc_t rv =
                    make_rsrc(a.Vpart + ((size_t)((par ^ 1u) * a.G + g) * c) * W * (TILE_B / 4), c * W * TILE_B);
                for (unsigned m = 0; m < c; ++m) {
#pragma unroll
                    for (int u = 0; u < MT; ++u) {
                        const f32x4 v = ld_sc1(rv, (m * W + w) * TILE_B + (u * 64 + lane) * 16u);
                        Vin[u] = m ? Vin[u] + v : v;
                    }
                }
            }
            if (err_slot >= 0) {      // ||x_t - v_t||^2 of this wavefront's 16 frames, summed as k_wide_err2 sums it
                double acc = 0.0;
#pragma unroll
                for (int u = 0; u < MT; ++u) {
                    const f32x4 x = xt[u * 64];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double xd = x[r], vd = Vin[u][r];
                        acc += (xd - vd) * (xd - vd);
                    }
                }
                acc += __shfl_xor(acc, 16, 64);
                acc += __shfl_xor(acc, 32, 64);
                const long t = 16L * ft + (lane & 15);
                if (lane < 16 && t < a.T_) a.err2s[(long)err_slot * a.err_stride + t] = acc;
            }
            if (kl && it > 0) {       // B operand of the KL numerator: X (/) max(V, eps)   (sklearn _nmf.py:560-575)
#pragma unroll
                for (int u = 0; u < MT; ++u) {
                    const f32x4 x = xt[u * 64];
#pragma unroll
                    for (int r = 0; r < 4; ++r) Vin[u][r] = x[r] / (Vin[u][r] < a.eps ? a.eps : Vin[u][r]);
                }
            }
            const int t = ft * 16 + i16;
            if (t < a.T_) {
                const int ut = a.frame_utt[t];
                if (ut >= 0) {
                    live = a.active[ut] != 0;
                    h0v = (float)a.h0[ut];
                }
            }
        }
        const __amdgpu_buffer_rsrc_t rh = make_rsrc(a.Hw + (size_t)(on ? ft : 0) * a.NB * 256, (unsigned)a.NB * 1024u);
        const __amdgpu_buffer_rsrc_t rp = make_rsrc(a.Pw + (size_t)(on ? ft : 0) * a.NB * 256, (unsigned)a.NB * 1024u);
        const __amdgpu_buffer_rsrc_t rhs = make_rsrc(
            (snap_slot >= 0 ? a.Hs + (size_t)snap_slot * a.hs_stride : a.Hw) + (size_t)(on ? ft : 0) * a.NB * 256, (unsigned)a.NB * 1024u);
        const bool load_h = it > 0 || !a.init_const, load_p = it > 0 && !kl;
        // start value of the denominator's accumulator: l1, and pymf's + eps (ADD); iteration 0 forms the bare P
        const float c0 = it == 0 ? 0.f : a.l1 + (a.mode == EVC_EPS_ADD ? a.eps : 0.f);
        const int n_edge = (a.N & 15) ? a.NB - 1 : -1;           // the block that reaches into the zero padding
        f32x4 hC = f32x4{0, 0, 0, 0}, pC = hC, hN = hC, pN = hC;
        if (on && nb > 0) {
            if (load_h) hC = ld_sc1(rh, (j0 * 64 + lane) * 16u);
            if (load_p) pC = ld_sc1(rp, (j0 * 64 + lane) * 16u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                          // block j0 is in stage 0
        WSTAMP(2);
        if (TG && s_ctl[2]) {                               // the sums of the previous iteration never arrived
            if (tid == 0) __hip_atomic_store(a.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }

        // Every memory operation of a block step gets a whole step to complete: the next block's images and H / P
        // tiles are requested at the top of the step, and the updated tile of the PREVIOUS step is stored there too
        // (a write-through store issued in the middle of a step was not done at the barrier that ends it).
        //
        // Stagger (-DEVC_WIDE_STAGGER, W == 8; tried, measured, OFF): between two barriers every wavefront runs the same
        // program, so a SIMD's two wavefronts multiply together and then run the update's ~50 VALU instructions
        // together, the matrix pipe idle meanwhile (taking the update out saves 9 % of a step).  With wavefronts 4-7
        // half a step late - in step i: V' of block i-1, then D and the update of block i; a block's images then live
        // for two steps, hence three LDS stages - one wavefront's update runs beside the other's MFMAs.  Measured at
        // the STFT flow, 16 utterances: 331 us of block steps per iteration against 312 without (256 VGPRs instead of
        // 206, and the early half's first fragments of a step are no longer requested ahead): dropped.
        constexpr int NP = (MT + 1) / 2;                  // pairs of bin tiles
        constexpr bool no_mem = EVC_WIDE_ABLATE == 1 || EVC_WIDE_ABLATE == 3, no_dma = EVC_WIDE_ABLATE == 2 || EVC_WIDE_ABLATE == 3;
        const bool late = STAGGER && w >= W / 2;          // (wave-uniform)
        const bool do_d = !(kl && it == 0);
        auto frag = [&](const char* base, int u) {
            return *reinterpret_cast<const f32x4*>(base + ((u < MT ? u : MT - 1) * 64 + lane) * 16);
        };
        // Both products walk the bin tiles in pairs; the fragments of the next pair are requested before the MFMAs of
        // this pair issue (two register sets), and the first pair of the NEXT product (`next`, if any) before the
        // last pair's: LDS latency runs beside the matrix pipe.  q0/q1 carry that first pair from product to product.
        f32x4 q0 = f32x4{0, 0, 0, 0}, q1 = q0;
        auto product_d = [&](const char* img, bool preloaded, const char* next) -> f32x4 {
            // D = A_j^T Vin: two accumulation chains (one wavefront per SIMD cannot issue dependent MFMAs back to back)
            f32x4 fa[2][2];
            f32x4 d0 = f32x4{c0, c0, c0, c0}, d1 = f32x4{0, 0, 0, 0};
            fa[0][0] = preloaded ? q0 : frag(img, 0);
            fa[0][1] = preloaded ? q1 : frag(img, 1);
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                const int u = 2 * pp;
                if (pp + 1 < NP) {
                    fa[(pp + 1) & 1][0] = frag(img, u + 2);
                    fa[(pp + 1) & 1][1] = frag(img, u + 3);
                } else if (next) {
                    fa[(pp + 1) & 1][0] = frag(next, 0);
                    fa[(pp + 1) & 1][1] = frag(next, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    d0 = Mma<float>::mma(fa[pp & 1][0][r], Vin[u][r], d0);
                    if (u + 1 < MT) d1 = Mma<float>::mma(fa[pp & 1][1][r], Vin[u + 1][r], d1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            q0 = fa[NP & 1][0];
            q1 = fa[NP & 1][1];
            return d0 + d1;
        };
        auto product_v = [&](const char* img, const f32x4& hv, bool preloaded, const char* next) {
            // V' += A_j H'_j: neighbouring accumulators alternate
            f32x4 fa[2][2];
            fa[0][0] = preloaded ? q0 : frag(img, 0);
            fa[0][1] = preloaded ? q1 : frag(img, 1);
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                const int u = 2 * pp;
                if (pp + 1 < NP) {
                    fa[(pp + 1) & 1][0] = frag(img, u + 2);
                    fa[(pp + 1) & 1][1] = frag(img, u + 3);
                } else if (next) {
                    fa[(pp + 1) & 1][0] = frag(next, 0);
                    fa[(pp + 1) & 1][1] = frag(next, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Vn[u] = Mma<float>::mma(fa[pp & 1][0][r], hv[r], Vn[u]);
                    if (u + 1 < MT) Vn[u + 1] = Mma<float>::mma(fa[pp & 1][1][r], hv[r], Vn[u + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            q0 = fa[NP & 1][0];
            q1 = fa[NP & 1][1];
        };
        f32x4 hS = f32x4{0, 0, 0, 0};      // the tile to store at the top of the next step
        bool have_s = false;
        // the update of block jb: H' from H, P and the denominator (iteration 0: the start values; P goes to memory)
        auto update = [&](int jb, const f32x4& D) -> f32x4 {
            f32x4 hn;
            if (it == 0) {
                if (a.init_const) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) hn[r] = (jb * 16 + 4 * q + r < a.N) ? h0v : 0.f;
                    hS = hn;
                    have_s = true;
                } else {
                    hn = hC;
                }
                if (!kl) st_sc1(rp, (jb * 64 + lane) * 16u, D);
            } else if (EVC_WIDE_ABLATE == 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) hn[r] = hC[r] + 1e-30f * D[r];
                hS = hn;
                have_s = true;
            } else {
                switch (a.mode) {
                    case EVC_EPS_ZERO_REPLACE:                       // sklearn _nmf.py:620-629
#pragma unroll
                        for (int r = 0; r < 4; ++r) hn[r] = hC[r] * (pC[r] / (D[r] == 0.f ? a.eps : D[r]));
                        break;
                    case EVC_EPS_CLAMP:                              // deComP batch_mu.py
#pragma unroll
                        for (int r = 0; r < 4; ++r) hn[r] = hC[r] * (pC[r] / (D[r] > a.eps ? D[r] : a.eps));
                        break;
                    case WIDE_KL:                                    // sklearn _nmf.py:556-606: H (.) (A/colsum)^T R
#pragma unroll
                        for (int r = 0; r < 4; ++r) hn[r] = hC[r] * D[r];
                        break;
                    default:                                         // ADD (eps already in D), NONE: (h p) / d
#pragma unroll
                        for (int r = 0; r < 4; ++r) hn[r] = (hC[r] * pC[r]) / D[r];
                        break;
                }
                if (jb == n_edge) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) hn[r] = (jb * 16 + 4 * q + r < a.N) ? hn[r] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) hn[r] = live ? hn[r] : hC[r];     // frozen (stopped utterance) / padding
                hS = hn;
                have_s = true;
            }
            return hn;
        };

        f32x4 hP = f32x4{0, 0, 0, 0};      // late wavefronts: H' of the previous block, its V' product still to come
        const char* s2P = nullptr;         // ... and that block's V' image
        bool qa2 = false;                  // q0 / q1 hold the first fragment pair of that image
        int st = 0;                        // LDS stage of block i
        for (int i = 0; i < nb; ++i) {
            const int jb = j0 + i, stn = st == NSTAGE - 1 ? 0 : st + 1;
            const char* sb = smem + (no_dma ? 0 : st) * IMG;
            const char* s2 = sb + MT * 1024;
            if (on && have_s && !no_mem) {
                st_sc1(rh, ((jb - 1) * 64 + lane) * 16u, hS);
                if (snap_slot >= 0) st_sc1(rhs, ((jb - 1) * 64 + lane) * 16u, hS);
            }
            if (i + 1 < nb) {
                if (!no_dma) stage_block(jb + 1, stn);
                if (on && !no_mem) {
                    if (load_h) hN = ld_sc1(rh, ((jb + 1) * 64 + lane) * 16u);
                    if (load_p) pN = ld_sc1(rp, ((jb + 1) * 64 + lane) * 16u);
                }
            }
            if (on) {
                if (!late) {
                    f32x4 D = f32x4{c0, c0, c0, c0};
                    if (do_d) D = product_d(sb, false, s2);
                    const f32x4 hn = update(jb, D);
                    product_v(s2, hn, do_d, nullptr);
                } else {
                    bool qa1 = false;
                    if (s2P) {
                        product_v(s2P, hP, qa2, do_d ? sb : nullptr);
                        qa1 = do_d;
                    }
                    f32x4 D = f32x4{c0, c0, c0, c0};
                    if (do_d) D = product_d(sb, qa1, s2);
                    qa2 = do_d;
                    hP = update(jb, D);
                    s2P = s2;
                }
            }
            __syncthreads();            // (waits vmcnt(0): the next block has landed; everybody is done with this stage)
            hC = hN;
            pC = pN;
            st = stn;
        }
        if (on && late && s2P) product_v(s2P, hP, qa2, nullptr);      // the late half's last V' product

        WSTAMP(3);
        // publish the partial V' of this range (and the last block's activations)
        if (on) {
            if (have_s) {
                st_sc1(rh, ((j1 - 1) * 64 + lane) * 16u, hS);
                if (snap_slot >= 0) st_sc1(rhs, ((j1 - 1) * 64 + lane) * 16u, hS);
            }
            const __amdgpu_buffer_rsrc_t rv =
                make_rsrc(a.Vpart + (((size_t)(par * a.G + g) * c + e) * W + w) * (TILE_B / 4), TILE_B);
#pragma unroll
            for (int u = 0; u < MT; ++u)
                st_sc1(rv, (u * 64 + lane) * 16u, TG ? tag_set(Vn[u], ((unsigned)it >> 1) & 1u) : Vn[u]);
        }
        if (TG) {
            // nobody waits for a counter: the partial's floats carry the epoch bit, the H' tiles are read again by this
            // workgroup only (static schedule), kernels behind the launch see everything
            lds_barrier();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        WSTAMP(4);
        if (tid == 0) {
            if (!TG) __hip_atomic_fetch_add(a.done + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!a.static_q) nxt = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ctl[0] = nxt;
        }
        if (TG) lds_barrier(); else __syncthreads();
        WSTAMP(5);
    }
}

// ------------------------------------------------------------------------------------------
// packing / unpacking (once per call; once per dictionary when it was prepared)
// ------------------------------------------------------------------------------------------
// Aw[jb][0][u][lane = 16 q + i][r]  = A1[bin 16 u + 4 q + r][exemplar 16 jb + i]      (A operand of D; A1 = A, or A / colsum for KL)
// Aw[jb][1][u][lane = 16 q + i][r]  = A [bin 16 u + i][exemplar 16 jb + 4 q + r]      (A operand of V')
// At1 / At2: exemplars as rows (n_rows x ld, zero padded), bins < ld
__global__ __launch_bounds__(256) void k_wide_pack_dict(const float* __restrict__ At1, const float* __restrict__ At2,
                                                        int ld, int n_rows, int NB, int MT, float* __restrict__ Aw) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_block = 2L * MT * 256;
    if (gid >= (long)NB * per_block) return;
    const long jb = gid / per_block;
    const int o = (int)(gid - jb * per_block);
    const int half = o / (MT * 256), o2 = o - half * MT * 256;
    const int u = o2 >> 8, lane = (o2 >> 2) & 63, r = o2 & 3, q = lane >> 4, i = lane & 15;
    const int bin = half ? 16 * u + i : 16 * u + 4 * q + r;
    const long n = half ? 16 * jb + 4 * q + r : 16 * jb + i;
    const float* src = half ? At2 : At1;
    Aw[gid] = (n < n_rows && bin < ld) ? src[n * ld + bin] : 0.f;
}

// Xw[ft][u][lane = 16 q + i][r] = X[frame 16 ft + i][bin 16 u + 4 q + r]      (Xt: frames as rows, rows x ld, zero padded)
__global__ __launch_bounds__(256) void k_wide_pack_x(const float* __restrict__ Xt, int ld, int rows, long tiles, int MT,
                                                     float* __restrict__ Xw) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= tiles * MT * 256) return;
    const long ft = gid / (MT * 256);
    const int o = (int)(gid - ft * MT * 256);
    const int u = o >> 8, lane = (o >> 2) & 63, r = o & 3, q = lane >> 4, i = lane & 15;
    const long t = 16 * ft + i;
    const int bin = 16 * u + 4 * q + r;
    Xw[gid] = (t < rows && bin < ld) ? Xt[t * ld + bin] : 0.f;
}

// Hw[ft][jb][lane = 16 q + i][r] <-> H[exemplar 16 jb + 4 q + r][frame 16 ft + i] of the caller (frame_major: H[t ldh + n],
// else H[n ldh + t]).  Import: zero outside; export: NaN everywhere when the solve was aborted.
template <bool IMPORT>
__global__ __launch_bounds__(256) void k_wide_h_io(float* __restrict__ H, long ldh, int frame_major, int T_, int N,
                                                   f32x4* __restrict__ Hw, long tiles, int NB, int NBw, const int* abort) {
    const int lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= tiles) return;
    const long ft = tile / NB, jb = tile % NB;
    const int q = lane >> 4, i = lane & 15;
    const long t = 16 * ft + i, n0 = 16 * jb + 4 * q;
    f32x4* hw = Hw + (ft * NBw + jb) * 64 + lane;
    if (IMPORT) {
        f32x4 v = f32x4{0, 0, 0, 0};
        if (t < T_) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n0 + r < N) v[r] = frame_major ? H[t * ldh + n0 + r] : H[(n0 + r) * ldh + t];
        }
        *hw = v;
    } else {
        if (t >= T_) return;
        f32x4 v = *hw;
        if (abort && *abort) v = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
        if (frame_major && n0 + 3 < N && ((reinterpret_cast<uintptr_t>(H + t * ldh + n0) & 15) == 0)) {
            *reinterpret_cast<f32x4*>(H + t * ldh + n0) = v;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n0 + r < N) {
                    if (frame_major) H[t * ldh + n0 + r] = v[r]; else H[(n0 + r) * ldh + t] = v[r];
                }
        }
    }
}

// err2[t] = sum_m (X - V)^2 of frame t, or 2 KL(X || V) (see k_frame_err_kl), V = the sum of the c partials of
// iteration `it` (or their reduced sum); one wavefront per frame tile
__global__ __launch_bounds__(256) void k_wide_err2(WideArgs a, int MT, int W, int it, double eps, int kl,
                                                   double* __restrict__ err2) {
    const int lane = threadIdx.x & 63;
    const long ft = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ft >= a.TT) return;
    const int g = (int)(ft / W), w = (int)(ft % W);
    const unsigned par = (unsigned)(it & 1);
    const size_t tile = (size_t)MT * 64;          // float4 per tile
    const f32x4* xt = reinterpret_cast<const f32x4*>(a.Xw) + (size_t)ft * tile + lane;
    double acc = 0.0;
    for (int u = 0; u < MT; ++u) {
        f32x4 v;
        if (a.rmode) {
            v = (reinterpret_cast<const f32x4*>(a.Vsum) + ((size_t)(par * a.G + g) * W + w) * tile)[u * 64 + lane];
            if (a.tagged) {                      // (the sums carry their epoch bit: k_fused_wide's readers clear it too)
                u32x4 b = __builtin_bit_cast(u32x4, v);
                b = b & ~1u;
                v = __builtin_bit_cast(f32x4, b);
            }
        } else {
            const f32x4* p = reinterpret_cast<const f32x4*>(a.Vpart) + ((size_t)(par * a.G + g) * a.c * W + w) * tile;
            v = p[u * 64 + lane];
            for (int m = 1; m < a.c; ++m) v += p[(size_t)m * W * tile + u * 64 + lane];
        }
        const f32x4 x = xt[u * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double xd = x[r], vd = v[r];
            if (kl) {
                acc += vd;
                if (xd > eps) acc += xd * log(xd / (vd < eps ? eps : vd)) - xd;
            } else {
                acc += (xd - vd) * (xd - vd);
            }
        }
    }
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    const long t = 16 * ft + (lane & 15);
    if (lane < 16 && t < a.T_) err2[t] = kl ? 2.0 * acc : acc;
}

// Hw <- Hs for the frames of utterances that stopped at the check whose iteration count is `target` (k_utt_check has just
// set active = 0 and n_iter = target): they were iterated on to the end of the launch, the snapshot holds what the
// reference holds.  One thread per 16 bytes of the activations.
__global__ __launch_bounds__(256) void k_wide_restore(f32x4* __restrict__ Hw, const f32x4* __restrict__ Hs,
                                                      const int* __restrict__ frame_utt, const int* __restrict__ active,
                                                      const int* __restrict__ n_iter, int target, int NB, long tiles, int T_) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= tiles * NB * 64) return;
    const long ft = gid / ((long)NB * 64);
    const long t = 16 * ft + (gid & 15);
    if (t >= T_) return;
    const int ut = frame_utt[t];
    if (ut >= 0 && active[ut] == 0 && n_iter[ut] == target) Hw[gid] = Hs[gid];
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static const int WIDE_MT_SET[] = {4, 6, 8, 10, 13};

size_t wide_ctl_words(const WideLayout& f) { return 4 + 2 * (size_t)f.G; }

bool wide_supported(int M, int N, int T_, int dtype, int algo) {
    return dtype == EVC_F32 && algo == EVC_ALGO_FACTORED && M > 32 && M <= 208 && N >= 16 && T_ >= 1;
}

WideLayout wide_layout(int M, int N, int T_, int n_cus, int c_req, int w_req) {
    WideLayout f{};
    const int mt = (M + 15) / 16;
    f.MT = 13;
    for (int v : WIDE_MT_SET)
        if (v >= mt) { f.MT = v; break; }
    f.NB = (N + 15) / 16;
    f.TT = (T_ + 15) / 16;
    if (n_cus <= 0) n_cus = 256;
    // 8 wavefronts per workgroup (two per SIMD cover each other's latencies; half the dictionary traffic per flop).
    // Rounds 2-3 took 4 below 512 frame tiles (more, smaller groups); with the static schedule 8 wins at every batch
    // size measured (1 / 2 / 4 / 8 utterances: 47.8 / 71.9 / 114.5 / 213.5 us per iteration against 55.5 / 81.8 / 148.0 /
    // 230.9 - profiles/r04_wide_small_batches.md)
    f.W = w_req == 4 || w_req == 8 ? w_req : 8;
    f.G = (f.TT + f.W - 1) / f.W;
    // ranges per frame group (round 4, profiles/r04_wide_small_batches.md):
    //  * 32 groups or more (six utterances of 688 frames on): ceil(CUs / G) <= 8 ranges, every sweep task adds up the
    //    partials of its group itself (no reduce tasks) and the ticket queue absorbs the few tasks beyond the CU count
    //    (8 utterances, 43 groups: 6 ranges = 258 tasks, 0.570 of the peak; 5 ranges on the static schedule: 0.529);
    //  * fewer groups: floor(CUs / G) ranges - never more sweep tasks than workgroups, so that the static schedule
    //    applies - and, beyond four ranges, a reduce slice per task so that the traffic stays linear in c
    //    (5 utterances, 27 groups: 9 ranges + slices 0.524; 10 ranges without: 0.501).
    int c = c_req > 0 ? c_req : (f.G >= n_cus ? 1 : (f.G * 8 >= n_cus ? (n_cus + f.G - 1) / f.G : n_cus / f.G));
    const int cmax = f.NB / 2 > 0 ? f.NB / 2 : 1;
    if (c > cmax) c = cmax;
    if (c > 64) c = 64;
    f.c = c;
    f.rmode = (c > 8 || (c > 4 && f.G * c <= n_cus)) ? 1 : 0;
    f.tagged = (f.rmode && f.G * c <= n_cus) ? 1 : 0;
#ifdef EVC_WIDE_STAMP      // (the stand-alone harness only)
    if (getenv("EVC_WIDE_RMODE_MIN")) f.rmode = c >= atoi(getenv("EVC_WIDE_RMODE_MIN")) ? 1 : 0;
    if (getenv("EVC_WIDE_NO_TAGS")) f.tagged = 0;
#endif
    const size_t tile = (size_t)f.MT * 256;      // floats per V tile
    f.aw = (size_t)f.NB * 2 * tile;
    f.xw = (size_t)f.G * f.W * tile;
    f.hw = (size_t)f.G * f.W * f.NB * 256;
    f.vpart = 2 * (size_t)f.G * f.c * f.W * tile;
    f.vsum = 2 * (size_t)f.G * f.W * tile;
    return f;
}

// element counts the workspace must provide so that the automatic layout AND the tuning overrides (wavefronts per
// workgroup 4 / 8, up to 8 ranges for small batches) fit
WideCaps wide_caps(int M, int N, int T_, int n_cus) {
    const WideLayout a4 = wide_layout(M, N, T_, n_cus, 0, 4), a8 = wide_layout(M, N, T_, n_cus, 0, 8);
    WideCaps k{};
    const size_t tile = (size_t)a4.MT * 256, ttp = (size_t)round_up(a4.TT, 8);
    int c_cap = a4.c > a8.c ? a4.c : a8.c;
    if (a4.TT <= 4096 && c_cap < 8) c_cap = 8;
    if (a4.TT <= 256) c_cap = 64;
    k.c_cap = c_cap;
    k.aw = a4.aw;
    k.xw = ttp * tile;
    k.hw = ttp * a4.NB * 256;
    k.vpart = 2 * ttp * c_cap * tile;
    k.vsum = 2 * ttp * tile;
    k.ctl = 4 + 2 * (size_t)((a4.TT + 3) / 4);
    return k;
}
bool wide_fits(const WideLayout& f, const WideCaps& k) {
    return f.aw <= k.aw && f.xw <= k.xw && f.hw <= k.hw && f.vpart <= k.vpart && f.vsum <= k.vsum &&
           wide_ctl_words(f) <= k.ctl;
}

hipError_t wide_pack_dict(const WideLayout& f, const float* At1, const float* At2, int ld, int n_rows, float* Aw,
                          hipStream_t s) {
    const long n = (long)f.aw;
    hipLaunchKernelGGL(k_wide_pack_dict, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, At1, At2, ld, n_rows, f.NB,
                       f.MT, Aw);
    return hipGetLastError();
}

hipError_t wide_pack_x(const WideLayout& f, const float* Xt, int ld, int rows, float* Xw, hipStream_t s) {
    const long tiles = (long)f.G * f.W, n = tiles * f.MT * 256;
    hipLaunchKernelGGL(k_wide_pack_x, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Xt, ld, rows, tiles, f.MT, Xw);
    return hipGetLastError();
}

hipError_t wide_import_h(const WideLayout& f, float* Hw, const float* H, long ldh, int frame_major, int T_, int N,
                         hipStream_t s) {
    const long tiles = (long)f.G * f.W * f.NB;
    hipLaunchKernelGGL((k_wide_h_io<true>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, const_cast<float*>(H), ldh,
                       frame_major, T_, N, reinterpret_cast<f32x4*>(Hw), tiles, f.NB, f.NB, (const int*)nullptr);
    return hipGetLastError();
}

hipError_t wide_export_h(const WideLayout& f, const float* Hw, float* H, long ldh, int frame_major, int T_, int N,
                         const int* abort, hipStream_t s) {
    const long tiles = (long)f.TT * f.NB;
    hipLaunchKernelGGL((k_wide_h_io<false>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, H, ldh, frame_major, T_, N,
                       reinterpret_cast<f32x4*>(const_cast<float*>(Hw)), tiles, f.NB, f.NB, abort);
    return hipGetLastError();
}

static WideArgs wide_args(const WideLayout& f, const WideBuffers& b, const UttState& u, int N, int T_, int mode,
                          double eps, double l1, int init_const) {
    WideArgs a{};
    a.Aw = b.Aw; a.Xw = b.Xw; a.Hw = b.Hw; a.Pw = b.Pw; a.Vpart = b.Vpart; a.Vsum = b.Vsum;
    a.ticket = b.ctl; a.done = b.ctl + 4; a.done_r = b.ctl + 4 + f.G; a.abort = reinterpret_cast<int*>(b.ctl + 1);
    a.frame_utt = u.frame_utt; a.active = u.active; a.h0 = u.h0;
    a.NB = f.NB; a.TT = f.TT; a.G = f.G; a.c = f.c; a.rmode = f.rmode;
    a.tagged = f.tagged;
    a.N = N; a.T_ = T_; a.mode = mode; a.eps = (float)eps; a.l1 = (float)l1; a.init_const = init_const;
    return a;
}


// zero the task counters and the abort flag: once per solve, before the first wide_iterate
hipError_t wide_begin(const WideLayout& f, const WideBuffers& b, hipStream_t s) {
    return hipMemsetAsync(b.ctl, 0, wide_ctl_words(f) * sizeof(unsigned), s);
}

template <int MT, int W>
static hipError_t wide_launch(const WideArgs& a, unsigned grid, hipStream_t s) {
    const size_t lds = 3 * (size_t)(2 * MT * 1024) + 16;
    if (lds > 48 * 1024) {
        hipError_t e = a.tagged ? hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_wide<MT, W, true>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                : hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_wide<MT, W, false>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (a.tagged)
        hipLaunchKernelGGL((k_fused_wide<MT, W, true>), dim3(grid), dim3(W * 64), lds, s, a);
    else
        hipLaunchKernelGGL((k_fused_wide<MT, W, false>), dim3(grid), dim3(W * 64), lds, s, a);
    return hipGetLastError();
}

template <int W>
static hipError_t wide_dispatch(int MT, const WideArgs& a, unsigned grid, hipStream_t s) {
    switch (MT) {
        case 4: return wide_launch<4, W>(a, grid, s);
        case 6: return wide_launch<6, W>(a, grid, s);
        case 8: return wide_launch<8, W>(a, grid, s);
        case 10: return wide_launch<10, W>(a, grid, s);
        case 13: return wide_launch<13, W>(a, grid, s);
        default: return hipErrorInvalidValue;
    }
}

// iterations [it_begin, it_end) in one launch (iteration 0: P = A^T X and V = A H0); one workgroup per CU
hipError_t wide_iterate(const WideLayout& f, const WideBuffers& b, const UttState& u, int N, int T_, int it_begin,
                        int it_end, int mode, double eps, double l1, int init_const, int n_cus, hipStream_t s,
                        int snap_every, int snap_first) {
    if (it_end <= it_begin) return hipSuccess;
    WideArgs a = wide_args(f, b, u, N, T_, mode, eps, l1, init_const);
    a.it_begin = it_begin; a.it_end = it_end;
    a.snap_every = (snap_every > 0 && b.Hs && b.err2s) ? snap_every : 0;
    a.snap_first = snap_first;
    a.Hs = b.Hs; a.hs_stride = b.hs_stride; a.err2s = b.err2s; a.err_stride = b.err_stride;
    if (a.snap_every > 0 && (it_end - 2 - snap_first) / a.snap_every >= b.snap_slots) return hipErrorInvalidValue;
    const long per_it = (long)f.G * f.c * (f.rmode ? 2 : 1);
    const long tasks = per_it * (it_end - it_begin);
    if (n_cus <= 0) n_cus = 256;
#ifdef EVC_WIDE_STAMP      // (the stand-alone harnesses only: the library reads nothing from the environment)
    static const int no_static = getenv("EVC_WIDE_NO_STATIC") ? atoi(getenv("EVC_WIDE_NO_STATIC")) : 0;
#else
    constexpr int no_static = 0;
#endif
    // (two 4-wavefront workgroups per CU, one sweeping while the other exchanges, were tried: 90 us per iteration at one
    // utterance against 48 - twice the partial sums to exchange, and the two share the matrix pipe)
    a.static_q = (f.G * f.c <= n_cus && !no_static) ? 1 : 0;
    a.tagged = (a.static_q && f.tagged) ? 1 : 0;
    if (a.tagged && it_begin == 0) {
        // stale floats must not carry the epoch bit of the first two iterations (0): ones everywhere
        hipError_t em = hipMemsetAsync(b.Vpart, 0xFF, f.vpart * sizeof(float), s);
        if (em == hipSuccess) em = hipMemsetAsync(b.Vsum, 0xFF, f.vsum * sizeof(float), s);
        if (em != hipSuccess) return em;
    }
    const unsigned grid = a.static_q ? (unsigned)(f.G * f.c) : (unsigned)(tasks < n_cus ? tasks : n_cus);
    // the ticket counter of this launch: tickets 0 .. grid-1 belong to the workgroups by index
    hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b.ctl), (int)grid, 1, s);
    if (e != hipSuccess) return e;
    return f.W == 8 ? wide_dispatch<8>(f.MT, a, grid, s) : wide_dispatch<4>(f.MT, a, grid, s);
}

hipError_t wide_restore(const WideLayout& f, const WideBuffers& b, const UttState& u, int slot, int target_iter, int T_,
                        hipStream_t s) {
    if (!b.Hs || slot < 0 || slot >= b.snap_slots) return hipErrorInvalidValue;
    const long tiles = (long)f.G * f.W, n = tiles * f.NB * 64;
    hipLaunchKernelGGL(k_wide_restore, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<f32x4*>(b.Hw),
                       reinterpret_cast<const f32x4*>(b.Hs + (size_t)slot * b.hs_stride), u.frame_utt, u.active, u.n_iter,
                       target_iter, f.NB, tiles, T_);
    return hipGetLastError();
}

// per-frame residuals of the activations after iteration `it` (the V' that iteration published)
hipError_t wide_err2(const WideLayout& f, const WideBuffers& b, const UttState& u, int N, int T_, int it, int kl,
                     double eps, double* err2, hipStream_t s) {
    WideArgs a = wide_args(f, b, u, N, T_, 0, eps, 0.0, 0);
    hipLaunchKernelGGL(k_wide_err2, dim3((unsigned)((f.TT + 3) / 4)), dim3(256), 0, s, a, f.MT, f.W, it, eps, kl, err2);
    return hipGetLastError();
}

}  // namespace evc
