// Register-resident variant of the fused FACTORED kernel (float64, M <= 32).
//
// evc_fused.hip streams every 16x16 activation tile through HBM once per iteration (one read,
// one write): 2*N*8 bytes per frame-iteration, which at N = 4096 puts that kernel against the
// HBM wall (4.4 TB/s measured at 666 k frames/s).  Here each of the 8 wavefronts of a workgroup
// keeps RES of its tiles in VGPRs for the whole launch (RES = 16: 8 x 16 x 2 KiB = 256 KiB of
// the CU's 512 KiB register file = half of a 16-frame column block at N = 4096) and streams only
// the others.  A wavefront walks its tiles k = 0, 1, 2, ... (tile index w + 8k); among the first
// 2*RES the even ones are resident, the odd ones streamed, so that
//   * a streamed tile's load is issued two units (~3300 cycles) before its first use at the cost
//     of one extra tile of registers;
//   * the dictionary fragments of unit k+1 are requested as soon as the last D/P MFMA of unit k
//     has issued (its operand registers are free from then on).
//
// This kernel is the straight-line fast path only; it requires
//   N >= 1024 (fused_layout pads the tile count to whole rounds of the 8 wavefronts from there on:
//   every wavefront owns the same number of tiles, >= 2*RES; padding exemplars are all-zero),
//   a guarded eps mode (not NONE),
//   every frame of the workgroup live (utterance still active) - otherwise the workgroup
//   returns immediately and evc_fused.hip's kernel, launched behind it with skip_all_live = 1,
//   processes it; that kernel also does the V pre-pass of the first launch.
// Padded frames of the last workgroup (x = 0, h = 0) need no masking: their denominators are 0,
// which sends the tile through mu_tile's exact path and leaves h at exactly 0.
// The quotient is always formed divide-first, h * (p / den); for the ADD mode this differs from
// pymf's (h*p)/den in the last bit only.
#include "evc_fused_common.h"

#ifndef EVC_RES_MAX
#define EVC_RES_MAX 16
#endif
#ifndef EVC_RES_PL
#define EVC_RES_PL 7      // numerator tiles per wavefront cached in LDS (8 x 7 x 2 KiB = 112 KiB)
#endif

namespace evc {

constexpr int RNW = 8;    // wavefronts per workgroup

// KL: the generalised Kullback-Leibler update (A1p holds the dictionary divided by its column sums; the
// per-unit work is one MFMA chain (A_j/colsum)^T (X / max(V, eps)) and a multiply, no division).
//
// COOP (few frame tiles, e.g. one utterance = 43 workgroups on 256 CUs): coop_c workgroups share one frame
// tile, workgroup `member` owning exemplar tiles [member NT/c, (member+1) NT/c).  After the intra-workgroup
// combine every thread publishes its element of the partial V' with an agent-scope atomic store (these bypass
// the per-XCD L2 both ways; an agent-scope release/acquire pair instead would write back and invalidate the
// whole L2 every iteration and lose the dictionary fragments cached there - measured 45 us per iteration) and
// polls the same element of its peers.  There is no separate flag: the lowest mantissa bit of the word is the
// buffer's epoch (readers clear it), so one memory round trip carries data and arrival; a counter + data
// protocol costs five serialised round trips (measured 19 us per iteration against 9 us of compute).  All
// members sum the c words in member order and obtain the bitwise identical V'.  Two buffers alternate by
// iteration parity: a member writes iteration i+2 only after it has read every peer's word of iteration i+1,
// which the peer wrote after reading this member's word of iteration i.  The host launches TT*c <= #CUs
// workgroups of one-per-CU size, so all members are resident; a poll budget bounds every wait, and a
// workgroup that exhausts it raises coop_abort, on which all leave (no hang); the host then redoes the solve
// with one workgroup per tile (two processes sharing a GPU can starve each other's members of CUs).
constexpr unsigned COOP_POLL_LIMIT = 1u << 18;     // a fraction of a second; a peer's sweep takes microseconds

template <int MSTEPS, int RES, int PL, bool KL, bool COOP = false>
__global__ __launch_bounds__(RNW * 64) void k_fused_res(FusedArgs a) {
    constexpr int MT = MSTEPS > 4 ? 2 : 1;
    constexpr int E = MT * 4 * 64;               // doubles in one V (accumulator order)
    extern __shared__ double lds[];
    double* red = lds;                           // [RNW][E]  partial V' of every wavefront
    double* vL = lds + RNW * E;                  // [MT*4][64]  V, B-operand order
    double* xL = vL + E;                         // [MT*4][64]  X, B-operand order
    double* rL = xL + E;                         // [MT*4][64]  X / max(V, eps): KL numerator operand
    // numerator tiles P = A_j^T X of the first PL tiles of every wavefront: computed in the first sweep
    // of a launch, read back afterwards (7 fewer MFMAs per unit; LDS would otherwise sit idle)
    f64x2* pL = reinterpret_cast<f64x2*>(rL + E) + (size_t)(threadIdx.x >> 6) * (PL * 128);   // [PL][2][64] per wavefront
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cc = COOP ? a.coop_c : 1;
    const long tt = COOP ? blockIdx.x / cc : blockIdx.x;
    const int member = COOP ? (int)(blockIdx.x % cc) : 0;
    const double* __restrict__ A1p = a.A1p;
    const double* __restrict__ A2p = a.A2p;
    f64x2* __restrict__ Hp = a.Hp;
    const int NT = a.NT;
    const int KT = NT / cc / RNW;                // tiles per wavefront (host guarantees >= 2*RES)
    const long tile0 = (long)member * (NT / cc);  // first exemplar tile of this workgroup

    {
        const long t = 16 * tt + (lane & 15);
        int u = -1;
        if (t < a.T_) u = a.frame_utt[t];
        const bool live = (t >= a.T_) || ((u >= 0) && (a.active[u] != 0));   // padding counts as live
        if (!__syncthreads_and(live)) return;    // left to the general kernel (skip_all_live)
    }

    for (int e = tid; e < E; e += RNW * 64) {
        const int s = e >> 6, l = e & 63;
        const bool in = s < MSTEPS;
        const double x = in ? a.Xp[(tt * MSTEPS + s) * 64 + l] : 0.0;
        const double v = in ? a.Vp[(tt * 8 + s) * 64 + l] : 0.0;
        xL[e] = x;
        vL[e] = v;
        rL[e] = x / (v < a.eps ? a.eps : v);
    }
    __syncthreads();

    // Per-tile base addresses are wave-uniform (SGPR base + unsigned per-lane offset -> the
    // global_load saddr form).  With per-lane 64-bit addresses the unrolled sweep needs one VGPR
    // pair per tile and array (the 16 KiB tile stride does not fit the instruction's immediate
    // offset) and spills massively.  `sw` is an opaque zero added to the offsets once per sweep so
    // that the ~100 tile addresses are re-derived with scalar adds instead of being hoisted out of
    // the iteration loop into SGPRs that then spill.
    const unsigned ul = (unsigned)lane;
    long sw = 0;
    constexpr int MSP = (MSTEPS + 1) & ~1;                            // k-steps padded to pairs in A1p
    const long a1w = (tile0 + w) * MSP * 64;                          // tile k: + k * RNW*MSP*64
    const long a2w = (tile0 + w) * MT * 256;                          // tile k: + k * RNW*MT*256
    const long hw = (tt * NT + tile0 + w) * 128;                      // tile k: + k * RNW*128
    auto load_a1 = [&](double (&a1)[MSTEPS], int k) {
        const f64x2* t = reinterpret_cast<const f64x2*>(A1p + (a1w + sw + (long)k * (RNW * MSP * 64)));
#pragma unroll
        for (int s = 0; s < MSTEPS; s += 2) {
            if (s + 1 < MSTEPS) {
                const f64x2 v = t[(s >> 1) * 64 + ul];
                a1[s] = v[0];
                a1[s + 1] = v[1];
            } else {
                // odd tail: load 8 bytes only - a 16-byte load would leave a dead register pair that the
                // allocator reuses while the load is still in flight (write-after-write => vmcnt(0))
                a1[s] = reinterpret_cast<const double*>(&t[(s >> 1) * 64 + ul])[0];
            }
        }
    };
    auto load_a2 = [&](double (&a2)[MT][4], int k) {
        const f64x2* t = reinterpret_cast<const f64x2*>(A2p + (a2w + sw + (long)k * (RNW * MT * 256)));
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f64x2 v = t[(u * 2 + (r >> 1)) * 64 + ul];
                a2[u][r] = v[0];
                a2[u][r + 1] = v[1];
            }
    };
    auto load_h = [&](HTile& h, int k) {
        const f64x2* t = Hp + (hw + sw + (long)k * (RNW * 128));
        const f64x2 h01 = t[ul], h23 = t[ul + 64];     // (non-temporal accesses measured 8 % slower)
        h[0] = h01[0]; h[1] = h01[1]; h[2] = h23[0]; h[3] = h23[1];
    };
    auto store_h = [&](const HTile& h, int k) {
        f64x2* t = Hp + (hw + sw + (long)k * (RNW * 128));
        t[ul] = f64x2{h[0], h[1]};
        t[ul + 64] = f64x2{h[2], h[3]};
    };
    auto vacc = [&](const double (&a2)[MT][4], const HTile& h, f64x4 (&vn)[MT]) {
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) vn[u] = Mma<double>::mma(a2[u][r], h[r], vn[u]);
    };

    const int mode = a.eps_mode;
    const double eps = a.eps;
    // l1 (sklearn _nmf.py:615-617) and pymf's +eps (nmf.py:68) ride in the accumulator's start value
    const double d0 = a.l1 + (mode == EVC_EPS_ADD ? a.eps : 0.0);
    const f64x4 dinit = {d0, d0, d0, d0};
    const unsigned lo = fast_lo(mode, eps);

    // one unit: D = A_j^T V (+l1), P = A_j^T X, h <- h * P / D, V' += A_j h
    int it = 0;
    // D (and P, unless tile k's numerator is cached in LDS: k < PL and not the first sweep)
    auto dp = [&](const double (&a1)[MSTEPS], f64x4& d, f64x4& p, int k) {
        if (KL) {
#pragma unroll
            for (int s = 0; s < MSTEPS; ++s) p = Mma<double>::mma(a1[s], rL[s * 64 + lane], p);
        } else if (k < PL && it > 0) {
#pragma unroll
            for (int s = 0; s < MSTEPS; ++s) d = Mma<double>::mma(a1[s], vL[s * 64 + lane], d);
            const f64x2 p01 = pL[(k * 2) * 64 + lane], p23 = pL[(k * 2 + 1) * 64 + lane];
            p = f64x4{p01[0], p01[1], p23[0], p23[1]};
        } else {
#pragma unroll
            for (int s = 0; s < MSTEPS; ++s) {
                d = Mma<double>::mma(a1[s], vL[s * 64 + lane], d);
                p = Mma<double>::mma(a1[s], xL[s * 64 + lane], p);
            }
            if (k < PL) {
                pL[(k * 2) * 64 + lane] = f64x2{p[0], p[1]};
                pL[(k * 2 + 1) * 64 + lane] = f64x2{p[2], p[3]};
            }
        }
    };
    auto unit = [&](double (&a1)[MSTEPS], double (&a2)[MT][4], HTile& h, f64x4 (&vn)[MT], int k,
                    bool more) {
        __builtin_amdgcn_sched_barrier(0);       // keep the unrolled units' loads where they are written
        load_a2(a2, k);
        f64x4 d = dinit, p = {0, 0, 0, 0};
        dp(a1, d, p, k);
        if (more) load_a1(a1, k + 1);            // operand registers are free once the MFMAs issued
        __builtin_amdgcn_sched_barrier(0);
        if (KL) { for (int r = 0; r < 4; ++r) h[r] *= p[r]; }
        else mu_tile<false>(h, p, d, mode, eps, lo);
    };

    HTile hres[RES];
#pragma unroll
    for (int k = 0; k < RES; ++k) load_h(hres[k], 2 * k);

    double a1[MSTEPS], a2[MT][4];
    // The two streamed-tile buffers are separate arrays on purpose: as one `hs[2][4]` they are
    // promoted to a single 16-register tuple, and the update of one half then copies - and waits
    // for - the other half, which has its prefetch in flight.
    static_assert(RES % 2 == 0, "the streamed buffers alternate in pairs of units");
    HTile hsa, hsb;
    for (it = 0; it < a.iters; ++it) {
        // opaque to the optimiser: the per-tile addresses are re-derived with scalar adds in every
        // sweep instead of being hoisted out of this loop into ~200 SGPRs (which then spill)
        asm volatile("" : "+s"(sw));
        f64x4 vn[MT];
#pragma unroll
        for (int u = 0; u < MT; ++u) vn[u] = f64x4{0, 0, 0, 0};
        if (it == 0) {
            load_a1(a1, 0);
            load_h(hsa, 1);
        }
        // resident tile 2*k2, then streamed tile 2*k2+1 held in `cur`; `nxt` receives tile 2*k2+3
        auto pair = [&](int k2, HTile& cur, HTile& nxt) {
            unit(a1, a2, hres[k2], vn, 2 * k2, true);
            vacc(a2, hres[k2], vn);
            const bool more = (2 * k2 + 2 < KT);
            __builtin_amdgcn_sched_barrier(0);
            load_a2(a2, 2 * k2 + 1);
            {
                f64x4 d = dinit, p = {0, 0, 0, 0};
                dp(a1, d, p, 2 * k2 + 1);
                if (more) load_a1(a1, 2 * k2 + 2);
                __builtin_amdgcn_sched_barrier(0);
                if (KL) { for (int r = 0; r < 4; ++r) cur[r] *= p[r]; }
                else mu_tile<false>(cur, p, d, mode, eps, lo);
            }
            // the next streamed tile's load goes out here, two units ahead of its use (after the update:
            // registers with a load in flight must not cross the update's rare-path merge)
            if (k2 + 1 < RES) load_h(nxt, 2 * k2 + 3);
            store_h(cur, 2 * k2 + 1);
            vacc(a2, cur, vn);
        };
#pragma unroll
        for (int k2 = 0; k2 < RES; k2 += 2) {
            pair(k2, hsa, hsb);
            pair(k2 + 1, hsb, hsa);
        }
        for (int k = 2 * RES; k < KT; ++k) {     // beyond the resident window: plain streaming
            HTile h;
            load_h(h, k);
            unit(a1, a2, h, vn, k, k + 1 < KT);
            store_h(h, k);
            vacc(a2, h, vn);
        }
        // the next sweep's first operands do not depend on V': request them before the combine
        if (it + 1 < a.iters) {
            load_a1(a1, 0);
            load_h(hsa, 1);        // tile 1 was stored earlier in this sweep by this very wavefront
        }
        // V' partials -> LDS -> each wavefront sums a slice over the partials in fixed order -> vL
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[w * E + (u * 4 + r) * 64 + lane] = vn[u][r];
        __syncthreads();
        long long* slot = nullptr;
        if (COOP) slot = reinterpret_cast<long long*>(a.coop_buf) + ((size_t)(it & 1) * a.TT + tt) * cc * E;
        const long long tag = (it >> 1) & 1;       // a buffer is reused every second iteration: its epoch bit flips
        bool ok = true;
        for (int e = w * 64 + lane; e < E; e += RNW * 64) {
            double acc = 0.0;
#pragma unroll
            for (int ww = 0; ww < RNW; ++ww) acc += red[ww * E + e];
            if (COOP) {
                // publish: the value with its lowest mantissa bit replaced by the epoch bit (readers clear it again,
                // so every member - this one included - sums the same c numbers, each within 1 ulp of the partial)
                const long long mine = (__double_as_longlong(acc) & ~1LL) | tag;
                __hip_atomic_store(slot + (size_t)member * E + e, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // gather: poll the peers' words until they carry this epoch (stale words carry the other bit: the
                // host fills the buffers with ones before the launch, iterations 2k and 2k+1 write epoch k & 1)
                double sum = 0.0;
                for (int m0 = 0; m0 < cc; m0 += 2) {
                    long long b0 = mine, b1 = mine;
                    unsigned polls = 0;
                    for (;;) {
                        if (m0 != member) b0 = __hip_atomic_load(slot + (size_t)m0 * E + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (m0 + 1 != member) b1 = __hip_atomic_load(slot + (size_t)(m0 + 1) * E + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (((b0 & 1) == tag) && ((b1 & 1) == tag)) break;
                        if (++polls > COOP_POLL_LIMIT ||
                            ((polls & 1023) == 0 &&
                             __hip_atomic_load(a.coop_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                            ok = false;
                            break;
                        }
                    }
                    if (!ok) break;
                    sum += __longlong_as_double(b0 & ~1LL);       // member order: identical on every member
                    sum += __longlong_as_double(b1 & ~1LL);
                }
                acc = sum;
            }
            vL[e] = acc;
            if (KL) rL[e] = xL[e] / (acc < a.eps ? a.eps : acc);     // sklearn _nmf.py:572-576
        }
        if (COOP) {
            if (!__syncthreads_and(ok)) {        // a peer never showed up: void the launch, let everybody leave
                if (tid == 0) __hip_atomic_store(a.coop_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        } else {
            __syncthreads();
        }
    }

#pragma unroll
    for (int k = 0; k < RES; ++k) store_h(hres[k], 2 * k);

    // carry V to the next launch; per-frame squared residual of the final activations
    if (COOP && member != 0) return;             // every member holds the same V: one writes it
    for (int e = tid; e < E; e += RNW * 64) {
        const int s = e >> 6, l = e & 63;
        if (s < MSTEPS) a.Vp[(tt * 8 + s) * 64 + l] = vL[e];
    }
    if (a.write_err && w == 0) {
        double e = 0.0;
#pragma unroll
        for (int s = 0; s < MSTEPS; ++s) {
            const double x = xL[s * 64 + lane], v = vL[s * 64 + lane];
            e += KL ? kl_terms(x, v, a.eps) : (x - v) * (x - v);
        }
        e += __shfl_xor(e, 16, 64);      // the 4 lane groups hold one frame's bins
        e += __shfl_xor(e, 32, 64);
        const long t = 16 * tt + lane;
        if (lane < 16 && t < a.T_) a.err2[t] = e;
    }
}

template <int MSTEPS, int RES, bool KL, bool COOP>
static hipError_t launch_res(const FusedArgs& a, hipStream_t s) {
    constexpr int MT = MSTEPS > 4 ? 2 : 1;
    constexpr int E = MT * 4 * 64;
    constexpr int PL = KL ? 0 : EVC_RES_PL;
    const size_t lds = (size_t)(RNW + 3) * E * sizeof(double) + (size_t)RNW * PL * 256 * sizeof(double);
    if (lds > 64 * 1024) {   // per launch: no mutable global state is kept
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_res<MSTEPS, RES, PL, KL, COOP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const unsigned grid = (unsigned)a.TT * (COOP ? (unsigned)a.coop_c : 1u);
    hipLaunchKernelGGL((k_fused_res<MSTEPS, RES, PL, KL, COOP>), dim3(grid), dim3(RNW * 64), lds, s, a);
    return hipGetLastError();
}

template <int MSTEPS, bool KL>
static hipError_t pick_res(const FusedArgs& a, hipStream_t s) {
    if (a.coop_c > 1) {
        // cooperative: NT / coop_c / 8 tiles per wavefront (the host picked coop_c so that this is >= 8)
        if (!a.coop_buf || !a.coop_cnt || !a.coop_abort || a.NT % (a.coop_c * RNW)) return hipErrorInvalidValue;
        if ((long)a.TT * a.coop_c > COOP_MAX_TILES) return hipErrorInvalidValue;
        // stale words must not carry the epoch bit of the first two iterations (0): fill with ones
        hipError_t e = hipMemsetAsync(a.coop_buf, 0xFF, sizeof(double) * 2 * a.TT * a.coop_c * 512, s);
        if (e != hipSuccess) return e;
        const int KT = a.NT / a.coop_c / RNW;
        if (KT >= 2 * EVC_RES_MAX) return launch_res<MSTEPS, EVC_RES_MAX, KL, true>(a, s);
        if (KT >= 16) return launch_res<MSTEPS, 8, KL, true>(a, s);
        if (KT >= 8) return launch_res<MSTEPS, 4, KL, true>(a, s);
        return hipErrorInvalidValue;
    }
    const int KT = a.NT / RNW;
    if (KT >= 2 * EVC_RES_MAX) return launch_res<MSTEPS, EVC_RES_MAX, KL, false>(a, s);
    if (KT >= 16) return launch_res<MSTEPS, 8, KL, false>(a, s);
    if (KT >= 8) return launch_res<MSTEPS, 4, KL, false>(a, s);
    return hipErrorInvalidValue;
}

// Largest power-of-two number of cooperating workgroups per frame tile such that every workgroup is resident
// (one per CU), every wavefront still owns >= 8 exemplar tiles, and at least half of the CUs would idle without it.
int fused_res_coop_factor(int NT, int TT, int n_cus) {
    if (n_cus > COOP_MAX_TILES) n_cus = COOP_MAX_TILES;
    int c = 1;
    while (2 * c * TT <= n_cus && NT % (2 * c * RNW) == 0 && NT / (2 * c * RNW) >= 8) c *= 2;
    return c;
}

template <int MSTEPS>
static hipError_t pick_loss(const FusedArgs& a, hipStream_t s) {
    return a.loss == EVC_LOSS_KL ? pick_res<MSTEPS, true>(a, s) : pick_res<MSTEPS, false>(a, s);
}

bool fused_res_supported(int N, int eps_mode, int exact_div) {
    return N >= 1024 && eps_mode != EVC_EPS_NONE && !exact_div;      // fused_layout pads N to 128 from there on
}

hipError_t fused_res_launch(int msteps, const FusedArgs& a, hipStream_t s) {
    switch (msteps) {
        case 1: return pick_loss<1>(a, s);
        case 2: return pick_loss<2>(a, s);
        case 3: return pick_loss<3>(a, s);
        case 4: return pick_loss<4>(a, s);
        case 5: return pick_loss<5>(a, s);
        case 6: return pick_loss<6>(a, s);
        case 7: return pick_loss<7>(a, s);
        case 8: return pick_loss<8>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace evc
