// Shared by evc_fused.hip (general streamed kernel) and evc_fused_res.hip (register-resident
// kernel): packed layouts, kernel arguments and the element-wise update.
#pragma once
#include "evc_internal.h"

#include <type_traits>

namespace evc {

typedef double f64x2 __attribute__((ext_vector_type(2)));

__host__ __device__ inline int fused_msteps(int M) { return M <= 16 ? (M + 3) / 4 : 4 + (M - 16 + 3) / 4; }
// bin handled by k-step s for lane group q
__device__ __forceinline__ int bin_of(int s, int q) { return 16 * (s >> 2) + q + 4 * (s & 3); }

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
    int skip_all_live;       // 1: workgroups whose frames are all live were done by k_fused_res
    int force_live;          // 1: ignore the utterances' active flags (synthesis pre-pass)
    int exact_div;           // 1: correctly rounded quotients on the fast path too (pymf's 2.2e-16 stop rule)
    int loss;                // EVC_LOSS_*; for KL A1p holds the dictionary pre-divided by its column sums
    int eps_mode;
    double eps, l1;
    // cooperative launch (k_fused_res, few frame tiles): coop_c workgroups share one 16-frame tile, each owning
    // NT / coop_c exemplar tiles; they exchange their V' partials through coop_buf once per iteration
    int coop_c;              // 1 = off
    double* coop_buf;        // [2][TT][coop_c][E]
    int* coop_cnt;           // [TT] arrival counters, zero at launch
    int* coop_abort;         // set when a wait timed out: the launch's results are void
    int groups;              // k_fused_all: groups of coop_c workgroups walking the frame tiles (set by the launcher)
    // k_fused_all, first launch of a solve with a constant start value per utterance (EVC_INIT_SKLEARN / CONST):
    // H = h0[utterance] (0 in the padding) and V = A H = h0 * rowsum(A) are formed in the kernel, so neither the
    // fill of the packed activations nor the V = A H pre-pass runs (2.6 of 184 ms at C2)
    int init_const;
    const double* h0;        // [n_utt]
    const double* rsum;      // [32] row sums of the dictionary (bins), 0 beyond M
    // k_fused_all, last launch of a solve in which no utterance can stop: the activations also go straight to the
    // caller's matrix (the separate export pass reads and writes all of H once more: 2.6 of 180 ms at C2)
    double* Hx;              // NULL: off
    long ldhx;
    int hx_frame_major;      // 1: Hx[t * ldhx + n], 0: Hx[n * ldhx + t]
    // k_fused_xy: bins of the dictionary (0: unknown) and the lane group of the last k-step that holds the spare bin M
    // in which the denominators' start value travels (-1: not used); see evc_fused_xy.hip
    int M, spare_q;
    long long stagger_cycles;    // k_fused_all: every second pair of groups starts this many shader cycles late (0: off)
#if defined(EVC_ALL_TIMING) || defined(EVC_XY_TIMING)
    long long* dbg;          // tools/ubench/fused_all_bench.hip: s_memtime stamps of the first round's steps
#endif
};

// One tile's four activations of a lane.  The elements sit 16 bytes apart on purpose: stored back to
// back, the optimiser merges their stores into vector stores and promotes the array to one 8-register
// tuple, and a tuple crosses the update's rare-path merge as a unit (register copies on the fast path).
struct HTile {
    double v[4][2];
    __device__ __forceinline__ double& operator[](int r) { return v[r][0]; }
    __device__ __forceinline__ const double& operator[](int r) const { return v[r][0]; }
};

// The update with the guard mode as a compile-time constant (the switch is hoisted out of the
// sweep).  On gfx950 an f64 MFMA and any VALU instruction of the same SIMD do not overlap
// (tools/ubench/mfma_valu_f64.hip: times add), so the VALU instruction count of this function is
// directly MFMA time lost.  Hence:
//   * l1 (and pymf's additive eps) are folded into the initial value of the D accumulator, so
//     `dacc` arrives as the finished denominator sum;
//   * one unsigned range test per tile (high words, v_max3) decides between the fast path and the
//     exact path.  Fast path = every denominator is a normal number in [lo, 2^250): then neither
//     the ==0 replacement nor the clamp can fire, and the four reciprocals of a tile come from ONE
//     v_rcp_f64 (a quarter-rate instruction) by batch inversion: R = 1/(d0 d1 d2 d3) with one Newton
//     step (v_rcp_f64 is good to 2^-24.4, one step gives 2.3e-15: tools/ubench/rcp_accuracy.hip),
//     1/d0 = R (d2 d3) d1 and so on - 3 + 2 + 6 multiplies/FMAs instead of 3 more v_rcp and 6 more
//     Newton FMAs; products of four numbers in [2^-250, 2^250) cannot leave the normal range.  Then
//     q = num * (1/d): no v_div_scale/fmas/fixup, no selects.  The quotient's ~3e-15 relative error
//     is the size of the rounding noise of the N-term denominator sums next to it;
//   * the exact path (zero / denormal / huge / NaN denominators, and the unguarded NONE mode
//     always) applies the guard literally and divides with IEEE semantics, so inf/NaN behaviour
//     is the reference's.
// num/den for a normal den in [2^-250, 2^250): v_rcp_f64, one Newton step, one residual correction.
// Correctly rounded on every sample of tools/ubench/rcp_accuracy.hip (2^30 quotients): the iteration
// then reaches the same floating-point fixed points as IEEE division, which pymf's stop rule
// (|ferr_i - ferr_{i-1}| / T < 2.2e-16, base.py:189-206) needs in order to fire at the same iteration.
__device__ __forceinline__ double exact_div(double num, double den) {
    double r = __builtin_amdgcn_rcp(den);
    const double e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double qv = num * r;
    const double rem = __builtin_fma(-den, qv, num);
    return __builtin_fma(rem, r, qv);
}
// 1/d[0..3] for four normal numbers in [2^-250, 2^250)
__device__ __forceinline__ void batch_rcp(const f64x4& d, double (&r)[4]) {
    const double a = d[0] * d[1], b = d[2] * d[3];
    const double ab = a * b;
    double R = __builtin_amdgcn_rcp(ab);
    const double e = __builtin_fma(-ab, R, 1.0);
    R = __builtin_fma(R, e, R);
    const double Ra = R * b, Rb = R * a;
    r[0] = Ra * d[1];
    r[1] = Ra * d[0];
    r[2] = Rb * d[3];
    r[3] = Rb * d[2];
}
__device__ __forceinline__ unsigned hi_word(double x) { return (unsigned)(__double_as_longlong(x) >> 32); }
// lowest admissible high word for the fast path: 2^-250, or one binade above eps when clamping
constexpr unsigned FAST_HI_WORD = 0x4F900000u;      // 2^250
__device__ __forceinline__ unsigned fast_lo(int mode, double eps) {
    unsigned lo = 0x30500000u;                      // 2^-250
    if (mode == EVC_EPS_CLAMP && eps > 0) {
        const unsigned e = hi_word(eps) + 0x00200000u;
        lo = e > lo ? e : lo;
    }
    return lo;
}
// MUL_FIRST: pymf / nmf_tool form (h*p)/den, else sklearn / deComP form h*(p/den).  `mode` is only
// consulted on the exact path.
template <bool MUL_FIRST, bool EXACT_DIV = false, class HT>
__device__ __forceinline__ void mu_tile(HT& h, const f64x4& p, const f64x4& dacc, int mode,
                                        double eps, unsigned lo) {
    const unsigned span = FAST_HI_WORD > lo ? FAST_HI_WORD - lo : 0u;
    const unsigned worst = max(max(hi_word(dacc[0]) - lo, hi_word(dacc[1]) - lo),
                               max(hi_word(dacc[2]) - lo, hi_word(dacc[3]) - lo));
    if (__builtin_expect(mode != EVC_EPS_NONE && __all(worst < span), 1)) {
        double rc[4];
        if (!EXACT_DIV) batch_rcp(dacc, rc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double num = MUL_FIRST ? h[r] * p[r] : p[r];
            const double qv = EXACT_DIV ? exact_div(num, dacc[r]) : num * rc[r];
            h[r] = MUL_FIRST ? qv : h[r] * qv;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double dn = dacc[r];
            dn = (mode == EVC_EPS_ZERO_REPLACE && dn == 0.0) ? eps : dn;   // sklearn _nmf.py:620
            dn = (mode == EVC_EPS_CLAMP && !(dn > eps)) ? eps : dn;         // deComP
            const double qv = (MUL_FIRST ? h[r] * p[r] : p[r]) / dn;
            h[r] = MUL_FIRST ? qv : h[r] * qv;
        }
    }
}

// mu_tile for the guarded modes only (k_fused_all, k_fused_xy never run EVC_EPS_NONE: no test of the mode in front of
// every tile - a scalar branch and two mask moves per unit), divide-first, and with the exact path taken one quotient at
// a time: it is the rare path (zero / denormal /
// huge denominators, tiles holding padding exemplars under sklearn's guard), and four interleaved IEEE divisions need ~40
// registers at the one point of the sweep where everything else is live too - enough to push activation tiles into
// scratch memory for the whole loop.
template <class HT>
__device__ __forceinline__ void mu_tile_guarded(HT& h, const f64x4& p, const f64x4& dacc, int mode, double eps, unsigned lo) {
    // the fast quotients are formed unconditionally (straight-line code behind the D chain); the wavefront overwrites
    // them on the exact path when any lane's denominators are out of the fast range
    const unsigned span = FAST_HI_WORD > lo ? FAST_HI_WORD - lo : 0u;
    const unsigned worst = max(max(hi_word(dacc[0]) - lo, hi_word(dacc[1]) - lo),
                               max(hi_word(dacc[2]) - lo, hi_word(dacc[3]) - lo));
    double rc[4], q[4];
    batch_rcp(dacc, rc);
#pragma unroll
    for (int r = 0; r < 4; ++r) q[r] = p[r] * rc[r];
    if (__builtin_expect(!__all(worst < span), 0)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double dn = dacc[r];
            dn = (mode == EVC_EPS_ZERO_REPLACE && dn == 0.0) ? eps : dn;   // sklearn _nmf.py:620
            dn = (mode == EVC_EPS_CLAMP && !(dn > eps)) ? eps : dn;         // deComP
            q[r] = p[r] / dn;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = h[r] * q[r];
}

// The same update split for software pipelining (k_fused_all): the quotients q = p / guard(d) of a tile.
// mu_quot_fast is branch-free - it belongs to the basic block of the MFMAs it is interleaved with - and
// returns whether every denominator of the lane allowed the fast path (its q is meaningless otherwise);
// mu_quot_exact is the literal guard + IEEE division, taken by the whole wavefront when any lane needs it.
// The update is then h <- h * q (divide first; for the ADD mode this differs from pymf's (h*p)/d in the last
// bit only, like in k_fused_res).
__device__ __forceinline__ bool mu_quot_fast(const f64x4& p, const f64x4& dacc, unsigned lo, double (&q)[4]) {
    const unsigned span = FAST_HI_WORD > lo ? FAST_HI_WORD - lo : 0u;
    const unsigned worst = max(max(hi_word(dacc[0]) - lo, hi_word(dacc[1]) - lo),
                               max(hi_word(dacc[2]) - lo, hi_word(dacc[3]) - lo));
    double rc[4];
    batch_rcp(dacc, rc);
#pragma unroll
    for (int r = 0; r < 4; ++r) q[r] = p[r] * rc[r];
    return worst < span;
}
// mu_quot_fast for two tiles at once, level by level: a dependent f64 VALU operation has ~28 cycles of latency
// on gfx950 (throughput ~6), the quotient chain is 8 levels deep, and nothing else runs beside it (one wavefront
// per SIMD, MFMAs do not overlap VALU) - so the second tile's chain fills the latency slots of the first.  The
// compiler's scheduler does not model that latency and would emit one chain after the other; the fences keep
// the levels apart.
__device__ __forceinline__ bool mu_quot_fast2(const f64x4& pa, const f64x4& da, const f64x4& pb, const f64x4& db,
                                              unsigned lo, double (&qa)[4], double (&qb)[4]) {
#define EVC_LV __builtin_amdgcn_sched_barrier(0)
    const unsigned span = FAST_HI_WORD > lo ? FAST_HI_WORD - lo : 0u;
    const double a0 = da[0] * da[1], a1 = da[2] * da[3], b0 = db[0] * db[1], b1 = db[2] * db[3];
    const unsigned wa = max(max(hi_word(da[0]) - lo, hi_word(da[1]) - lo), max(hi_word(da[2]) - lo, hi_word(da[3]) - lo));
    const unsigned wb = max(max(hi_word(db[0]) - lo, hi_word(db[1]) - lo), max(hi_word(db[2]) - lo, hi_word(db[3]) - lo));
    EVC_LV;
    const double aa = a0 * a1, bb = b0 * b1;
    EVC_LV;
    double Ra = __builtin_amdgcn_rcp(aa), Rb = __builtin_amdgcn_rcp(bb);
    EVC_LV;
    const double ea = __builtin_fma(-aa, Ra, 1.0), eb = __builtin_fma(-bb, Rb, 1.0);
    EVC_LV;
    Ra = __builtin_fma(Ra, ea, Ra); Rb = __builtin_fma(Rb, eb, Rb);
    EVC_LV;
    const double Ra1 = Ra * a1, Rb1 = Rb * b1, Ra0 = Ra * a0, Rb0 = Rb * b0;
    EVC_LV;
    const double ra0 = Ra1 * da[1], rb0 = Rb1 * db[1], ra1 = Ra1 * da[0], rb1 = Rb1 * db[0];
    const double ra2 = Ra0 * da[3], rb2 = Rb0 * db[3], ra3 = Ra0 * da[2], rb3 = Rb0 * db[2];
    EVC_LV;
    qa[0] = pa[0] * ra0; qb[0] = pb[0] * rb0; qa[1] = pa[1] * ra1; qb[1] = pb[1] * rb1;
    qa[2] = pa[2] * ra2; qb[2] = pb[2] * rb2; qa[3] = pa[3] * ra3; qb[3] = pb[3] * rb3;
    EVC_LV;
#undef EVC_LV
    return max(wa, wb) < span;
}

// mu_quot_fast with its VALU instructions placed by hand in the gaps of NM MFMAs (mf(i) issues the i-th).
// The 23 instructions form a dependent chain of seven levels (range test + 2 products | product + v_rcp | fma |
// fma | 2 | 4 | 4 multiplies): one level per gap, fenced by sched_barrier so that the compiler keeps the order -
// every level's operands are ready when its gap comes round (an f64 MFMA takes 64 cycles), and neither the
// wavefront nor the matrix pipe behind it ever waits on VALU latency.  (sched_group_barrier does not achieve
// this placement: it counts instructions, it does not know the chain.)
template <int NM, class F>
__device__ __forceinline__ bool mu_quot_shadowed(F&& mf, const f64x4& p, const f64x4& d, unsigned lo,
                                                 double (&q)[4]) {
#define EVC_MF(i)                                                    \
    do {                                                             \
        if ((i) < NM) { mf(i); __builtin_amdgcn_sched_barrier(0); }  \
    } while (0)
#ifdef EVC_DBG_NOQ       // diagnostic (tools/ubench): the MFMAs alone, no quotient arithmetic
    for (int i = 0; i < NM; ++i) mf(i);
    for (int r = 0; r < 4; ++r) q[r] = p[r];
    return true;
#endif
    EVC_MF(0);
    const unsigned span = FAST_HI_WORD > lo ? FAST_HI_WORD - lo : 0u;
    const unsigned worst = max(max(hi_word(d[0]) - lo, hi_word(d[1]) - lo),
                               max(hi_word(d[2]) - lo, hi_word(d[3]) - lo));
    const double a = d[0] * d[1], b = d[2] * d[3];
    __builtin_amdgcn_sched_barrier(0);
    EVC_MF(1);
    const double ab = a * b;
    double R = __builtin_amdgcn_rcp(ab);
    __builtin_amdgcn_sched_barrier(0);
    EVC_MF(2);
    const double e = __builtin_fma(-ab, R, 1.0);
    __builtin_amdgcn_sched_barrier(0);
    EVC_MF(3);
    R = __builtin_fma(R, e, R);
    __builtin_amdgcn_sched_barrier(0);
    EVC_MF(4);
    const double Ra = R * b, Rb = R * a;
    __builtin_amdgcn_sched_barrier(0);
    EVC_MF(5);
    const double r0 = Ra * d[1], r1 = Ra * d[0], r2 = Rb * d[3], r3 = Rb * d[2];
    __builtin_amdgcn_sched_barrier(0);
    EVC_MF(6);
    q[0] = p[0] * r0; q[1] = p[1] * r1; q[2] = p[2] * r2; q[3] = p[3] * r3;
    __builtin_amdgcn_sched_barrier(0);
    EVC_MF(7);
#undef EVC_MF
    static_assert(NM <= 8, "at most 8 MFMAs per shadow");
    return worst < span;
}
__device__ __forceinline__ void mu_quot_exact(const f64x4& p, const f64x4& dacc, int mode, double eps,
                                              double (&q)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double dn = dacc[r];
        dn = (mode == EVC_EPS_ZERO_REPLACE && dn == 0.0) ? eps : dn;   // sklearn _nmf.py:620
        dn = (mode == EVC_EPS_CLAMP && !(dn > eps)) ? eps : dn;         // deComP
        q[r] = p[r] / dn;
    }
}

hipError_t fused_res_launch(int msteps, const FusedArgs& a, hipStream_t s);
// evc_fused_all.hip: every activation and numerator tile register-resident, NT / 32 workgroups per frame tile
hipError_t fused_all_launch(int msteps, const FusedArgs& a, int n_cus, hipStream_t s);
// evc_fused_xy.hip: the same with two frame tiles per member and the exchange inside the sweeps, NT / 16 workgroups per pair
hipError_t fused_xy_launch(int msteps, const FusedArgs& a, int n_cus, hipStream_t s);

// per-frame share of 2 KL(X || A H) from the B-operand images of X and V held by one lane
// (sklearn _nmf.py:136-160: log term only where x > eps, V floored at eps there, plus sum(V))
__device__ __forceinline__ double kl_terms(double x, double v, double eps) {
    double e = v;
    if (x > eps) e += x * log(x / (v < eps ? eps : v)) - x;
    return 2.0 * e;
}

}  // namespace evc
