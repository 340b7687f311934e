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
#ifdef EVC_ALL_TIMING
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

hipError_t fused_res_launch(int msteps, const FusedArgs& a, hipStream_t s);
// evc_fused_all.hip: every activation and numerator tile register-resident, NT / 32 workgroups per frame tile
hipError_t fused_all_launch(int msteps, const FusedArgs& a, int n_cus, hipStream_t s);

// per-frame share of 2 KL(X || A H) from the B-operand images of X and V held by one lane
// (sklearn _nmf.py:136-160: log term only where x > eps, V floored at eps there, plus sum(V))
__device__ __forceinline__ double kl_terms(double x, double v, double eps) {
    double e = v;
    if (x > eps) e += x * log(x / (v < eps ? eps : v)) - x;
    return 2.0 * e;
}

}  // namespace evc
