// Shared by evc_fused.hip (general streamed kernel) and evc_fused_res.hip (register-resident
// kernel): packed layouts, kernel arguments and the element-wise update.
#pragma once
#include "evc_internal.h"

#include <type_traits>

namespace evc {

typedef double f64x2 __attribute__((ext_vector_type(2)));

// tell the compiler a pointer is wave-uniform (it then lives in SGPRs and global accesses use the
// saddr + per-lane-offset form)
template <typename P>
__device__ __forceinline__ P* uniform_ptr(P* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (P*)(((unsigned long long)hi << 32) | lo);
}

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
    int loss;                // EVC_LOSS_*; for KL A1p holds the dictionary pre-divided by its column sums
    int eps_mode;
    double eps, l1;
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
//     exact path.  Fast path = every denominator is a normal number in [lo, 2^928): then neither
//     the ==0 replacement nor the clamp can fire, and the quotient is formed as
//     r = v_rcp_f64(den) (2^-24.4), one Newton step (2^-48.8 = 2.3e-15 relative, measured by
//     tools/ubench/rcp_accuracy.hip), q = num*r: 5 VALU per element, no v_div_scale/fmas/fixup and
//     no selects.  The 10-ulp quotient is the same size as the rounding noise of the N-term
//     denominator sums next to it; a residual correction would make the quotient correctly
//     rounded for 2 more VALU per element (2.7 % of the kernel) and moves the end-to-end
//     difference to the oracle by less than its spread - not taken;
//   * the exact path (zero / denormal / huge / NaN denominators, and the unguarded NONE mode
//     always) applies the guard literally and divides with IEEE semantics, so inf/NaN behaviour
//     is the reference's.
__device__ __forceinline__ double fast_div(double num, double den) {
    double r = __builtin_amdgcn_rcp(den);
    const double e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    return num * r;
}
__device__ __forceinline__ unsigned hi_word(double x) { return (unsigned)(__double_as_longlong(x) >> 32); }
// lowest admissible high word for the fast path: 2^-928, or one binade above eps when clamping
__device__ __forceinline__ unsigned fast_lo(int mode, double eps) {
    unsigned lo = 0x05F00000u;
    if (mode == EVC_EPS_CLAMP && eps > 0) {
        const unsigned e = hi_word(eps) + 0x00200000u;
        lo = e > lo ? e : lo;
    }
    return lo;
}
// MUL_FIRST: pymf / nmf_tool form (h*p)/den, else sklearn / deComP form h*(p/den).  `mode` is only
// consulted on the exact path.
template <bool MUL_FIRST, class HT>
__device__ __forceinline__ void mu_tile(HT& h, const f64x4& p, const f64x4& dacc, int mode,
                                        double eps, unsigned lo) {
    const unsigned span = 0x79F00000u - lo;
    const unsigned worst = max(max(hi_word(dacc[0]) - lo, hi_word(dacc[1]) - lo),
                               max(hi_word(dacc[2]) - lo, hi_word(dacc[3]) - lo));
    if (__builtin_expect(mode != EVC_EPS_NONE && __all(worst < span), 1)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double qv = fast_div(MUL_FIRST ? h[r] * p[r] : p[r], dacc[r]);
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

// per-frame share of 2 KL(X || A H) from the B-operand images of X and V held by one lane
// (sklearn _nmf.py:136-160: log term only where x > eps, V floored at eps there, plus sum(V))
__device__ __forceinline__ double kl_terms(double x, double v, double eps) {
    double e = v;
    if (x > eps) e += x * log(x / (v < eps ? eps : v)) - x;
    return 2.0 * e;
}
bool fused_res_supported(int N, int eps_mode);

}  // namespace evc
