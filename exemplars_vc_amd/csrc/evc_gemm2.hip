// k_gemm2: the contractions of the generic path (M > 32 bins, float32, GRAM / LITERAL), second generation.
//
//   C[f][r] = sum_k L[f][k] R[r][k]          L: frames x K, R: rows (exemplars or bins) x K, both k-contiguous
//
// What k_gemm_nt (evc_gemm.hip) left on the table, by its profile (profiles/r02_base_*: 44-63 % matrix-pipe
// occupancy, one LDS bank conflict per LDS instruction at float32):
//   * k-major LDS with one scalar ds_read per operand and k-step -> here the LDS image is row-major with the
//     row's 16-byte slots XOR-swizzled by the row number, every fragment read is a conflict-free ds_read_b128
//     that feeds 2 (f64) or 4 (f32) k-steps (the MFMA's k assignment is free as long as both operands use
//     the same one: lane group q takes the k-values of slot 4 kk + q);
//   * k-slabs of 16 with a barrier each -> slabs of 128 or 256 bytes per row, one barrier per slab;
//   * 4 scalar loads/stores per accumulator tile in the update epilogue -> the operand roles are swapped (rows
//     of R on the MFMA's i axis, frames on j) and, for f64, the rows of a 16-tile are permuted on the way out
//     of LDS, so that a lane's four results are four CONSECUTIVE columns of one frame row: H, P and H' move as
//     one 16- or 32-byte access per lane and tile;
//   * a single tile shape and no look at how the grid fills 256 CUs -> shapes and split-K chosen per call.
#include "evc_internal.h"
#include <type_traits>

// Diagnostic builds only (tools/ubench/gemm2_bench.hip, -DEVC_G2_ABLATE=n): what bounds the kernel is found by
// taking one part out.  1: the epilogue reads no H / P (update on constants); 2: main loop without global loads
// (LDS + MFMA pipeline alone); 3: no MFMAs (loads, LDS traffic and barriers alone); 4: no epilogue stores.
#ifndef EVC_G2_ABLATE
#define EVC_G2_ABLATE 0
#endif

// H, P and H' stream through the update kernel once per iteration (541 MB at the STFT flow) beside a dictionary that
// should stay in the L2: EVC_G2_NT=1 (diagnostic) marks them non-temporal - measured 11 % SLOWER (277 vs 250 us)
#if defined(EVC_G2_NT) && EVC_G2_NT
#define EVC_NT_LOAD(p) __builtin_nontemporal_load(p)
#define EVC_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define EVC_NT_LOAD(p) (*(p))
#define EVC_NT_STORE(v, p) (*(p) = (v))
#endif

#ifdef EVC_G2_STAMP      // diagnostic build (tools/ubench/gemm2_bench.hip): per-wavefront s_memtime stamps of the update kernel
__device__ long long* evc_g2_dbg = nullptr;
#define G2_STAMP(i) do { if (MU && evc_g2_dbg && lane == 0) { const unsigned l_ = blockIdx.x + gridDim.x * blockIdx.y; \
        if (l_ < 2048u) evc_g2_dbg[(l_ * 8 + (tid >> 6)) * 8 + (i)] = (i) == 7 ? (long long)__builtin_amdgcn_s_memrealtime() : (long long)__builtin_amdgcn_s_memtime(); } } while (0)
#else
#define G2_STAMP(i)
#endif

namespace evc {

constexpr int EPI_KL = 100, EPI_STORE = 101;      // epilogue bodies besides the four eps modes

template <typename T> struct G2T;
template <> struct G2T<double> { typedef double vec __attribute__((ext_vector_type(2))); typedef double vec4 __attribute__((ext_vector_type(4))); };
template <> struct G2T<float> { typedef float vec __attribute__((ext_vector_type(4))); typedef float vec4 __attribute__((ext_vector_type(4))); };

// BF x BR block (frames x rows of R), WF x WR per wavefront, BK elements of k per slab, MINW: waves per SIMD the
// register allocation is held to (2 workgroups per CU when the LDS image is 64 KiB)
template <typename T, int BF, int BR, int WF, int WR, int BK, bool MU, int MINW, int DEPTH>
__global__ __launch_bounds__((BF / WF) * (BR / WR) * 64, MINW) void k_gemm2(
    const T* __restrict__ L, int ldl, const T* __restrict__ R, int ldr, T* __restrict__ C, int ldc, int Kd,
    MuEpilogue<T> ep, long slab, int jv) {
    typedef typename Mma<T>::acc_t acc_t;
    typedef typename G2T<T>::vec vec;
    typedef typename G2T<T>::vec4 vec4;
    constexpr int EPV = 16 / (int)sizeof(T);             // elements per 16-byte slot
    constexpr int SLOTS = BK / EPV;                       // slots per LDS row (8 or 16)
    constexpr int NWF = BF / WF, NWR = BR / WR, NTHR = NWF * NWR * 64;
    constexpr int FI = WF / 16, RI = WR / 16;
    constexpr int ROWS = BF + BR;
    constexpr int RPP = NTHR / SLOTS;                     // rows staged per pass of the workgroup
    constexpr int PL = BF / RPP, PR = BR / RPP;           // passes over the L rows / the R rows
    constexpr int KK = SLOTS / 4;                         // fragment reads per slab and operand tile
    static_assert(SLOTS == 8 || SLOTS == 16, "LDS rows of 128 or 256 bytes");
    static_assert(NTHR % SLOTS == 0 && RPP % SLOTS == 0 && BF % RPP == 0 && BR % RPP == 0, "staging shape");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* sm = reinterpret_cast<T*>(smem_raw);               // [DEPTH + 1][ROWS][BK]

    // Block coordinates.  Workgroups go to the 8 XCDs round-robin by linear id, each XCD with its own L2.  The plain
    // product (V = H Am^T: few R blocks, each frame block's rows of H wanted by all of them) renumbers the grid so
    // that the R blocks of one frame block are neighbours in time on ONE XCD: H then crosses the fabric once
    // instead of once per R block.  The update kernel keeps the plain order: there an XCD sees every 8th exemplar
    // block only, i.e. one eighth of the dictionary, and the shared operand (V) is small.
    if (ep.gate && *ep.gate == 0) return;      // (uniform: every utterance has stopped)
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (!MU) {
        const unsigned gx = gridDim.x, gxy = gx * gridDim.y, total = gxy * gridDim.z;
        const unsigned lin = bx + gx * by + gxy * bz, xcd = lin & 7u;
        const unsigned v = xcd * (total >> 3) + (xcd < (total & 7u) ? xcd : (total & 7u)) + (lin >> 3);
        bz = v / gxy;
        const unsigned rem = v - bz * gxy;
        by = rem / gx;
        bx = rem - by * gx;
    }
    // split-K: block z owns k in [z Kd, (z+1) Kd) and writes its partial product to slab z
    L += (long)bz * Kd;
    R += (long)bz * Kd;
    C += (long)bz * slab;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wf = w % NWF, wr = w / NWF;
    const int i16 = lane & 15, q = lane >> 4;
    const long bf0 = (long)by * BF, br0 = (long)bx * BR;
    // rows of R at and beyond jv are zero padding (the bins of V = H Am^T rounded up to the block width): their
    // 16-row groups are not multiplied - the wavefront's matrix-pipe time goes to the CU's other workgroups
    int nri = ((int)(jv - (br0 + wr * WR)) + 15) / 16;
    nri = nri < 0 ? 0 : (nri > RI ? RI : nri);

#ifdef EVC_G2_STAGGER
    // diagnostic: every second workgroup of the first round starts late, so that the workgroups sharing a CU are
    // not all in their main loop, then all in their epilogue
    if (MU) {
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
        if (lin < 1024u && (lin & 1u))
            for (int i = 0; i < EVC_G2_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    if (MU) {
        // a block whose frames all belong to stopped utterances only carries H over
        int any = 0;
        if (tid < BF) {
            const int u = ep.frame_utt[bf0 + tid];
            any = (u >= 0) && (ep.active[u] != 0);
        }
        if (!__syncthreads_or(any)) {
            if (C != ep.Hin) {
                for (int e = tid; e < BF * BR; e += NTHR) {
                    const long r = bf0 + e / BR, c = br0 + e % BR;
                    C[r * ldc + c] = ep.Hin[r * ep.ldh + c];
                }
            }
            return;
        }
    }

    // staging: thread -> (row0 + pass * RPP, slot); the swizzled slot is the same in every pass (RPP % SLOTS == 0)
    const int row0 = tid / SLOTS, slot = tid % SLOTS;
    // swizzle keys.  L rows (frames): the row number.  R rows: the MFMA row i that reads the row - for f64 the
    // rows of a 16-tile are permuted on the way out (see `ar` below), and with 128-byte rows two LDS rows share
    // one bank row, so the key is made of the bits of i that keep the 16 lanes of a ds_read_b128 group apart
    auto key_of = [](int i) { return SLOTS == 16 ? i : ((i & 3) | ((i >> 3) << 2)); };
    const int r15 = row0 & 15;
    const int pslot = slot ^ (row0 & (SLOTS - 1));
    const int pslotR = sizeof(T) == 8 ? slot ^ key_of(4 * (r15 & 3) + (r15 >> 2)) : pslot;
    const T* gl = L + (bf0 + row0) * (long)ldl;
    const T* gr = R + (br0 + row0) * (long)ldr;
    typedef vec StgSet[PL + PR];
    StgSet stg[DEPTH];
    auto fetch = [&](StgSet& st, int k0) {
        if (EVC_G2_ABLATE == 2) {
#pragma unroll
            for (int i = 0; i < PL + PR; ++i) st[i] = vec(T(1e-3));
            return;
        }
        // (Kd is a multiple of 16 elements: a slot is all in or all out.)  A slot beyond Kd re-reads the row's last
        // one - its products are skipped (nkk in compute) - so that no select stands between a load and its use:
        // with one, the compiler waits for the previous slab's loads before it issues the next ones
        int e = k0 + slot * EPV;
        e = e > Kd - EPV ? Kd - EPV : e;
#pragma unroll
        for (int i = 0; i < PL; ++i) st[i] = *reinterpret_cast<const vec*>(gl + (long)i * RPP * ldl + e);
#pragma unroll
        for (int i = 0; i < PR; ++i) st[PL + i] = *reinterpret_cast<const vec*>(gr + (long)i * RPP * ldr + e);
    };
    auto stash = [&](const StgSet& st, int buf) {
        T* b = sm + buf * (ROWS * BK);
#pragma unroll
        for (int i = 0; i < PL; ++i) *reinterpret_cast<vec*>(b + (row0 + i * RPP) * BK + pslot * EPV) = st[i];
#pragma unroll
        for (int i = 0; i < PR; ++i) *reinterpret_cast<vec*>(b + (BF + row0 + i * RPP) * BK + pslotR * EPV) = st[PL + i];
    };

    acc_t acc[RI][FI];
#pragma unroll
    for (int a = 0; a < RI; ++a)
#pragma unroll
        for (int b = 0; b < FI; ++b) acc[a][b] = acc_t{0, 0, 0, 0};

    // fragment rows.  R on the MFMA's i axis; for f64 the accumulator row is q + 4 r, so the tile's rows are
    // permuted (i -> 4 (i & 3) + (i >> 2)): a lane's registers r = 0..3 are then rows 4 q + r, as for f32.
    const int ar = sizeof(T) == 8 ? 4 * (i16 & 3) + (i16 >> 2) : i16;
    const int keyA = sizeof(T) == 8 ? key_of(i16) : (i16 & (SLOTS - 1)), keyB = i16 & (SLOTS - 1);
    int rowA[RI], rowB[FI];
#pragma unroll
    for (int ri = 0; ri < RI; ++ri) rowA[ri] = BF + wr * WR + 16 * ri + ar;
#pragma unroll
    for (int fi = 0; fi < FI; ++fi) rowB[fi] = wf * WF + 16 * fi + i16;

    // The update's operands (H and P tiles) are requested before the last slab is multiplied, when they fit the
    // register budget: their latency then hides behind that slab's MFMAs instead of heading the epilogue.
    constexpr int TREGS = RI * FI * 4 * (int)sizeof(T) / 4;    // VGPRs of one operand's tiles
    constexpr bool PREF = MU && TREGS <= 16, PREFP = PREF;
    vec4 hv[PREF ? RI : 1][PREF ? FI : 1], pv[PREFP ? RI : 1][PREFP ? FI : 1];

    const int nslab = (Kd + BK - 1) / BK;
    auto prefetch_update_operands = [&]() {
#pragma unroll
        for (int fi = 0; fi < (PREF ? FI : 0); ++fi)
#pragma unroll
            for (int ri = 0; ri < (PREF ? RI : 0); ++ri) {
                const long o = (bf0 + wf * WF + 16 * fi + i16) * ep.ldh + br0 + wr * WR + 16 * ri + 4 * q;
                hv[ri][fi] = EVC_NT_LOAD(reinterpret_cast<const vec4*>(ep.Hin + o));
                if (PREFP && !ep.kl) pv[PREFP ? ri : 0][PREFP ? fi : 0] = EVC_NT_LOAD(reinterpret_cast<const vec4*>(ep.P + o));
            }
    };
    // One slab's MFMAs for the first NR 16-row groups of R.  NR is a compile-time count: a run-time bound inside the
    // unrolled loops puts a branch between every two MFMAs (measured: the skipped products then bought nothing).
    auto compute_n = [&](int buf, int sl, auto nr_tag) {
        constexpr int NR = decltype(nr_tag)::value;
        const T* b = sm + buf * (ROWS * BK);
        const int rest = Kd - sl * BK;
        const int nkk = rest >= BK ? KK : rest / (4 * EPV);
        // the fragments of group kk + 1 are requested before the MFMAs of group kk issue (two register sets; reads
        // beyond the k-tail fetch slots whose products are skipped): LDS latency runs beside the matrix pipe
        constexpr int NS = MINW >= 4 ? 1 : 2;       // (the 128 x 128 two-per-CU shape has no registers for a second set)
        vec fa[NS][NR ? NR : 1], fb[NS][FI];
        auto read_group = [&](int kk, vec (&A)[NR ? NR : 1], vec (&B)[FI]) {
#pragma unroll
            for (int ri = 0; ri < NR; ++ri)
                A[ri] = *reinterpret_cast<const vec*>(b + rowA[ri] * BK + ((4 * kk + q) ^ keyA) * EPV);
#pragma unroll
            for (int fi = 0; fi < FI; ++fi)
                B[fi] = *reinterpret_cast<const vec*>(b + rowB[fi] * BK + ((4 * kk + q) ^ keyB) * EPV);
        };
        if (NS == 2) read_group(0, fa[0], fb[0]);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            if (NS == 2) {
                if (kk + 1 < KK) read_group(kk + 1, fa[(kk + 1) & (NS - 1)], fb[(kk + 1) & (NS - 1)]);
                __builtin_amdgcn_sched_barrier(0);
            } else if (kk < nkk) {
                read_group(kk, fa[0], fb[0]);
            }
            if (kk < nkk) {
#pragma unroll
                for (int e = 0; e < EPV; ++e)
#pragma unroll
                    for (int ri = 0; ri < NR; ++ri)
#pragma unroll
                        for (int fi = 0; fi < FI; ++fi) {
                            if (EVC_G2_ABLATE == 3) acc[ri][fi][e & 3] += fa[kk & (NS - 1)][ri][e] * fb[kk & (NS - 1)][fi][e];
                            else acc[ri][fi] = Mma<T>::mma(fa[kk & (NS - 1)][ri][e], fb[kk & (NS - 1)][fi][e], acc[ri][fi]);
                        }
            }
            if (NS == 2) __builtin_amdgcn_sched_barrier(0);
        }
    };
    // all groups (every block but the last one along R), or the block that reaches into the padding: 1 .. RI - 1
    // groups by halving steps (RI is 2 or 4), or none at all
    auto compute = [&](int buf, int sl) {
        if (nri == RI) compute_n(buf, sl, std::integral_constant<int, RI>{});
        else if (RI > 2 && nri == 3) compute_n(buf, sl, std::integral_constant<int, (RI > 2 ? 3 : 1)>{});
        else if (RI > 2 && nri == 2) compute_n(buf, sl, std::integral_constant<int, (RI > 2 ? 2 : 1)>{});
        else if (nri >= 1) compute_n(buf, sl, std::integral_constant<int, 1>{});
    };
    G2_STAMP(0);
    G2_STAMP(7);
    if (DEPTH == 1) {
        // two LDS buffers; the next slab's loads fly while this one feeds the MFMAs
        fetch(stg[0], 0);
        stash(stg[0], 0);
        __syncthreads();
        for (int sl = 0; sl < nslab; ++sl) {
            const bool more = sl + 1 < nslab;
            if (more) fetch(stg[0], (sl + 1) * BK);
            if (PREF && !more) prefetch_update_operands();
            compute(sl & 1, sl);
            if (more) stash(stg[0], (sl + 1) & 1);
            __syncthreads();
        }
    } else {
        // Short slabs on small tiles (32 MFMAs per wavefront and slab against ~2 us of load latency): TWO slabs are
        // in flight.  Three LDS buffers; slab sl is in buffer sl % 3, slab sl + 1 in register set (sl + 1) & 1, and
        // slab sl + 2 is requested into the set slab sl came from.
        fetch(stg[0], 0);
        if (nslab > 1) fetch(stg[DEPTH - 1], BK);
        stash(stg[0], 0);
        __syncthreads();
        auto step = [&](StgSet& free_set, const StgSet& next_set, int sl) {
            if (sl + 2 < nslab) fetch(free_set, (sl + 2) * BK);
            if (PREF && sl + 1 == nslab) prefetch_update_operands();
            compute(sl % 3, sl);
            if (sl + 1 < nslab) stash(next_set, (sl + 1) % 3);
            __syncthreads();
        };
        for (int sl = 0; sl < nslab; sl += 2) {
            step(stg[0], stg[DEPTH - 1], sl);
            if (sl + 1 < nslab) step(stg[DEPTH - 1], stg[0], sl + 1);
        }
    }

    // epilogue: lane (i16, q) holds, per tile, rows 4 q .. 4 q + 3 of R (consecutive columns of C) of frame i16.
    // The guard mode is a template argument of the epilogue body: one uniform switch in front of it instead of one
    // per element (which left 650 basic blocks and each element's division chain on its own), a stopped frame is a
    // select, and the zero padding of the last exemplar block a wave-uniform case.
    auto epilogue_body = [&](auto mode_tag, auto edge_tag) {
        constexpr int MODE = decltype(mode_tag)::value;          // eps mode; EPI_KL; EPI_STORE
        constexpr bool edge = decltype(edge_tag)::value;         // the block reaches into the zero padding of H
#pragma unroll
        for (int fi = 0; fi < FI; ++fi) {
            const long t = bf0 + wf * WF + 16 * fi + i16;
            bool live = true;
            if (MU) {
                const int u = ep.frame_utt[t];
                live = (u >= 0) && (ep.active[u] != 0);
            }
#pragma unroll
            for (int ri = 0; ri < RI; ++ri) {
                const long n0 = br0 + wr * WR + 16 * ri + 4 * q;
                vec4 out;
                if (MODE == EPI_STORE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[r] = acc[ri][fi][r];
                } else {
                    const vec4 h = EVC_G2_ABLATE == 1 ? vec4(T(0.5)) : PREF ? hv[PREF ? ri : 0][PREF ? fi : 0]
                                        : EVC_NT_LOAD(reinterpret_cast<const vec4*>(ep.Hin + t * ep.ldh + n0));
                    if (MODE == EPI_KL) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) out[r] = h[r] * acc[ri][fi][r];
                    } else {
                        const vec4 p = EVC_G2_ABLATE == 1 ? vec4(T(0.25)) : PREFP ? pv[PREFP ? ri : 0][PREFP ? fi : 0]
                                             : EVC_NT_LOAD(reinterpret_cast<const vec4*>(ep.P + t * ep.ldh + n0));
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            out[r] = mu_update<T>(h[r], p[r], acc[ri][fi][r], MODE, ep.eps, ep.l1);
                    }
                    if (edge) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n0 + r >= ep.N) out[r] = T(0);   // keep the zero padding exact (0/0 modes)
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[r] = live ? out[r] : h[r];
                }
                if (EVC_G2_ABLATE != 4 || out[0] == T(-1.2345)) {
                    if (MU) EVC_NT_STORE(out, reinterpret_cast<vec4*>(C + t * ldc + n0));
                    else *reinterpret_cast<vec4*>(C + t * ldc + n0) = out;
                }
            }
        }
    };
    auto epilogue = [&](auto mode_tag) {
        G2_STAMP(1);
        if (MU && br0 + BR > ep.N) epilogue_body(mode_tag, std::true_type{});
        else epilogue_body(mode_tag, std::false_type{});
        __builtin_amdgcn_s_waitcnt(0);
        G2_STAMP(2);
    };
    if (!MU) {
        epilogue(std::integral_constant<int, EPI_STORE>{});
    } else if (ep.kl) {
        epilogue(std::integral_constant<int, EPI_KL>{});
    } else {
        switch (ep.eps_mode) {
            case EVC_EPS_ADD: epilogue(std::integral_constant<int, EVC_EPS_ADD>{}); break;
            case EVC_EPS_ZERO_REPLACE: epilogue(std::integral_constant<int, EVC_EPS_ZERO_REPLACE>{}); break;
            case EVC_EPS_CLAMP: epilogue(std::integral_constant<int, EVC_EPS_CLAMP>{}); break;
            default: epilogue(std::integral_constant<int, EVC_EPS_NONE>{}); break;
        }
    }
}

template <typename T, int BF, int BR, int WF, int WR, int BK, bool MU, int MINW, int DEPTH = 1>
static hipError_t launch2(const T* L, int ldl, const T* R, int ldr, T* C, int ldc, int I, int J, int Kd,
                          const MuEpilogue<T>& ep, hipStream_t s, int splits = 1, long slab = 0, int jv = 0) {
    constexpr int NTHR = (BF / WF) * (BR / WR) * 64;
    const size_t lds = (size_t)(DEPTH + 1) * (BF + BR) * BK * sizeof(T);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2<T, BF, BR, WF, WR, BK, MU, MINW, DEPTH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid(J / BR, I / BF, splits), block(NTHR);
    hipLaunchKernelGGL((k_gemm2<T, BF, BR, WF, WR, BK, MU, MINW, DEPTH>), grid, block, lds, s, L, ldl, R, ldr, C, ldc,
                       Kd / splits, ep, slab, jv > 0 ? jv : J);
    return hipGetLastError();
}

template <typename T> static bool aligned16(const T* p, int ld) {
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ((size_t)ld * sizeof(T)) % 16 == 0;
}

// can k_gemm2 take this contraction?  (workspace operands always can; Griffin-Lim's strided frames too)
template <typename T>
bool gemm2_ok(const T* L, int ldl, const T* R, int ldr, const T* C, int ldc, int I, int J, int Kd) {
    return I > 0 && J > 0 && Kd > 0 && I % 64 == 0 && J % 64 == 0 && Kd % 16 == 0 && aligned16(L, ldl) &&
           aligned16(R, ldr) && aligned16(C, ldc);
}

// LDS row length per dtype: 256-byte rows (deep slab, one 128x128 workgroup per CU) / 128-byte rows
template <typename T> struct G2K { static constexpr int BIG = 256 / (int)sizeof(T), SMALL = 128 / (int)sizeof(T); };

// Shapes (frames x rows of R, threads, LDS, workgroups per CU):
//   BIG    128 x 128, 512, 128 KiB, 1   deep slabs: long contractions with plenty of blocks (GRAM)
//   MID    128 x 128, 512,  64 KiB, 2   (float32 only: float64 accumulators do not fit 128 VGPRs)
//   SMALL   64 x 128, 256,  48 KiB, 3   finer tile quantisation on 256 CUs, one's epilogue beside the others' MFMAs
//   SMALL64 64 x  64, 256,  48 KiB, 3 (two slabs in flight)  R with a multiple of 64 (not 128) rows: V = H Am^T for M <= 64 mod 128
template <typename T, bool MU>
static hipError_t launch_shape(int shape, const T* L, int ldl, const T* R, int ldr, T* C, int ldc, int I, int J, int Kd,
                               const MuEpilogue<T>& ep, hipStream_t s, int splits, long slab, int jv = 0) {
    switch (shape) {
        case 0: return launch2<T, 128, 128, 32, 64, G2K<T>::BIG, MU, 2>(L, ldl, R, ldr, C, ldc, I, J, Kd, ep, s, splits, slab, jv);
        case 1: return launch2<T, 128, 128, 32, 64, G2K<T>::SMALL, MU, sizeof(T) == 4 ? 4 : 2>(L, ldl, R, ldr, C, ldc, I, J, Kd, ep, s, splits, slab, jv);
        case 2: return launch2<T, 64, 128, 32, 64, G2K<T>::SMALL, MU, 3>(L, ldl, R, ldr, C, ldc, I, J, Kd, ep, s, splits, slab, jv);
        default: return launch2<T, 64, 64, 32, 32, G2K<T>::SMALL, MU, 3, 2>(L, ldl, R, ldr, C, ldc, I, J, Kd, ep, s, splits, slab, jv);
    }
}

static double round_eff(long wg, long slots) { return (double)wg / (double)(((wg + slots - 1) / slots) * slots); }

// C = L R^T.  With `scratch` (and ldc == J) a short grid is split over k into slabs; see gemm_nt for splits_out.
template <typename T>
hipError_t gemm2(const T* L, int ldl, const T* R, int ldr, T* C, int ldc, int I, int J, int Kd, hipStream_t s,
                 T* scratch, size_t scratch_elems, int* splits_out, int n_cus, int jv, const int* gate) {
    if (splits_out) *splits_out = 0;
    MuEpilogue<T> ep{};
    ep.gate = gate;
    if (n_cus <= 0) n_cus = 256;
    // 64 x 128 blocks (three per CU) when R has a multiple of 128 rows and the grid fills the CUs; 64 x 64 blocks
    // (four per CU) otherwise: a grid of less than one round runs as long as ONE workgroup does, so the smallest
    // tile is the fastest
#ifdef EVC_G2_PLAIN_SHAPE      // diagnostic builds: force a block shape of the plain product (0 / 1: 128 x 128)
    const bool wide = EVC_G2_PLAIN_SHAPE != 3;
    const int shape = EVC_G2_PLAIN_SHAPE;
    const long blocks = shape <= 1 ? (long)(I / 128) * (J / 128) : (long)(I / 64) * (J / (wide ? 128 : 64));
    const long slots = (long)n_cus * (shape == 0 ? 1 : (shape == 1 ? 2 : (wide ? 3 : 4)));
#else
    const bool wide = J % 128 == 0 && (long)(I / 64) * (J / 128) >= 3L * n_cus;
    const int shape = wide ? 2 : 3;
    const long blocks = (long)(I / 64) * (J / (wide ? 128 : 64));
    // (48 KiB of LDS hold three workgroups per CU for either shape; counting four for the narrow one makes the rule
    // split a little earlier, which measured better: STFT flow, 16 utterances, 157.8 against 155.5 kframes/s)
    const long slots = (long)n_cus * (wide ? 3 : 4);
#endif
    // split-K so that the grid fills the CUs' workgroup slots in whole rounds: the smallest split whose rounds
    // are >= 85 % full (each workgroup keeps >= 128 of k)
    int splits = 1;
    const long slab = (long)I * ldc;
    if (scratch && ldc == J) {
        double best = round_eff(blocks, slots);
        for (int sp = 2; sp <= 32 && best < 0.85; ++sp) {
            if (Kd % (sp * 16) || Kd / sp < 128 || (size_t)sp * slab > scratch_elems) continue;
            const double eff = round_eff(blocks * sp, slots);
            if (eff > best + 0.03) { best = eff; splits = sp; }
        }
    }
    T* out = splits > 1 ? scratch : C;
    hipError_t e = launch_shape<T, false>(shape, L, ldl, R, ldr, out, ldc, I, J, Kd, ep, s, splits, slab, jv);
    if (e != hipSuccess || splits == 1) return e;
    if (splits_out) {          // the caller's next kernel sums the slabs itself (scratch + z * I * ldc, z < splits)
        *splits_out = splits;
        return hipSuccess;
    }
    return sum_slabs<T>(scratch, slab, splits, C, s, gate);
}

// the same contraction with the multiplicative update as epilogue: Hout = mu(Hin, P, L R^T)
template <typename T>
hipError_t gemm2_mu(const T* L, int ldl, const T* R, int ldr, T* Hout, int I, int J, int Kd,
                    const MuEpilogue<T>& ep, hipStream_t s) {
    // The epilogue moves 3 elements per output against 2 Kd flops: for short contractions (Kd of a few hundred) it is as
    // long as the main loop, so several workgroups share a CU and one's epilogue runs beside the others' MFMAs.
    // float32: 128 x 128 blocks, two per CU, when there are enough of them for whole rounds; else (and float64,
    // whose accumulators need the registers) 64 x 128 blocks, three per CU.
    // A grid of less than one round of 64 x 128 blocks takes the 64 x 64 shape (four per CU): it then runs as long
    // as one small workgroup does.
    int dev = 0, n_cus = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cus <= 0)
        n_cus = 256;
    const long b128 = (long)(I / 128) * (J / 128), b64 = (long)(I / 64) * (J / 128);
#ifdef EVC_G2_MU_SHAPE
    const int shape = EVC_G2_MU_SHAPE;       // diagnostic builds: force a block shape
    (void)b128; (void)b64;
#else
    const int shape = (sizeof(T) == 4 && I % 128 == 0 && b128 >= 8L * n_cus) ? 1 : (b64 >= 3L * n_cus ? 2 : 3);
#endif
    return launch_shape<T, true>(shape, L, ldl, R, ldr, Hout, ep.ldh, I, J, Kd, ep, s, 1, 0);
}

#define EVC_INST2(T)                                                                                              \
    template bool gemm2_ok<T>(const T*, int, const T*, int, const T*, int, int, int, int);                        \
    template hipError_t gemm2<T>(const T*, int, const T*, int, T*, int, int, int, int, hipStream_t, T*, size_t,  \
                                 int*, int, int, const int*);                                                       \
    template hipError_t gemm2_mu<T>(const T*, int, const T*, int, T*, int, int, int, const MuEpilogue<T>&, hipStream_t);
EVC_INST2(double)
EVC_INST2(float)

}  // namespace evc
