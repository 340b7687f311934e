// k_fused_wide64: the fused FACTORED multiplicative update for wide float64 spectra (144 < M <= 528 bins: the
// 513-bin STFT magnitudes of BASELINE C3 / C5_513, 04_align_n_nmf.py:315-326 with fft_size 1024; the 201-bin
// |Re STFT| flow when its spectra are float64, 04_align_n_nmf.py:398,422 - 3 whole bin tiles per wavefront + the split one).
//
//   per iteration and frame:  D = A^T V (+ l1, + eps)      V = A H of the previous iteration
//                             H' = H (.) P (/) guard(D)     P = A^T X, formed once
//                             V' = A H'
//
// Same task queue as k_fused_wide (evc_wide.hip: tickets in iteration-major order, per-group counters, partial V'
// published per exemplar range, reduce tasks when the ranges are many, sc1 hand-off); what differs is who holds
// what, because V and V' of 16 frames at M = 513 are 528 registers per lane:
//
// * A WORKGROUP (4 wavefronts, one per SIMD, 512 registers each) owns 32 frames = 2 frame tiles.  The bins are split
//   over its wavefronts: wavefront w holds bin tiles 4 k + w, k < TPW, of V and of V' for BOTH frame tiles (2 x TPW
//   accumulator tiles each).  For the 16x16x4 f64 MFMA the accumulator layout (lane = (q, frame), register r <->
//   row q + 4 r) is the B-operand layout of k-step r (rows 4 r + q): V' feeds the next D product and D / H' feed
//   V' += A_j H'_j without a shuffle, as in the float32 kernel.
// * One bin tile more than the wavefronts hold whole (bins 64 TPW ... 64 TPW + 15: the 513th bin of a 1024-point
//   spectrum would cost every wavefront a ninth tile) is split over them by k-step: wavefront w multiplies rows
//   4 w .. 4 w + 3 of it into its partial D, and k-step w (exemplars 4 w .. 4 w + 3) of the block into a partial
//   sum of that tile's V'; the four partial sums are added up by whoever reads V next.  One fragment position and
//   four MFMAs per block instead of sixteen.
// * D = A_j^T V needs all bins: every wavefront multiplies its own bins (2 TPW k-steps of 4 per frame tile), the four
//   partial 16 x 16 tiles meet in LDS (16 KiB), and wavefront w reduces and updates ONE QUARTER of the two H tiles
//   (frame tile w / 2, register pair w % 2: it alone loads that quarter of H and P and stores that quarter of H'),
//   the four quarters of H' meet in LDS again (4 KiB).  Two LDS-only barriers per block, no global drain.
// * The dictionary goes global -> LDS -> registers with nothing to share: the A-operand fragments of a wavefront's own
//   bins are private to it (its three neighbours hold other bins).  A wavefront's fragments are stored in the order
//   it consumes them (Aw[w][chunk b] = D image of block b, the split tile's position, V' image of block b - 1), so its
//   stream is one linear walk; it runs through a private ring of 32 slots of 1 KiB filled by LDS-DMA (global_load_lds_dwordx4, no registers,
//   32 KiB in flight per wavefront - 32 frames per CU means the whole 75 MB image passes every CU once per
//   iteration, ~16 B/clk/CU at the matrix rate, and latency x bandwidth needs that much in flight); one 16-byte
//   ds_read per lane feeds two k-steps x two frame tiles = four 64-cycle MFMAs.
// * The products are software-pipelined across blocks: step i runs D of block i, then V' of block i - 1 - the
//   partial D tiles travel through LDS, are reduced and turned into H' while V' of the previous block occupies the
//   matrix pipe.
#include "evc_internal.h"

#include <stdlib.h>
#include <type_traits>

namespace evc {

typedef unsigned w64_u32x4 __attribute__((ext_vector_type(4)));
typedef double w64_d2 __attribute__((ext_vector_type(2)));

constexpr long W64_SPIN_LIMIT = 1L << 25;
constexpr int WIDE64_RING = 32;                    // (a power of two, <= 64: vmcnt is six bits)

struct Wide64Args {
    const double* Aw;        // [4][NB + 1][2 NL + 1][64][2] per wavefront and chunk b: D image of block b (NL = 2 TPW positions), the
                             // extra tile's position (D fragment of block b, V' fragment of block b - 1), V' image of block b - 1
    const double* Xw;        // [G][4][2][TPW + 1][2][64][2]    frames in the V chunk layout (slot TPW: the extra tile, partial per wavefront)
    double* Hw;              // [2 G][NB][2][64][2]         activations, quarter-major accumulator order
    double* Pw;              // the same layout: numerators A^T X
    double* Vpart;           // [2][G][c][4][2][TPW + 1][2][64][2]
    double* Vsum;            // [2][G][4][2][TPW + 1][2][64][2]   (reduce mode)
    unsigned* ticket;
    unsigned* done;          // [G]
    unsigned* done_r;        // [G]
    int* abort;
    const int* frame_utt;
    const int* active;
    const double* h0;
    int NB, TT, G, c, rmode;
    int it_begin, it_end;
    int N, T_;
    int mode;                // EVC_EPS_*
    double eps, l1;
    int init_const;
    int exact;               // 1: correctly rounded quotients (else one shared reciprocal per lane, <= 2 ulp)
    int static_q;            // 1: workgroup b runs sweep task b and reduce slice b of every iteration (G c <= CUs)
};

__device__ __forceinline__ w64_d2 ld2_sc1(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(w64_d2, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 16));
}
__device__ __forceinline__ w64_d2 ld2(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(w64_d2, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
__device__ __forceinline__ void st2_sc1(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, w64_d2 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(w64_u32x4, v), rs, voff, soff, 16);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc64(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// workgroup barrier that orders LDS traffic only (the global loads in flight stay in flight)
__device__ __forceinline__ void lds_barrier() {
#if defined(EVC_W64_ABLATE) && EVC_W64_ABLATE == 3
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return;
#endif
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// -DEVC_W64_ABLATE=n (diagnostic builds of tools/ubench/wide64_bench.hip, wrong results): 1: no waits for the fragment
// loads; 2: no fragment loads at all; 3: no workgroup barriers inside a step; 4: no update arithmetic; 5: no MFMAs
#ifndef EVC_W64_ABLATE
#define EVC_W64_ABLATE 0
#endif
#ifdef EVC_WIDE_STAMP
__device__ unsigned long long* evc_wide64_dbg = nullptr;        // [tasks of the launch][10]
#define W64STAMP(k) do { if (tid == 0 && evc_wide64_dbg) evc_wide64_dbg[(size_t)tk * 10 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define W64NOTE(k, v) do { if (tid == 0 && evc_wide64_dbg) evc_wide64_dbg[(size_t)tk * 10 + (k)] = (unsigned long long)(v); } while (0)
#else
#define W64STAMP(k)
#define W64NOTE(k, v)
#endif

// The MFMAs are asm: V (the B operand of D) is pinned to the AGPR half of the register file, every accumulator and the
// fragments to the VGPR half.  Left to the register allocator (either MFMA form), V' tiles travelled between the two
// halves around their MFMAs - 150 to 480 copies per block step, each pair of them a dependency stall next to a
// 64-cycle instruction (V' product at 85 to 103 cycles per MFMA instead of 64).  Wait states the hazard recogniser
// cannot see: s_nop 1 opens every string (a VALU-written operand or zeroed accumulator -> MFMA: 2 states); an
// accumulate chain needs none; w64_settle() stands between the last MFMA of an accumulator and any other reader
// (16-pass DGEMM result -> LDS / memory / VALU read: 18 states).
__device__ __forceinline__ void w64_mma_a(f64x4& acc, double x, double y_agpr) {
#if defined(EVC_W64_ABLATE) && EVC_W64_ABLATE == 5
    acc[0] += x * y_agpr;
#else
    asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "a"(y_agpr));
#endif
}
__device__ __forceinline__ void w64_mma_v(f64x4& acc, double x, double y) {
#if defined(EVC_W64_ABLATE) && EVC_W64_ABLATE == 5
    acc[0] += x * y;
#else
    asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
#endif
}
__device__ __forceinline__ void w64_settle() { asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); }

template <int TPW>
__global__ __launch_bounds__(256, 1) void k_fused_wide64(Wide64Args a) {
    constexpr int NL = 2 * TPW;                       // 16-byte fragment loads per product and wavefront
    constexpr int KH = TPW == 3 ? 1 : (TPW + 1) / 2;  // bin tiles of the V' product that run before the first barrier (the second
                                                      // half keeps >= 3 fragment positions for the update's three pieces)
    constexpr int TS = TPW + 1;                       // tile slots per frame tile in a wavefront's V chunk (the last: the extra tile)
    constexpr int CP = 2 * NL + 1;                    // fragment positions per block (chunk)
    constexpr unsigned WCH = 2u * TS * 2048u;         // bytes of a wavefront's V chunk (2 frame tiles x TS slots x 2 KiB)
    constexpr unsigned GCH = 4u * WCH;                // ... of a frame group's
    constexpr int R = WIDE64_RING;                    // 1 KiB slots of a wavefront's fragment ring
    extern __shared__ __attribute__((aligned(16))) char s_ring[];     // [4][R][1024]
    __shared__ w64_d2 s_d[4][2][2][64];               // partial D: [source wavefront][frame tile][register pair][lane]
    __shared__ w64_d2 s_h[2][2][64];                  // H' of the block: [frame tile][register pair][lane]
    __shared__ w64_d2 s_hp[4][2][64];                 // per wavefront: its quarter of the block's H, then of P
    __shared__ volatile unsigned s_ctl[4];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, i16 = lane & 15;
    const int ftq = w >> 1, xq = w & 1;               // the quarter this wavefront reduces and updates
    const unsigned GC = (unsigned)(a.G * a.c);
    const unsigned per_it = a.rmode ? 2u * GC : GC;
    const unsigned total = per_it * (unsigned)(a.it_end - a.it_begin);
    const unsigned c = (unsigned)a.c;
    const unsigned lane16 = (unsigned)lane * 16u;
    char* const ring_w = s_ring + w * (R * 1024);                     // this wavefront's ring
    const unsigned ring_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring_w + lane16;
    const unsigned ring_s = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring_w);
    const unsigned hp_s = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&s_hp[w][0][0]);

    // (the first ticket is the workgroup's index and the counter starts at the grid size: see k_fused_wide)
    unsigned nxt = 0;
    if (tid == 0) {
        nxt = blockIdx.x;
        s_ctl[0] = nxt;
        s_ctl[1] = 1u;
    }
    __syncthreads();

    auto wait_for = [&](const unsigned* ctr, unsigned need) -> bool {
        if (tid == 0) {
            unsigned ok = 1u;
            long spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                __builtin_amdgcn_s_sleep(2);
                ++spins;
                if ((spins & 63) == 0 && __hip_atomic_load(a.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    ok = 0u;
                    break;
                }
                if (spins > W64_SPIN_LIMIT) {
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

    for (;;) {
        const unsigned tk = __builtin_amdgcn_readfirstlane(s_ctl[0]);
        if (tk >= total) break;
        __syncthreads();
        const unsigned itl = tk / per_it, rem = tk - itl * per_it;
        // (static schedule: this workgroup's reduce slice, then its sweep of the next iteration; ticket queue: drawn when
        // the task is finished - see k_fused_wide)
        if (tid == 0 && a.static_q) nxt = (a.rmode && rem < GC) ? tk + GC : (itl + 1) * per_it + blockIdx.x;
        const int it = a.it_begin + (int)itl;
        const bool reduce = rem >= GC;
        const unsigned idx = reduce ? rem - GC : rem;
        const int g = (int)(idx / c), e = (int)(idx - (unsigned)g * c);
        const unsigned par = (unsigned)(it & 1);
        W64STAMP(0);
        W64NOTE(6, (reduce ? 1u : 0u) | ((unsigned)blockIdx.x << 8));
        W64NOTE(7, ((unsigned long long)it << 32) | (unsigned)(g * 256 + e));

        if (reduce) {
            // ---- reduce task: slice e of the group's V' = sum over the c ranges, in range order ----
            if (!wait_for(a.done + g, c * (unsigned)(it + 1))) break;
            W64STAMP(1);
            constexpr unsigned U = GCH / 16u;                     // 16-byte units of a group's partial
            const unsigned lo = (unsigned)((unsigned long)e * U / c), hi = (unsigned)((unsigned long)(e + 1) * U / c);
            const __amdgpu_buffer_rsrc_t rin = rsrc64(a.Vpart + ((size_t)(par * a.G + g) * c) * (GCH / 8), c * GCH);
            const __amdgpu_buffer_rsrc_t rout = rsrc64(a.Vsum + (size_t)(par * a.G + g) * (GCH / 8), GCH);
            for (unsigned un = lo + tid; un < hi; un += 256) {
                w64_d2 acc = w64_d2{0, 0};
                for (unsigned m0 = 0; m0 < c; m0 += 16) {
                    w64_d2 v[16];
#pragma unroll
                    for (unsigned k = 0; k < 16; ++k) {
                        const unsigned m = m0 + k < c ? m0 + k : c - 1;
                        v[k] = ld2_sc1(rin, un * 16u, m * GCH);
                    }
#pragma unroll
                    for (unsigned k = 0; k < 16; ++k)
                        if (m0 + k < c) acc = (m0 + k) ? acc + v[k] : v[k];
                }
                st2_sc1(rout, un * 16u, 0, acc);
            }
            W64STAMP(3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            W64STAMP(4);
            if (tid == 0) {
                __hip_atomic_fetch_add(a.done_r + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!a.static_q) nxt = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_ctl[0] = nxt;
            }
            __syncthreads();
            W64STAMP(5);
            continue;
        }

        // ---- sweep task (iteration it, frame group g, exemplar range e) ----
        const int j0 = (int)((long)e * a.NB / a.c), j1 = (int)((long)(e + 1) * a.NB / a.c), nb = j1 - j0;
        // The fragment stream of this task: chunks j0 .. j1 of this wavefront's image, 2 NL + 1 KiB each (the V' half of the
        // first and the D half of the last belong to the neighbouring ranges: fetched, not used).  Position p lives in
        // ring slot p % R; R positions are always in flight (the image is padded by R KiB).
        const char* gnext = reinterpret_cast<const char*>(a.Aw) +
                            ((size_t)w * (a.NB + 1) + (size_t)j0) * (CP * 1024u);             // (wave-uniform)
        unsigned pi = 0, pc = 0;                       // positions issued / consumed
        // LDS-DMA in asm: a DMA the compiler sees makes it wait for ALL outstanding ones before any LDS read it cannot
        // prove disjoint, i.e. it would drain the ring at every fragment read.  Hidden from it, the ring's LDS reads are
        // ordinary loads whose registers and lgkmcnt waits the compiler handles itself (an asm ds_read's destination
        // would count as written at once: under register pressure hipcc copied such a pending fragment into an AGPR
        // and reused the register - wrong values, and a fault when the late data landed on an address).  M0 is
        // saved and restored around it; s_nop 0: M0 written by SALU -> LDS-DMA reads it.
        auto dma = [&](const char* gbase, unsigned voff, unsigned lds_addr) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(gbase) : "memory");
        };
        auto issue = [&]() {
            if (EVC_W64_ABLATE == 2 || EVC_W64_ABLATE == 7) return;
            dma(gnext, lane16, ring_s + (pi & (R - 1)) * 1024u);
            gnext += 1024;
            ++pi;
        };
        // the 16 bytes of this lane in slot p % R (asynchronous: the caller's lgkmcnt wait names the destination)
        auto fetch = [&](unsigned p) -> w64_d2 {
            w64_d2 v;
            if (EVC_W64_ABLATE == 6 || EVC_W64_ABLATE == 7) return w64_d2{1.0 + p, 2.0};
            asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(ring_a + (p & (R - 1)) * 1024u) : "memory");
            return v;
        };
#pragma unroll
        for (int l = 0; l < R; ++l) issue();          // the dictionary does not depend on anybody
        if (it > 0 && !wait_for(a.rmode ? a.done_r + g : a.done + g, c * (unsigned)it)) break;
        W64STAMP(1);

        // the quarter of H / P this wavefront owns: frame tile 2 g + ftq, registers 2 xq, 2 xq + 1
        const int ft = 2 * g + ftq;
        bool live = false;
        double h0v = 0.0;
        {
            const int t = ft * 16 + i16;
            if (t < a.T_) {
                const int ut = a.frame_utt[t];
                if (ut >= 0) {
                    live = a.active[ut] != 0;
                    h0v = a.h0[ut];
                }
            }
        }
        // A frame group whose utterances have all stopped (or that is padding) does not sweep: its H stays, and so does
        // V' = A H - the task republishes the partial it published one iteration ago (the same bits a sweep would give).
        if (it > 0 && !__syncthreads_or(live ? 1 : 0)) {
            const __amdgpu_buffer_rsrc_t rsrc = rsrc64(a.Vpart + (((size_t)((par ^ 1u) * a.G + g) * c + e) * 4 + w) * (WCH / 8), WCH);
            const __amdgpu_buffer_rsrc_t rdst = rsrc64(a.Vpart + (((size_t)(par * a.G + g) * c + e) * 4 + w) * (WCH / 8), WCH);
#pragma unroll 4
            for (unsigned o = lane16; o < WCH; o += 1024u) st2_sc1(rdst, o, 0, ld2_sc1(rsrc, o, 0));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the ring's loads of this task too)
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(a.done + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!a.static_q) nxt = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_ctl[0] = nxt;
            }
            __syncthreads();
            continue;
        }
        // V of this wavefront's bins, both frame tiles
        f64x4 Vin[2][TPW], Vn[2][TPW], Vo[2];
        double Vodd[2] = {0.0, 0.0};                   // rows 4 w + q of the extra tile of V (B operand of its k-step w)
        {
            const double* src;
            unsigned nsum = 1, stride = 0;
            if (it == 0) {
                src = a.Xw + (size_t)g * (GCH / 8);
            } else if (a.rmode) {
                src = a.Vsum + (size_t)((par ^ 1u) * a.G + g) * (GCH / 8);
            } else {
                src = a.Vpart + ((size_t)((par ^ 1u) * a.G + g) * c) * (GCH / 8);
                nsum = c;
                stride = GCH;
            }
            const __amdgpu_buffer_rsrc_t rv = rsrc64(src, nsum * GCH);
            const unsigned wbase = (unsigned)w * WCH;
            for (unsigned m = 0; m < nsum; ++m) {
#pragma unroll
                for (int f = 0; f < 2; ++f) {
#pragma unroll
                    for (int k = 0; k < TPW; ++k) {
                        const unsigned o = wbase + (unsigned)((f * TS + k) * 2) * 1024u + lane16;
                        const w64_d2 v0 = ld2_sc1(rv, o, m * stride), v1 = ld2_sc1(rv, o + 1024u, m * stride);
                        const f64x4 v = f64x4{v0[0], v0[1], v1[0], v1[1]};
                        Vin[f][k] = m ? Vin[f][k] + v : v;
                    }
                    // the extra tile: register w of the sum of the four wavefronts' partial tiles (range order, then
                    // wavefront order)
#pragma unroll
                    for (unsigned ws = 0; ws < 4; ++ws) {
                        const w64_d2 v = ld2_sc1(rv, ws * WCH + (unsigned)((f * TS + TPW) * 2 + (w >> 1)) * 1024u + lane16, m * stride);
                        Vodd[f] += (w & 1) ? v[1] : v[0];
                    }
                }
            }
#pragma unroll
            for (int f = 0; f < 2; ++f) {
#pragma unroll
                for (int k = 0; k < TPW; ++k) Vn[f][k] = f64x4{0, 0, 0, 0};
                Vo[f] = f64x4{0, 0, 0, 0};
            }
        }
        const __amdgpu_buffer_rsrc_t rh = rsrc64(a.Hw + (size_t)ft * a.NB * 256, (unsigned)a.NB * 2048u);
        const __amdgpu_buffer_rsrc_t rp = rsrc64(a.Pw + (size_t)ft * a.NB * 256, (unsigned)a.NB * 2048u);
        const unsigned qoff = (unsigned)xq * 1024u + lane16;          // within a 2 KiB tile
        const char* const hbase = reinterpret_cast<const char*>(a.Hw + (size_t)ft * a.NB * 256);     // (wave-uniform)
        const char* const pbase = reinterpret_cast<const char*>(a.Pw + (size_t)ft * a.NB * 256);
        auto dma_sc1 = [&](const char* gbase, unsigned voff, unsigned lds_addr) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 sc1\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(gbase) : "memory");
        };
        const bool load_h = it > 0 || !a.init_const, load_p = it > 0;
        const int n_edge = (a.N & 15) ? a.NB - 1 : -1;
        f64x4 hf[2];                                   // H' of the previous block, both frame tiles (B operand of V')
        double hfw[2] = {0.0, 0.0};                    // ... and its register w (B operand of the extra tile's k-step w)
        hf[0] = f64x4{0, 0, 0, 0};
        hf[1] = hf[0];
        s_h[ftq][xq][lane] = w64_d2{0, 0};             // (the first step's V' product multiplies by H' = 0)
        lds_barrier();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 1) : "memory");
        w64_d2 f0 = fetch(0);                          // (the ring was filled before the dependency wait)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0)::"memory");
        W64STAMP(2);
        W64NOTE(8, __builtin_amdgcn_s_memtime());
        // One section of a product: stream positions pc .. pc + n - 1, four MFMAs each (mf(l, j, fragment), j = 0..3).
        // A wavefront alone on its SIMD issues in order: whatever stands between two MFMAs runs in the shadow of the
        // first (64 cycles), whatever follows the last one of a group delays the next group.  So the other work of a
        // position is dealt out between its MFMAs: after the first the read of position p + 1 from LDS (waited for
        // after the fourth: it has returned by then), after the second the refill of the slot R positions ahead.
        // The fragment of position p + 1 is requested from LDS behind the first MFMA of position p and waited for
        // behind its fourth.  The read is asm with the wait as a second statement naming the registers: the same read
        // as a C++ load costs 8 cycles per MFMA (tools/ubench/mfma64_fill.hip against mfma64_ports.hip: 72.0 / 66.8
        // ticks).  Between the two statements the compiler must not touch the destination (it counts as written at
        // once): tools/asm_audit.py checks the generated code for exactly that - the code THIS container's hipcc
        // (ROCm 7.2) generates; another compiler release needs the audit run again (tests/test_wide64_build.py runs
        // it on every CPU test pass, on whatever hipcc builds the library).  f0, the fragment of position pc,
        // is carried from section to section and step to step.
        // vmcnt counts in order: with R - 2 younger loads issued, vmcnt <= R - 2 says position pc + 1 has landed; other
        // loads and stores in between only make the wait stricter.
        auto section = [&](auto nc, auto&& mf, auto&& between) {
            constexpr int n = decltype(nc)::value;
#pragma unroll
            for (int l = 0; l < n; ++l) {
                mf(l, 0, f0);
                __builtin_amdgcn_sched_barrier(0);
                if (EVC_W64_ABLATE != 1 && EVC_W64_ABLATE != 2 && EVC_W64_ABLATE != 7) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 2) : "memory");
                w64_d2 f1 = fetch(pc + 1);
                between(l, 0);
                __builtin_amdgcn_sched_barrier(0);
                mf(l, 1, f0);
                __builtin_amdgcn_sched_barrier(0);
                issue();
                between(l, 1);
                __builtin_amdgcn_sched_barrier(0);
                mf(l, 2, f0);
                __builtin_amdgcn_sched_barrier(0);
                between(l, 2);
                __builtin_amdgcn_sched_barrier(0);
                mf(l, 3, f0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f1)::"memory");
                between(l, 3);               // (branchy work belongs here: outside the window in which f1 is pending)
                __builtin_amdgcn_sched_barrier(0);
                f0 = f1;
                ++pc;
            }
        };
        auto nothing = [](int, int) {};
        auto mfma_v = [&](int l, int j, const w64_d2& fr) {
            const int k = l >> 1, h = l & 1, f = j & 1, x = j >> 1;
            w64_mma_v(Vn[f][k], fr[x], hf[f][2 * h + x]);
        };

        // step i = 0 .. nb: D of block j0 + i, then V' of block j0 + i - 1.  ONE loop body for every step (peeled first
        // and last steps made the register allocator shuffle all of V' between two homes in every step): the first
        // step multiplies the previous range's V' image by H' = 0, the last one forms a D nobody reads (one block's
        // worth of MFMAs per task).  H and P of the block are requested at the top of the step and used behind the
        // first barrier, half a step later (nothing loaded is carried from step to step: a loop-carried load would
        // make the compiler drain the ring).
#ifdef EVC_W64_TIMERS
        unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define W64TICK(k) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tacc[k] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define W64TICK(k)
#endif
#pragma clang loop unroll(disable)
        for (int i = 0; i <= nb; ++i) {
            const bool has_d = i < nb;                 // (wave-uniform)
            const int jb = j0 + i;
            // H and P quarter of the block -> LDS (sc1: written by another workgroup in the previous iteration); they are the
            // oldest memory operations of the step, NL + 1 + 2 KH fragment loads younger when they are read
            if (has_d) {
                if (load_h) dma_sc1(hbase + (size_t)jb * 2048u, qoff, hp_s);
                if (load_p) dma_sc1(pbase + (size_t)jb * 2048u, qoff, hp_s + 1024u);
            }
            // ---- D partial of block jb over this wavefront's bins
            f64x4 da[2];
            da[0] = f64x4{0, 0, 0, 0};
            da[1] = da[0];
            section(std::integral_constant<int, NL>{}, [&](int l, int j, const w64_d2& fr) {
                const int k = l >> 1, h = l & 1, f = j & 1, x = j >> 1;
                w64_mma_a(da[f], fr[x], Vin[f][k][2 * h + x]);
            }, [&](int l, int j) {
                // H' of the previous block (the B operand of the V' product that follows): read here, three positions
                // before it is needed - behind the second barrier of the previous step its latency stood in the open
                if (l == NL - 3 && j == 2) {
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const w64_d2 v0 = s_h[f][0][lane], v1 = s_h[f][1][lane];
                        hf[f] = f64x4{v0[0], v0[1], v1[0], v1[1]};
                        hfw[f] = reinterpret_cast<const double*>(&s_h[f][w >> 1][lane])[w & 1];     // (register w of it)
                    }
                }
            });
            // ---- the extra tile's position: rows 4 w + q of it into the partial D; k-step w of block jb - 1 into its V'
            section(std::integral_constant<int, 1>{}, [&](int, int j, const w64_d2& fr) {
                if (j < 2) w64_mma_v(da[j], fr[0], Vodd[j]);
                else w64_mma_v(Vo[j - 2], fr[1], hfw[j - 2]);
            }, nothing);
            W64TICK(0);
            w64_settle();
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const f64x4 d = da[f];
                s_d[w][f][0][lane] = w64_d2{d[0], d[1]};
                s_d[w][f][1][lane] = w64_d2{d[2], d[3]};
            }
            // ---- V' += A_(jb-1) H'_(jb-1), first half of the bin tiles
            section(std::integral_constant<int, 2 * KH>{}, [&](int l, int j, const w64_d2& fr) { mfma_v(l, j, fr); }, nothing);
            W64TICK(1);
            lds_barrier();                              // the four partial D tiles are in LDS
            W64TICK(2);
            // ---- the second half of V' of block jb - 1; between its MFMAs, piece by piece: this wavefront's quarter of D
            // summed in wavefront order, the update, the quarter of H' to memory and to LDS
            // f64 VALU work does not hide behind f64 MFMAs (measured: the update costs the same ~1.2 k cycles per step in one
            // piece, or cut into thirteen pieces dealt out over thirteen gaps - the matrix instruction and the vector
            // unit's f64 arithmetic do not overlap), so it is kept SHORT rather than spread: three pieces (loads; the
            // arithmetic; stores), and the two quotients of a lane share one reciprocal, 1 / (den0 den1), refined by two
            // Newton steps (<= 2 ulp; lanes whose product leaves [1e-280, 1e280], and callers that ask for the
            // correctly rounded quotient - `exact` - take the division).
            w64_d2 d0, d1, d2, d3, hn, hC = w64_d2{0, 0}, pC = hC;
            auto update_piece = [&](int l, int j) {
                if (!has_d) return;
                const int gap = 4 * l + j;
                if (gap == 3) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL + 1 + 2 * KH) : "memory");
                    if (load_h) hC = s_hp[w][0][lane];
                    if (load_p) pC = s_hp[w][1][lane];
                    d0 = s_d[0][ftq][xq][lane];
                    d1 = s_d[1][ftq][xq][lane];
                    d2 = s_d[2][ftq][xq][lane];
                    d3 = s_d[3][ftq][xq][lane];
                } else if (gap == 7) {
                    d0 = ((d0 + d1) + d2) + d3;
                    if (it == 0) {
                        if (a.init_const) {
#pragma unroll
                            for (int y = 0; y < 2; ++y) hn[y] = (jb * 16 + q + 4 * (2 * xq + y) < a.N) ? h0v : 0.0;
                        } else {
                            hn = hC;
                        }
                    } else if (EVC_W64_ABLATE == 4) {
                        hn = hC + 1e-300 * d0;
                    } else {
                        // numerator and denominator of the surface's update (mu_update, evc_internal.h)
                        w64_d2 num, den, quo;
#pragma unroll
                        for (int y = 0; y < 2; ++y) {
                            const double dl = d0[y] + a.l1;
                            switch (a.mode) {
                                case EVC_EPS_ADD: num[y] = hC[y] * pC[y]; den[y] = dl + a.eps; break;
                                case EVC_EPS_ZERO_REPLACE: num[y] = pC[y]; den[y] = dl == 0.0 ? a.eps : dl; break;
                                case EVC_EPS_CLAMP: num[y] = pC[y]; den[y] = dl > a.eps ? dl : a.eps; break;
                                default: num[y] = hC[y] * pC[y]; den[y] = dl; break;
                            }
                        }
                        const double m = den[0] * den[1];
                        if (!a.exact && m > 1e-280 && m < 1e280) {
                            double r = __builtin_amdgcn_rcp(m);
                            r = __builtin_fma(__builtin_fma(-m, r, 1.0), r, r);
                            r = __builtin_fma(__builtin_fma(-m, r, 1.0), r, r);
                            quo[0] = num[0] * (r * den[1]);
                            quo[1] = num[1] * (r * den[0]);
                        } else {
                            quo[0] = num[0] / den[0];
                            quo[1] = num[1] / den[1];
                        }
#pragma unroll
                        for (int y = 0; y < 2; ++y) {
                            hn[y] = (a.mode == EVC_EPS_ZERO_REPLACE || a.mode == EVC_EPS_CLAMP) ? hC[y] * quo[y] : quo[y];
                            if (jb == n_edge) hn[y] = (jb * 16 + q + 4 * (2 * xq + y) < a.N) ? hn[y] : 0.0;
                            hn[y] = live ? hn[y] : hC[y];
                        }
                    }
                } else if (gap == 11) {
                    if (it == 0) st2_sc1(rp, qoff, (unsigned)jb * 2048u, d0);
                    if (it != 0 || a.init_const) st2_sc1(rh, qoff, (unsigned)jb * 2048u, hn);
                    s_h[ftq][xq][lane] = hn;
                }
            };
            static_assert(NL - 2 * KH >= 3, "the update needs three positions of the second half");
            section(std::integral_constant<int, NL - 2 * KH>{}, [&](int l, int j, const w64_d2& fr) { mfma_v(l + 2 * KH, j, fr); },
                    update_piece);
            W64TICK(3);
            lds_barrier();                              // the four quarters of H' are in LDS
            W64TICK(4);
        }

#ifdef EVC_W64_TIMERS
        if (tid == 0 && evc_wide64_dbg) {           // cycles per step: D | V' first half | barrier | V' second half + update | barrier
            evc_wide64_dbg[(size_t)tk * 10 + 1] = tacc[0] / (nb + 1);
            evc_wide64_dbg[(size_t)tk * 10 + 4] = tacc[1] / (nb + 1);
            evc_wide64_dbg[(size_t)tk * 10 + 6] = tacc[2] / (nb + 1);
            evc_wide64_dbg[(size_t)tk * 10 + 7] = tacc[3] / (nb + 1);
            evc_wide64_dbg[(size_t)tk * 10 + 8] = tacc[4] / (nb + 1);
            evc_wide64_dbg[(size_t)tk * 10 + 2] = tacc[5] / (nb + 1);
            evc_wide64_dbg[(size_t)tk * 10 + 3] = tacc[6] / (nb + 1);
        }
#endif
        w64_settle();
        W64STAMP(3);
        W64NOTE(9, __builtin_amdgcn_s_memtime());
        // publish the partial V' of this range
        {
            const __amdgpu_buffer_rsrc_t rv =
                rsrc64(a.Vpart + (((size_t)(par * a.G + g) * c + e) * 4 + w) * (WCH / 8), WCH);
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int k = 0; k < TS; ++k) {
                    const f64x4 v = k < TPW ? Vn[f][k < TPW ? k : 0] : Vo[f];
                    const unsigned o = (unsigned)((f * TS + k) * 2) * 1024u + lane16;
                    st2_sc1(rv, o, 0, w64_d2{v[0], v[1]});
                    st2_sc1(rv, o + 1024u, 0, w64_d2{v[2], v[3]});
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        W64STAMP(4);
        if (tid == 0) {
            __hip_atomic_fetch_add(a.done + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!a.static_q) nxt = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ctl[0] = nxt;
        }
        __syncthreads();
        W64STAMP(5);
    }
}

// ------------------------------------------------------------------------------------------
// packing / unpacking
// ------------------------------------------------------------------------------------------
// Aw[w][b][position][lane = 16 q + i][x], position = 0 .. 2 NL (NL = 2 TPW):
//   position 2 k + h           : A[bin 16 (4 k + w) + 4 (2 h + x) + q][exemplar 16 b + i]              (A operand of D, block b)
//   position NL                : x = 0: A[bin 64 TPW + 4 w + q][exemplar 16 b + i]                     (the extra tile: D, rows 4 w + q)
//                                x = 1: A[bin 64 TPW + i][exemplar 16 (b - 1) + 4 w + q]               (... V' of block b - 1, k-step w)
//   position NL + 1 + 2 k + h  : A[bin 16 (4 k + w) + i][exemplar 16 (b - 1) + 4 (2 h + x) + q]        (A operand of V', block b - 1)
// b = 0 .. NB (block -1 and block NB: zeros), then WIDE64_RING KiB of zeros.  At: exemplars as rows (n_rows x ld, zero
// padded), bins < ld
__global__ __launch_bounds__(256) void k_wide64_pack_dict(const double* __restrict__ At, int ld, int n_rows, int NB,
                                                          int TPW, long total, double* __restrict__ Aw) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int NL = 2 * TPW;
    const long per_chunk = (2L * NL + 1) * 128, per_wave = (NB + 1) * per_chunk;
    double v = 0.0;
    if (gid < 4 * per_wave) {
        const int w = (int)(gid / per_wave);
        const long o1 = gid - w * per_wave, b = o1 / per_chunk;
        const int o = (int)(o1 - b * per_chunk);
        const int x = o & 1, lane = (o >> 1) & 63, pos = o >> 7;
        const int q = lane >> 4, i = lane & 15;
        int bin;
        long n;
        if (pos < NL) {
            const int k = pos >> 1, h = pos & 1;
            bin = 16 * (4 * k + w) + 4 * (2 * h + x) + q;
            n = 16 * b + i;
        } else if (pos == NL) {
            bin = x ? 64 * TPW + i : 64 * TPW + 4 * w + q;
            n = x ? 16 * (b - 1) + 4 * w + q : 16 * b + i;
        } else {
            const int k = (pos - NL - 1) >> 1, h = (pos - NL - 1) & 1;
            bin = 16 * (4 * k + w) + i;
            n = 16 * (b - 1) + 4 * (2 * h + x) + q;
        }
        const long jb = n >> 4;          // (n >= -16: block -1 is n < 0)
        if (n >= 0 && jb < NB && n < n_rows && bin < ld) v = At[n * ld + bin];
    }
    Aw[gid] = v;
}

// Xw[g][w][f][k][h][lane = 16 q + i][x] = X[frame 16 (2 g + f) + i][bin 16 (4 k + w) + q + 4 (2 h + x)], k < TPW;
// slot k = TPW: the extra tile, X[frame][bin 64 TPW + q + 4 (2 h + x)] in wavefront 0's slot, zeros in the others
__global__ __launch_bounds__(256) void k_wide64_pack_x(const double* __restrict__ Xt, int ld, int rows, long G, int TPW,
                                                       double* __restrict__ Xw) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int TS = TPW + 1;
    const long per_group = 4L * 2 * TS * 256;
    if (gid >= G * per_group) return;
    const long g = gid / per_group;
    int o = (int)(gid - g * per_group);
    const int x = o & 1, lane = (o >> 1) & 63, h = (o >> 7) & 1;
    o >>= 8;
    const int k = o % TS, f = (o / TS) & 1, w = o / (2 * TS);
    const int q = lane >> 4, i = lane & 15;
    const long t = 16 * (2 * g + f) + i;
    const int bin = k < TPW ? 16 * (4 * k + w) + q + 4 * (2 * h + x) : 64 * TPW + q + 4 * (2 * h + x);
    Xw[gid] = (t < rows && bin < ld && (k < TPW || w == 0)) ? Xt[t * ld + bin] : 0.0;
}

// Hw[ft][jb][x][lane = 16 q + i][y] <-> H[exemplar 16 jb + q + 4 (2 x + y)][frame 16 ft + i]
template <bool IMPORT>
__global__ __launch_bounds__(256) void k_wide64_h_io(double* __restrict__ H, long ldh, int frame_major, int T_, int N,
                                                     w64_d2* __restrict__ Hw, long units, int NB, const int* abort) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;       // one 16-byte unit each
    if (gid >= units) return;
    const int lane = (int)(gid & 63), x = (int)((gid >> 6) & 1);
    const long tile = gid >> 7, ft = tile / NB, jb = tile % NB;
    const int q = lane >> 4, i = lane & 15;
    const long t = 16 * ft + i;
    if (IMPORT) {
        w64_d2 v = w64_d2{0, 0};
        if (t < T_) {
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const long n = 16 * jb + q + 4 * (2 * x + y);
                if (n < N) v[y] = frame_major ? H[t * ldh + n] : H[n * ldh + t];
            }
        }
        Hw[gid] = v;
    } else {
        if (t >= T_) return;
        w64_d2 v = Hw[gid];
        if (abort && *abort) v = w64_d2{__builtin_nan(""), __builtin_nan("")};
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const long n = 16 * jb + q + 4 * (2 * x + y);
            if (n < N) {
                if (frame_major) H[t * ldh + n] = v[y]; else H[n * ldh + t] = v[y];
            }
        }
    }
}

// err2[t] = sum_m (X - V)^2 of frame t, V = the sum of the c partials of iteration `it` (or their reduced sum);
// one wavefront per frame tile
__global__ __launch_bounds__(256) void k_wide64_err2(Wide64Args a, int TPW, int it, double* __restrict__ err2) {
    const int lane = threadIdx.x & 63;
    const long ft = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ft >= a.TT) return;
    const int g = (int)(ft >> 1), f = (int)(ft & 1);
    const unsigned par = (unsigned)(it & 1);
    const int TS = TPW + 1;
    const size_t wch = (size_t)2 * TS * 128, gch = 4 * wch;           // 16-byte units
    const w64_d2* xw = reinterpret_cast<const w64_d2*>(a.Xw) + (size_t)g * gch;
    const w64_d2* vs = reinterpret_cast<const w64_d2*>(a.Vsum) + (size_t)(par * a.G + g) * gch;
    const w64_d2* vp = reinterpret_cast<const w64_d2*>(a.Vpart) + (size_t)(par * a.G + g) * a.c * gch;
    auto v_at = [&](size_t o) {
        if (a.rmode) return vs[o];
        w64_d2 v = vp[o];
        for (int m = 1; m < a.c; ++m) v += vp[(size_t)m * gch + o];
        return v;
    };
    double acc = 0.0;
    for (int w = 0; w < 4; ++w)
        for (int kh = 0; kh < 2 * TPW; ++kh) {
            const size_t o = (size_t)w * wch + ((size_t)f * 2 * TS + kh) * 64 + lane;
            const w64_d2 v = v_at(o), x = xw[o];
            acc += (x[0] - v[0]) * (x[0] - v[0]) + (x[1] - v[1]) * (x[1] - v[1]);
        }
    for (int h = 0; h < 2; ++h) {            // the extra tile: the four wavefronts' partial sums first
        w64_d2 v = w64_d2{0, 0}, x = v;
        for (int w = 0; w < 4; ++w) {
            const size_t o = (size_t)w * wch + ((size_t)f * 2 * TS + 2 * TPW + h) * 64 + lane;
            v += v_at(o);
            x += xw[o];
        }
        acc += (x[0] - v[0]) * (x[0] - v[0]) + (x[1] - v[1]) * (x[1] - v[1]);
    }
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    const long t = 16 * ft + (lane & 15);
    if (lane < 16 && t < a.T_) err2[t] = acc;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static const int WIDE64_TPW_SET[] = {3, 4, 5, 7, 8};      // whole bin tiles per wavefront; + one tile split over the four

size_t wide_ctl_words(const Wide64Layout& f) { return 4 + 2 * (size_t)f.G; }

bool wide64_supported(int M, int N, int T_, int dtype, int algo, int loss) {
    return dtype == EVC_F64 && algo == EVC_ALGO_FACTORED && loss == EVC_LOSS_FROBENIUS && M > 144 && M <= 528 &&
           N >= 16 && T_ >= 1;
}

Wide64Layout wide64_layout(int M, int N, int T_, int n_cus, int c_req, int tpw_req) {
    Wide64Layout f{};
    f.TPW = 8;
    for (int v : WIDE64_TPW_SET)
        if (64 * v + 16 >= M && v >= tpw_req) { f.TPW = v; break; }
    f.NB = (N + 15) / 16;
    f.TT = (T_ + 15) / 16;
    f.G = (f.TT + 1) / 2;
    if (n_cus <= 0) n_cus = 256;
    // (evc_wide.hip's round-4 rule - from 32 groups on up to 8 ranges without reduce tasks - was measured here too and
    // lost: two utterances of the C3 shape, 43 groups: 6 ranges through the queue 0.481 of the peak, 5 ranges with
    // reduce slices on the static schedule 0.547)
    int c = c_req > 0 ? c_req : (f.G >= n_cus ? 1 : (f.G * 4 >= n_cus ? (n_cus + f.G - 1) / f.G : n_cus / f.G));
    const int cmax = f.NB / 2 > 0 ? f.NB / 2 : 1;
    if (c > cmax) c = cmax;
    if (c > 64) c = 64;
    f.c = c;
    f.rmode = c > 4 ? 1 : 0;
    const size_t gch = (size_t)4 * 2 * (f.TPW + 1) * 256;       // doubles per group chunk
    f.aw = (size_t)4 * (f.NB + 1) * (4 * f.TPW + 1) * 128 + (size_t)WIDE64_RING * 128;
    f.xw = (size_t)f.G * gch;
    f.hw = (size_t)f.G * 2 * f.NB * 256;
    f.vpart = 2 * (size_t)f.G * f.c * gch;
    f.vsum = 2 * (size_t)f.G * gch;
    return f;
}

Wide64Caps wide64_caps(int M, int N, int T_, int n_cus) {
    const Wide64Layout a = wide64_layout(M, N, T_, n_cus, 0, 0);
    Wide64Caps k{};
    int c_cap = a.c;
    if (a.TT <= 4096 && c_cap < 8) c_cap = 8;
    if (a.TT <= 256) c_cap = 64;
    const int cmax = a.NB / 2 > 0 ? a.NB / 2 : 1;
    if (c_cap > cmax) c_cap = cmax;
    k.c_cap = c_cap;
    const size_t gch = (size_t)4 * 2 * (a.TPW + 1) * 256;
    k.aw = a.aw;
    k.xw = a.xw;
    k.hw = a.hw;
    k.vpart = 2 * (size_t)a.G * c_cap * gch;
    k.vsum = a.vsum;
    k.ctl = wide_ctl_words(a);
    return k;
}
bool wide_fits(const Wide64Layout& f, const Wide64Caps& k) {
    return f.aw <= k.aw && f.xw <= k.xw && f.hw <= k.hw && f.vpart <= k.vpart && f.vsum <= k.vsum &&
           wide_ctl_words(f) <= k.ctl;
}

hipError_t wide_pack_dict(const Wide64Layout& f, const double* At, const double*, int ld, int n_rows, double* Aw,
                          hipStream_t s) {
    const long n = (long)f.aw;
    hipLaunchKernelGGL(k_wide64_pack_dict, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, At, ld, n_rows, f.NB, f.TPW,
                       n, Aw);
    return hipGetLastError();
}

hipError_t wide_pack_x(const Wide64Layout& f, const double* Xt, int ld, int rows, double* Xw, hipStream_t s) {
    const long n = (long)f.xw;
    hipLaunchKernelGGL(k_wide64_pack_x, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Xt, ld, rows, (long)f.G, f.TPW,
                       Xw);
    return hipGetLastError();
}

hipError_t wide_import_h(const Wide64Layout& f, double* Hw, const double* H, long ldh, int frame_major, int T_, int N,
                           hipStream_t s) {
    const long units = (long)f.G * 2 * f.NB * 128;
    hipLaunchKernelGGL((k_wide64_h_io<true>), dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, const_cast<double*>(H),
                       ldh, frame_major, T_, N, reinterpret_cast<w64_d2*>(Hw), units, f.NB, (const int*)nullptr);
    return hipGetLastError();
}

hipError_t wide_export_h(const Wide64Layout& f, const double* Hw, double* H, long ldh, int frame_major, int T_, int N,
                           const int* abort, hipStream_t s) {
    const long units = (long)f.TT * f.NB * 128;
    hipLaunchKernelGGL((k_wide64_h_io<false>), dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, H, ldh, frame_major,
                       T_, N, reinterpret_cast<w64_d2*>(const_cast<double*>(Hw)), units, f.NB, abort);
    return hipGetLastError();
}

static Wide64Args wide64_args(const Wide64Layout& f, const Wide64Buffers& b, const UttState& u, int N, int T_, int mode,
                              double eps, double l1, int init_const, int exact = 0) {
    Wide64Args a{};
    a.Aw = b.Aw; a.Xw = b.Xw; a.Hw = b.Hw; a.Pw = b.Pw; a.Vpart = b.Vpart; a.Vsum = b.Vsum;
    a.ticket = b.ctl; a.done = b.ctl + 4; a.done_r = b.ctl + 4 + f.G; a.abort = reinterpret_cast<int*>(b.ctl + 1);
    a.frame_utt = u.frame_utt; a.active = u.active; a.h0 = u.h0;
    a.NB = f.NB; a.TT = f.TT; a.G = f.G; a.c = f.c; a.rmode = f.rmode;
    a.N = N; a.T_ = T_; a.mode = mode; a.eps = eps; a.l1 = l1; a.init_const = init_const; a.exact = exact;
    return a;
}

hipError_t wide_begin(const Wide64Layout& f, const Wide64Buffers& b, hipStream_t s) {
    return hipMemsetAsync(b.ctl, 0, wide_ctl_words(f) * sizeof(unsigned), s);
}

template <int TPW>
static hipError_t wide64_launch(const Wide64Args& a, unsigned grid, hipStream_t s) {
    const size_t lds = (size_t)4 * WIDE64_RING * 1024;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_wide64<TPW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_fused_wide64<TPW>), dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t wide_iterate(const Wide64Layout& f, const Wide64Buffers& b, const UttState& u, int N, int T_, int it_begin,
                          int it_end, int mode, double eps, double l1, int init_const, int n_cus, hipStream_t s) {
    if (it_end <= it_begin) return hipSuccess;
    // (mode: EVC_EPS_*, + 0x1000 when the caller wants correctly rounded quotients)
    Wide64Args a = wide64_args(f, b, u, N, T_, mode & 0xfff, eps, l1, init_const, (mode >> 12) & 1);
    a.it_begin = it_begin; a.it_end = it_end;
    const long per_it = (long)f.G * f.c * (f.rmode ? 2 : 1);
    const long tasks = per_it * (it_end - it_begin);
    if (n_cus <= 0) n_cus = 256;
#ifdef EVC_WIDE_STAMP      // (the stand-alone harnesses only: the library reads nothing from the environment)
    static const int no_static = getenv("EVC_WIDE_NO_STATIC") ? atoi(getenv("EVC_WIDE_NO_STATIC")) : 0;
#else
    constexpr int no_static = 0;
#endif
    a.static_q = (f.G * f.c <= n_cus && !no_static) ? 1 : 0;
    const unsigned grid = a.static_q ? (unsigned)(f.G * f.c) : (unsigned)(tasks < n_cus ? tasks : n_cus);
    hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b.ctl), (int)grid, 1, s);
    if (e != hipSuccess) return e;
    switch (f.TPW) {
        case 3: return wide64_launch<3>(a, grid, s);
        case 4: return wide64_launch<4>(a, grid, s);
        case 5: return wide64_launch<5>(a, grid, s);
        case 7: return wide64_launch<7>(a, grid, s);
        case 8: return wide64_launch<8>(a, grid, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t wide_err2(const Wide64Layout& f, const Wide64Buffers& b, const UttState& u, int N, int T_, int it, int,
                     double, double* err2, hipStream_t s) {
    Wide64Args a = wide64_args(f, b, u, N, T_, 0, 0.0, 0.0, 0);
    hipLaunchKernelGGL(k_wide64_err2, dim3((unsigned)((f.TT + 3) / 4)), dim3(256), 0, s, a, f.TPW, it, err2);
    return hipGetLastError();
}

}  // namespace evc
