// Internal declarations shared by the translation units of libevc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/evc.h"

namespace evc {

// ------------------------------------------------------------------------------------------
// MFMA 16x16x4 wrappers.  One operand element per lane for A and B:
//   A-operand lane l holds Aop[i = l & 15][k = l >> 4]
//   B-operand lane l holds Bop[k = l >> 4][j = l & 15]
// C/D (4 values per lane, register r):  column j = l & 15 and
//   f64: row i = (l >> 4) + 4 r          (v_mfma_f64_16x16x4_f64)
//   f32: row i = 4 (l >> 4) + r          (v_mfma_f32_16x16x4_f32)
// ------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mma;
template <> struct Mma<double> {
    typedef f64x4 acc_t;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Mma<float> {
    typedef f32x4 acc_t;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};

// ------------------------------------------------------------------------------------------
// The element-wise multiplicative update, one statement per reference surface.
//   h: current activation, p: numerator (A^T X), d: denominator (A^T A H)
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T mu_update(T h, T p, T d, int eps_mode, T eps, T l1) {
    d += l1;                                        // sklearn _nmf.py:615-617 (l1 == 0: identity)
    switch (eps_mode) {
        case EVC_EPS_ADD:                           // pymf nmf.py:68-70
            return (h * p) / (d + eps);
        case EVC_EPS_ZERO_REPLACE:                  // sklearn _nmf.py:620-629
            d = (d == T(0)) ? eps : d;
            return h * (p / d);
        case EVC_EPS_CLAMP:                         // deComP batch_mu.py
            return h * (p / (d > eps ? d : eps));
        default:                                    // EVC_EPS_NONE, nmf_tool nmf.py:39
            return (h * p) / d;
    }
}

// per-utterance bookkeeping that lives in the workspace
struct UttState {
    int* frame_utt;      // [Tp]  utterance index of each frame (padding frames: -1)
    int* offsets;        // [n_utt+1]
    int* active;         // [n_utt + 1]: per utterance; [n_utt]: how many are active (the gate of the generic path's kernels)
    int* n_iter;         // [n_utt]
    double* err_init;    // [n_utt]
    double* err_prev;    // [n_utt]
    double* h0;          // [n_utt]   initial activation value (INIT_SKLEARN / CONST)
    double* trace;       // [n_utt][n_slots]
    int n_slots;
};

template <typename T> struct MuEpilogue {
    const T* Hin;        // [Tp][ldh]
    const T* P;          // [Tp][ldh]
    const int* frame_utt;
    const int* active;
    int ldh;
    int N, T_;           // true (unpadded) sizes
    int eps_mode;
    T eps, l1;
    int kl;              // 1: Hout = Hin * acc (the KL numerator over a pre-scaled dictionary); P unused
    const int* gate;     // optional: the kernel returns at once when *gate == 0 (no utterance is active any more)
};

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ----- evc_gemm.hip -----
// C[I x J] = L[I x Kd] * R[J x Kd]^T, all row-major, I % 128 == 0, J % 64 == 0, Kd % 16 == 0.
// scratch (optional, scratch_elems elements): lets a small-grid, long-K product be split over K.
template <typename T>
hipError_t gemm_nt(const T* L, int ldl, const T* R, int ldr, T* C, int ldc, int I, int J, int Kd,
                   hipStream_t s, T* scratch = nullptr, size_t scratch_elems = 0, int* splits_out = nullptr,
                   int j_valid = 0, const int* gate = nullptr);
// gate: optional device word; the launched kernels return at once when it is 0 (the stop rules have stopped every
// utterance: the launches the host has still queued cost a few microseconds each instead of a contraction)
// j_valid: rows of R from j_valid on are known to be zero (padding up to the block width); their products are skipped
// splits_out: when the contraction was split over k into slabs in `scratch` (slab z at scratch + z * I * ldc),
// *splits_out = their number and C is NOT written - the caller's next kernel sums them in order; else 0.
// Same contraction with the multiplicative update as epilogue: C = mu(Hin, P, L R^T).
template <typename T>
hipError_t gemm_nt_mu(const T* L, int ldl, const T* R, int ldr, T* Hout, int I, int J, int Kd,
                      const MuEpilogue<T>& ep, hipStream_t s);
// ----- evc_gemm2.hip: second-generation contraction kernel (swizzled row-major LDS, 16-byte fragment reads,
// vector epilogue); gemm_nt / gemm_nt_mu route to it whenever shapes and alignment allow -----
template <typename T>
bool gemm2_ok(const T* L, int ldl, const T* R, int ldr, const T* C, int ldc, int I, int J, int Kd);
template <typename T>
hipError_t gemm2(const T* L, int ldl, const T* R, int ldr, T* C, int ldc, int I, int J, int Kd, hipStream_t s,
                 T* scratch, size_t scratch_elems, int* splits_out, int n_cus, int j_valid, const int* gate = nullptr);
template <typename T>
hipError_t gemm2_mu(const T* L, int ldl, const T* R, int ldr, T* Hout, int I, int J, int Kd,
                    const MuEpilogue<T>& ep, hipStream_t s);
// C = sum_z part[z * slab + .]  (fixed order)
template <typename T>
hipError_t sum_slabs(const T* part, long slab, int splits, T* C, hipStream_t s, const int* gate = nullptr);
// Bounds-checked general-stride contraction on caller memory (used by evc_synthesize):
// C[i*csi + j*csj] = sum_k L[i*lsi + k*lsk] * R[j*rsj + k*rsk]
template <typename T>
hipError_t gemm_strided(const T* L, long lsi, long lsk, const T* R, long rsj, long rsk, T* C,
                        long csi, long csj, int I, int J, int Kd, hipStream_t s);
// Y = B H for Mb <= 64 bins (evc_synthesize): H(t,n) = H[t hst + n hsn], B(n,mb) = B[n bsn + mb bsm], Y(t,mb) = Y[t yst + mb ysm]
template <typename T>
hipError_t synth_skinny(const T* H, long hst, long hsn, const T* B, long bsn, long bsm, T* Y, long yst,
                        long ysm, int T_, int Mb, int N, hipStream_t s);

// ----- evc_aux.hip -----
// dst (dst_rows x dst_cols, row stride dst_ld) = src (src_rows x src_cols), zero outside src.
// src_trans: element (r,c) of the logical matrix is src[c*src_ld + r] instead of src[r*src_ld + c];
// dst_trans likewise for the destination.
// element-type conversion of a row-major matrix (the float32 surfaces ride the float64 fused path)
template <typename S, typename D>
hipError_t cvt2d(const S* src, long lds_, long rows, long cols, D* dst, long ldd, hipStream_t s);
template <typename T>
hipError_t copy2d(const T* src, long src_ld, int src_rows, int src_cols, int src_trans, T* dst,
                  long dst_ld, int dst_rows, int dst_cols, int dst_trans, hipStream_t s);
hipError_t utt_single(const UttState& u, int T_, hipStream_t s);
hipError_t utt_setup(const UttState& u, int n_utt, int T_, int Tp, int iters, hipStream_t s);
template <typename T>
hipError_t utt_sklearn_h0(const T* Xt, int ldx, int M, int N, const UttState& u, int n_utt,
                          hipStream_t s);
hipError_t utt_const_h0(const UttState& u, int n_utt, double v, hipStream_t s);
template <typename T>
hipError_t fill_h0(T* Ht, int ldh, int Tp, int N, int T_, const UttState& u, hipStream_t s);
template <typename T>
hipError_t frame_err2(const T* Xt, int ldx, const T* Vt, int ldv, int M, int T_, double* err2,
                      hipStream_t s);
// generalised KL variant: dictionary scaled by 1/colsum, ratio X / max(V, eps), per-frame 2*KL
template <typename T>
hipError_t kl_scale_dict(const T* At, int ld, int M, int rows, double eps, T* Akl, hipStream_t s);
template <typename T>
hipError_t kl_ratio(const T* Xt, int ldx, const T* Vt, int ldv, int M, long Tp, double eps, T* Rt, int ldr,
                    hipStream_t s);
template <typename T>
hipError_t frame_err_kl(const T* Xt, int ldx, const T* Vt, int ldv, int M, int T_, double eps, double* err2,
                        hipStream_t s);
// evaluate the stopping rule after check number `c` (c == 0: error at init)
hipError_t utt_check(const double* err2, const UttState& u, int n_utt, int c, int check_every,
                     int stop_rule, double tol, hipStream_t s);

// ----- evc_fused.hip -----
// Persistent fused FACTORED kernel for small dictionaries heights (M <= 32), float64.
struct FusedLayout {
    int NT;              // exemplar tiles of 16
    int TT, TTp;         // frame tiles of 16 (TTp: padded to a multiple of 4)
    int msteps;          // k-steps of 4 bins actually issued
    int mtiles;          // 1 (M <= 16) or 2
    int M;               // bins
    size_t a1, a2, xp, hp, vp;   // element counts of the packed arrays
};
constexpr int COOP_MAX_TILES = 256;       // frame tiles x cooperating workgroups never exceeds this (one per CU)
constexpr int ALL_MAX_WGS = 1024;         // resident workgroups of k_fused_all the exchange buffers are sized for
// k_fused_all with more than 8 members per group: the summed slices live behind the partials in coop_buf
constexpr int ALL_MAX_MEMBERS = 128;      // members per frame tile (N <= 65536)
constexpr int ALL_RS_STRIDE = 640;        // words per member in a group's partials: C slices x ceil(NE / C) <= NE + C - 1
constexpr long ALL_SLICE_OFFSET = 2L * ALL_MAX_WGS * ALL_RS_STRIDE;
constexpr long ALL_SLICE_ELEMS = 2L * (ALL_MAX_WGS / 2) * 512;    // [2][groups <= ALL_MAX_WGS / 2][512]
int fused_res_coop_factor(int NT, int TT, int n_cus);
// workgroups per frame tile the all-resident kernel (k_fused_all) uses for this problem; 0: it does not apply
int fused_all_members(int NT, int N, int eps_mode, int exact_div, int loss);
// workgroups per PAIR of frame tiles k_fused_xy (evc_fused_xy.hip) uses for this problem; 0: it does not apply
int fused_xy_members(int NT, int N, int eps_mode, int exact_div, int loss);
bool fused_res_supported(int N, int eps_mode, int exact_div);
struct FusedBuffers {
    double *A1p, *A2p, *Xp, *Hp, *Vp;
    double* coop_buf;      // exchange buffers of the cooperative launch (see k_fused_res)
    int* coop_cnt;         // [COOP_MAX_TILES] arrival counters, then one abort flag
    int coop_c;            // cooperating workgroups per frame tile chosen for this call (1 = off)
    int all_c;             // k_fused_all: workgroups per frame tile (0 = that kernel is not used)
    int xy_c;              // k_fused_xy: workgroups per pair of frame tiles (0 = that kernel is not used)
    double* rsum;          // [32] row sums of the dictionary (k_fused_all's in-kernel start)
    int init_const;        // 1: the first launch forms H = h0 and V = h0 rowsum(A) itself (no fill, no pre-pass)
    double* Hx;            // k_fused_all's last launch also writes the caller's H (NULL: off); ldhx, hx_frame_major
    long ldhx;
    int hx_frame_major;
    int n_cus;             // compute units of the device (sizes k_fused_all's persistent grid)
};
bool fused_supported(int M, int N, int T_, int dtype);
FusedLayout fused_layout(int M, int N, int T_);
// At[n][m] / Xt[t][m]: the zero-padded frames-as-rows workspace arrays
// (either destination may be NULL: only the other fragment order is written); At has n_rows rows: the exemplar
// slots of the (possibly further padded) tile grid beyond them are written as zeros
// ones_bin >= 0: that (padding) bin of the D-operand image holds 1.0 in every exemplar slot, so that a constant placed in
// the same bin of V's image is added to every denominator by the D product itself (k_fused_xy; the images of X and V
// hold zeros there otherwise, so every other kernel is unaffected)
hipError_t fused_pack_dict(const FusedLayout& f, double* A1p, double* A2p, const double* At, int ldA, int n_rows,
                           hipStream_t s, int ones_bin = -1);
hipError_t fused_pack_frames(const FusedLayout& f, double* Xp, const double* Xt, int ldx, hipStream_t s);
// rsum[m] = sum_n At[n][m] for m < M (fixed order), 0 for M <= m < 32
hipError_t fused_rowsum(const double* At, int ldA, int M, int N, double* rsum, hipStream_t s);
// packed activations <-> the caller's H (frame_major: H[t*ldh+n], else H[n*ldh+t]); per-utterance constant fill
hipError_t fused_import_h(const FusedLayout& f, double* Hp, const double* H, long ldh, int frame_major, int T_,
                          int N, hipStream_t s);
hipError_t fused_export_h(const FusedLayout& f, const double* Hp, double* H, long ldh, int frame_major, int T_,
                          int N, hipStream_t s);
hipError_t fused_fill_h(const FusedLayout& f, double* Hp, int N, int T_, const UttState& u, hipStream_t s);
// Y = B H from the packed activations (fB: layout for (Mb, N, T); B2p: B's V'-operand fragments)
hipError_t fused_synthesize(const FusedLayout& fB, const double* B2p, const double* Hp, double* Yp,
                            const UttState& u, int N, int T_, int Mb, double* Y, long ldy, int frame_major,
                            hipStream_t s);
// `iters` updates in one launch.  first: V is built from H by a pre-pass (else carried over in
// Vp from the previous launch); write_err: per-frame squared residuals of the final H -> err2.
hipError_t fused_iterate(const FusedLayout& f, const FusedBuffers& b, const UttState& u, int N, int T_,
                         int iters, int first, int write_err, double* err2, int eps_mode, double eps,
                         double l1, int c_req, int all_live_known, int loss, int exact_div, hipStream_t s);

// ----- evc_wide.hip -----
// Fused FACTORED kernel for wide spectra (32 < M <= 208 bins, float32): k_fused_wide, a task queue over
// (iteration, frame group, exemplar range).
struct WideLayout {
    int MT;              // bin tiles of 16 (template instance: >= ceil(M / 16))
    int W;               // wavefronts (= frame tiles of 16) per workgroup: 4 or 8
    int NB;              // exemplar blocks of 16
    int TT, G;           // frame tiles, frame groups of W tiles
    int c, rmode;        // exemplar ranges per group; 1: a reduce task sums the partial V' (c > 4)
    int tagged;          // 1: static schedule with reduce slices - the hand-offs carry their arrival flag in the data (evc_wide.hip)
    size_t aw, xw, hw, vpart, vsum;      // element counts (Pw has hw elements)
};
struct WideBuffers {
    float *Aw, *Xw, *Hw, *Pw, *Vpart, *Vsum;
    unsigned* ctl;       // [4 + 2 G]: ticket, abort flag, -, -, done[G], done_r[G]
    // several stop checks per launch (round 4): snapshots of the activations at the checks inside a launch and the
    // residuals the following iteration's tasks leave (NULL / 0: one check per launch)
    float* Hs;           // [snap_slots][hs_stride]
    size_t hs_stride;
    double* err2s;       // [snap_slots][err_stride]
    long err_stride;
    int snap_slots;
};
bool wide_supported(int M, int N, int T_, int dtype, int algo);
// c_req / w_req: 0 = automatic (tuning and tests: ranges per group, wavefronts per workgroup)
WideLayout wide_layout(int M, int N, int T_, int n_cus, int c_req, int w_req);
size_t wide_ctl_words(const WideLayout& f);
struct WideCaps { size_t aw, xw, hw, vpart, vsum, ctl; int c_cap; };
WideCaps wide_caps(int M, int N, int T_, int n_cus);
bool wide_fits(const WideLayout& f, const WideCaps& k);
// At1: the dictionary the numerator / denominator contraction uses (A, or A / colsum for KL), At2: A; exemplars as rows
hipError_t wide_pack_dict(const WideLayout& f, const float* At1, const float* At2, int ld, int n_rows, float* Aw,
                          hipStream_t s);
hipError_t wide_pack_x(const WideLayout& f, const float* Xt, int ld, int rows, float* Xw, hipStream_t s);
hipError_t wide_import_h(const WideLayout& f, float* Hw, const float* H, long ldh, int frame_major, int T_, int N,
                         hipStream_t s);
hipError_t wide_export_h(const WideLayout& f, const float* Hw, float* H, long ldh, int frame_major, int T_, int N,
                         const int* abort, hipStream_t s);
hipError_t wide_begin(const WideLayout& f, const WideBuffers& b, hipStream_t s);
// snap_every > 0: the iterations snap_first, snap_first + snap_every, ... < it_end - 1 are stop checks inside the launch
// (b.Hs / b.err2s receive the activations and the residuals of check k in slot k; wide_restore puts a slot back)
hipError_t wide_iterate(const WideLayout& f, const WideBuffers& b, const UttState& u, int N, int T_, int it_begin,
                        int it_end, int mode, double eps, double l1, int init_const, int n_cus, hipStream_t s,
                        int snap_every = 0, int snap_first = 0);
hipError_t wide_restore(const WideLayout& f, const WideBuffers& b, const UttState& u, int slot, int target_iter, int T_,
                        hipStream_t s);
hipError_t wide_err2(const WideLayout& f, const WideBuffers& b, const UttState& u, int N, int T_, int it, int kl,
                     double eps, double* err2, hipStream_t s);

// ----- evc_wide64.hip -----
// The same task queue for wide float64 spectra (144 < M <= 528, Frobenius): k_fused_wide64; a workgroup of four
// wavefronts owns 32 frames, the bins are split over its wavefronts.
struct Wide64Layout {
    int TPW;             // whole bin tiles of 16 per wavefront (template instance: 64 TPW + 16 >= M; the last tile is split)
    int NB;              // exemplar blocks of 16
    int TT, G;           // frame tiles, frame groups of 2 tiles
    int c, rmode;        // exemplar ranges per group; 1: a reduce task sums the partial V' (c > 4)
    size_t aw, xw, hw, vpart, vsum;      // element counts (Pw has hw elements)
};
struct Wide64Buffers {
    double *Aw, *Xw, *Hw, *Pw, *Vpart, *Vsum;
    unsigned* ctl;       // [4 + 2 G]: ticket, abort flag, -, -, done[G], done_r[G]
};
struct Wide64Caps { size_t aw, xw, hw, vpart, vsum, ctl; int c_cap; };
bool wide64_supported(int M, int N, int T_, int dtype, int algo, int loss);
// c_req: 0 = automatic; tpw_req: 0 = the narrowest instance that holds M bins (tests: a wider one)
Wide64Layout wide64_layout(int M, int N, int T_, int n_cus, int c_req, int tpw_req);
size_t wide_ctl_words(const Wide64Layout& f);
Wide64Caps wide64_caps(int M, int N, int T_, int n_cus);
bool wide_fits(const Wide64Layout& f, const Wide64Caps& k);
// (the second dictionary pointer, the KL-scaled rows of the float32 kernel, is unused: Frobenius only)
hipError_t wide_pack_dict(const Wide64Layout& f, const double* At, const double* unused, int ld, int n_rows, double* Aw,
                          hipStream_t s);
hipError_t wide_pack_x(const Wide64Layout& f, const double* Xt, int ld, int rows, double* Xw, hipStream_t s);
hipError_t wide_import_h(const Wide64Layout& f, double* Hw, const double* H, long ldh, int frame_major, int T_, int N,
                           hipStream_t s);
hipError_t wide_export_h(const Wide64Layout& f, const double* Hw, double* H, long ldh, int frame_major, int T_, int N,
                           const int* abort, hipStream_t s);
hipError_t wide_begin(const Wide64Layout& f, const Wide64Buffers& b, hipStream_t s);
hipError_t wide_iterate(const Wide64Layout& f, const Wide64Buffers& b, const UttState& u, int N, int T_, int it_begin,
                          int it_end, int mode, double eps, double l1, int init_const, int n_cus, hipStream_t s);
hipError_t wide_err2(const Wide64Layout& f, const Wide64Buffers& b, const UttState& u, int N, int T_, int it, int kl,
                     double eps, double* err2, hipStream_t s);

// ----- evc_gl.hip -----
size_t gl_workspace_bytes(long T_total, int n_utt, int F, int hop, int iters);
int stft_frames(long L, int hop, bool center, int F);
size_t stft_workspace_bytes(long L, int F, int hop, bool center);
hipError_t stft_run(const double* x, long L, int F, int hop, bool center, double* re, long ldre, double* im,
                    long ldim, void* ws, hipStream_t s);
hipError_t gl_run(const double* mag, long ldm, const int* frame_offsets, int n_utt, int F, int hop, int iters, double* x,
                  void* ws, double* rmse_host, hipStream_t s);

// ----- evc_dtw.hip -----
size_t dtw_workspace_bytes(const int* aoff, const int* boff, int n_pairs);
int dtw_max_frames();
hipError_t dtw_run(const double* A, long lda, const int* aoff, const double* B, long ldb, const int* boff,
                   int D, int n_pairs, int* path_a, int* path_b, int* path_len, double* total, void* ws,
                   hipStream_t s);
// aligned-frame gather (evc_dtw.hip): exclusive scan of the path lengths, then rows by the paths
hipError_t dtw_path_scan(const int* path_len, int n_pairs, int* row_start, hipStream_t s);
template <typename T>
hipError_t dtw_gather(const T* src, long ld_src, int elem_stride, const int* path, const int* path_len, const int* src_off,
                      const int* pair_off, const int* row_start, int n_pairs, int cols, int op, T* dst, long ld_dst,
                      hipStream_t s);

}  // namespace evc
