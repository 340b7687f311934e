// C ABI of libevc_hip.so (see include/evc.h): argument checking, workspace carving and the
// launch sequence of one activation solve.  No allocation, no global state, no exceptions.
#include "evc_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

using namespace evc;

namespace {

enum { ST_OK = 0, ST_BADARG = -1, ST_WORKSPACE = -2, ST_UNSUPPORTED = -3, ST_COOP_TIMEOUT = -4 };

struct Carver {
    char* base;
    size_t off;
    template <typename U> U* take(size_t count) {
        off = (off + 255) & ~size_t(255);
        U* p = base ? reinterpret_cast<U*>(base + off) : nullptr;
        off += count * sizeof(U);
        return p;
    }
};

struct Dims {
    int M, N, T_, n_utt, Mb;
    int Mk, Mj, Np, Tp;
};

Dims make_dims(int esize, int M, int N, int T_, int n_utt, int Mb = 0) {
    Dims d;
    d.M = M; d.N = N; d.T_ = T_; d.n_utt = n_utt; d.Mb = Mb;
    d.Mk = round_up(M, 16);
    d.Mj = round_up(M, 64);
    d.Np = round_up(N, 128);
    // frames are padded to the contraction kernels' frame tile: 64 where k_gemm2 is in charge (float32) and for short
    // float64 batches (<= 2048 frames: k_gemm_nt then runs 64-row blocks anyway, and one 688-frame utterance is 704
    // rows instead of 768), 128 otherwise (diagnostic builds: see use_gemm2 in evc_gemm.hip)
#if defined(EVC_DIAG_GEMM_V1)
    d.Tp = round_up(T_, 128);
#elif defined(EVC_DIAG_GEMM2_F64)
    d.Tp = round_up(T_, 64);
#else
    d.Tp = round_up(T_, (esize == 4 || T_ <= 2048) ? 64 : 128);
#endif
    return d;
}

constexpr int DICT_MAGIC = 0x45564344;      // "EVCD"

// What a solve needs of the dictionary, whatever the frames are.  Carved from the call's workspace and filled on every
// call, or - with a prepared dictionary (evc_dict_prepare) - carved from its image, filled once.
struct DictPlan {
    bool fused;          // float64 fused kernels (M <= 32): operand fragments, row sums
    bool packed_b;       // ... and B's fragments for the synthesis from packed tiles (1 <= Mb <= 32)
    bool wide;           // k_fused_wide's block images (float32, 32 < M <= 208)
    bool wide64;         // k_fused_wide64's block images (float64, 144 < M <= 528, Frobenius)
    bool kl;             // the dictionary divided by its column sums
    bool bc;             // a compact exemplars-as-rows copy of B (prepared images: the caller's B is not consulted)
};
template <typename T> struct DictArrays {
    T *At, *Am, *Akl;
    double *A1p, *A2p, *rsum;
    T* Bt;
    double *B1p, *B2p;
    T* Bc;
    float* Aw;
    double* Aw64;
};
template <typename T> DictPlan dict_plan(int M, int Mb, int N, int loss, bool prepared) {
    DictPlan p{};
    p.kl = loss == EVC_LOSS_KL;
    p.fused = sizeof(T) == 8 && fused_supported(M, N, 1, EVC_F64);
    p.packed_b = p.fused && Mb >= 1 && Mb <= 32;
    p.wide = sizeof(T) == 4 && wide_supported(M, N, 1, EVC_F32, EVC_ALGO_FACTORED);
    p.wide64 = sizeof(T) == 8 && wide64_supported(M, N, 1, EVC_F64, EVC_ALGO_FACTORED, loss);
    p.bc = prepared && Mb >= 1;
    return p;
}
template <typename T> DictArrays<T> take_dict(Carver& c, const Dims& d, const DictPlan& p) {
    DictArrays<T> a{};
    a.At = c.take<T>((size_t)d.Np * d.Mk);
    a.Am = c.take<T>((size_t)d.Mj * d.Np);
    a.Akl = c.take<T>((size_t)d.Np * d.Mk);
    if (p.fused) {
        const FusedLayout fl = fused_layout(d.M, d.N, 1);
        a.A1p = c.take<double>(fl.a1);
        a.A2p = c.take<double>(fl.a2);
        a.rsum = c.take<double>(32);
    }
    if (p.packed_b) {
        const FusedLayout flB = fused_layout(d.Mb, d.N, 1);
        a.Bt = c.take<T>((size_t)d.Np * 32);
        a.B1p = c.take<double>(flB.a1);
        a.B2p = c.take<double>(flB.a2);
    }
    if (p.wide) a.Aw = c.take<float>(wide_layout(d.M, d.N, 1, 256, 0, 0).aw);
    if (p.wide64) a.Aw64 = c.take<double>(wide64_layout(d.M, d.N, 1, 256, 0, 0).aw);
    if (p.bc) a.Bc = c.take<T>((size_t)d.N * d.Mb);
    return a;
}

template <typename T> struct Workspace {
    T *At, *Am, *Xt, *H0, *H1, *Pt, *G, *Vt;
    T *Akl, *Rt;         // KL: dictionary / column sums, and X / max(V, eps)
    T* Vsplit;           // split-K slabs of V = H Am^T when there are few frames
    size_t vsplit_elems;
    double* err2;
    UttState u;
    FusedLayout fl;
    FusedBuffers fb;
    // synthesis from the packed activations (fused path, Mb <= 32)
    FusedLayout flB;
    T* Bt;
    double *B1p, *B2p, *Yp;
    bool fused, packed_synth;
    size_t bytes;
};

int n_slots_for(int iters, int check_every) { return 1 + (check_every > 0 ? iters / check_every : 0); }

template <typename T>
Workspace<T> carve(void* base, const Dims& d, int algo, int n_slots, bool fused, const DictArrays<T>* ext = nullptr) {
    Workspace<T> w;
    Carver c{static_cast<char*>(base), 0};
    const bool gram = (algo == EVC_ALGO_GRAM || algo == EVC_ALGO_LITERAL);
    w.fused = fused;
    // the dictionary's arrays: from the prepared image, or from this workspace (then filled on every call)
    DictPlan plan = dict_plan<T>(d.M, d.Mb, d.N, EVC_LOSS_KL, false);
    plan.fused = fused;
    plan.packed_b = fused && d.Mb >= 1 && d.Mb <= 32;
    plan.wide = false;
    plan.wide64 = false;
    const DictArrays<T> da = ext ? *ext : take_dict<T>(c, d, plan);
    w.At = da.At;
    w.Am = da.Am;
    w.Akl = da.Akl;
    w.Xt = c.take<T>((size_t)d.Tp * d.Mk);
    w.Rt = fused ? nullptr : c.take<T>((size_t)d.Tp * d.Mk);
    w.packed_synth = fused && d.Mb >= 1 && d.Mb <= 32;
    // the fused path keeps the activations in the packed layout only; a frames-as-rows copy is
    // needed by the generic path, and by a synthesis that cannot run from the packed tiles
    const bool need_h0 = !fused || (d.Mb > 32);
    w.H0 = need_h0 ? c.take<T>((size_t)d.Tp * d.Np) : nullptr;
    w.Pt = fused ? nullptr : c.take<T>((size_t)d.Tp * d.Np);
    w.Vt = fused ? nullptr : c.take<T>((size_t)d.Tp * d.Mj);
    // slabs for the split-K form of V = H Am^T: its T x Mj output has few tiles (Mj is 64..576) against a long
    // contraction (N), so short batches need the split to fill the CUs
    const int vslabs = d.Tp <= 2048 ? 32 : (d.Tp <= 16384 ? 4 : (d.Tp <= 65536 ? 2 : 0));
    w.vsplit_elems = fused ? 0 : (size_t)vslabs * d.Tp * d.Mj;
    w.Vsplit = w.vsplit_elems ? c.take<T>(w.vsplit_elems) : nullptr;
    w.fl = FusedLayout{};
    w.fb = FusedBuffers{};
    if (fused) {
        w.fl = fused_layout(d.M, d.N, d.T_);
        w.fb.A1p = da.A1p;
        w.fb.A2p = da.A2p;
        w.fb.Xp = c.take<double>(w.fl.xp);
        w.fb.Hp = c.take<double>(w.fl.hp);
        w.fb.Vp = c.take<double>(w.fl.vp);
        static_assert(ALL_MAX_WGS >= COOP_MAX_TILES, "coop_buf is sized by k_fused_all's layout");
        w.fb.coop_buf = c.take<double>((size_t)(ALL_SLICE_OFFSET + ALL_SLICE_ELEMS));
        w.fb.coop_cnt = c.take<int>(COOP_MAX_TILES + 1);
        w.fb.rsum = da.rsum;
        w.fb.coop_c = 1;
    }
    w.flB = FusedLayout{};
    w.Bt = nullptr; w.B1p = w.B2p = w.Yp = nullptr;
    if (w.packed_synth) {
        w.flB = fused_layout(d.Mb, d.N, d.T_);
        // (a prepared dictionary without B: these come from the workspace and are filled per call)
        w.Bt = da.Bt ? da.Bt : c.take<T>((size_t)d.Np * 32);
        w.B1p = da.B1p ? da.B1p : c.take<double>(w.flB.a1);
        w.B2p = da.B2p ? da.B2p : c.take<double>(w.flB.a2);
        w.Yp = c.take<double>(w.flB.vp);
    }
    w.H1 = gram ? c.take<T>((size_t)d.Tp * d.Np) : nullptr;
    w.G = gram ? c.take<T>((size_t)d.Np * d.Np) : nullptr;
    w.err2 = c.take<double>(d.Tp);
    w.u.frame_utt = c.take<int>(d.Tp);
    w.u.offsets = c.take<int>(d.n_utt + 1);
    w.u.active = c.take<int>(d.n_utt + 1);
    w.u.n_iter = c.take<int>(d.n_utt);
    w.u.err_init = c.take<double>(d.n_utt);
    w.u.err_prev = c.take<double>(d.n_utt);
    w.u.h0 = c.take<double>(d.n_utt);
    w.u.trace = c.take<double>((size_t)d.n_utt * n_slots);
    w.u.n_slots = n_slots;
    w.bytes = (c.off + 255) & ~size_t(255);
    return w;
}

bool use_fused(int M, int N, int T_, int dtype, int algo) {
    return algo == EVC_ALGO_FACTORED && fused_supported(M, N, T_, dtype);
}

#define HIP_TRY(expr)                              \
    do {                                           \
        hipError_t e__ = (expr);                   \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

// The worst-case slot count is bounded by iters+1; evc_workspace_bytes has no iters argument,
// so the trace region is sized for MAX_SLOTS checks and evc_nmf_solve rejects more.
constexpr int MAX_SLOTS = 4097;

struct SynthArgs {          // optional Y = B H appended to a solve (evc_nmf_convert)
    const void* B; int ldb; void* Y; int ldy; int Mb;
    int b_rows;             // 1: B is exemplars-as-rows whatever the call's layout (a prepared dictionary's copy)
    int b_packed;           // 1: B's fragments are already in the dictionary image (nothing to import)
};

// hand back the per-utterance results
int copy_back(const UttState& u, int n_utt, int n_slots, int* n_iter_out, double* err_out, hipStream_t s) {
    if (n_iter_out || err_out) {
        if (n_iter_out)
            HIP_TRY(hipMemcpyAsync(n_iter_out, u.n_iter, sizeof(int) * n_utt, hipMemcpyDeviceToHost, s));
        if (err_out)
            HIP_TRY(hipMemcpyAsync(err_out, u.trace, sizeof(double) * (size_t)n_utt * n_slots,
                                   hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return ST_OK;
}

// Y = B H on frames-as-rows activations Hc (row stride ldc) and the caller's B / Y
template <typename T>
int synth_rows(const T* Hc, long ldc, const SynthArgs& y, int N, int T_, bool fm, hipStream_t s) {
    const T* B = static_cast<const T*>(y.B);
    T* Y = static_cast<T*>(y.Y);
    // B(n, mb) = B[n bsn + mb bsm]
    const long bsn = (fm || y.b_rows) ? y.ldb : 1, bsm = (fm || y.b_rows) ? 1 : y.ldb;
    if (fm)   // Y[t][mb] = sum_n Hc[t][n] B[n][mb]
        HIP_TRY(gemm_strided<T>(Hc, ldc, 1, B, bsm, bsn, Y, y.ldy, 1, T_, y.Mb, N, s));
    else      // Y[mb][t] = sum_n B[mb][n] Hc[t][n]
        HIP_TRY(gemm_strided<T>(B, bsm, bsn, Hc, ldc, 1, Y, y.ldy, 1, y.Mb, T_, N, s));
    return ST_OK;
}

// Bring the caller's dictionary into the arrays of `a` (every layout the plan names).  B may be NULL.
template <typename T>
int prepare_dict(const DictArrays<T>& a, const DictPlan& p, const T* A, int lda, const T* B, int ldb, const Dims& d,
                 bool fm, double eps, bool want_rowsum, hipStream_t s) {
    HIP_TRY(copy2d<T>(A, lda, d.N, d.M, fm ? 0 : 1, a.At, d.Mk, d.Np, d.Mk, 0, s));
    HIP_TRY(copy2d<T>(A, lda, d.M, d.N, fm ? 1 : 0, a.Am, d.Np, d.Mj, d.Np, 0, s));
    if (p.kl) HIP_TRY(kl_scale_dict<T>(a.At, d.Mk, d.M, d.Np, eps, a.Akl, s));
    return ST_OK;
}
int prepare_dict_fused(const DictArrays<double>& a, const DictPlan& p, const double* B, int ldb, const Dims& d, bool fm,
                       bool want_rowsum, hipStream_t s) {
    if (p.fused) {
        const FusedLayout fl = fused_layout(d.M, d.N, 1);
        if (p.kl) {      // D/P operand order from the scaled dictionary, V' operand order from A
            HIP_TRY(fused_pack_dict(fl, a.A1p, nullptr, a.Akl, d.Mk, d.Np, s));
            HIP_TRY(fused_pack_dict(fl, nullptr, a.A2p, a.At, d.Mk, d.Np, s));
        } else {
            HIP_TRY(fused_pack_dict(fl, a.A1p, a.A2p, a.At, d.Mk, d.Np, s, (d.M % 4) ? d.M : -1));
        }
        if (want_rowsum) HIP_TRY(fused_rowsum(a.At, d.Mk, d.M, d.N, a.rsum, s));
    }
    if (p.packed_b && B && a.Bt) {
        // Bt[n][mb] (zero padded to 32 bins) -> B's V'-operand fragments
        const FusedLayout flB = fused_layout(d.Mb, d.N, 1);
        HIP_TRY(copy2d<double>(B, ldb, d.N, d.Mb, fm ? 0 : 1, a.Bt, 32, d.Np, 32, 0, s));
        HIP_TRY(fused_pack_dict(flB, a.B1p, a.B2p, a.Bt, 32, d.Np, s));
    }
    return ST_OK;
}
// the arrays of a prepared dictionary (arithmetic type T); `skip`: bytes of staging in front of them
template <typename T> DictArrays<T> dict_arrays(const evc_dict* dk, size_t skip) {
    const Dims dd = make_dims((int)sizeof(T), dk->M, dk->N, 1, 1, dk->Mb);
    Carver c{static_cast<char*>(dk->mem), skip};
    return take_dict<T>(c, dd, dict_plan<T>(dk->M, dk->Mb, dk->N, dk->loss, true));
}
template <typename T> DictArrays<T> dict_arrays_at(void* mem, size_t skip, int M, int Mb, int N, int loss) {
    const Dims dd = make_dims((int)sizeof(T), M, N, 1, 1, Mb);
    Carver c{static_cast<char*>(mem), skip};
    return take_dict<T>(c, dd, dict_plan<T>(M, Mb, N, loss, true));
}
// staging of a float32 dictionary that rides the float64 fused kernels: A and B widened once
size_t dict_f64_staging(int M, int Mb, int N) {
    return ((((size_t)N * M + (size_t)N * Mb) * sizeof(double) + 512) + 255) & ~size_t(255);
}

template <typename T> size_t dict_image_bytes(int M, int Mb, int N, int loss, size_t skip) {
    const Dims dd = make_dims((int)sizeof(T), M, N, 1, 1, Mb);
    Carver c{nullptr, skip};
    take_dict<T>(c, dd, dict_plan<T>(M, Mb, N, loss, true));
    return (c.off + 255) & ~size_t(255);
}

template <typename T>
int dict_prepare_typed(const T* A, int lda, const T* B, int ldb, int M, int Mb, int N, bool fm, int loss, double eps,
                              void* mem, size_t skip, hipStream_t s) {
    const Dims dd = make_dims((int)sizeof(T), M, N, 1, 1, Mb);
    const DictPlan p = dict_plan<T>(M, Mb, N, loss, true);
    Carver c{static_cast<char*>(mem), skip};
    const DictArrays<T> a = take_dict<T>(c, dd, p);
    int st = prepare_dict<T>(a, p, A, lda, B, ldb, dd, fm, eps, true, s);
    if (st) return st;
    if (B && a.Bc) HIP_TRY(copy2d<T>(B, ldb, N, Mb, fm ? 0 : 1, a.Bc, Mb, N, Mb, 0, s));
    if (p.wide) {
        const WideLayout fl = wide_layout(M, N, 1, 256, 0, 0);
        HIP_TRY(wide_pack_dict(fl, p.kl ? reinterpret_cast<const float*>(a.Akl) : reinterpret_cast<const float*>(a.At),
                               reinterpret_cast<const float*>(a.At), dd.Mk, dd.Np, a.Aw, s));
    }
    if (p.wide64) {
        const Wide64Layout fl = wide64_layout(M, N, 1, 256, 0, 0);
        HIP_TRY(wide_pack_dict(fl, reinterpret_cast<const double*>(a.At), nullptr, dd.Mk, dd.Np, a.Aw64, s));
    }
    return ST_OK;
}

// The fused persistent path (float64, M <= 32): one launch per `check_every` iterations (or a
// single launch when no residual is requested); V is carried between launches.
template <typename T>
int solve_fused(const Workspace<T>& w, const Dims& d, const evc_solve_opts& o, int n_utt, hipStream_t s,
                int* coop_used, T* H_out, int ldh, int* exported, evc_solve_info* inf) {
    return ST_UNSUPPORTED;
}
// one host round trip: did a cooperative launch of this call give up waiting for a peer workgroup?
template <typename T>
int coop_timed_out(const Workspace<T>& w, hipStream_t s, int* aborted) {
    *aborted = 0;
    HIP_TRY(hipMemcpyAsync(aborted, w.fb.coop_cnt + COOP_MAX_TILES, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return ST_OK;
}
template <>
int solve_fused<double>(const Workspace<double>& w, const Dims& d, const evc_solve_opts& o, int n_utt,
                        hipStream_t s, int* coop_used, double* H_out, int ldh, int* exported, evc_solve_info* inf) {
    *exported = 0;
    const int c_override = (o.reserved >> 8) & 0xff;     // 0 = automatic, 1 / 2 = general kernel
    // pymf's stop rule compares successive errors against 2.2e-16: it only fires at the reference's
    // iteration if the update reaches the same floating-point fixed point, i.e. with correctly rounded
    // quotients (reserved bit 1 asks for them explicitly)
    const int exact_div = (o.stop_rule == EVC_STOP_PYMF || (o.reserved & 2)) ? 1 : 0;
    if (!o.dict) {     // (a prepared dictionary holds the fragments already)
        DictArrays<double> da{};
        da.At = w.At; da.Akl = w.Akl; da.A1p = w.fb.A1p; da.A2p = w.fb.A2p; da.rsum = w.fb.rsum;
        DictPlan pl{};
        pl.fused = true; pl.kl = o.loss == EVC_LOSS_KL;
        int st = prepare_dict_fused(da, pl, nullptr, 0, d, true, false, s);
        if (st) return st;
    }
    HIP_TRY(fused_pack_frames(w.fl, w.fb.Xp, w.Xt, d.Mk, s));
    // few frame tiles (one or two utterances): several workgroups share a tile and split the exemplars
    // (k_fused_res COOP); reserved bit 2 switches it off
    FusedBuffers fb = w.fb;
    fb.coop_c = 1;
    fb.all_c = 0;
    fb.xy_c = 0;
    int dev = 0, cus = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    fb.n_cus = cus;
    // every activation and numerator tile register-resident, NT / 32 workgroups per frame tile (k_fused_all);
    // reserved bit 4 switches that kernel off, bit 2 every form of inter-workgroup exchange
    if (c_override == 0 && !(o.reserved & 16)) {
        const int c = fused_all_members(w.fl.NT, d.N, o.eps_mode, exact_div, o.loss);
        if (c == 1 || (c > 1 && c <= cus && !(o.reserved & 4))) fb.all_c = c;
        // two frame tiles per member, exchange inside the sweeps (k_fused_xy, round 4): measured slower than k_fused_all
        // (profiles/r04_xy_notes.md), so only on request - reserved bit 5, EVC_FLAG_PAIR_TILES
        const int cx = fused_xy_members(w.fl.NT, d.N, o.eps_mode, exact_div, o.loss);
        if (cx >= 2 && cx <= 2 * cus && !(o.reserved & 4) && (o.reserved & 32)) { fb.xy_c = cx; fb.all_c = 0; }
    }
    if (!fb.all_c && !fb.xy_c && !(o.reserved & 4) && c_override == 0 && fused_res_supported(d.N, o.eps_mode, exact_div))
        fb.coop_c = fused_res_coop_factor(w.fl.NT, w.fl.TT, cus);
    // start values: caller-given ones were imported by the caller of this function; constants are either written
    // into the packed tiles here, or - first launch on k_fused_all, no residual wanted at init - formed by that kernel
    fb.init_const = 0;
    if (o.init_mode != EVC_INIT_GIVEN) {
        if ((fb.all_c >= 1 || fb.xy_c >= 2) && o.iters > 0 && !(o.check_every > 0 && o.stop_rule == EVC_STOP_SKLEARN)) {
            if (!o.dict) HIP_TRY(fused_rowsum(w.At, d.Mk, d.M, d.N, fb.rsum, s));
            fb.init_const = 1;
        } else {
            HIP_TRY(fused_fill_h(w.fl, fb.Hp, d.N, d.T_, w.u, s));
        }
    }
    const bool exchanges = fb.coop_c > 1 || fb.all_c > 1 || fb.xy_c > 1;
    inf->kernel = fb.xy_c >= 2 ? EVC_KERNEL_FUSED_XY : fb.all_c >= 1 ? EVC_KERNEL_FUSED_ALL
                  : ((c_override == 0 && fused_res_supported(d.N, o.eps_mode, exact_div)) ? EVC_KERNEL_FUSED_RES
                                                                                          : EVC_KERNEL_FUSED_MU);
    inf->members = fb.xy_c >= 2 ? fb.xy_c : fb.all_c >= 1 ? fb.all_c : fb.coop_c;
    inf->exchange = exchanges ? 1 : 0;
    int* coop_abort = fb.coop_cnt + COOP_MAX_TILES;
    // tests only (evc_solve_opts.test_abort_at, 0 in production): k > 0 raises the abort flag in front of the k-th
    // launch of the iteration loop (-1: the call starts with it raised), as a timed-out wait would; the call then
    // takes the caller's retry path after a partially completed solve
    const int fake_at = o.test_abort_at < 0 ? 0 : (o.test_abort_at > 0 ? o.test_abort_at : -1);
    if (exchanges) HIP_TRY(hipMemsetAsync(coop_abort, fake_at == 0 ? 1 : 0, sizeof(int), s));
    int first = 1, launch_no = 0;
    if (o.check_every > 0 && o.stop_rule == EVC_STOP_SKLEARN) {   // error_at_init
        HIP_TRY(fused_iterate(w.fl, fb, w.u, d.N, d.T_, 0, 1, 1, w.err2, o.eps_mode, o.eps, o.l1,
                              c_override, 1, o.loss, exact_div, s));
        HIP_TRY(utt_check(w.err2, w.u, n_utt, 0, o.check_every, o.stop_rule, o.tol, s));
        first = 0;
    }
    if (o.ev_loop_start) HIP_TRY(hipEventRecord((hipEvent_t)o.ev_loop_start, s));
    int done = 0;
    while (done < o.iters) {
        int n = o.iters - done;
        bool check = false;
        if (o.check_every > 0 && n >= o.check_every) { n = o.check_every; check = true; }
        // the last launch of a solve in which nothing can stop writes the caller's H itself (k_fused_all)
        if (H_out && (fb.all_c >= 1 || fb.xy_c >= 2) && c_override == 0 && o.stop_rule == EVC_STOP_NONE && done + n == o.iters) {
            fb.Hx = H_out; fb.ldhx = ldh; fb.hx_frame_major = o.layout == EVC_FRAME_MAJOR ? 1 : 0;
            *exported = 1;
        }
        if (exchanges && fake_at > 0 && launch_no == fake_at) HIP_TRY(hipMemsetAsync(coop_abort, 1, sizeof(int), s));
        ++launch_no;
        ++inf->launches;
        HIP_TRY(fused_iterate(w.fl, fb, w.u, d.N, d.T_, n, first, check ? 1 : 0, w.err2, o.eps_mode,
                              o.eps, o.l1, c_override, o.stop_rule == EVC_STOP_NONE ? 1 : 0, o.loss, exact_div, s));
        first = 0;
        done += n;
        if (check)
            HIP_TRY(utt_check(w.err2, w.u, n_utt, done / o.check_every, o.check_every, o.stop_rule,
                              o.tol, s));
    }
    if (o.ev_loop_stop) HIP_TRY(hipEventRecord((hipEvent_t)o.ev_loop_stop, s));
    *coop_used = exchanges ? 1 : 0;         // the caller checks the abort flag (one host round trip)
    return ST_OK;
}

// tail of the fused path: H out of the packed tiles, Y from them
template <typename T>
int finish_fused(const Workspace<T>& w, const Dims& d, const evc_solve_opts& o, T* H, int ldh,
                 const SynthArgs* y, hipStream_t s) {
    return ST_UNSUPPORTED;
}
template <>
int finish_fused<double>(const Workspace<double>& w, const Dims& d, const evc_solve_opts& o, double* H, int ldh,
                         const SynthArgs* y, hipStream_t s) {
    const bool fm = (o.layout == EVC_FRAME_MAJOR);
    if (H) HIP_TRY(fused_export_h(w.fl, w.fb.Hp, H, ldh, fm ? 1 : 0, d.T_, d.N, s));
    if (!y) return ST_OK;
    if (w.packed_synth) {
        // Bt[n][mb] (zero padded to 32 bins) -> B's V'-operand fragments -> pre-pass -> Y
        if (!y->b_packed) {
            HIP_TRY(copy2d<double>(static_cast<const double*>(y->B), y->ldb, d.N, y->Mb, (fm || y->b_rows) ? 0 : 1, w.Bt, 32,
                                   d.Np, 32, 0, s));
            HIP_TRY(fused_pack_dict(w.flB, w.B1p, w.B2p, w.Bt, 32, d.Np, s));
        }
        HIP_TRY(fused_synthesize(w.flB, w.B2p, w.fb.Hp, w.Yp, w.u, d.N, d.T_, y->Mb,
                                 static_cast<double*>(y->Y), y->ldy, fm ? 1 : 0, s));
        return ST_OK;
    }
    HIP_TRY(fused_export_h(w.fl, w.fb.Hp, w.H0, d.Np, 1, d.T_, d.N, s));
    return synth_rows<double>(w.H0, d.Np, *y, d.N, d.T_, fm, s);
}

// ------------------------------------------------------------------------------------------------------
// The wide fused path (float32, 32 < M <= 208, FACTORED): k_fused_wide (evc_wide.hip).  One launch per
// `check_every` iterations (a single launch when no residual is wanted); iteration 0 forms P and V = A H0.
// ------------------------------------------------------------------------------------------------------
template <typename T> struct WideKind;
template <> struct WideKind<float> {
    typedef WideLayout Layout;
    typedef WideBuffers Buffers;
    typedef WideCaps Caps;
    static constexpr int kernel = EVC_KERNEL_FUSED_WIDE;
    static Layout layout(int M, int N, int T_, int n_cus, int c_req, int w_req) {
        return wide_layout(M, N, T_, n_cus, c_req, w_req);
    }
    static Caps caps(int M, int N, int T_, int n_cus) { return wide_caps(M, N, T_, n_cus); }
    static float* image(const DictArrays<float>& a) { return a.Aw; }
};
template <> struct WideKind<double> {
    typedef Wide64Layout Layout;
    typedef Wide64Buffers Buffers;
    typedef Wide64Caps Caps;
    static constexpr int kernel = EVC_KERNEL_FUSED_WIDE64;
    static Layout layout(int M, int N, int T_, int n_cus, int c_req, int w_req) {
        return wide64_layout(M, N, T_, n_cus, c_req, w_req);
    }
    static Caps caps(int M, int N, int T_, int n_cus) { return wide64_caps(M, N, T_, n_cus); }
    static double* image(const DictArrays<double>& a) { return a.Aw64; }
};
template <typename T> struct WideWs {
    T *At, *Akl, *Xt, *H0;
    typename WideKind<T>::Buffers fb;
    typename WideKind<T>::Caps caps;
    double* err2;
    UttState u;
    size_t bytes;
};
int device_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    return cus;
}
template <typename T>
WideWs<T> carve_wide(void* base, const Dims& d, int n_slots, int n_cus, bool with_synth, bool kl) {
    WideWs<T> w{};
    Carver c{static_cast<char*>(base), 0};
    w.caps = WideKind<T>::caps(d.M, d.N, d.T_, n_cus);
    w.At = c.take<T>((size_t)d.Np * d.Mk);
    w.Akl = kl ? c.take<T>((size_t)d.Np * d.Mk) : nullptr;
    w.Xt = c.take<T>((size_t)d.Tp * d.Mk);
    w.H0 = with_synth ? c.take<T>((size_t)d.Tp * d.Np) : nullptr;
    w.fb.Aw = c.take<T>(w.caps.aw);
    w.fb.Xw = c.take<T>(w.caps.xw);
    w.fb.Hw = c.take<T>(w.caps.hw);
    w.fb.Pw = c.take<T>(w.caps.hw);
    w.fb.Vpart = c.take<T>(w.caps.vpart);
    w.fb.Vsum = c.take<T>(w.caps.vsum);
    w.fb.ctl = c.take<unsigned>(w.caps.ctl);
    w.err2 = c.take<double>(d.Tp);
    w.u.frame_utt = c.take<int>(d.Tp);
    w.u.offsets = c.take<int>(d.n_utt + 1);
    w.u.active = c.take<int>(d.n_utt + 1);
    w.u.n_iter = c.take<int>(d.n_utt);
    w.u.err_init = c.take<double>(d.n_utt);
    w.u.err_prev = c.take<double>(d.n_utt);
    w.u.h0 = c.take<double>(d.n_utt);
    w.u.trace = c.take<double>((size_t)d.n_utt * n_slots);
    w.u.n_slots = n_slots;
    if constexpr (sizeof(T) == 4) {
        // k_fused_wide, several stop checks per launch: up to 4 snapshots of the activations (5 checks per launch), as
        // many as fit in 2 GiB (a 16-utterance STFT batch: 180 MB each)
        const size_t one = w.caps.hw * sizeof(float);
        int slots = one ? (int)(((size_t)2 << 30) / one) : 0;
        slots = slots > 4 ? 4 : slots;
        w.fb.snap_slots = slots;
        w.fb.hs_stride = w.caps.hw;
        w.fb.err_stride = d.Tp;
        w.fb.Hs = slots ? c.take<float>((size_t)slots * w.caps.hw) : nullptr;
        w.fb.err2s = slots ? c.take<double>((size_t)slots * d.Tp) : nullptr;
    }
    w.bytes = (c.off + 255) & ~size_t(255);
    return w;
}
// Routing between the fused task-queue kernels and the two contractions, from the measured table
// profiles/r04_routing_table.md (tools/tune_routing.py on the final build: whole calls, K = 20 and K = 80, N in {1024, 4096,
// 16384}, 1 .. 64 utterances of 688 frames; fractions of the matrix peak, fused / two contractions at K = 80):
//  * float32 (k_fused_wide, 32 < M <= 208): ahead from one utterance on at every N and M measured (M = 201: 0.52 / 0.42 at
//    N = 16384, 0.34 / 0.29 at 4096, 0.16 / 0.13 at 1024), and by more and more towards 64 utterances (M = 201, N = 4096:
//    0.71 / 0.53).
//  * float64 (k_fused_wide64, 144 < M <= 528): small dictionaries (N = 1024) from one utterance on (M = 513: 0.26 / 0.22,
//    three utterances 0.47 / 0.35) up to ~32 (64: 0.64 / 0.69).  N >= 2048: two utterances lose (M = 513, N = 4096:
//    0.50 / 0.61 - five ranges with reduce slices against the contractions' best case), three win (0.60 / 0.53: 260 tasks
//    through the queue) and so do four to ~24 (16: 0.66 / 0.62; 32: 0.67 / 0.67; 64: 0.67 / 0.71).  At M = 257 (4 bin tiles
//    per wavefront: fewer MFMAs per block against the same fixed work) the window closes earlier: 2048 <= N < 8192 up to
//    ~14 utterances (16: 0.55 / 0.55), from N = 8192 on the two contractions win or tie throughout (six utterances at
//    N = 16384: 0.55 / 0.59).  176 < M <= 208 (3 bin tiles per wavefront, 0.49 - 0.52): N < 8192 like the others up to ~45
//    utterances (32: 0.51 / 0.49, 64: 0.51 / 0.58), from 8192 on only where the contractions' tile counts fall badly
//    (12 - 32 utterances: 0.51 - 0.52 / 0.40 - 0.47); M <= 176 pads more than a thirteenth of the tile slots and stays out
//    (M = 160: 0.42).
// The tuning bits (ranges, wavefronts / bin tiles) force the fused kernel at any size.
bool use_wide(int M, int N, int T_, int dtype, int algo, int loss, int reserved) {
    if (reserved & EVC_FLAG_NO_FUSED) return false;
    // the task queues hand partial sums from workgroup to workgroup inside a launch and the call reads one word back at
    // the end (a wait that ran out): exactly what EVC_FLAG_NO_EXCHANGE rules out
    if (reserved & EVC_FLAG_NO_EXCHANGE) return false;
    const bool forced = ((reserved >> 8) & 0xff) != 0 || ((reserved >> 16) & 0xf) != 0;
    const int tiles = (T_ + 15) / 16;
    if (dtype == EVC_F64) {
        if (!wide64_supported(M, N, T_, dtype, algo, loss)) return false;
        if (forced) return true;
        const int lo64 = N < 2048 ? 43 : 100;
        if (M <= 208) {
            if (M <= 176) return false;
            return tiles >= (N < 8192 ? lo64 : 500) && tiles <= 2000;
        }
        const int hi = N < 2048 ? 1400 : (M >= 400 ? 1000 : (N < 8192 ? 600 : 0));
        return tiles >= lo64 && tiles <= hi;
    }
    if (!wide_supported(M, N, T_, dtype, algo)) return false;
    if (forced) return true;
    // (with tagged hand-offs on the static schedule - the last change of round 4 - the fused kernel is ahead from one
    // utterance of 688 frames on at every (M, N) measured: M = 201, N = 4096: 0.34 / 0.29 at one, 0.49 / 0.34 at two;
    // N = 1024: 0.16 / 0.13; M = 64, N = 4096: 0.21 / 0.15.  Shorter inputs were not measured and stay where they were.)
    return tiles >= 43;
}

template <typename T>
int solve_wide(const T* A, int lda, const T* X, int ldx, T* H, int ldh, int M, int N, int T_,
               const int* utt_offsets, int n_utt, const evc_solve_opts& o, void* ws, size_t ws_bytes,
               int* n_iter_out, double* err_out, const SynthArgs* y, hipStream_t s, evc_solve_info* inf) {
    typedef WideKind<T> K;
    const Dims d = make_dims((int)sizeof(T), M, N, T_, n_utt, y ? y->Mb : 0);
    const int n_slots = n_slots_for(o.iters, o.check_every);
    if (n_slots > MAX_SLOTS) return ST_UNSUPPORTED;
    const int n_cus = device_cus();
    const bool kl = o.loss == EVC_LOSS_KL, fm = o.layout == EVC_FRAME_MAJOR;
    WideWs<T> w = carve_wide<T>(ws, d, MAX_SLOTS, n_cus, true, sizeof(T) == 4);
    if (w.bytes > ws_bytes) return ST_WORKSPACE;
    w.u.n_slots = n_slots;
    // tuning / tests: reserved bits 8..15 = exemplar ranges per frame group, bits 16..19 = wavefronts per workgroup
    const typename K::Layout fl = K::layout(M, N, T_, n_cus, (o.reserved >> 8) & 0xff, (o.reserved >> 16) & 0xf);
    // (a prepared dictionary holds the block images of the DEFAULT layout: a tuning override of the wavefront / tile
    // count does not fit it - unsupported, not a workspace problem)
    if (!wide_fits(fl, w.caps)) return (o.dict && ((o.reserved >> 16) & 0xf)) ? ST_UNSUPPORTED : ST_WORKSPACE;

    if (utt_offsets)
        HIP_TRY(hipMemcpyAsync(w.u.offsets, utt_offsets, sizeof(int) * (n_utt + 1), hipMemcpyHostToDevice, s));
    else
        HIP_TRY(utt_single(w.u, T_, s));
    HIP_TRY(utt_setup(w.u, n_utt, T_, d.Tp, o.iters, s));
    SynthArgs ydict;
    if (o.dict) {                 // the block images (and B) come from the prepared dictionary
        const DictArrays<T> ext = dict_arrays<T>(o.dict, 0);
        w.fb.Aw = K::image(ext);
        if (y && o.dict->Mb > 0) {
            ydict = *y;
            ydict.B = ext.Bc; ydict.ldb = o.dict->Mb; ydict.b_rows = 1;
            y = &ydict;
        }
        inf->prepared = 1;
    } else {
        HIP_TRY(copy2d<T>(A, lda, N, M, fm ? 0 : 1, w.At, d.Mk, d.Np, d.Mk, 0, s));
        if (kl) HIP_TRY(kl_scale_dict<T>(w.At, d.Mk, M, d.Np, o.eps, w.Akl, s));
        HIP_TRY(wide_pack_dict(fl, kl ? w.Akl : w.At, w.At, d.Mk, d.Np, w.fb.Aw, s));
    }
    HIP_TRY(copy2d<T>(X, ldx, T_, M, fm ? 0 : 1, w.Xt, d.Mk, d.Tp, d.Mk, 0, s));
    if (o.init_mode == EVC_INIT_SKLEARN) HIP_TRY(utt_sklearn_h0<T>(w.Xt, d.Mk, M, N, w.u, n_utt, s));
    else if (o.init_mode == EVC_INIT_CONST) HIP_TRY(utt_const_h0(w.u, n_utt, o.init_value, s));
    HIP_TRY(wide_pack_x(fl, w.Xt, d.Mk, d.Tp, w.fb.Xw, s));
    const int init_const = o.init_mode == EVC_INIT_GIVEN ? 0 : 1;
    if (!init_const) HIP_TRY(wide_import_h(fl, w.fb.Hw, H, ldh, fm ? 1 : 0, T_, N, s));
    HIP_TRY(wide_begin(fl, w.fb, s));
    // (float64: pymf's stop rule compares successive errors against 2.2e-16 and reserved bit 1 asks for it explicitly -
    // correctly rounded quotients, as in solve_fused)
    const bool exact = sizeof(T) == 8 && (o.stop_rule == EVC_STOP_PYMF || (o.reserved & 2));
    const int mode = (kl ? 100 : o.eps_mode) | (exact ? 0x1000 : 0);

    inf->kernel = K::kernel;
    inf->members = fl.c;
    inf->exchange = 1;               // tasks wait for tasks of other workgroups, whatever the number of ranges
    // tests only (evc_solve_opts.test_abort_at, 0 in production): k > 0 raises the abort flag in front of the k-th
    // launch of the iteration loop (-1: the call starts with it raised), as a wait that ran out would
    const int fake_at = o.test_abort_at < 0 ? 0 : (o.test_abort_at > 0 ? o.test_abort_at : -1);
    int* abort_w = reinterpret_cast<int*>(w.fb.ctl + 1);
    if (fake_at == 0) HIP_TRY(hipMemsetAsync(abort_w, 1, sizeof(int), s));
    int launch_no = 0;
    int next_it = 0;                 // first iteration not yet run (0 = the pass that forms P and V = A H0)
    auto run_to = [&](int it_end, int snap_every = 0, int snap_first = 0) -> int {      // iterations [next_it, it_end)
        if (it_end <= next_it) return 0;
        if constexpr (sizeof(T) == 4)
            HIP_TRY(wide_iterate(fl, w.fb, w.u, N, T_, next_it, it_end, mode, o.eps, o.l1, init_const, n_cus, s, snap_every,
                                 snap_first));
        else
            HIP_TRY(wide_iterate(fl, w.fb, w.u, N, T_, next_it, it_end, mode, o.eps, o.l1, init_const, n_cus, s));
        ++inf->launches;
        next_it = it_end;
        return 0;
    };
    // k_fused_wide: up to 1 + snap_slots stop checks per launch.  A launch boundary costs the task queue about one task
    // time (the tail of one launch and the head of the next do not overlap) plus the residual kernel: with a check every
    // 10 iterations that was 28 % of the STFT flow's default call at 16 utterances (profiles/r03_default_call_wide.jsonl).
    // The checks inside a launch are evaluated after it, in order; an utterance that stopped at one of them gets the
    // snapshot of that check back (k_wide_restore) - results are those of one launch per check, bit for bit.
    int checks_per_launch = 1;
    if constexpr (sizeof(T) == 4)
        if (o.check_every > 0 && !kl) checks_per_launch = 1 + w.fb.snap_slots;
    auto check = [&](int c) -> int {
        HIP_TRY(wide_err2(fl, w.fb, w.u, N, T_, next_it - 1, kl ? 1 : 0, o.eps, w.err2, s));
        HIP_TRY(utt_check(w.err2, w.u, n_utt, c, o.check_every, o.stop_rule, o.tol, s));
        return 0;
    };
    if (o.check_every > 0 && o.stop_rule == EVC_STOP_SKLEARN) {   // error_at_init, sklearn _nmf.py:827
        int st = run_to(1);
        if (st) return st;
        st = check(0);
        if (st) return st;
    }
    if (o.ev_loop_start) HIP_TRY(hipEventRecord((hipEvent_t)o.ev_loop_start, s));
    int done = 0;
    while (done < o.iters) {
        int n = o.iters - done;
        bool chk = false;
        int L = 1;
        if (o.check_every > 0 && n >= o.check_every) {
            // the checks of a launch are speculation: iterations behind a check at which every utterance stopped are
            // thrown away.  The number grows with the iterations already done (1, 1, 2, 3, 4 ...: launches end at 10, 20,
            // 40, 70, 110, 150 of the default call), so at most a third of a solve's iterations are wasted
            L = 1 + done / (2 * o.check_every);
            if (L > checks_per_launch) L = checks_per_launch;
            if (L > n / o.check_every) L = n / o.check_every;
            n = L * o.check_every;
            chk = true;
        }
        if (fake_at > 0 && launch_no == fake_at) HIP_TRY(hipMemsetAsync(abort_w, 1, sizeof(int), s));
        ++launch_no;
        int st = L > 1 ? run_to(done + n + 1, o.check_every, done + o.check_every) : run_to(done + n + 1);
        if (st) return st;
        if constexpr (sizeof(T) == 4) {
            for (int k = 0; k + 1 < L; ++k) {      // the checks inside the launch, oldest first
                const int cno = done / o.check_every + 1 + k;
                HIP_TRY(utt_check(w.fb.err2s + (size_t)k * w.fb.err_stride, w.u, n_utt, cno, o.check_every, o.stop_rule,
                                  o.tol, s));
                if (o.stop_rule != EVC_STOP_NONE) HIP_TRY(wide_restore(fl, w.fb, w.u, k, cno * o.check_every, T_, s));
            }
        }
        done += n;
        if (chk) {
            st = check(done / o.check_every);
            if (st) return st;
        }
    }
    {
        int st = run_to(1);           // iters == 0: the start values still have to exist
        if (st) return st;
    }
    if (o.ev_loop_stop) HIP_TRY(hipEventRecord((hipEvent_t)o.ev_loop_stop, s));
    // A bounded wait that ran out (a wedged or oversubscribed device) voids the solve: the flag is read back before
    // anything reaches the caller's H or Y - one host round trip per solve, as for the other exchanging kernels
    // (include/evc.h, Host synchronisation) - and the caller of this function redoes the solve on the two contractions
    // from the untouched inputs.  (Round 3 exported NaN under status 0.)
    {
        int aborted = 0;
        HIP_TRY(hipMemcpyAsync(&aborted, abort_w, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (aborted) return ST_COOP_TIMEOUT;
    }
    const int* abort = abort_w;
    if (H) HIP_TRY(wide_export_h(fl, w.fb.Hw, H, ldh, fm ? 1 : 0, T_, N, abort, s));
    if (y) {
        HIP_TRY(wide_export_h(fl, w.fb.Hw, w.H0, d.Np, 1, T_, N, abort, s));
        int st = synth_rows<T>(w.H0, d.Np, *y, N, T_, fm, s);
        if (st) return st;
    }
    return copy_back(w.u, n_utt, n_slots, n_iter_out, err_out, s);
}

template <typename T> int gemm_kernel_id() {
#if defined(EVC_DIAG_GEMM_V1)
    return EVC_KERNEL_GEMM_NT;
#elif defined(EVC_DIAG_GEMM2_F64)
    return EVC_KERNEL_GEMM2;
#else
    return sizeof(T) == 4 ? EVC_KERNEL_GEMM2 : EVC_KERNEL_GEMM_NT;
#endif
}

template <typename T>
int solve_typed(const void* A_, int lda, const void* X_, int ldx, void* H_, int ldh, int M, int N,
                int T_, const int* utt_offsets, int n_utt, const evc_solve_opts& o, void* ws,
                size_t ws_bytes, int* n_iter_out, double* err_out, const SynthArgs* y, hipStream_t s,
                evc_solve_info* inf) {
    const T* A = static_cast<const T*>(A_);
    const T* X = static_cast<const T*>(X_);
    T* H = static_cast<T*>(H_);
    const Dims d = make_dims((int)sizeof(T), M, N, T_, n_utt, y ? y->Mb : 0);
    int algo = o.algo == EVC_ALGO_AUTO ? EVC_ALGO_FACTORED : o.algo;
    const int n_slots = n_slots_for(o.iters, o.check_every);
    if (n_slots > MAX_SLOTS) return ST_UNSUPPORTED;
    const bool fused = use_fused(M, N, T_, sizeof(T) == 8 ? EVC_F64 : EVC_F32, algo) && !(o.reserved & 1);
    // dictionary arrays: from the prepared image (float32 callers riding the float64 kernels: behind its staging)
    DictArrays<T> ext{};
    if (o.dict) ext = dict_arrays<T>(o.dict, (sizeof(T) == 8 && o.dict->dtype == EVC_F32) ? dict_f64_staging(M, o.dict->Mb, N) : 0);
    Workspace<T> w = carve<T>(ws, d, algo, MAX_SLOTS, fused, o.dict ? &ext : nullptr);
    if (w.bytes > ws_bytes) return ST_WORKSPACE;
    inf->prepared = o.dict ? 1 : 0;
    SynthArgs ydict;              // synthesis from the prepared copy of B
    if (o.dict && y && o.dict->Mb > 0) {
        ydict = *y;
        ydict.B = ext.Bc; ydict.ldb = o.dict->Mb; ydict.b_rows = 1; ydict.b_packed = ext.B2p ? 1 : 0;
        y = &ydict;
    }
    w.u.n_slots = n_slots;
    const bool fm = (o.layout == EVC_FRAME_MAJOR);

    // ---- utterances ----
    if (utt_offsets)
        HIP_TRY(hipMemcpyAsync(w.u.offsets, utt_offsets, sizeof(int) * (n_utt + 1), hipMemcpyHostToDevice, s));
    else
        HIP_TRY(utt_single(w.u, T_, s));
    HIP_TRY(utt_setup(w.u, n_utt, T_, d.Tp, o.iters, s));

    // ---- import the caller's matrices into zero-padded frames-as-rows workspace arrays ----
    // At[n][m], Am[m][n], Xt[t][m], Ht[t][n]
    const bool kl = (o.loss == EVC_LOSS_KL);
    if (!o.dict) {
        DictArrays<T> da{};
        da.At = w.At; da.Am = w.Am; da.Akl = w.Akl;
        DictPlan pl{};
        pl.kl = kl;
        int st = prepare_dict<T>(da, pl, A, lda, nullptr, 0, d, fm, o.eps, false, s);
        if (st) return st;
    }
    HIP_TRY(copy2d<T>(X, ldx, T_, M, fm ? 0 : 1, w.Xt, d.Mk, d.Tp, d.Mk, 0, s));
    if (o.init_mode == EVC_INIT_SKLEARN) HIP_TRY(utt_sklearn_h0<T>(w.Xt, d.Mk, M, N, w.u, n_utt, s));
    else if (o.init_mode == EVC_INIT_CONST) HIP_TRY(utt_const_h0(w.u, n_utt, o.init_value, s));

    if (fused) {     // activations live in the packed tile layout from start to finish
        evc_solve_opts oo = o;
        // A cooperative launch that gave up waiting for a peer workgroup (another process or stream held the
        // CUs it needed) leaves void results, and the solve is redone with one workgroup per frame tile.  The
        // flag costs a host round trip: when the start values can be regenerated it is read after the results
        // have been exported (they are exported again by the redo); with caller-given start values it is read
        // before anything is written to the caller's H.
        const bool check_first = (o.init_mode == EVC_INIT_GIVEN);
        for (int attempt = 0; attempt < 2; ++attempt) {
            if (o.init_mode == EVC_INIT_GIVEN)
                HIP_TRY(fused_import_h(w.fl, w.fb.Hp, reinterpret_cast<const double*>(H), ldh, fm ? 1 : 0, T_, N, s));
            int coop_used = 0, aborted = 0;       // (constant start values: solve_fused)
            int exported = 0;
            // (with caller-given start values the abort flag is read before anything goes to the caller's H: no
            // direct export then)
            int st = solve_fused(w, d, oo, n_utt, s, &coop_used, check_first ? (T*)nullptr : H, ldh, &exported, inf);
            if (st) return st;
            if (coop_used && check_first) {
                st = coop_timed_out(w, s, &aborted);
                if (st) return st;
            }
            if (!aborted) {
                st = finish_fused<T>(w, d, o, exported ? (T*)nullptr : H, ldh, y, s);
                if (st) return st;
                if (coop_used && !check_first) {
                    st = coop_timed_out(w, s, &aborted);
                    if (st) return st;
                }
            }
            if (!aborted) break;
            if (attempt == 1) return ST_COOP_TIMEOUT;      // cannot happen: the redo is not cooperative
            oo.reserved |= 4;
            oo.test_abort_at = 0;
            inf->redo = 1;
            HIP_TRY(utt_setup(w.u, n_utt, T_, d.Tp, o.iters, s));
            if (o.init_mode == EVC_INIT_SKLEARN) HIP_TRY(utt_sklearn_h0<T>(w.Xt, d.Mk, M, N, w.u, n_utt, s));
            else if (o.init_mode == EVC_INIT_CONST) HIP_TRY(utt_const_h0(w.u, n_utt, o.init_value, s));
        }
        return copy_back(w.u, n_utt, n_slots, n_iter_out, err_out, s);
    }

    if (o.init_mode == EVC_INIT_GIVEN)
        HIP_TRY(copy2d<T>(H, ldh, T_, N, fm ? 0 : 1, w.H0, d.Np, d.Tp, d.Np, 0, s));
    else
        HIP_TRY(fill_h0<T>(w.H0, d.Np, d.Tp, N, T_, w.u, s));

    // ---- numerator (and Gram matrix) ----
    const bool gram = (algo == EVC_ALGO_GRAM || algo == EVC_ALGO_LITERAL);
    if (gram) HIP_TRY(gemm_nt<T>(w.At, d.Mk, w.At, d.Mk, w.G, d.Np, d.Np, d.Np, d.Mk, s));
    if (!kl) HIP_TRY(gemm_nt<T>(w.Xt, d.Mk, w.At, d.Mk, w.Pt, d.Np, d.Tp, d.Np, d.Mk, s));

    T* Hc = w.H0;       // current activations
    T* Hn = w.H1;       // ping-pong partner (GRAM only)
    bool v_valid = false;

    auto residual_check = [&](int c) -> int {
        if (!v_valid) HIP_TRY(gemm_nt<T>(Hc, d.Np, w.Am, d.Np, w.Vt, d.Mj, d.Tp, d.Mj, d.Np, s, w.Vsplit, w.vsplit_elems, nullptr, d.Mk));
        v_valid = true;
        if (kl) HIP_TRY(frame_err_kl<T>(w.Xt, d.Mk, w.Vt, d.Mj, M, T_, o.eps, w.err2, s));
        else HIP_TRY(frame_err2<T>(w.Xt, d.Mk, w.Vt, d.Mj, M, T_, w.err2, s));
        HIP_TRY(utt_check(w.err2, w.u, n_utt, c, o.check_every, o.stop_rule, o.tol, s));
        return 0;
    };

    if (o.check_every > 0 && o.stop_rule == EVC_STOP_SKLEARN) {
        int st = residual_check(0);     // error_at_init, sklearn _nmf.py:827
        if (st) return st;
    }

    MuEpilogue<T> ep;
    ep.P = w.Pt; ep.frame_utt = w.u.frame_utt; ep.active = w.u.active; ep.ldh = d.Np;
    ep.N = N; ep.T_ = T_; ep.eps_mode = o.eps_mode; ep.eps = (T)o.eps; ep.l1 = (T)o.l1; ep.kl = kl ? 1 : 0;
    // Once the stop rules have stopped every utterance the launches still queued return at once (the host does not
    // read the flags back: the call stays asynchronous).  In-place path only: GRAM swaps its two H buffers per launch.
    const int* gate = (!gram && o.check_every > 0 && o.stop_rule != EVC_STOP_NONE) ? w.u.active + n_utt : nullptr;
    ep.gate = gate;

    inf->kernel = gemm_kernel_id<T>();
    inf->members = 1;
    inf->launches = o.iters * (gram ? 1 : 2);
    if (o.ev_loop_start) HIP_TRY(hipEventRecord((hipEvent_t)o.ev_loop_start, s));
    for (int it = 1; it <= o.iters; ++it) {
        if (algo == EVC_ALGO_LITERAL) {   // pymf nmf.py:68-69 recomputes both every iteration
            HIP_TRY(gemm_nt<T>(w.At, d.Mk, w.At, d.Mk, w.G, d.Np, d.Np, d.Np, d.Mk, s));
            HIP_TRY(gemm_nt<T>(w.Xt, d.Mk, w.At, d.Mk, w.Pt, d.Np, d.Tp, d.Np, d.Mk, s));
        }
        if (gram) {
            ep.Hin = Hc;                  // H' = mu(H, P, H G^T)   (G symmetric)
            HIP_TRY(gemm_nt_mu<T>(Hc, d.Np, w.G, d.Np, Hn, d.Tp, d.Np, d.Np, ep, s));
            T* tmp = Hc; Hc = Hn; Hn = tmp;
        } else {
            if (!v_valid) HIP_TRY(gemm_nt<T>(Hc, d.Np, w.Am, d.Np, w.Vt, d.Mj, d.Tp, d.Mj, d.Np, s, w.Vsplit, w.vsplit_elems, nullptr, d.Mk, gate));
            ep.Hin = Hc;
            if (kl) {                     // H' = H (.) (X (/) max(V, eps)) (A / colsum)   sklearn _nmf.py:556-606
                HIP_TRY(kl_ratio<T>(w.Xt, d.Mk, w.Vt, d.Mj, M, d.Tp, o.eps, w.Rt, d.Mk, s));
                HIP_TRY(gemm_nt_mu<T>(w.Rt, d.Mk, w.Akl, d.Mk, Hc, d.Tp, d.Np, d.Mk, ep, s));
            } else {                      // H' = mu(H, P, V At^T), V = H Am^T ; in place
                HIP_TRY(gemm_nt_mu<T>(w.Vt, d.Mj, w.At, d.Mk, Hc, d.Tp, d.Np, d.Mk, ep, s));
            }
        }
        v_valid = false;
        if (o.check_every > 0 && it % o.check_every == 0) {
            int st = residual_check(it / o.check_every);
            if (st) return st;
        }
    }

    if (o.ev_loop_stop) HIP_TRY(hipEventRecord((hipEvent_t)o.ev_loop_stop, s));
    if (H) HIP_TRY(copy2d<T>(Hc, d.Np, d.T_, d.N, 0, H, ldh, d.T_, d.N, fm ? 0 : 1, s));
    if (y) {
        int st = synth_rows<T>(Hc, d.Np, *y, d.N, d.T_, fm, s);
        if (st) return st;
    }
    return copy_back(w.u, n_utt, n_slots, n_iter_out, err_out, s);
}

template <typename T> size_t workspace_typed(int M, int Mb, int N, int T_, int n_utt, int algo) {
    const Dims d = make_dims((int)sizeof(T), M, N, T_, n_utt, Mb);
    const int al = algo == EVC_ALGO_AUTO ? EVC_ALGO_FACTORED : algo;
    const int dt = sizeof(T) == 8 ? EVC_F64 : EVC_F32;
    // callers may disable the fused path per call, so the query covers both carvings
    const size_t a = carve<T>(nullptr, d, al, MAX_SLOTS, false).bytes;
    const size_t b = use_fused(M, N, T_, dt, al) ? carve<T>(nullptr, d, al, MAX_SLOTS, true).bytes : 0;
    return a > b ? a : b;
}

bool bad_ld(int layout, int ld, int rows_fm, int cols_fm) {
    // FRAME_MAJOR: rows_fm x cols_fm with ld >= cols_fm; BIN_MAJOR: the transpose
    return layout == EVC_FRAME_MAJOR ? ld < cols_fm : ld < rows_fm;
}

}  // namespace

extern "C" {

int evc_version(void) { return EVC_VERSION; }

const char* evc_strerror(int status) {
    switch (status) {
        case ST_OK: return "ok";
        case ST_BADARG: return "invalid argument";
        case ST_WORKSPACE: return "workspace too small (see evc_workspace_bytes)";
        case ST_UNSUPPORTED: return "unsupported option combination";
        case ST_COOP_TIMEOUT: return "cooperative launch timed out waiting for a peer workgroup (results void)";
        default: break;
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "unknown status";
}

int evc_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    return e == hipSuccess ? n : -(int)e;
}

// float32 callers with a small bin count: the float64 fused kernels are 2.5x faster than the float32
// generic path (and exact to float32 rounding), so their matrices are widened into a staging region, solved by
// the float64 route and narrowed on the way out.  The reference's float32 surface (nmf_tool, TF1) is compared
// against a float64 restatement anyway: computing in float64 only moves the result towards it.
static bool f32_rides_f64(int M, int N, int T, int algo, int reserved) {
    return (algo == EVC_ALGO_AUTO || algo == EVC_ALGO_FACTORED) && !(reserved & 1) &&
           fused_supported(M, N, T, EVC_F64);
}
static size_t f32_staging_bytes(int M, int Mb, int N, int T) {
    const size_t n = (size_t)N * M + (size_t)T * M + (size_t)T * N + (size_t)N * Mb + (size_t)T * Mb;
    return ((n * sizeof(double) + 6 * 256) + 255) & ~size_t(255);   // the inner workspace starts 256-byte aligned
}

size_t evc_workspace_bytes(int M, int Mb, int N, int T, int n_utt, int dtype, int algo) {
    if (M < 0 || Mb < 0 || N < 0 || T < 0 || n_utt < 1) return 0;
    if (algo < EVC_ALGO_GRAM || algo > EVC_ALGO_AUTO) return 0;
    if (dtype == EVC_F64) {
        size_t b = workspace_typed<double>(M, Mb, N, T, n_utt, algo);
        if (wide64_supported(M, N, T, dtype, algo == EVC_ALGO_AUTO ? EVC_ALGO_FACTORED : algo, EVC_LOSS_FROBENIUS)) {
            const Dims d = make_dims(8, M, N, T, n_utt, Mb);
            const size_t wb = carve_wide<double>(nullptr, d, MAX_SLOTS, device_cus(), true, false).bytes;
            if (wb > b) b = wb;
        }
        return b;
    }
    if (dtype == EVC_F32) {
        size_t b = workspace_typed<float>(M, Mb, N, T, n_utt, algo);
        if (wide_supported(M, N, T, dtype, algo == EVC_ALGO_AUTO ? EVC_ALGO_FACTORED : algo)) {
            const Dims d = make_dims(4, M, N, T, n_utt, Mb);
            const size_t wb = carve_wide<float>(nullptr, d, MAX_SLOTS, device_cus(), true, true).bytes;
            if (wb > b) b = wb;
        }
        if (f32_rides_f64(M, N, T, algo, 0)) {
            const size_t c = f32_staging_bytes(M, Mb, N, T) + workspace_typed<double>(M, Mb, N, T, n_utt, algo);
            if (c > b) b = c;
        }
        return b;
    }
    return 0;
}

static int solve_f32_on_f64(const void* A, int lda, const void* X, int ldx, void* H, int ldh, int M, int N, int T,
                            const int* utt_offsets, int n_utt, const evc_solve_opts& o, void* ws, size_t ws_bytes,
                            int* n_iter_out, double* err_out, const SynthArgs* y, hipStream_t s,
                            evc_solve_info* inf) {
    const bool fm = (o.layout == EVC_FRAME_MAJOR);
    const int Mb = y ? y->Mb : 0;
    const size_t stage = f32_staging_bytes(M, Mb, N, T);
    if (ws_bytes < stage) return ST_WORKSPACE;
    // staged matrices keep the caller's orientation with compact rows: (outer, inner) per layout
    const long aR = fm ? N : M, aC = fm ? M : N, xR = fm ? T : M, xC = fm ? M : T, hR = fm ? T : N, hC = fm ? N : T;
    const long bR = fm ? N : Mb, bC = fm ? Mb : N, yR = fm ? T : Mb, yC = fm ? Mb : T;
    Carver c{static_cast<char*>(ws), 0};
    double* A64 = c.take<double>((size_t)N * M);
    double* X64 = c.take<double>((size_t)T * M);
    double* H64 = c.take<double>((size_t)T * N);
    double* B64 = c.take<double>((size_t)N * Mb);
    double* Y64 = c.take<double>((size_t)T * Mb);
    if (!o.dict) HIP_TRY((cvt2d<float, double>(static_cast<const float*>(A), lda, aR, aC, A64, aC, s)));
    HIP_TRY((cvt2d<float, double>(static_cast<const float*>(X), ldx, xR, xC, X64, xC, s)));
    if (H && o.init_mode == EVC_INIT_GIVEN)
        HIP_TRY((cvt2d<float, double>(static_cast<const float*>(H), ldh, hR, hC, H64, hC, s)));
    SynthArgs y64{};
    if (y) {
        if (!(o.dict && o.dict->Mb > 0))
            HIP_TRY((cvt2d<float, double>(static_cast<const float*>(y->B), y->ldb, bR, bC, B64, bC, s)));
        y64 = SynthArgs{B64, (int)bC, Y64, (int)yC, Mb};
    }
    evc_solve_opts o64 = o;
    o64.dtype = EVC_F64;
    const int st = solve_typed<double>(A64, (int)aC, X64, (int)xC, H ? H64 : nullptr, (int)hC, M, N, T, utt_offsets,
                                       n_utt, o64, static_cast<char*>(ws) + stage, ws_bytes - stage, n_iter_out,
                                       err_out, y ? &y64 : nullptr, s, inf);
    if (st) return st;
    if (H) HIP_TRY((cvt2d<double, float>(H64, hC, hR, hC, static_cast<float*>(H), ldh, s)));
    if (y) HIP_TRY((cvt2d<double, float>(Y64, yC, yR, yC, static_cast<float*>(y->Y), y->ldy, s)));
    if (n_iter_out || err_out) HIP_TRY(hipStreamSynchronize(s));    // those calls are synchronous: so are H and Y
    return ST_OK;
}

// ---- prepared dictionaries ----
size_t evc_dict_bytes(int M, int Mb, int N, int dtype, int loss) {
    if (M < 1 || Mb < 0 || N < 1) return 0;
    if (loss != EVC_LOSS_FROBENIUS && loss != EVC_LOSS_KL) return 0;
    if (dtype == EVC_F64) return dict_image_bytes<double>(M, Mb, N, loss, 0);
    if (dtype != EVC_F32) return 0;
    if (f32_rides_f64(M, N, 1, EVC_ALGO_FACTORED, 0))     // widened once, then the float64 image
        return dict_image_bytes<double>(M, Mb, N, loss, dict_f64_staging(M, Mb, N));
    return dict_image_bytes<float>(M, Mb, N, loss, 0);
}

int evc_dict_prepare(const void* A, int lda, const void* B, int ldb, int M, int Mb, int N, int layout, int dtype, int loss,
                     double eps, void* mem, size_t mem_bytes, evc_dict* dict, evc_stream_t stream) {
    if (!A || !mem || !dict || M < 1 || N < 1 || Mb < 0 || (Mb > 0) != (B != nullptr)) return ST_BADARG;
    if (layout != EVC_FRAME_MAJOR && layout != EVC_BIN_MAJOR) return ST_BADARG;
    if (bad_ld(layout, lda, N, M) || (B && bad_ld(layout, ldb, N, Mb))) return ST_BADARG;
    if (loss == EVC_LOSS_KL && !(eps > 0.0)) return ST_UNSUPPORTED;
    const size_t need = evc_dict_bytes(M, Mb, N, dtype, loss);
    if (need == 0) return ST_BADARG;
    if (mem_bytes < need || (reinterpret_cast<uintptr_t>(mem) & 255)) return ST_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool fm = layout == EVC_FRAME_MAJOR;
    int st;
    if (dtype == EVC_F64) {
        st = dict_prepare_typed<double>(static_cast<const double*>(A), lda, static_cast<const double*>(B), ldb, M, Mb, N, fm,
                                        loss, eps, mem, 0, s);
        if (!st) {
            const Dims dd = make_dims(8, M, N, 1, 1, Mb);
            st = prepare_dict_fused(dict_arrays_at<double>(mem, 0, M, Mb, N, loss), dict_plan<double>(M, Mb, N, loss, true),
                                    static_cast<const double*>(B), ldb, dd, fm, true, s);
        }
    } else if (f32_rides_f64(M, N, 1, EVC_ALGO_FACTORED, 0)) {
        // widen A and B once (compact rows in the caller's orientation), then the float64 image behind them
        const size_t skip = dict_f64_staging(M, Mb, N);
        Carver c{static_cast<char*>(mem), 0};
        double* A64 = c.take<double>((size_t)N * M);
        double* B64 = c.take<double>((size_t)N * Mb);
        const long aR = fm ? N : M, aC = fm ? M : N, bR = fm ? N : Mb, bC = fm ? Mb : N;
        HIP_TRY((cvt2d<float, double>(static_cast<const float*>(A), lda, aR, aC, A64, aC, s)));
        if (B) HIP_TRY((cvt2d<float, double>(static_cast<const float*>(B), ldb, bR, bC, B64, bC, s)));
        st = dict_prepare_typed<double>(A64, (int)aC, B ? B64 : nullptr, (int)bC, M, Mb, N, fm, loss, eps, mem, skip, s);
        if (!st) {
            const Dims dd = make_dims(8, M, N, 1, 1, Mb);
            st = prepare_dict_fused(dict_arrays_at<double>(mem, skip, M, Mb, N, loss), dict_plan<double>(M, Mb, N, loss, true),
                                    B ? B64 : nullptr, (int)bC, dd, fm, true, s);
        }
    } else {
        st = dict_prepare_typed<float>(static_cast<const float*>(A), lda, static_cast<const float*>(B), ldb, M, Mb, N, fm,
                                       loss, eps, mem, 0, s);
    }
    if (st) return st;
    dict->struct_bytes = (int)sizeof(evc_dict);
    dict->magic = DICT_MAGIC;
    dict->M = M; dict->Mb = Mb; dict->N = N; dict->dtype = dtype; dict->loss = loss; dict->reserved = 0;
    dict->eps = loss == EVC_LOSS_KL ? eps : 0.0;
    dict->mem = mem;
    dict->bytes = need;
    return ST_OK;
}

static int solve_checked(const void* A, int lda, const void* X, int ldx, void* H, int ldh, int M, int N,
                        int T, const int* utt_offsets, int n_utt, const evc_solve_opts* opts,
                        void* workspace, size_t workspace_bytes, int* n_iter_out, double* err_out,
                        const SynthArgs* y, evc_stream_t stream) {
    if (!opts || opts->struct_bytes != (int)sizeof(evc_solve_opts)) return ST_BADARG;
    const evc_solve_opts& o = *opts;
    if (M < 1 || N < 1 || T < 0 || n_utt < 1 || o.iters < 0) return ST_BADARG;
    if (o.dtype != EVC_F64 && o.dtype != EVC_F32) return ST_BADARG;
    if (o.layout != EVC_FRAME_MAJOR && o.layout != EVC_BIN_MAJOR) return ST_BADARG;
    if (o.algo < EVC_ALGO_GRAM || o.algo > EVC_ALGO_AUTO) return ST_BADARG;
    if (o.eps_mode < EVC_EPS_ADD || o.eps_mode > EVC_EPS_CLAMP) return ST_BADARG;
    if (o.init_mode < EVC_INIT_GIVEN || o.init_mode > EVC_INIT_CONST) return ST_BADARG;
    if (o.stop_rule < EVC_STOP_NONE || o.stop_rule > EVC_STOP_PYMF) return ST_BADARG;
    if (o.check_every < 0) return ST_BADARG;
    if (o.stop_rule != EVC_STOP_NONE && o.check_every == 0) return ST_BADARG;
    if (!(o.l1 >= 0.0)) return ST_BADARG;
    if (o.loss != EVC_LOSS_FROBENIUS && o.loss != EVC_LOSS_KL) return ST_BADARG;
    if (o.loss == EVC_LOSS_KL) {   // sklearn's KL update: its guards, no Gram shortcut, no (quirky) L1
        if (o.eps_mode != EVC_EPS_ZERO_REPLACE || o.l1 != 0.0 || !(o.eps > 0.0)) return ST_UNSUPPORTED;
        if (o.algo == EVC_ALGO_GRAM || o.algo == EVC_ALGO_LITERAL) return ST_UNSUPPORTED;
    }
    if (y && y->Mb < 1) return ST_BADARG;
    if (o.info && o.info->struct_bytes != (int)sizeof(evc_solve_info)) return ST_BADARG;
    if (T == 0) {
        if (n_iter_out) for (int i = 0; i < n_utt; ++i) n_iter_out[i] = 0;
        return ST_OK;
    }
    const evc_dict* dk = o.dict;
    if (dk) {         // a prepared dictionary replaces the A (and B) arguments
        if (dk->struct_bytes != (int)sizeof(evc_dict) || dk->magic != DICT_MAGIC || !dk->mem) return ST_BADARG;
        if (dk->M != M || dk->N != N || dk->dtype != o.dtype || dk->loss != o.loss) return ST_BADARG;
        if (o.loss == EVC_LOSS_KL && dk->eps != o.eps) return ST_BADARG;
        if (y && dk->Mb > 0 && dk->Mb != y->Mb) return ST_BADARG;
        if (dk->bytes < evc_dict_bytes(M, dk->Mb, N, dk->dtype, dk->loss)) return ST_BADARG;
        // a float32 dictionary with M <= 32 was widened for the float64 fused kernels: only that route is prepared
        if (o.dtype == EVC_F32 && f32_rides_f64(M, N, T, EVC_ALGO_FACTORED, 0) && !f32_rides_f64(M, N, T, o.algo, o.reserved))
            return ST_UNSUPPORTED;
    }
    if ((!dk && !A) || !X || !workspace) return ST_BADARG;
    if (!H && (!y || o.init_mode == EVC_INIT_GIVEN)) return ST_BADARG;   // H may be omitted by evc_nmf_convert only
    if ((!dk && bad_ld(o.layout, lda, N, M)) || bad_ld(o.layout, ldx, T, M) || (H && bad_ld(o.layout, ldh, T, N)))
        return ST_BADARG;
    if (y && (!y->Y || bad_ld(o.layout, y->ldy, T, y->Mb))) return ST_BADARG;
    if (y && !(dk && dk->Mb > 0) && (!y->B || bad_ld(o.layout, y->ldb, N, y->Mb))) return ST_BADARG;
    if (utt_offsets) {
        if (utt_offsets[0] != 0 || utt_offsets[n_utt] != T) return ST_BADARG;
        for (int i = 0; i < n_utt; ++i)
            if (utt_offsets[i + 1] < utt_offsets[i]) return ST_BADARG;
    } else if (n_utt != 1) {
        return ST_BADARG;
    }
    {   // tuning bits 16..19: wavefronts per workgroup of k_fused_wide (4 | 8), whole bin tiles per wavefront of
        // k_fused_wide64 (4 | 5 | 7 | 8: the narrowest instance >= the request that holds M); anything else is an error
        const int tw = (o.reserved >> 16) & 0xf;
        if (M > 32 && tw != 0 && !(o.dtype == EVC_F32 ? (tw == 4 || tw == 8) : (tw == 3 || tw == 4 || tw == 5 || tw == 7 || tw == 8)))
            return ST_BADARG;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    evc_solve_info inf{};
    int st;
    const int algo_eff = o.algo == EVC_ALGO_AUTO ? EVC_ALGO_FACTORED : o.algo;
    // a task-queue solve whose bounded wait ran out is redone on the two contractions (nothing has reached H or Y)
    evc_solve_opts redo_o = o;
    redo_o.reserved |= EVC_FLAG_NO_FUSED;
    redo_o.test_abort_at = 0;
    if (o.dtype == EVC_F64 && use_wide(M, N, T, o.dtype, algo_eff, o.loss, o.reserved)) {
        st = solve_wide<double>(static_cast<const double*>(A), lda, static_cast<const double*>(X), ldx,
                                static_cast<double*>(H), ldh, M, N, T, utt_offsets, n_utt, o, workspace, workspace_bytes,
                                n_iter_out, err_out, y, s, &inf);
        if (st == ST_COOP_TIMEOUT) {
            const int launches = inf.launches;
            inf = evc_solve_info{};
            st = solve_typed<double>(A, lda, X, ldx, H, ldh, M, N, T, utt_offsets, n_utt, redo_o, workspace,
                                     workspace_bytes, n_iter_out, err_out, y, s, &inf);
            inf.redo = 1;
            inf.launches += launches;
        }
    } else if (o.dtype == EVC_F64)
        st = solve_typed<double>(A, lda, X, ldx, H, ldh, M, N, T, utt_offsets, n_utt, o, workspace,
                                 workspace_bytes, n_iter_out, err_out, y, s, &inf);
    else if (f32_rides_f64(M, N, T, o.algo, o.reserved))
        st = solve_f32_on_f64(A, lda, X, ldx, H, ldh, M, N, T, utt_offsets, n_utt, o, workspace, workspace_bytes,
                              n_iter_out, err_out, y, s, &inf);
    else if (use_wide(M, N, T, o.dtype, algo_eff, o.loss, o.reserved)) {
        st = solve_wide<float>(static_cast<const float*>(A), lda, static_cast<const float*>(X), ldx, static_cast<float*>(H),
                        ldh, M, N, T, utt_offsets, n_utt, o, workspace, workspace_bytes, n_iter_out, err_out, y, s, &inf);
        if (st == ST_COOP_TIMEOUT) {
            const int launches = inf.launches;
            inf = evc_solve_info{};
            st = solve_typed<float>(A, lda, X, ldx, H, ldh, M, N, T, utt_offsets, n_utt, redo_o, workspace,
                                    workspace_bytes, n_iter_out, err_out, y, s, &inf);
            inf.redo = 1;
            inf.launches += launches;
        }
    } else
        st = solve_typed<float>(A, lda, X, ldx, H, ldh, M, N, T, utt_offsets, n_utt, o, workspace,
                                workspace_bytes, n_iter_out, err_out, y, s, &inf);
    if (o.info && o.info->struct_bytes == (int)sizeof(evc_solve_info)) {
        inf.struct_bytes = (int)sizeof(evc_solve_info);
        *o.info = inf;
    }
    return st;
}

int evc_nmf_solve(const void* A, int lda, const void* X, int ldx, void* H, int ldh, int M, int N,
                  int T, const int* utt_offsets, int n_utt, const evc_solve_opts* opts,
                  void* workspace, size_t workspace_bytes, int* n_iter_out, double* err_out,
                  evc_stream_t stream) {
    return solve_checked(A, lda, X, ldx, H, ldh, M, N, T, utt_offsets, n_utt, opts, workspace,
                         workspace_bytes, n_iter_out, err_out, nullptr, stream);
}

int evc_nmf_convert(const void* A, int lda, const void* X, int ldx, const void* B, int ldb, void* H,
                    int ldh, void* Y, int ldy, int M, int Mb, int N, int T, const int* utt_offsets,
                    int n_utt, const evc_solve_opts* opts, void* workspace, size_t workspace_bytes,
                    int* n_iter_out, double* err_out, evc_stream_t stream) {
    const SynthArgs y{B, ldb, Y, ldy, Mb};
    return solve_checked(A, lda, X, ldx, H, ldh, M, N, T, utt_offsets, n_utt, opts, workspace,
                         workspace_bytes, n_iter_out, err_out, &y, stream);
}

int evc_synthesize(const void* B, int ldb, const void* H, int ldh, void* Y, int ldy, int Mb, int N,
                   int T, int layout, int dtype, evc_stream_t stream) {
    if (Mb < 1 || N < 0 || T < 0) return ST_BADARG;
    if (layout != EVC_FRAME_MAJOR && layout != EVC_BIN_MAJOR) return ST_BADARG;
    if (dtype != EVC_F64 && dtype != EVC_F32) return ST_BADARG;
    if (T == 0) return ST_OK;
    if (!Y || (N > 0 && (!B || !H))) return ST_BADARG;
    if (bad_ld(layout, ldb, N, Mb) || bad_ld(layout, ldh, T, N) || bad_ld(layout, ldy, T, Mb))
        return ST_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t e;
    if (Mb <= 64) {     // a handful of bins: one pass over H, wavefronts split the exemplars (k_synth_skinny)
        const bool fm = layout == EVC_FRAME_MAJOR;
        // FRAME_MAJOR: H[t][n], B[n][mb], Y[t][mb];  BIN_MAJOR: H[n][t], B[mb][n], Y[mb][t]
        const long hst = fm ? ldh : 1, hsn = fm ? 1 : ldh, bsn = fm ? ldb : 1, bsm = fm ? 1 : ldb;
        const long yst = fm ? ldy : 1, ysm = fm ? 1 : ldy;
        if (dtype == EVC_F64)
            e = synth_skinny<double>((const double*)H, hst, hsn, (const double*)B, bsn, bsm, (double*)Y, yst, ysm, T, Mb, N, s);
        else
            e = synth_skinny<float>((const float*)H, hst, hsn, (const float*)B, bsn, bsm, (float*)Y, yst, ysm, T, Mb, N, s);
    } else if (layout == EVC_FRAME_MAJOR) {
        // Y[t][mb] = sum_n H[t][n] B[n][mb]           (np.matmul(H.T, B), 04_align_n_nmf.py:391)
        if (dtype == EVC_F64)
            e = gemm_strided<double>((const double*)H, ldh, 1, (const double*)B, 1, ldb, (double*)Y, ldy, 1, T, Mb, N, s);
        else
            e = gemm_strided<float>((const float*)H, ldh, 1, (const float*)B, 1, ldb, (float*)Y, ldy, 1, T, Mb, N, s);
    } else {
        // Y[mb][t] = sum_n B[mb][n] H[n][t]
        if (dtype == EVC_F64)
            e = gemm_strided<double>((const double*)B, ldb, 1, (const double*)H, 1, ldh, (double*)Y, ldy, 1, Mb, T, N, s);
        else
            e = gemm_strided<float>((const float*)B, ldb, 1, (const float*)H, 1, ldh, (float*)Y, ldy, 1, Mb, T, N, s);
    }
    return (int)e;
}

int evc_residual(const void* A, int lda, const void* X, int ldx, const void* H, int ldh, int M,
                 int N, int T, int layout, int dtype, double* err2_out, void* workspace,
                 size_t workspace_bytes, evc_stream_t stream) {
    if (M < 1 || N < 1 || T < 0) return ST_BADARG;
    if (layout != EVC_FRAME_MAJOR && layout != EVC_BIN_MAJOR) return ST_BADARG;
    if (dtype != EVC_F64 && dtype != EVC_F32) return ST_BADARG;
    if (T == 0) return ST_OK;
    if (!A || !X || !H || !err2_out || !workspace) return ST_BADARG;
    if (bad_ld(layout, lda, N, M) || bad_ld(layout, ldx, T, M) || bad_ld(layout, ldh, T, N))
        return ST_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const Dims d = make_dims(dtype == EVC_F64 ? 8 : 4, M, N, T, 1, 0);
    const bool fm = (layout == EVC_FRAME_MAJOR);
#define EVC_RESID(TT)                                                                              \
    {                                                                                              \
        Workspace<TT> w = carve<TT>(workspace, d, EVC_ALGO_FACTORED, MAX_SLOTS, false);             \
        if (w.bytes > workspace_bytes) return ST_WORKSPACE;                                        \
        HIP_TRY(copy2d<TT>((const TT*)A, lda, M, N, fm ? 1 : 0, w.Am, d.Np, d.Mj, d.Np, 0, s));     \
        HIP_TRY(copy2d<TT>((const TT*)X, ldx, T, M, fm ? 0 : 1, w.Xt, d.Mk, d.Tp, d.Mk, 0, s));     \
        HIP_TRY(copy2d<TT>((const TT*)H, ldh, T, N, fm ? 0 : 1, w.H0, d.Np, d.Tp, d.Np, 0, s));     \
        HIP_TRY(gemm_nt<TT>(w.H0, d.Np, w.Am, d.Np, w.Vt, d.Mj, d.Tp, d.Mj, d.Np, s, nullptr, 0, nullptr, d.Mk));              \
        HIP_TRY(frame_err2<TT>(w.Xt, d.Mk, w.Vt, d.Mj, M, T, err2_out, s));                         \
    }
    if (dtype == EVC_F64) EVC_RESID(double) else EVC_RESID(float)
#undef EVC_RESID
    return ST_OK;
}

size_t evc_griffin_lim_workspace_bytes(int T, int fft_size, int hop, int iters) {
    if (T < 1 || fft_size < 2 || (fft_size & 1) || hop < 1 || iters < 0) return 0;
    return gl_workspace_bytes(T, 1, fft_size, hop, iters);
}

int evc_griffin_lim(const void* mag, int ldm, int T, int fft_size, int hop, int iters, void* x,
                    void* workspace, size_t workspace_bytes, double* rmse_out, evc_stream_t stream) {
    if (T < 0) return ST_BADARG;
    const int off[2] = {0, T};
    return evc_griffin_lim_batch(mag, ldm, off, 1, fft_size, hop, iters, x, workspace, workspace_bytes, rmse_out,
                                 stream);
}

static bool gl_offsets_ok(const int* off, int n_utt) {
    if (!off || n_utt < 1 || off[0] != 0) return false;
    for (int u = 0; u < n_utt; ++u)
        if (off[u + 1] < off[u]) return false;
    return true;
}

size_t evc_griffin_lim_batch_workspace_bytes(const int* frame_offsets, int n_utt, int fft_size, int hop, int iters) {
    if (!gl_offsets_ok(frame_offsets, n_utt) || frame_offsets[n_utt] < 1 || fft_size < 2 || (fft_size & 1) ||
        hop < 1 || iters < 0)
        return 0;
    return gl_workspace_bytes(frame_offsets[n_utt], n_utt, fft_size, hop, iters);
}

int evc_griffin_lim_batch(const void* mag, int ldm, const int* frame_offsets, int n_utt, int fft_size, int hop,
                          int iters, void* x, void* workspace, size_t workspace_bytes, double* rmse_out,
                          evc_stream_t stream) {
    if (!gl_offsets_ok(frame_offsets, n_utt) || fft_size < 2 || (fft_size & 1) || hop < 1 || iters < 0)
        return ST_BADARG;
    if (frame_offsets[n_utt] == 0) return ST_OK;
    if (!mag || !x || !workspace || ldm < fft_size / 2 + 1) return ST_BADARG;
    const size_t need = gl_workspace_bytes(frame_offsets[n_utt], n_utt, fft_size, hop, iters);
    if (need == 0) return ST_BADARG;
    if (workspace_bytes < need) return ST_WORKSPACE;
    return (int)gl_run(static_cast<const double*>(mag), ldm, frame_offsets, n_utt, fft_size, hop, iters,
                       static_cast<double*>(x), workspace, rmse_out, reinterpret_cast<hipStream_t>(stream));
}

int evc_stft_frames(long n_samples, int fft_size, int hop, int center) {
    if (n_samples < 1 || fft_size < 2 || (fft_size & 1) || hop < 1) return 0;
    return stft_frames(n_samples, hop, center != 0, fft_size);
}

size_t evc_stft_workspace_bytes(long n_samples, int fft_size, int hop, int center) {
    if (n_samples < 1 || fft_size < 2 || (fft_size & 1) || hop < 1) return 0;
    return stft_workspace_bytes(n_samples, fft_size, hop, center != 0);
}

int evc_stft(const void* x, long n_samples, int fft_size, int hop, int center, void* re, int ldre, void* im,
             int ldim, void* workspace, size_t workspace_bytes, evc_stream_t stream) {
    if (n_samples < 0 || fft_size < 2 || (fft_size & 1) || hop < 1) return ST_BADARG;
    if (n_samples > (1L << 31) - 4096) return ST_BADARG;        // frame counts are ints
    if (n_samples == 0 || stft_frames(n_samples, hop, center != 0, fft_size) == 0) return ST_OK;
    const int nb = fft_size / 2 + 1;
    if (!x || !re || !im || !workspace || ldre < nb || ldim < nb) return ST_BADARG;
    if (workspace_bytes < stft_workspace_bytes(n_samples, fft_size, hop, center != 0)) return ST_WORKSPACE;
    return (int)stft_run(static_cast<const double*>(x), n_samples, fft_size, hop, center != 0,
                         static_cast<double*>(re), ldre, static_cast<double*>(im), ldim, workspace,
                         reinterpret_cast<hipStream_t>(stream));
}

static bool dtw_offsets_ok(const int* off, int n_pairs) {
    if (!off || off[0] != 0) return false;
    for (int p = 0; p < n_pairs; ++p)
        if (off[p + 1] < off[p] || off[p + 1] - off[p] > dtw_max_frames()) return false;
    return true;
}

size_t evc_dtw_workspace_bytes(const int* a_offsets, const int* b_offsets, int n_pairs) {
    if (n_pairs < 1 || !dtw_offsets_ok(a_offsets, n_pairs) || !dtw_offsets_ok(b_offsets, n_pairs)) return 0;
    return dtw_workspace_bytes(a_offsets, b_offsets, n_pairs);
}

int evc_dtw_align(const void* A, int lda, const int* a_offsets, const void* B, int ldb,
                  const int* b_offsets, int D, int n_pairs, int* path_a, int* path_b, int* path_len,
                  double* total, void* workspace, size_t workspace_bytes, evc_stream_t stream) {
    if (n_pairs < 1 || D < 1 || lda < D || ldb < D) return ST_BADARG;
    if (D > 512 || n_pairs > 65535) return ST_UNSUPPORTED;     // LDS tile of the cost kernel; grid z
    if (!dtw_offsets_ok(a_offsets, n_pairs) || !dtw_offsets_ok(b_offsets, n_pairs)) return ST_BADARG;
    if (!A || !B || !path_a || !path_b || !path_len || !workspace) return ST_BADARG;
    if (workspace_bytes < dtw_workspace_bytes(a_offsets, b_offsets, n_pairs)) return ST_WORKSPACE;
    return (int)dtw_run(static_cast<const double*>(A), lda, a_offsets, static_cast<const double*>(B), ldb,
                        b_offsets, D, n_pairs, path_a, path_b, path_len, total, workspace,
                        reinterpret_cast<hipStream_t>(stream));
}


int evc_dtw_path_rows(const int* path_len, int n_pairs, int* row_start, int* n_rows_out, evc_stream_t stream) {
    if (!path_len || !row_start || n_pairs < 1 || n_pairs > 65535) return ST_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(dtw_path_scan(path_len, n_pairs, row_start, s));
    if (n_rows_out) {
        HIP_TRY(hipMemcpyAsync(n_rows_out, row_start + n_pairs, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return ST_OK;
}

int evc_dtw_gather_rows(const void* src, long ld_src, int elem_stride, const int* path, const int* path_len,
                        const int* src_offsets, const int* pair_offsets, const int* row_start, int n_pairs, int cols,
                        int op, void* dst, long ld_dst, int dtype, evc_stream_t stream) {
    if (!src || !path || !path_len || !src_offsets || !pair_offsets || !row_start || !dst) return ST_BADARG;
    if (n_pairs < 1 || n_pairs > 65535 || cols < 1 || elem_stride < 1 || ld_dst < cols) return ST_BADARG;
    if (ld_src < (long)(cols - 1) * elem_stride + 1) return ST_BADARG;
    if (op != EVC_GATHER_COPY && op != EVC_GATHER_ABS) return ST_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == EVC_F64)
        HIP_TRY(dtw_gather<double>(static_cast<const double*>(src), ld_src, elem_stride, path, path_len, src_offsets,
                                   pair_offsets, row_start, n_pairs, cols, op, static_cast<double*>(dst), ld_dst, s));
    else if (dtype == EVC_F32)
        HIP_TRY(dtw_gather<float>(static_cast<const float*>(src), ld_src, elem_stride, path, path_len, src_offsets,
                                  pair_offsets, row_start, n_pairs, cols, op, static_cast<float*>(dst), ld_dst, s));
    else
        return ST_BADARG;
    return ST_OK;
}

}  // extern "C"
