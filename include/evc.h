/*
 * evc.h - C ABI of libevc_hip.so: the MI355X (gfx950) exemplar-NMF activation solver.
 *
 * The reference (entn-at/exemplars_vc) is pure Python and has no FFI of its own; the
 * boundary of its hot path is three plain-Python call surfaces.  This header is the
 * native boundary those surfaces bind to (see INTEGRATION.md for the ctypes stubs):
 *
 *   evc_nmf_solve   replaces the multiplicative-update loop behind
 *                     - _factorize()                    04_align_n_nmf.py:194-215
 *                       (sklearn _fit_multiplicative_update, _nmf.py:731-893,
 *                        _multiplicative_update_w beta=2 branch, _nmf.py:526-556,612-631)
 *                     - pymf NMF._update_h / factorize  pymf/nmf.py:66-70, pymf/base.py:208-270
 *                     - nmf_tool NMF.NMF (mu, initW)    nmf_tool/nmf.py:36-40,57-67
 *   evc_synthesize  replaces np.matmul(H.T, B) in convert()  04_align_n_nmf.py:371-373,391
 *   evc_nmf_convert both of the above back to back (factorize() + convert(), :452-455)
 *   evc_griffin_lim replaces reconstruct_signal_griffin_lim()  zz_audio_utilities.py:258-292
 *   evc_stft        replaces librosa.core.stft(...) of 04_align_n_nmf.py:422
 *   evc_dtw_align   replaces _dtw_alignment() / dtw_alignment()  01_make_dict_parallel.py:215-249
 *   evc_dtw_path_rows, evc_dtw_gather_rows  replace align_sp_ap_f0() + the stacking  04_align_n_nmf.py:100-169,230-246
 *   evc_residual    replaces sklearn _beta_divergence(beta=2, square_root=True)
 *                   (_nmf.py:85-135) and pymf frobenius_norm (pymf/base.py:144-165)
 *
 * Conventions
 *   Math (BASELINE.json north_star): X is M x T (bins x frames), A is M x N (source
 *   exemplar dictionary), B is Mb x N (parallel target dictionary), H is N x T, Y = B H.
 *   All pointers are DEVICE pointers unless marked "host".  Nothing here allocates,
 *   frees or throws; every function returns a status (0 ok; -1 invalid argument, -2 workspace
 *   too small, -3 unsupported combination, -4 cooperative launch timed out twice (not reachable:
 *   the redo is not cooperative); >0 a hipError_t value).  Work is enqueued on `stream`.
 *   Host synchronisation - exactly these cases, nothing else waits:
 *     (1) n_iter_out / err_out / rmse_out non-NULL: the call returns after copying them back;
 *     (2) evc_nmf_solve / evc_nmf_convert on the float64 fused path (M <= 32) when several workgroups
 *         share a frame tile and exchange partial sums inside a launch (k_fused_all for N > 512,
 *         the cooperative k_fused_res launch for one or two utterances): one round trip at the end
 *         of the call reads the flag that tells whether a workgroup gave up waiting for its peers (then
 *         the solve is redone without any exchange).  EVC_FLAG_NO_EXCHANGE keeps such a call fully
 *         asynchronous (at about half the speed for a lone utterance, ~10 % less for large batches);
 *     (3) evc_nmf_solve / evc_nmf_convert on the task-queue kernels for wide spectra (k_fused_wide: float32,
 *         32 < M <= 208, from one utterance (43 frame tiles) on; k_fused_wide64: float64,
 *         176 < M <= 528, 3 .. ~30 utterances, small dictionaries from one):
 *         the same round trip, taken BEFORE anything is written to H or Y, so that a solve whose wait ran out is
 *         redone on the two-contraction path from the untouched inputs (evc_solve_info.redo = 1).  Round 3 delivered
 *         NaN under status 0 there.  EVC_FLAG_NO_EXCHANGE routes away from these kernels too.  Batches of up to ~5
 *         utterances run these kernels on a static schedule (one task per workgroup and iteration) that needs all its
 *         workgroups resident at once, like k_fused_all's exchange: two such solves started concurrently on two streams
 *         can starve each other until the bounded waits run out (seconds) and both are redone - give concurrent
 *         small solves EVC_FLAG_NO_EXCHANGE (the compat layer's side streams do).  Larger batches draw tasks from a
 *         queue and depend on nobody's residency.
 *   No global mutable state: calls on distinct streams/devices are independent and the
 *   caller's current device (hipSetDevice) is honoured.  Nothing is read from the process environment.
 *   Host arrays (utt_offsets, frame_offsets, a_offsets / b_offsets) are consumed before the call returns: they are
 *   copied to the device by hipMemcpyAsync from pageable memory, which HIP stages at enqueue time; keep them valid
 *   until the call returns, not longer.
 *   k_fused_all's exchange carries its arrival flag in the lowest mantissa bit of every partial sum it publishes
 *   (readers clear it): each partial V' is truncated by at most one ulp, infinities and NaNs pass unchanged.  So does
 *   k_fused_wide on its static schedule (float32, batches of up to ~5 utterances): one float32 ulp per partial sum and
 *   per summed slice.
 */
#ifndef EVC_H
#define EVC_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVC_VERSION 100 /* 0.1.0 */

/* hipStream_t, spelled without the HIP headers so that C callers can include this file. */
typedef struct ihipStream_t* evc_stream_t;

/* element type of every matrix in a call (the arithmetic is carried out in the same type) */
enum { EVC_F64 = 0, EVC_F32 = 1 };

/* storage order of the caller's matrices
 *   FRAME_MAJOR : the reference scripts' orientation (frames / exemplars as rows):
 *                 X[t*ldx+m]  A[n*lda+m]  B[n*ldb+mb]  H[t*ldh+n]  Y[t*ldy+mb]
 *                 (X = `X` T x M, A = `W` N x M of _factorize; H = sklearn's W, T x N)
 *   BIN_MAJOR   : the north_star / pymf / nmf_tool orientation (bins as rows):
 *                 X[m*ldx+t]  A[m*lda+n]  B[mb*ldb+n] H[n*ldh+t]  Y[mb*ldy+t]          */
enum { EVC_FRAME_MAJOR = 0, EVC_BIN_MAJOR = 1 };

/* how the denominator D = A^T A H (+ l1) is guarded, and the multiply/divide order
 *   ADD          H <- (H*P) / (D + eps)            pymf nmf.py:68-70      (eps = 1e-9)
 *   ZERO_REPLACE D[D==0] = eps; H <- H * (P/D)     sklearn _nmf.py:620-629 (eps = 1.1920929e-7)
 *   NONE         H <- H * P / D                    nmf_tool nmf.py:39
 *   CLAMP        H <- H * (P / max(D, eps))        deComP batch_mu.py:8-26 (eps = 1e-15)   */
enum { EVC_EPS_ADD = 0, EVC_EPS_ZERO_REPLACE = 1, EVC_EPS_NONE = 2, EVC_EPS_CLAMP = 3 };

/* which algebra evaluates the denominator
 *   GRAM      G = A^T A and P = A^T X once, then G H per iteration (sklearn's hoisting)
 *   FACTORED  A^T (A H) per iteration: 4MN instead of 2N^2 flop per frame-iteration
 *   LITERAL   G and P recomputed in every iteration, as pymf/nmf_tool literally do
 *   AUTO      FACTORED                                                                    */
enum { EVC_ALGO_GRAM = 0, EVC_ALGO_FACTORED = 1, EVC_ALGO_LITERAL = 2, EVC_ALGO_AUTO = 3 };

/* initial activations
 *   GIVEN    H holds H0 on entry (pymf / nmf_tool: random, caller-seeded)
 *   SKLEARN  every entry of utterance u starts at sqrt(mean(X_u) / N)  (_nmf.py:1228-1231)
 *   CONST    every entry starts at init_value                                              */
enum { EVC_INIT_GIVEN = 0, EVC_INIT_SKLEARN = 1, EVC_INIT_CONST = 2 };

/* stopping rule, evaluated per utterance every `check_every` iterations on
 * err = ||X_u - A H_u||_F
 *   NONE     errors are recorded (if check_every > 0) but never stop the loop
 *   SKLEARN  stop when (err_prev - err) / err_at_init < tol       (_nmf.py:871-884)
 *   PYMF     from the third recorded error on, stop when |err - err_prev| / T_u < tol
 *            (pymf/base.py:189-206,266-270)                                               */
enum { EVC_STOP_NONE = 0, EVC_STOP_SKLEARN = 1, EVC_STOP_PYMF = 2 };

/* divergence minimised by the multiplicative update
 *   FROBENIUS  H <- H (.) A^T X (/) (A^T A H)                       (what the scripts force, :210)
 *   KL         H <- H (.) A^T (X (/) max(A H, eps)) (/) colsum(A)   generalised Kullback-Leibler: the
 *              default of _factorize's signature (04_align_n_nmf.py:194), sklearn _nmf.py:556-606;
 *              requires EVC_EPS_ZERO_REPLACE (sklearn's guards) and l1 == 0; the residual reported
 *              to the stopping rule is sqrt(2 KL(X || A H)) (_nmf.py:136-160)                      */
enum { EVC_LOSS_FROBENIUS = 0, EVC_LOSS_KL = 1 };

/* evc_solve_opts.reserved
 *   NO_FUSED         the generic two-contraction path instead of the fused persistent kernels (M <= 32)
 *   EXACT_DIV        correctly rounded quotients in the fused float64 kernels (always on with EVC_STOP_PYMF;
 *                    else a shared / refined reciprocal, <= 2 ulp)
 *   NO_EXCHANGE      no kernel in which workgroups exchange data inside a launch (k_fused_all with more than one member,
 *                    cooperative k_fused_res, the task queues k_fused_wide / k_fused_wide64): the call is then fully
 *                    asynchronous (see "Host synchronisation" above); a latency / determinism knob
 *   NO_ALL_RESIDENT  keep k_fused_all out (k_fused_res, with its cooperative launch for few frame tiles)
 *   PAIR_TILES       tuning / experiments: k_fused_xy (two frame tiles per member, exchange phases inside the sweeps)
 *                    instead of k_fused_all where both apply; measured slower (profiles/r04_xy_notes.md) */
enum { EVC_FLAG_NO_FUSED = 1, EVC_FLAG_EXACT_DIV = 2, EVC_FLAG_NO_EXCHANGE = 4, EVC_FLAG_NO_ALL_RESIDENT = 16, EVC_FLAG_PAIR_TILES = 32 };

struct evc_solve_info;
struct evc_dict;
typedef struct evc_solve_opts {
    int struct_bytes;  /* sizeof(evc_solve_opts), for forward compatibility */
    int dtype;         /* EVC_F64 | EVC_F32 */
    int layout;        /* EVC_FRAME_MAJOR | EVC_BIN_MAJOR */
    int algo;          /* EVC_ALGO_* */
    int iters;         /* maximum number of multiplicative updates (>= 0) */
    int eps_mode;      /* EVC_EPS_* */
    int init_mode;     /* EVC_INIT_* */
    int check_every;   /* 0: never evaluate the residual; k>0: every k iterations */
    int stop_rule;     /* EVC_STOP_* */
    int reserved;      /* flags, 0 = defaults: an OR of EVC_FLAG_* (below); bits 8..15: tuning only - M <= 32: 1 | 2 force
                          the general streamed kernel with that many frame tiles per workgroup; M > 32: that many exemplar
                          ranges per frame group in k_fused_wide / k_fused_wide64, whatever the batch size; bits 16..19,
                          tuning only: k_fused_wide with 4 | 8 wavefronts per workgroup, k_fused_wide64 with at least 3 | 4 | 5 | 7 | 8
                          whole bin tiles per wavefront (the narrowest instance that holds M); other values: status -1; with
                          a prepared dictionary whose images do not fit the override: status -3 */
    int loss;          /* EVC_LOSS_* */
    int test_abort_at; /* 0 in production.  Tests only: k > 0 pretends, in front of the k-th launch of the iteration
                          loop, that a workgroup gave up waiting for its peers (the abort flag is raised as a timed-out
                          wait would raise it); -1: the call starts with the flag raised.  The solve then takes the
                          documented redo path and evc_solve_info.redo reports 1. */
    double eps;        /* guard value for eps_mode */
    double l1;         /* added to the denominator (sklearn l1_reg_W = M*alpha_W*l1_ratio) */
    double tol;        /* threshold of stop_rule */
    double init_value; /* EVC_INIT_CONST */
    /* optional hipEvent_t pair recorded on `stream` immediately before / after the launches of
     * the iteration loop (the dominant kernel); NULL = not recorded.  Used by bench.py to time
     * that kernel live with HIP events. */
    void* ev_loop_start;
    void* ev_loop_stop;
    /* optional evc_solve_info* (host, caller-owned, struct_bytes set by the caller; NULL = not wanted): filled before the
     * call returns with what the library actually ran.  No extra synchronisation: `redo` is known from the round trip an
     * exchanging solve performs anyway. */
    struct evc_solve_info* info;
    /* optional prepared dictionary (host struct filled by evc_dict_prepare; NULL = import A / B from the caller's
     * matrices on every call, as the reference's scripts effectively do).  With a prepared dictionary the A and B
     * arguments of evc_nmf_solve / evc_nmf_convert are ignored (may be NULL); M, Mb, N, dtype, loss and layout-independent
     * options must match what it was prepared for, else the call returns -1. */
    const struct evc_dict* dict;
} evc_solve_opts;

/* the kernel that carried the iteration loop of a solve */
enum {
    EVC_KERNEL_NONE = 0,
    EVC_KERNEL_GEMM_NT = 1,     /* generic path, k_gemm_nt x2 per iteration (float64, M > 32; GRAM / LITERAL) */
    EVC_KERNEL_GEMM2 = 2,       /* generic path, k_gemm2 x2 per iteration (float32) */
    EVC_KERNEL_FUSED_MU = 3,    /* k_fused_mu: fused FACTORED, everything streamed (M <= 32) */
    EVC_KERNEL_FUSED_RES = 4,   /* k_fused_res: half of H register-resident (members > 1: its cooperative launch) */
    EVC_KERNEL_FUSED_ALL = 5,   /* k_fused_all: H and P register-resident, `members` workgroups per frame tile */
    EVC_KERNEL_FUSED_WIDE = 6,  /* k_fused_wide: fused FACTORED for float32, 32 < M <= 208: task queue over (frame group, exemplar range) */
    EVC_KERNEL_FUSED_WIDE64 = 7, /* k_fused_wide64: the same for float64, 144 < M <= 528 (bins split over a workgroup's wavefronts) */
    EVC_KERNEL_FUSED_XY = 8     /* k_fused_xy: H and P register-resident, `members` workgroups per PAIR of frame tiles, the
                                   exchange phases inside the sweeps (round 4; on request only: EVC_FLAG_PAIR_TILES) */
};

typedef struct evc_solve_info {
    int struct_bytes;  /* in: sizeof(evc_solve_info) */
    int kernel;        /* EVC_KERNEL_* of the (last) attempt whose results were delivered */
    int members;       /* workgroups / tasks sharing one frame tile or frame group (1: no sharing) */
    int launches;      /* kernel launches of the iteration loop (all attempts) */
    int redo;          /* 1: an exchange wait ran out and the solve was redone on kernels without exchange */
    int exchange;      /* 1: the delivered results come from a kernel whose workgroups exchange partial sums in a launch */
    int prepared;      /* 1: the dictionary came from an evc_dict_prepare image (no per-call import / packing) */
    int reserved;
} evc_solve_info;

/* A dictionary imported once.  The reference builds A and B once per run (04_align_n_nmf.py:230-246,350-361) and the
 * dictionary is fixed across utterances; every evc_nmf_solve / evc_nmf_convert call nevertheless has to bring the caller's
 * matrices into the layouts its kernels read (zero-padded transposes, MFMA operand fragments, KL column scaling, row
 * sums).  evc_dict_prepare does that once into caller-owned device memory; calls that pass the handle
 * in evc_solve_opts.dict skip it (results are bitwise those of the unprepared call).  The struct is plain host data:
 * copyable, nothing to free besides `mem`, which the caller owns and must keep alive and unmodified while it is used. */
typedef struct evc_dict {
    int struct_bytes;  /* sizeof(evc_dict), set by evc_dict_prepare */
    int magic;
    int M, Mb, N;      /* Mb = 0: no target dictionary B was given (evc_nmf_convert then needs its B argument) */
    int dtype, loss, reserved;
    double eps;        /* KL: the guard the column sums were formed with (must equal evc_solve_opts.eps) */
    void* mem;         /* device memory, evc_dict_bytes() bytes, 256-byte aligned */
    size_t bytes;
} evc_dict;

int evc_version(void);
const char* evc_strerror(int status);

/* bytes of device memory a prepared dictionary of this size needs (0: invalid arguments) */
size_t evc_dict_bytes(int M, int Mb, int N, int dtype, int loss);
/* Import A (M x N) and, if B != NULL, B (Mb x N), both in `layout`, into `mem` and fill *dict.  Asynchronous on `stream`
 * (the handle may be used by later calls on the same stream at once).  eps: the KL guard (ignored for Frobenius).
 * The image serves every algebra and kernel route of evc_nmf_solve / evc_nmf_convert (the Gram matrix of
 * EVC_ALGO_GRAM / LITERAL is still formed per call). */
int evc_dict_prepare(const void* A, int lda, const void* B, int ldb, int M, int Mb, int N, int layout, int dtype,
                     int loss, double eps, void* mem, size_t mem_bytes, evc_dict* dict, evc_stream_t stream);

/* number of GPUs visible to the library (hipGetDeviceCount); <0 on failure */
int evc_device_count(void);

/* Bytes of device workspace evc_nmf_solve / evc_nmf_convert / evc_residual need for a problem of
 * this size.  Mb: bins of the target dictionary B for evc_nmf_convert, 0 otherwise.
 * n_utt is the number of utterances the T frames are split into (>= 1). */
size_t evc_workspace_bytes(int M, int Mb, int N, int T, int n_utt, int dtype, int algo);

/* Solve X ~ A H for H >= 0 with A fixed: `iters` multiplicative updates
 *   H <- H (.) A^T X (/) guard(A^T A H + l1).
 *
 * The T frames are the concatenation of n_utt utterances; utt_offsets (host, n_utt+1
 * ascending ints, utt_offsets[0]=0, utt_offsets[n_utt]=T) delimits them, NULL means one
 * utterance.  Utterances only matter for EVC_INIT_SKLEARN and for the stopping rule, both
 * of which the reference applies per call, i.e. per utterance: frames of a stopped
 * utterance are frozen while the others go on.
 *
 * n_iter_out (host, n_utt ints or NULL): updates applied to each utterance.
 * err_out    (host, n_utt * (1 + iters/check_every) doubles or NULL): per utterance, the
 *            residual at init followed by the residual at each check that was evaluated
 *            (unevaluated slots are NaN).
 * When both are NULL the call is fully asynchronous.                                    */
int evc_nmf_solve(const void* A, int lda, const void* X, int ldx, void* H, int ldh,
                  int M, int N, int T,
                  const int* utt_offsets, int n_utt,
                  const evc_solve_opts* opts,
                  void* workspace, size_t workspace_bytes,
                  int* n_iter_out, double* err_out,
                  evc_stream_t stream);

/* evc_nmf_solve followed by Y = B H in one launch sequence: factorize() + convert() of
 * 04_align_n_nmf.py:218-333,336-393 (the __main__ sequence :452-455).  The synthesis reads the
 * activations in the solver's own tile layout, so H is not re-read in the caller's layout; H may be
 * NULL when only Y is wanted (not with EVC_INIT_GIVEN).  B: Mb x N target exemplars, Y: Mb x T,
 * both in opts->layout.  Workspace: evc_workspace_bytes(M, Mb, N, T, ...). */
int evc_nmf_convert(const void* A, int lda, const void* X, int ldx, const void* B, int ldb,
                    void* H, int ldh, void* Y, int ldy,
                    int M, int Mb, int N, int T,
                    const int* utt_offsets, int n_utt,
                    const evc_solve_opts* opts,
                    void* workspace, size_t workspace_bytes,
                    int* n_iter_out, double* err_out,
                    evc_stream_t stream);

/* Y = B H  (04_align_n_nmf.py:391: np.matmul(H.T, B) in FRAME_MAJOR orientation). */
int evc_synthesize(const void* B, int ldb, const void* H, int ldh, void* Y, int ldy,
                   int Mb, int N, int T, int layout, int dtype, evc_stream_t stream);

/* err2_out[t] (device, T values of `dtype`... always double) = sum_m (X[m,t] - (A H)[m,t])^2.
 * The Frobenius residual of a set of frames is sqrt of the sum of its entries.
 * workspace: evc_workspace_bytes(M, 0, N, T, 1, dtype, EVC_ALGO_GRAM) suffices.            */
int evc_residual(const void* A, int lda, const void* X, int ldx, const void* H, int ldh,
                 int M, int N, int T, int layout, int dtype,
                 double* err2_out, void* workspace, size_t workspace_bytes,
                 evc_stream_t stream);

/* STFT front end - the feature extraction that feeds the path when the scripts run on STFT features:
 * librosa.core.stft(y, n_fft=400, hop_length=80, window='hann') at 04_align_n_nmf.py:422 and
 * 03_a_b_r_parallel.py:103 (librosa is a third-party dependency absent here; its published algorithm
 * is restated: frames centred by reflect-padding n_fft/2 samples, periodic Hann window, rfft).  float64.
 *   x      : n_samples doubles (device)
 *   re, im : n_frames x (fft_size/2 + 1), rows are time slices (device) - i.e. the transposed `.T`
 *            the scripts store; n_frames = evc_stft_frames(n_samples, fft_size, hop, center)
 *   center : 1 = librosa's default (reflect padding), 0 = frames start at sample 0            */
int evc_stft_frames(long n_samples, int fft_size, int hop, int center);
size_t evc_stft_workspace_bytes(long n_samples, int fft_size, int hop, int center);
int evc_stft(const void* x, long n_samples, int fft_size, int hop, int center, void* re, int ldre,
             void* im, int ldim, void* workspace, size_t workspace_bytes, evc_stream_t stream);

/* Griffin-Lim phase reconstruction - the back end that follows the path when the scripts run on
 * STFT magnitudes: reconstruct_signal_griffin_lim(), zz_audio_utilities.py:258-292 (with its
 * stft_for_reconstruction / istft_for_reconstruction, :181-218), called from synthesize2(),
 * 04_align_n_nmf.py:182-191.  float64.
 *   mag : T x (fft_size/2 + 1) magnitudes, rows are time slices, row stride ldm (device)
 *   x   : T*hop + fft_size samples (device); in: the initial signal (the reference draws
 *         np.random.randn), out: the reconstruction after `iters` iterations
 *   rmse_out (host, iters doubles or NULL): the per-iteration RMSE the reference prints; non-NULL
 *         makes the call synchronous.   fft_size must be even. */
size_t evc_griffin_lim_workspace_bytes(int T, int fft_size, int hop, int iters);
int evc_griffin_lim(const void* mag, int ldm, int T, int fft_size, int hop, int iters, void* x,
                    void* workspace, size_t workspace_bytes, double* rmse_out, evc_stream_t stream);
/* The same for a batch of utterances in one call (the reference reconstructs one file per call of
 * synthesize2(), 04_align_n_nmf.py:182-191; a batch fills the GPU, one 688-frame utterance does not).
 *   frame_offsets : host, n_utt + 1 ints, frame_offsets[0] = 0: utterance u owns rows frame_offsets[u] ..
 *                   frame_offsets[u+1]-1 of `mag` (T_u frames)
 *   x             : the signals back to back: utterance u's T_u*hop + fft_size samples start at sample
 *                   hop*frame_offsets[u] + u*fft_size (device; in: initial signals, out: reconstructions)
 *   rmse_out      : host, n_utt x iters doubles ([u][iteration]) or NULL; non-NULL makes the call synchronous
 * Every utterance's result equals that of a call of its own up to the summation order of the contractions
 * (the split of the 400-deep sums over workgroups depends on the number of rows). */
size_t evc_griffin_lim_batch_workspace_bytes(const int* frame_offsets, int n_utt, int fft_size, int hop,
                                             int iters);
int evc_griffin_lim_batch(const void* mag, int ldm, const int* frame_offsets, int n_utt, int fft_size, int hop,
                          int iters, void* x, void* workspace, size_t workspace_bytes, double* rmse_out,
                          evc_stream_t stream);

/* Dynamic-time-warping alignment of parallel utterance pairs - the step that builds the parallel
 * dictionary: _dtw_alignment(), 01_make_dict_parallel.py:215-228, i.e. the third-party call
 * dtw(feat_A.T, feat_B.T, dist=lambda x, y: sum(np.square(x - y))) (accumulated cost with steps
 * (i-1,j-1), (i-1,j), (i,j-1); trace-back with ties to the diagonal, then to i-1).  float64.
 *   A, B      : frames as rows (row strides lda, ldb), D features each; pair p owns rows
 *               a_offsets[p] .. a_offsets[p+1]-1 of A and b_offsets[p] .. of B (host arrays, n_pairs+1)
 *   path_a/b  : device int arrays of sum_p (Ta_p + Tb_p) entries; pair p's path starts at
 *               a_offsets[p] + b_offsets[p] and has path_len[p] (device, n_pairs) entries
 *   total     : device, n_pairs doubles or NULL: accumulated cost of the last cell
 * Frames per utterance are limited by the LDS border buffers of the tiled wavefront (7680). */
size_t evc_dtw_workspace_bytes(const int* a_offsets, const int* b_offsets, int n_pairs);
int evc_dtw_align(const void* A, int lda, const int* a_offsets, const void* B, int ldb,
                  const int* b_offsets, int D, int n_pairs, int* path_a, int* path_b, int* path_len,
                  double* total, void* workspace, size_t workspace_bytes, evc_stream_t stream);


/* Gather of the aligned frames - align_sp_ap_f0(), 04_align_n_nmf.py:100-169, and the stacking of the aligned frames
 * into the dictionary, :230-246,320-324,350-361: the dictionary's rows are the frames the DTW paths name, pair after
 * pair.  With these two calls the paths evc_dtw_align left on the device are consumed there: the feature frames make no
 * round trip through the host, only the number of rows N (one int) comes back, because N sizes the dictionary.
 *
 * evc_dtw_path_rows: row_start[p] = first dictionary row of pair p (exclusive scan of path_len), row_start[n_pairs] = N.
 *   path_len   : device, n_pairs ints (evc_dtw_align)          row_start : device, n_pairs + 1 ints
 *   n_rows_out : host int or NULL; non-NULL makes the call synchronous (it returns N)
 * evc_dtw_gather_rows: dst[row_start[p] + k][c] = op(src[src_offsets[p] + path[pair_offsets[p] + k]][c * elem_stride])
 *   for k < path_len[p], c < cols.
 *   src          : frames as rows, row stride ld_src (elements); elem_stride = 2 picks the real parts of an interleaved
 *                  complex matrix (the script's np.abs(real(stft)), :320-324, with op = EVC_GATHER_ABS)
 *   path         : device, path_a or path_b of evc_dtw_align;  pair_offsets : device, n_pairs ints, a_offsets[p] + b_offsets[p]
 *   src_offsets  : device, n_pairs ints, first row of pair p's utterance in src
 *   dst          : N x cols, row stride ld_dst;  dtype: EVC_F64 | EVC_F32 (src and dst alike) */
enum { EVC_GATHER_COPY = 0, EVC_GATHER_ABS = 1 };
int evc_dtw_path_rows(const int* path_len, int n_pairs, int* row_start, int* n_rows_out, evc_stream_t stream);
int evc_dtw_gather_rows(const void* src, long ld_src, int elem_stride, const int* path, const int* path_len,
                        const int* src_offsets, const int* pair_offsets, const int* row_start, int n_pairs, int cols,
                        int op, void* dst, long ld_dst, int dtype, evc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EVC_H */
