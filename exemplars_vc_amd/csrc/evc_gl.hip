// Griffin-Lim phase reconstruction: the back end that follows the activation solve when
// the scripts run on STFT magnitudes (04_align_n_nmf.py:182-191 -> zz_audio_utilities.py:181-218,
// 258-292).  float64; one utterance or a batch of them per call.
//
// With fft_size = 400 (not a power of two) and a few hundred frames, the transforms are small dense
// contractions, so both directions run on the fp64 matrix cores through k_gemm_nt:
//   S = frames(x) W_f     frames(x)[t][n] = x[hop t + n] is just x read with row stride `hop`
//                         (overlapping rows, no framing pass); W_f = hanning (.) [cos | -sin]
//   x' = overlap_add(P W_i)   W_i = irfft basis (DC/Nyquist weight 1, others 2, /fft_size) (.) hanning
// Between them one small kernel replaces magnitudes (P = mag * exp(j angle(S))), after them one
// gathers the overlap-add (each sample sums its <= fft/hop frames in frame order: deterministic).
//
// Batch: one utterance is 688 frames x 400 samples - 84 output tiles, a third of the CUs, 5 launches of ~20 us per
// iteration.  A batch is laid out as ONE virtual signal: utterance u starts at virtual frame row
// r_u = frame_offsets[u] + u G, G = ceil(fft / hop), i.e. at sample hop r_u, so that the frames of all utterances
// are still rows of one strided matrix (row stride hop).  The G rows between two utterances straddle both signals:
// their magnitudes are taken as zero, so they add nothing to the overlap-add, and the samples between two signals
// stay zero.  Every kernel below then works on the batch as on one long utterance (0.7 % more rows for 688-frame
// utterances), and the contractions have thousands of rows.
#include "evc_internal.h"

namespace evc {

struct GlDims {
    int T_, F, hop, nb;       // frames, fft size, hop, bins = F/2 + 1
    int Tp, K1, J1, K2, J2;   // padded GEMM extents
    long L, Lp;               // signal length T*hop + F, and padded buffer length
};

static GlDims gl_dims(int T_, int F, int hop) {
    GlDims d;
    d.T_ = T_; d.F = F; d.hop = hop; d.nb = F / 2 + 1;
    d.Tp = round_up(T_, 128);
    d.K1 = round_up(F, 16);
    d.J1 = round_up(2 * d.nb, 64);
    d.K2 = round_up(2 * d.nb, 16);
    d.J2 = round_up(F, 64);
    d.L = (long)T_ * hop + F;
    d.Lp = (long)hop * (d.Tp - 1) + d.K1;
    if (d.Lp < d.L) d.Lp = d.L;
    d.Lp = (d.Lp + 15) & ~15L;
    return d;
}

// np.hanning(F)[n] (symmetric: period F-1, zz_audio_utilities.py:196,212) or, periodic, scipy's
// get_window('hann', F) = the window librosa.stft applies (period F)
__device__ __forceinline__ double hanning(int n, int F, bool periodic = false) {
    return F < 2 ? 1.0 : 0.5 - 0.5 * cospi(2.0 * n / (double)(periodic ? F : F - 1));
}

// W_f[j][n] (J1 x K1) and W_i[n][k] (J2 x K2), zero in the padding
__global__ __launch_bounds__(256) void k_gl_tables(GlDims d, double* __restrict__ Wf, double* __restrict__ Wi,
                                                   bool periodic) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long n1 = (long)d.J1 * d.K1, n2 = (long)d.J2 * d.K2;
    if (gid < n1) {
        const int j = (int)(gid / d.K1), n = (int)(gid % d.K1);
        double v = 0.0;
        if (n < d.F && j < 2 * d.nb) {
            const int k = j < d.nb ? j : j - d.nb;
            const double ang = 2.0 * (double)(((long)k * n) % d.F) / (double)d.F;   // exact argument reduction
            v = hanning(n, d.F, periodic) * (j < d.nb ? cospi(ang) : -sinpi(ang));
        }
        Wf[gid] = v;
    } else if (Wi && gid < n1 + n2) {
        const long g = gid - n1;
        const int n = (int)(g / d.K2), kk = (int)(g % d.K2);
        double v = 0.0;
        if (n < d.F && kk < 2 * d.nb) {
            const int k = kk < d.nb ? kk : kk - d.nb;
            const double wk = (k == 0 || k == d.nb - 1) ? 1.0 : 2.0;     // irfft: Hermitian half spectrum
            const double ang = 2.0 * (double)(((long)k * n) % d.F) / (double)d.F;
            v = hanning(n, d.F) * wk / (double)d.F * (kk < d.nb ? cospi(ang) : -sinpi(ang));
        }
        Wi[g] = v;
    }
}

// P = mag * exp(1j * angle(S))   (zz_audio_utilities.py:284-286); S = [Re | Im] per row
// S may arrive as `splits` k-slabs (slab z at S + z * slab) that are summed here, in order
__global__ __launch_bounds__(256) void k_gl_project(const double* __restrict__ S, int lds_, long slab, int splits,
                                                    const double* __restrict__ mag, long ldm, GlDims d,
                                                    const int* __restrict__ row_src, double* __restrict__ P) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)d.Tp * d.K2) return;
    const long t = gid / d.K2;
    const int c = (int)(gid % d.K2);
    double v = 0.0;
    const int src = t < d.T_ ? row_src[t] : -1;       // the row of `mag` behind virtual row t (-1: between utterances)
    if (src >= 0 && c < 2 * d.nb) {
        const int k = c < d.nb ? c : c - d.nb;
        double re = S[t * lds_ + k], im = S[t * lds_ + d.nb + k];
        for (int z = 1; z < splits; ++z) {
            re += S[z * slab + t * lds_ + k];
            im += S[z * slab + t * lds_ + d.nb + k];
        }
        // exp(1j * angle(S)) = S / |S| (angle(0) = 0 -> 1): hypot and a division instead of atan2 + cos / sin, which
        // were 14 % of a batched iteration; agrees with the libm form to ~1e-16 (re = im = 0, -0 included: cos = 1,
        // sin = 0 like numpy's angle(0) = 0; NaN / inf components propagate as NaN like cos / sin of a NaN angle)
        const double r = hypot(re, im);
        const double cs = r > 0.0 ? re / r : (r == 0.0 ? 1.0 : r - r);
        const double sn = r > 0.0 ? im / r : (r == 0.0 ? 0.0 : r - r);
        v = mag[(long)src * ldm + k] * (c < d.nb ? cs : sn);
    }
    P[gid] = v;
}

// x'[s] = sum over the frames f covering s of Fr[f][s - hop f]   (zz_audio_utilities.py:214-217)
__global__ __launch_bounds__(256) void k_gl_overlap_add(const double* __restrict__ Fr, int ldf, long slab, int splits,
                                                        GlDims d, double* __restrict__ xn) {
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= d.Lp) return;
    double acc = 0.0;
    if (s < d.L) {
        long f0 = (s - d.F + d.hop) / d.hop;          // first frame with s - hop f < F
        if (s - d.F + 1 <= 0) f0 = 0;
        long f1 = s / d.hop;                          // last frame with s - hop f >= 0
        if (f1 > d.T_ - 1) f1 = d.T_ - 1;
        for (long f = f0; f <= f1; ++f) {
            double v = Fr[f * ldf + (s - d.hop * f)];
            for (int z = 1; z < splits; ++z) v += Fr[z * slab + f * ldf + (s - d.hop * f)];    // k-slabs, in order
            acc += v;
        }
    }
    xn[s] = acc;                                      // zero beyond the signal: the padded GEMM rows read it
}

// virtual frame row / sample position of utterance u (G gap rows in front of every utterance but the first)
__device__ __forceinline__ long gl_row0(const int* off, int u, int G) { return (long)off[u] + (long)u * G; }

// row_src[r] = row of `mag` behind virtual row r, or -1
__global__ __launch_bounds__(256) void k_gl_rows(const int* __restrict__ off, int n_utt, int G, int Tp,
                                                 int* __restrict__ row_src) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= Tp) return;
    int src = -1;
    for (int u = 0; u < n_utt; ++u) {
        const long r0 = gl_row0(off, u, G);
        if (r >= r0 && r < r0 + (off[u + 1] - off[u])) src = off[u] + (int)(r - r0);
    }
    row_src[r] = src;
}

// rmse[u] = sqrt(sum((x' - x)^2) / L_u) over utterance u's samples (zz_audio_utilities.py:289); one block per
// utterance, fixed order
__global__ __launch_bounds__(256) void k_gl_rmse(const double* __restrict__ xn, const double* __restrict__ xo,
                                                 const int* __restrict__ off, int G, int hop, int F, int iters,
                                                 double* __restrict__ out) {
    __shared__ double red[4];
    const int u = blockIdx.x;
    const long s0 = gl_row0(off, u, G) * hop, L = (long)(off[u + 1] - off[u]) * hop + F;
    double acc = 0.0;
    for (long s = threadIdx.x; s < L; s += 256) {
        const double df = xn[s0 + s] - xo[s0 + s];
        acc += df * df;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[(long)u * iters] = sqrt((red[0] + red[1] + red[2] + red[3]) / (double)L);
}

// caller's concatenated signals (utterance u at sample hop off[u] + u F) <-> the virtual signal; one block row per
// utterance.  to_virtual: xv must have been zeroed (the samples between utterances stay zero).
__global__ __launch_bounds__(256) void k_gl_move_x(double* __restrict__ xc, double* __restrict__ xv,
                                                   const int* __restrict__ off, int G, int hop, int F, bool to_virtual) {
    const int u = blockIdx.y;
    const long L = (long)(off[u + 1] - off[u]) * hop + F;
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= L) return;
    const long c = (long)off[u] * hop + (long)u * F + s, v = gl_row0(off, u, G) * hop + s;
    if (to_virtual) xv[v] = xc[c];
    else xc[c] = xv[v];
}

static int gl_gap(int F, int hop) { return (F + hop - 1) / hop; }
// virtual rows of a batch whose utterances hold T_total frames
static long gl_virtual_rows(long T_total, int n_utt, int F, int hop) { return T_total + (long)(n_utt - 1) * gl_gap(F, hop); }

size_t gl_workspace_bytes(long T_total, int n_utt, int F, int hop, int iters) {
    const long R = gl_virtual_rows(T_total, n_utt, F, hop);
    if (R > (1L << 30)) return 0;
    const GlDims d = gl_dims((int)R, F, hop);
    size_t n = (size_t)d.J1 * d.K1 + (size_t)d.J2 * d.K2       // tables
             + (size_t)d.Tp * d.J1 + (size_t)d.Tp * d.K2 + (size_t)d.Tp * d.J2   // S, P, Fr
             + 2 * (size_t)d.Lp + (size_t)(iters > 0 ? iters : 1) * n_utt        // x ping-pong, rmse trace
             + 8 * (size_t)d.Tp * (d.J1 > d.J2 ? d.J1 : d.J2)                     // split-K slabs of the contractions
             + ((size_t)d.Tp + n_utt + 1 + 1) / 2;                                // row map, offsets (ints)
    return n * sizeof(double) + 12 * 256;
}

// x: the utterances' signals back to back (utterance u: T_u hop + F samples at offset hop off[u] + u F); in = initial
// signals (the reference draws randn), out = reconstructions.  off: host, n_utt + 1 frame offsets into `mag`.
// rmse_host (host, [n_utt][iters], may be NULL): per-iteration sqrt(mean((x_new - x_old)^2)) of every utterance.
__global__ void k_gl_single_offsets(int* off, int T_) {
    off[0] = 0;
    off[1] = T_;
}

hipError_t gl_run(const double* mag, long ldm, const int* off, int n_utt, int F, int hop, int iters, double* x,
                  void* ws, double* rmse_host, hipStream_t s) {
    const int G = gl_gap(F, hop);
    const GlDims d = gl_dims((int)gl_virtual_rows(off[n_utt], n_utt, F, hop), F, hop);
    double* p = static_cast<double*>(ws);
    auto take = [&](size_t n) { double* q = p; p += (n + 31) & ~size_t(31); return q; };
    double* Wf = take((size_t)d.J1 * d.K1);
    double* Wi = take((size_t)d.J2 * d.K2);
    double* S = take((size_t)d.Tp * d.J1);
    double* P = take((size_t)d.Tp * d.K2);
    double* Fr = take((size_t)d.Tp * d.J2);
    double* xa = take((size_t)d.Lp);
    double* xb = take((size_t)d.Lp);
    double* tr = take((size_t)(iters > 0 ? iters : 1) * n_utt);
    const size_t nsplit = 8 * (size_t)d.Tp * (d.J1 > d.J2 ? d.J1 : d.J2);
    double* split = take(nsplit);
    int* doff = reinterpret_cast<int*>(take(((size_t)n_utt + 2) / 2));
    int* row_src = reinterpret_cast<int*>(take(((size_t)d.Tp + 1) / 2));

    // The offsets reach the device by value for a single utterance (evc_griffin_lim hands a stack array: nothing
    // may read it after this function returns); a batch's host array is copied asynchronously - the caller keeps it
    // valid until the call returns (include/evc.h), and HIP stages a pageable source before hipMemcpyAsync returns.
    hipError_t e = hipSuccess;
    if (n_utt == 1) {
        hipLaunchKernelGGL(k_gl_single_offsets, dim3(1), dim3(1), 0, s, doff, off[1]);
        e = hipGetLastError();
    } else {
        e = hipMemcpyAsync(doff, off, sizeof(int) * (n_utt + 1), hipMemcpyHostToDevice, s);
    }
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(xa, 0, sizeof(double) * d.Lp, s);
    if (e != hipSuccess) return e;
    int Tmax = 0;
    for (int u = 0; u < n_utt; ++u) Tmax = off[u + 1] - off[u] > Tmax ? off[u + 1] - off[u] : Tmax;
    const dim3 mv_grid((unsigned)(((long)Tmax * hop + F + 255) / 256), (unsigned)n_utt);
    const long nt = (long)d.J1 * d.K1 + (long)d.J2 * d.K2;
    hipLaunchKernelGGL(k_gl_tables, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, d, Wf, Wi, false);
    hipLaunchKernelGGL(k_gl_rows, dim3((unsigned)((d.Tp + 255) / 256)), dim3(256), 0, s, doff, n_utt, G, d.Tp, row_src);
    hipLaunchKernelGGL(k_gl_move_x, mv_grid, dim3(256), 0, s, x, xa, doff, G, hop, F, true);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    double* xc = xa;
    double* xn = xb;
    for (int it = 0; it < iters; ++it) {
        // S[t][:] = x[hop t : hop t + F] W_f     (rows of L overlap: row stride = hop)
        int sp = 0;
        e = gemm_nt<double>(xc, hop, Wf, d.K1, S, d.J1, d.Tp, d.J1, d.K1, s, split, nsplit, &sp);
        if (e != hipSuccess) return e;
        const long np_ = (long)d.Tp * d.K2;
        hipLaunchKernelGGL(k_gl_project, dim3((unsigned)((np_ + 255) / 256)), dim3(256), 0, s, sp ? split : S, d.J1,
                           (long)d.Tp * d.J1, sp ? sp : 1, mag, ldm, d, row_src, P);
        e = gemm_nt<double>(P, d.K2, Wi, d.K2, Fr, d.J2, d.Tp, d.J2, d.K2, s, split, nsplit, &sp);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_gl_overlap_add, dim3((unsigned)((d.Lp + 255) / 256)), dim3(256), 0, s, sp ? split : Fr,
                           d.J2, (long)d.Tp * d.J2, sp ? sp : 1, d, xn);
        if (rmse_host)
            hipLaunchKernelGGL(k_gl_rmse, dim3((unsigned)n_utt), dim3(256), 0, s, xn, xc, doff, G, hop, F, iters, tr + it);
        double* t = xc; xc = xn; xn = t;
    }
    hipLaunchKernelGGL(k_gl_move_x, mv_grid, dim3(256), 0, s, x, xc, doff, G, hop, F, false);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (rmse_host && iters > 0) {
        e = hipMemcpyAsync(rmse_host, tr, sizeof(double) * iters * n_utt, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

// ---- STFT front end: librosa.core.stft(y, n_fft, hop_length, window='hann') as called at
// 04_align_n_nmf.py:422 and 03_a_b_r_parallel.py:103 - centred frames over the reflect-padded signal,
// periodic Hann window, rfft.  Same contraction as the Griffin-Lim analysis step: S = frames(xp) W_f.

// xp[i] = x[reflect(i - pad)] (numpy 'reflect': the edge sample is not repeated), zero beyond the padded
// signal (the padded GEMM rows read it)
__global__ __launch_bounds__(256) void k_stft_pad(const double* __restrict__ x, long L, int pad, long Lp,
                                                  double* __restrict__ xp) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= Lp) return;
    double v = 0.0;
    if (i < L + 2L * pad) {
        long s = i - pad;
        if (L == 1) s = 0;
        else {
            const long period = 2 * (L - 1);
            s %= period;
            if (s < 0) s += period;
            if (s >= L) s = period - s;
        }
        v = x[s];
    }
    xp[i] = v;
}

// S may arrive as `splits` k-slabs (slab z at S + z * slab) that are summed here, in order
__global__ __launch_bounds__(256) void k_stft_split(const double* __restrict__ S, int lds_, long slab, int splits,
                                                    int T_, int nb, double* __restrict__ re, long ldre,
                                                    double* __restrict__ im, long ldim) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)T_ * nb) return;
    const long t = gid / nb;
    const int k = (int)(gid % nb);
    double r = S[t * lds_ + k], i = S[t * lds_ + nb + k];
    for (int z = 1; z < splits; ++z) {
        r += S[z * slab + t * lds_ + k];
        i += S[z * slab + t * lds_ + nb + k];
    }
    re[t * ldre + k] = r;
    im[t * ldim + k] = i;
}

int stft_frames(long L, int hop, bool center, int F) {
    if (center) return (int)(1 + L / hop);
    return L < F ? 0 : (int)(1 + (L - F) / hop);
}

static GlDims stft_dims(long L, int F, int hop, bool center) {
    GlDims d = gl_dims(stft_frames(L, hop, center, F) > 0 ? stft_frames(L, hop, center, F) : 1, F, hop);
    const long need = L + (center ? 2L * (F / 2) : 0);
    if (d.Lp < need) d.Lp = (need + 15) & ~15L;
    return d;
}

size_t stft_workspace_bytes(long L, int F, int hop, bool center) {
    const GlDims d = stft_dims(L, F, hop, center);
    return ((size_t)d.J1 * d.K1 + 9 * (size_t)d.Tp * d.J1 + (size_t)d.Lp + 128) * sizeof(double) + 4 * 256;
}

hipError_t stft_run(const double* x, long L, int F, int hop, bool center, double* re, long ldre, double* im,
                    long ldim, void* ws, hipStream_t s) {
    const int T_ = stft_frames(L, hop, center, F);
    if (T_ <= 0) return hipSuccess;
    const GlDims d = stft_dims(L, F, hop, center);
    double* p = static_cast<double*>(ws);
    auto take = [&](size_t n) { double* q = p; p += (n + 31) & ~size_t(31); return q; };
    double* Wf = take((size_t)d.J1 * d.K1);
    double* S = take((size_t)d.Tp * d.J1);
    double* xp = take((size_t)d.Lp);
    const size_t nsplit = 8 * (size_t)d.Tp * d.J1;     // split-K slabs: one utterance is 84 output tiles on 256 CUs
    double* split = take(nsplit);
    const long nt = (long)d.J1 * d.K1;
    hipLaunchKernelGGL(k_gl_tables, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, d, Wf, (double*)nullptr, true);
    hipLaunchKernelGGL(k_stft_pad, dim3((unsigned)((d.Lp + 255) / 256)), dim3(256), 0, s, x, L, center ? F / 2 : 0,
                       d.Lp, xp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int sp = 0;
    e = gemm_nt<double>(xp, hop, Wf, d.K1, S, d.J1, d.Tp, d.J1, d.K1, s, split, nsplit, &sp);
    if (e != hipSuccess) return e;
    const long n = (long)T_ * d.nb;
    hipLaunchKernelGGL(k_stft_split, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sp ? split : S, d.J1,
                       (long)d.Tp * d.J1, sp ? sp : 1, T_, d.nb, re, ldre, im, ldim);
    return hipGetLastError();
}

}  // namespace evc
