// Griffin-Lim phase reconstruction: the back end that follows the activation solve when
// the scripts run on STFT magnitudes (04_align_n_nmf.py:182-191 -> zz_audio_utilities.py:181-218,
// 258-292).  float64, one utterance per call.
//
// With fft_size = 400 (not a power of two) and a few hundred frames, the transforms are small dense
// contractions, so both directions run on the fp64 matrix cores through k_gemm_nt:
//   S = frames(x) W_f     frames(x)[t][n] = x[hop t + n] is just x read with row stride `hop`
//                         (overlapping rows, no framing pass); W_f = hanning (.) [cos | -sin]
//   x' = overlap_add(P W_i)   W_i = irfft basis (DC/Nyquist weight 1, others 2, /fft_size) (.) hanning
// Between them one small kernel replaces magnitudes (P = mag * exp(j angle(S))), after them one
// gathers the overlap-add (each sample sums its <= fft/hop frames in frame order: deterministic).
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
                                                    double* __restrict__ P) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)d.Tp * d.K2) return;
    const long t = gid / d.K2;
    const int c = (int)(gid % d.K2);
    double v = 0.0;
    if (t < d.T_ && c < 2 * d.nb) {
        const int k = c < d.nb ? c : c - d.nb;
        double re = S[t * lds_ + k], im = S[t * lds_ + d.nb + k];
        for (int z = 1; z < splits; ++z) {
            re += S[z * slab + t * lds_ + k];
            im += S[z * slab + t * lds_ + d.nb + k];
        }
        const double ang = atan2(im, re);
        v = mag[t * ldm + k] * (c < d.nb ? cos(ang) : sin(ang));
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

// rmse = sqrt(sum((x' - x)^2) / L)   (zz_audio_utilities.py:289); single block, fixed order
__global__ __launch_bounds__(256) void k_gl_rmse(const double* __restrict__ xn, const double* __restrict__ xo, long L,
                                                 double* __restrict__ out) {
    __shared__ double red[4];
    double acc = 0.0;
    for (long s = threadIdx.x; s < L; s += 256) {
        const double df = xn[s] - xo[s];
        acc += df * df;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = sqrt((red[0] + red[1] + red[2] + red[3]) / (double)L);
}

__global__ __launch_bounds__(256) void k_gl_load_x(const double* __restrict__ x0, long L, long Lp, double* __restrict__ xb) {
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s < Lp) xb[s] = s < L ? x0[s] : 0.0;
}

size_t gl_workspace_bytes(int T_, int F, int hop, int iters) {
    const GlDims d = gl_dims(T_, F, hop);
    size_t n = (size_t)d.J1 * d.K1 + (size_t)d.J2 * d.K2       // tables
             + (size_t)d.Tp * d.J1 + (size_t)d.Tp * d.K2 + (size_t)d.Tp * d.J2   // S, P, Fr
             + 2 * (size_t)d.Lp + (size_t)(iters > 0 ? iters : 1)                 // x ping-pong, rmse trace
             + 8 * (size_t)d.Tp * (d.J1 > d.J2 ? d.J1 : d.J2);                    // split-K slabs of the contractions
    return n * sizeof(double) + 10 * 256;
}

// x: in = initial signal (the reference draws randn), out = reconstruction; length T*hop + F.
// rmse_dev (device, iters doubles, may be NULL): per-iteration sqrt(mean((x_new - x_old)^2)).
hipError_t gl_run(const double* mag, long ldm, int T_, int F, int hop, int iters, double* x, void* ws,
                  double* rmse_host, hipStream_t s) {
    const GlDims d = gl_dims(T_, F, hop);
    double* p = static_cast<double*>(ws);
    auto take = [&](size_t n) { double* q = p; p += (n + 31) & ~size_t(31); return q; };
    double* Wf = take((size_t)d.J1 * d.K1);
    double* Wi = take((size_t)d.J2 * d.K2);
    double* S = take((size_t)d.Tp * d.J1);
    double* P = take((size_t)d.Tp * d.K2);
    double* Fr = take((size_t)d.Tp * d.J2);
    double* xa = take((size_t)d.Lp);
    double* xb = take((size_t)d.Lp);
    double* tr = take((size_t)(iters > 0 ? iters : 1));
    const size_t nsplit = 8 * (size_t)d.Tp * (d.J1 > d.J2 ? d.J1 : d.J2);
    double* split = take(nsplit);

    const long nt = (long)d.J1 * d.K1 + (long)d.J2 * d.K2;
    hipLaunchKernelGGL(k_gl_tables, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, d, Wf, Wi, false);
    hipLaunchKernelGGL(k_gl_load_x, dim3((unsigned)((d.Lp + 255) / 256)), dim3(256), 0, s, x, d.L, d.Lp, xa);
    hipError_t e = hipGetLastError();
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
                           (long)d.Tp * d.J1, sp ? sp : 1, mag, ldm, d, P);
        e = gemm_nt<double>(P, d.K2, Wi, d.K2, Fr, d.J2, d.Tp, d.J2, d.K2, s, split, nsplit, &sp);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_gl_overlap_add, dim3((unsigned)((d.Lp + 255) / 256)), dim3(256), 0, s, sp ? split : Fr,
                           d.J2, (long)d.Tp * d.J2, sp ? sp : 1, d, xn);
        if (rmse_host) hipLaunchKernelGGL(k_gl_rmse, dim3(1), dim3(256), 0, s, xn, xc, d.L, tr + it);
        double* t = xc; xc = xn; xn = t;
    }
    e = hipMemcpyAsync(x, xc, sizeof(double) * d.L, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
    if (rmse_host && iters > 0) {
        e = hipMemcpyAsync(rmse_host, tr, sizeof(double) * iters, hipMemcpyDeviceToHost, s);
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

__global__ __launch_bounds__(256) void k_stft_split(const double* __restrict__ S, int lds_, int T_, int nb,
                                                    double* __restrict__ re, long ldre, double* __restrict__ im,
                                                    long ldim) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)T_ * nb) return;
    const long t = gid / nb;
    const int k = (int)(gid % nb);
    re[t * ldre + k] = S[t * lds_ + k];
    im[t * ldim + k] = S[t * lds_ + nb + k];
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
    return ((size_t)d.J1 * d.K1 + (size_t)d.Tp * d.J1 + (size_t)d.Lp + 96) * sizeof(double) + 4 * 256;
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
    const long nt = (long)d.J1 * d.K1;
    hipLaunchKernelGGL(k_gl_tables, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, d, Wf, (double*)nullptr, true);
    hipLaunchKernelGGL(k_stft_pad, dim3((unsigned)((d.Lp + 255) / 256)), dim3(256), 0, s, x, L, center ? F / 2 : 0,
                       d.Lp, xp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = gemm_nt<double>(xp, hop, Wf, d.K1, S, d.J1, d.Tp, d.J1, d.K1, s);
    if (e != hipSuccess) return e;
    const long n = (long)T_ * d.nb;
    hipLaunchKernelGGL(k_stft_split, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, S, d.J1, T_, d.nb, re, ldre,
                       im, ldim);
    return hipGetLastError();
}

}  // namespace evc
