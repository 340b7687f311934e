// Layout import/export, per-utterance initialisation and the two stopping rules.
//
// None of these kernels is on the critical path (they run once per call or once per
// check); they exist so that the hot kernels only ever see zero-padded, frames-as-rows
// workspace arrays, and so that the reference's per-call semantics (sklearn's constant
// initialisation and stop test, pymf's stop test) hold per utterance inside one batch.
#include "evc_internal.h"

namespace evc {

// ------------------------------------------------------------------------------------------
// copy2d: logical matrix (r, c); 32x32 tiles through LDS so that both the read and the write
// are coalesced whichever side is transposed.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_copy2d(const T* __restrict__ src, long src_ld, int src_rows,
                                                int src_cols, int src_trans, T* __restrict__ dst,
                                                long dst_ld, int dst_rows, int dst_cols,
                                                int dst_trans) {
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int tiles_x = (dst_cols + 31) / 32;
    const long r0 = (long)(blockIdx.x / tiles_x) * 32, c0 = (long)(blockIdx.x % tiles_x) * 32;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int a = ty + 8 * e;
        // the fast thread index walks the contiguous direction of the source
        const long r = src_trans ? r0 + tx : r0 + a;
        const long c = src_trans ? c0 + a : c0 + tx;
        T v = T(0);
        if (r < src_rows && c < src_cols) v = src_trans ? src[c * src_ld + r] : src[r * src_ld + c];
        tile[r - r0][c - c0] = v;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int a = ty + 8 * e;
        const long r = dst_trans ? r0 + tx : r0 + a;
        const long c = dst_trans ? c0 + a : c0 + tx;
        if (r < dst_rows && c < dst_cols) {
            const T v = tile[r - r0][c - c0];
            if (dst_trans) dst[c * dst_ld + r] = v; else dst[r * dst_ld + c] = v;
        }
    }
}

template <typename T>
hipError_t copy2d(const T* src, long src_ld, int src_rows, int src_cols, int src_trans, T* dst,
                  long dst_ld, int dst_rows, int dst_cols, int dst_trans, hipStream_t s) {
    if (dst_rows <= 0 || dst_cols <= 0) return hipSuccess;
    const long tiles = (long)((dst_cols + 31) / 32) * ((dst_rows + 31) / 32);
    if (tiles > 0x7fffffffL) return hipErrorInvalidValue;
    dim3 grid((unsigned)tiles), block(256);
    hipLaunchKernelGGL((k_copy2d<T>), grid, block, 0, s, src, src_ld, src_rows, src_cols, src_trans,
                       dst, dst_ld, dst_rows, dst_cols, dst_trans);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// utterance bookkeeping
// ------------------------------------------------------------------------------------------
__global__ void k_utt_setup(UttState u, int n_utt, int T_, int Tp, int iters) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int stride = gridDim.x * blockDim.x;
    for (int t = gid; t < Tp; t += stride) {
        int id = -1;
        if (t < T_) {  // largest id with offsets[id] <= t
            int lo = 0, hi = n_utt - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (u.offsets[mid] <= t) lo = mid; else hi = mid - 1;
            }
            id = lo;
        }
        u.frame_utt[t] = id;
    }
    for (int i = gid; i < n_utt; i += stride) {
        u.active[i] = (u.offsets[i + 1] > u.offsets[i]) ? 1 : 0;
        u.n_iter[i] = iters;
        u.err_init[i] = 0.0;
        u.err_prev[i] = 0.0;
        u.h0[i] = 0.0;
    }
    if (gid == 0) {                   // how many utterances are active: the gate of the generic path's kernels
        int n = 0;
        for (int i = 0; i < n_utt; ++i) n += (u.offsets[i + 1] > u.offsets[i]) ? 1 : 0;
        u.active[n_utt] = n;
    }
    const double nan = __longlong_as_double(0x7ff8000000000000ULL);
    for (long i = gid; i < (long)n_utt * u.n_slots; i += stride) u.trace[i] = nan;
}

__global__ void k_utt_single(UttState u, int T_) {
    u.offsets[0] = 0;
    u.offsets[1] = T_;
}
hipError_t utt_single(const UttState& u, int T_, hipStream_t s) {
    hipLaunchKernelGGL(k_utt_single, dim3(1), dim3(1), 0, s, u, T_);
    return hipGetLastError();
}

hipError_t utt_setup(const UttState& u, int n_utt, int T_, int Tp, int iters, hipStream_t s) {
    hipLaunchKernelGGL(k_utt_setup, dim3(64), dim3(256), 0, s, u, n_utt, T_, Tp, iters);
    return hipGetLastError();
}

// fixed-order block reduction of one double per thread (256 threads): wave shuffles, then LDS
__device__ __forceinline__ double block_sum_256(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    double tot = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return tot;
}

// h0[u] = sqrt(mean(X_u) / N)   (sklearn _nmf.py:1228-1231); one block per utterance
template <typename T>
__global__ __launch_bounds__(1024) void k_utt_sklearn_h0(const T* __restrict__ Xt, int ldx, int M,
                                                         int N, UttState u) {
    // one block of 16 wavefronts per utterance; every thread keeps 4 independent partial sums (a lone
    // utterance is a latency problem: 17 200 values), combined in a fixed order
    __shared__ double red[16];
    const int id = blockIdx.x;
    const long t0 = u.offsets[id], t1 = u.offsets[id + 1];
    const long cnt = (t1 - t0) * M;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (long e0 = threadIdx.x; e0 < cnt; e0 += 4096) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const long e = e0 + 1024L * v;
            if (e < cnt) acc[v] += (double)Xt[(t0 + e / M) * ldx + (int)(e % M)];
        }
    }
    double a = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < 16; ++w) tot += red[w];
        u.h0[id] = cnt > 0 ? sqrt(tot / (double)cnt / (double)N) : 0.0;
    }
}

template <typename T>
hipError_t utt_sklearn_h0(const T* Xt, int ldx, int M, int N, const UttState& u, int n_utt,
                          hipStream_t s) {
    hipLaunchKernelGGL((k_utt_sklearn_h0<T>), dim3(n_utt), dim3(1024), 0, s, Xt, ldx, M, N, u);
    return hipGetLastError();
}

__global__ void k_utt_const_h0(UttState u, int n_utt, double v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_utt) u.h0[i] = v;
}
hipError_t utt_const_h0(const UttState& u, int n_utt, double v, hipStream_t s) {
    hipLaunchKernelGGL(k_utt_const_h0, dim3((n_utt + 255) / 256), dim3(256), 0, s, u, n_utt, v);
    return hipGetLastError();
}

template <typename T>
__global__ __launch_bounds__(256) void k_fill_h0(T* __restrict__ Ht, int ldh, int N, int T_,
                                                 UttState u) {
    const long t = blockIdx.x;
    const int id = u.frame_utt[t];
    const T v = (id >= 0 && t < T_) ? (T)u.h0[id] : T(0);
    for (int n = blockIdx.y * 256 + threadIdx.x; n < ldh; n += gridDim.y * 256)
        Ht[t * ldh + n] = (n < N) ? v : T(0);
}

template <typename T>
hipError_t fill_h0(T* Ht, int ldh, int Tp, int N, int T_, const UttState& u, hipStream_t s) {
    dim3 grid(Tp, min(8, (ldh + 255) / 256));
    hipLaunchKernelGGL((k_fill_h0<T>), grid, dim3(256), 0, s, Ht, ldh, N, T_, u);
    return hipGetLastError();
}

// err2[t] = sum_m (X[t][m] - V[t][m])^2 ; one wavefront per frame, shuffle reduction
template <typename T>
__global__ __launch_bounds__(256) void k_frame_err2(const T* __restrict__ Xt, int ldx,
                                                    const T* __restrict__ Vt, int ldv, int M, int T_,
                                                    double* __restrict__ err2) {
    const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T_) return;
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int m = lane; m < M; m += 64) {
        const double d = (double)Xt[t * ldx + m] - (double)Vt[t * ldv + m];
        acc += d * d;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) err2[t] = acc;
}

template <typename T>
hipError_t frame_err2(const T* Xt, int ldx, const T* Vt, int ldv, int M, int T_, double* err2,
                      hipStream_t s) {
    if (T_ <= 0) return hipSuccess;
    hipLaunchKernelGGL((k_frame_err2<T>), dim3((T_ + 3) / 4), dim3(256), 0, s, Xt, ldx, Vt, ldv, M,
                       T_, err2);
    return hipGetLastError();
}

// ---- generalised Kullback-Leibler variant (sklearn _nmf.py:556-606, 136-160) ----
// Akl[n][m] = At[n][m] / colsum_n, colsum_n == 0 -> eps (the constant denominator of the KL update,
// folded into the dictionary once per call); one wavefront per exemplar
template <typename T>
__global__ __launch_bounds__(256) void k_kl_scale_dict(const T* __restrict__ At, int ld, int M, int rows,
                                                       T eps, T* __restrict__ Akl) {
    const long n = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= rows) return;
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int m = lane; m < M; m += 64) acc += (double)At[n * ld + m];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    T den = (T)acc;
    den = (den == T(0)) ? eps : den;
    for (int m = lane; m < ld; m += 64) Akl[n * ld + m] = (m < M) ? At[n * ld + m] / den : T(0);
}
template <typename T>
hipError_t kl_scale_dict(const T* At, int ld, int M, int rows, double eps, T* Akl, hipStream_t s) {
    hipLaunchKernelGGL((k_kl_scale_dict<T>), dim3((rows + 3) / 4), dim3(256), 0, s, At, ld, M, rows, (T)eps, Akl);
    return hipGetLastError();
}

// R[t][m] = X[t][m] / max(V[t][m], eps)   (m < M, zero in the padding)
template <typename T>
__global__ __launch_bounds__(256) void k_kl_ratio(const T* __restrict__ Xt, int ldx, const T* __restrict__ Vt,
                                                  int ldv, int M, long Tp, T eps, T* __restrict__ Rt, int ldr) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= Tp * ldr) return;
    const long t = gid / ldr;
    const int m = (int)(gid % ldr);
    T r = T(0);
    if (m < M) {
        T v = Vt[t * ldv + m];
        v = (v < eps) ? eps : v;
        r = Xt[t * ldx + m] / v;
    }
    Rt[gid] = r;
}
template <typename T>
hipError_t kl_ratio(const T* Xt, int ldx, const T* Vt, int ldv, int M, long Tp, double eps, T* Rt, int ldr,
                    hipStream_t s) {
    const long n = Tp * ldr;
    hipLaunchKernelGGL((k_kl_ratio<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Xt, ldx, Vt, ldv, M,
                       Tp, (T)eps, Rt, ldr);
    return hipGetLastError();
}

// err2[t] = 2 * ( sum_{m: x > eps} (x log(x / max(v, eps)) - x) + sum_m v ): the per-frame share of
// 2 KL(X || A H), so that sqrt(max(sum_t err2, 0)) is sklearn's _beta_divergence(beta=1, square_root=True)
template <typename T>
__global__ __launch_bounds__(256) void k_frame_err_kl(const T* __restrict__ Xt, int ldx, const T* __restrict__ Vt,
                                                      int ldv, int M, int T_, double eps, double* __restrict__ err2) {
    const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T_) return;
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int m = lane; m < M; m += 64) {
        const double x = (double)Xt[t * ldx + m], v = (double)Vt[t * ldv + m];
        acc += v;
        if (x > eps) acc += x * log(x / (v < eps ? eps : v)) - x;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) err2[t] = 2.0 * acc;
}
template <typename T>
hipError_t frame_err_kl(const T* Xt, int ldx, const T* Vt, int ldv, int M, int T_, double eps, double* err2,
                        hipStream_t s) {
    if (T_ <= 0) return hipSuccess;
    hipLaunchKernelGGL((k_frame_err_kl<T>), dim3((T_ + 3) / 4), dim3(256), 0, s, Xt, ldx, Vt, ldv, M, T_, eps, err2);
    return hipGetLastError();
}

// One block per utterance: err = sqrt(sum_t err2[t]); record it; apply the stopping rule.
//   sklearn _nmf.py:871-884:  (previous_error - error) / error_at_init < tol  -> stop
//   pymf base.py:189-206,266-270: for the third and later errors,
//                                 |ferr[i] - ferr[i-1]| / num_samples < eps  -> stop
__global__ __launch_bounds__(256) void k_utt_check(const double* __restrict__ err2, UttState u,
                                                   int c, int check_every, int stop_rule,
                                                   double tol) {
    __shared__ double red[4];
    const int id = blockIdx.x;
    if (!u.active[id]) return;   // uniform per block
    const long t0 = u.offsets[id], t1 = u.offsets[id + 1];
    double acc = 0.0;
    for (long t = t0 + threadIdx.x; t < t1; t += 256) acc += err2[t];
    const double err = sqrt(fmax(block_sum_256(acc, red), 0.0));   // (KL: rounding can leave a tiny negative sum)
    if (threadIdx.x != 0) return;
    u.trace[(long)id * u.n_slots + c] = err;
    if (c == 0) {
        u.err_init[id] = err;
        u.err_prev[id] = err;
        return;
    }
    const double prev = u.err_prev[id];
    bool stop = false;
    if (stop_rule == EVC_STOP_SKLEARN) {
        stop = (prev - err) / u.err_init[id] < tol;
    } else if (stop_rule == EVC_STOP_PYMF) {
        stop = (c >= 3) && (fabs(err - prev) / (double)(t1 - t0) < tol);
    }
    if (stop) {
        u.active[id] = 0;
        atomicSub(&u.active[gridDim.x], 1);
        u.n_iter[id] = c * check_every;
    } else {
        u.err_prev[id] = err;
    }
}

hipError_t utt_check(const double* err2, const UttState& u, int n_utt, int c, int check_every,
                     int stop_rule, double tol, hipStream_t s) {
    hipLaunchKernelGGL(k_utt_check, dim3(n_utt), dim3(256), 0, s, err2, u, c, check_every, stop_rule,
                       tol);
    return hipGetLastError();
}

// element-type conversion of a row-major matrix (rows x cols, leading dimensions in elements)
template <typename S, typename D>
__global__ __launch_bounds__(256) void k_cvt2d(const S* __restrict__ src, long lds_, long rows, long cols,
                                               D* __restrict__ dst, long ldd) {
    const long n = rows * cols;
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < n; g += (long)gridDim.x * 256) {
        const long r = g / cols, c = g % cols;
        dst[r * ldd + c] = (D)src[r * lds_ + c];
    }
}
template <typename S, typename D>
hipError_t cvt2d(const S* src, long lds_, long rows, long cols, D* dst, long ldd, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return hipSuccess;
    const long n = rows * cols;
    const unsigned grid = (unsigned)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    hipLaunchKernelGGL((k_cvt2d<S, D>), dim3(grid), dim3(256), 0, s, src, lds_, rows, cols, dst, ldd);
    return hipGetLastError();
}
template hipError_t cvt2d<float, double>(const float*, long, long, long, double*, long, hipStream_t);
template hipError_t cvt2d<double, float>(const double*, long, long, long, float*, long, hipStream_t);

#define EVC_INST(T)                                                                                 \
    template hipError_t copy2d<T>(const T*, long, int, int, int, T*, long, int, int, int, hipStream_t); \
    template hipError_t utt_sklearn_h0<T>(const T*, int, int, int, const UttState&, int, hipStream_t); \
    template hipError_t fill_h0<T>(T*, int, int, int, int, const UttState&, hipStream_t);           \
    template hipError_t frame_err2<T>(const T*, int, const T*, int, int, int, double*, hipStream_t);        \
    template hipError_t kl_scale_dict<T>(const T*, int, int, int, double, T*, hipStream_t);                \
    template hipError_t kl_ratio<T>(const T*, int, const T*, int, int, long, double, T*, int, hipStream_t); \
    template hipError_t frame_err_kl<T>(const T*, int, const T*, int, int, int, double, double*, hipStream_t);
EVC_INST(double)
EVC_INST(float)

}  // namespace evc
