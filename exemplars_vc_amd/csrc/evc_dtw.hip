// Dynamic-time-warping alignment of parallel utterance pairs: the step that builds the
// parallel exemplar dictionary before the activation solve - _dtw_alignment(), 01_make_dict_parallel.py:
// 215-228, which calls the third-party `dtw` package with a squared-Euclidean frame distance.
//
// One workgroup per utterance pair (the corpus is 162 pairs: one round on the 256 CUs):
//   k_dtw_cost        C[i][j] = sum_d (a[i][d] - b[j][d])^2, summed left to right without FMA contraction,
//                     exactly the arithmetic of `sum(np.square(x - y))`
//   k_dtw_accumulate  D[i][j] = C[i][j] + min(D[i-1][j-1], D[i][j-1], D[i-1][j]) by anti-diagonal wavefront
//                     (cells of one anti-diagonal are independent); the three live diagonals are kept in
//                     LDS, the full matrix goes to HBM for the trace-back, which follows the package's
//                     rule: argmin over (diagonal, i-1, j-1), first minimum wins.
// Integer/index work: results are bit-exact against the restated algorithm (oracle.dtw_align); PARITY
// with the package itself is UNPINNED (not installable here, no alignment fixture in the reference).
#include "evc_internal.h"

namespace evc {

struct DtwArgs {
    const double* A; long lda;
    const double* B; long ldb;
    const int* aoff;        // [n_pairs+1] frame offsets into A
    const int* boff;        // [n_pairs+1]
    const long* doff;       // [n_pairs+1] element offsets of the accumulated-cost matrices
    double* Dm;
    int* path_a; int* path_b; int* path_len; double* total;
    int D;
};

// One thread per column j and DTW_IC consecutive rows i: a lane's b-row is a strided (uncoalesced) read, so it is
// read once per DTW_IC cells instead of once per cell (the first version: 3.3 ms for the 162-pair corpus, all of it
// waiting for those reads); the DTW_IC rows of `a` sit in LDS (broadcast reads).  Every cell still sums its D
// squares left to right, without FMA contraction.
constexpr int DTW_IC = 16;

__global__ __launch_bounds__(256) void k_dtw_cost(DtwArgs g) {
#pragma clang fp contract(off)
    extern __shared__ double sA[];          // [DTW_IC][D]
    const int pair = blockIdx.z;
    const int Ta = g.aoff[pair + 1] - g.aoff[pair], Tb = g.boff[pair + 1] - g.boff[pair];
    const int i0 = blockIdx.y * DTW_IC;
    if (i0 >= Ta || (long)blockIdx.x * 256 >= Tb) return;              // (uniform)
    const int D = g.D;
    for (int e = threadIdx.x; e < DTW_IC * D; e += 256) {
        const int ii = e / D, d = e - ii * D;
        sA[e] = i0 + ii < Ta ? g.A[(long)(g.aoff[pair] + i0 + ii) * g.lda + d] : 0.0;
    }
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Tb) return;
    const double* b = g.B + (long)(g.boff[pair] + j) * g.ldb;
    double acc[DTW_IC];
#pragma unroll
    for (int ii = 0; ii < DTW_IC; ++ii) acc[ii] = 0.0;
    for (int d = 0; d < D; ++d) {
        const double bv = b[d];
#pragma unroll
        for (int ii = 0; ii < DTW_IC; ++ii) {
            const double df = sA[ii * D + d] - bv;
            const double sq = df * df;
            acc[ii] = acc[ii] + sq;
        }
    }
    double* out = g.Dm + g.doff[pair] + (long)i0 * Tb + j;
#pragma unroll
    for (int ii = 0; ii < DTW_IC; ++ii)
        if (i0 + ii < Ta) out[(long)ii * Tb] = acc[ii];
}

constexpr int DTW_THREADS = 1024;

__global__ __launch_bounds__(DTW_THREADS) void k_dtw_accumulate(DtwArgs g) {
    extern __shared__ double diag[];        // 3 x (Ta + 1): rolling anti-diagonals, indexed by i + 1
    const int pair = blockIdx.x;
    const int Ta = g.aoff[pair + 1] - g.aoff[pair], Tb = g.boff[pair + 1] - g.boff[pair];
    const int tid = threadIdx.x;
    const long pbase = (long)g.aoff[pair] + g.boff[pair];   // path buffers: capacity Ta + Tb per pair
    if (Ta <= 0 || Tb <= 0) {
        if (tid == 0) { g.path_len[pair] = 0; if (g.total) g.total[pair] = 0.0; }
        return;
    }
    double* Dm = g.Dm + g.doff[pair];
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    const int W = Ta + 1;
    double* d0 = diag;            // diagonal k
    double* d1 = diag + W;        // diagonal k-1
    double* d2 = diag + 2 * W;    // diagonal k-2
    for (int i = tid; i < 3 * W; i += DTW_THREADS) diag[i] = inf;
    __syncthreads();
    // The local cost of diagonal k + 1 is requested while diagonal k is worked on and the workgroup meets at its
    // barrier: the chain of Ta + Tb - 1 diagonals then carries LDS and barrier latency only, not a global-memory
    // round trip each (1.0 us per diagonal before).  One cell per thread and pass; `cn` holds the next pass-0 cost.
    auto cost_at = [&](int k, int p) {        // local cost of this thread's cell on diagonal k, pass p (0 if none)
        const int ilo = k - (Tb - 1) > 0 ? k - (Tb - 1) : 0;
        const int ihi = k < Ta - 1 ? k : Ta - 1;
        const int i = ilo + tid + p * DTW_THREADS;
        return (k <= Ta + Tb - 2 && i <= ihi) ? Dm[(long)i * Tb + (k - i)] : 0.0;
    };
    constexpr int PD = 8;                     // diagonals whose costs are in flight (a request takes ~1 us)
    double cn[PD];
#pragma unroll
    for (int d = 0; d < PD; ++d) cn[d] = cost_at(d, 0);
    auto diagonal = [&](int k, double c0) {
        const int ilo = k - (Tb - 1) > 0 ? k - (Tb - 1) : 0;
        const int ihi = k < Ta - 1 ? k : Ta - 1;
        int p = 0;
        for (int i = ilo + tid; i <= ihi; i += DTW_THREADS, ++p) {
            const int j = k - i;
            // slot i+1 of a diagonal holds D[i][.]; slot 0 is the inf border; D0[0][0] = 0 for the first cell
            const double dg = (i == 0 && j == 0) ? 0.0 : ((i > 0 && j > 0) ? d2[i] : inf);
            const double lf = j > 0 ? d1[i + 1] : inf;      // D[i][j-1]
            const double up = i > 0 ? d1[i] : inf;          // D[i-1][j]
            double m = dg < lf ? dg : lf;                   // min(D0[i,j], D0[i+1,j], D0[i,j+1])
            m = m < up ? m : up;
            const double v = (p == 0 ? c0 : Dm[(long)i * Tb + j]) + m;      // (further passes: utterances > 1024 frames)
            d0[i + 1] = v;
            Dm[(long)i * Tb + j] = v;
        }
        // the diagonals live in LDS: the barrier waits for LDS traffic only.  (__syncthreads() also waits for the
        // global store above and the cost requests in flight - a memory round trip per diagonal.)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        double* t = d2; d2 = d1; d1 = d0; d0 = t;           // rotate: the oldest diagonal is overwritten next
        // cells of the recycled buffer outside the next diagonal's range must read as "no cell": the
        // range test in dg/lf/up above already guards them, so no clearing is needed
    };
    const int last = Ta + Tb - 2;
    int k0 = 0;
    for (; k0 + PD - 1 <= last; k0 += PD) {           // whole groups of PD diagonals: ring slots are fixed registers
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const double c0 = cn[d];
            cn[d] = cost_at(k0 + d + PD, 0);
            diagonal(k0 + d, c0);
        }
    }
#pragma unroll
    for (int d = 0; d < PD; ++d)                      // the last, partial group
        if (k0 + d <= last) diagonal(k0 + d, cn[d]);
    __syncthreads();      // every store of the accumulated matrix has completed
    // trace-back (dtw package _traceback): one lane walks the path backwards into the end of the buffer
    __shared__ int s_start;
    const int cap = Ta + Tb;
    int* pa = g.path_a + pbase;
    int* pb = g.path_b + pbase;
    if (tid == 0) {
        int i = Ta - 1, j = Tb - 1, pos = cap - 1;
        pa[pos] = i; pb[pos] = j;
        while (i > 0 || j > 0) {
            const double dg = (i > 0 && j > 0) ? Dm[(long)(i - 1) * Tb + (j - 1)] : inf;   // D0[i, j]
            const double up = i > 0 ? Dm[(long)(i - 1) * Tb + j] : inf;                     // D0[i, j+1]
            const double lf = j > 0 ? Dm[(long)i * Tb + (j - 1)] : inf;                     // D0[i+1, j]
            int tb = 0;
            double m = dg;
            if (up < m) { m = up; tb = 1; }
            if (lf < m) { m = lf; tb = 2; }
            if (tb == 0) { --i; --j; } else if (tb == 1) { --i; } else { --j; }
            --pos;
            pa[pos] = i; pb[pos] = j;
        }
        s_start = pos;
        g.path_len[pair] = cap - pos;
        if (g.total) g.total[pair] = Dm[(long)(Ta - 1) * Tb + (Tb - 1)];
    }
    __syncthreads();
    // move the path to the front of its buffer (ranges overlap: chunked, read - barrier - write)
    const int start = s_start, len = cap - start;
    for (int c0 = 0; c0 < len; c0 += DTW_THREADS) {
        const int e = c0 + tid;
        int va = 0, vb = 0;
        if (e < len) { va = pa[start + e]; vb = pb[start + e]; }
        __syncthreads();
        if (e < len) { pa[e] = va; pb[e] = vb; }
        __syncthreads();
    }
}

// doff[p] = sum over pairs before p of Ta * Tb (one wavefront; the corpus has 162 pairs)
__global__ void k_dtw_offsets(const int* __restrict__ aoff, const int* __restrict__ boff, long* __restrict__ doff,
                              int n_pairs) {
    if (threadIdx.x != 0) return;
    long acc = 0;
    doff[0] = 0;
    for (int q = 0; q < n_pairs; ++q) {
        acc += (long)(aoff[q + 1] - aoff[q]) * (long)(boff[q + 1] - boff[q]);
        doff[q + 1] = acc;
    }
}

size_t dtw_workspace_bytes(const int* aoff, const int* boff, int n_pairs) {
    size_t cells = 0;
    for (int p = 0; p < n_pairs; ++p)
        cells += (size_t)(aoff[p + 1] - aoff[p]) * (size_t)(boff[p + 1] - boff[p]);
    return cells * sizeof(double) + (size_t)(n_pairs + 1) * (2 * sizeof(int) + sizeof(long)) + 1024;
}

int dtw_max_frames() { return (160 * 1024 - 64) / (3 * (int)sizeof(double)) - 1; }   // LDS: 3 diagonals

hipError_t dtw_run(const double* A, long lda, const int* aoff, const double* B, long ldb, const int* boff,
                   int D, int n_pairs, int* path_a, int* path_b, int* path_len, double* total, void* ws,
                   hipStream_t s) {
    char* p = static_cast<char*>(ws);
    int* d_aoff = reinterpret_cast<int*>(p); p += (((size_t)(n_pairs + 1) * sizeof(int)) + 255) & ~size_t(255);
    int* d_boff = reinterpret_cast<int*>(p); p += (((size_t)(n_pairs + 1) * sizeof(int)) + 255) & ~size_t(255);
    long* d_doff = reinterpret_cast<long*>(p); p += (((size_t)(n_pairs + 1) * sizeof(long)) + 255) & ~size_t(255);
    double* Dm = reinterpret_cast<double*>(p);
    // launch geometry from the caller's offsets; the element offsets of the per-pair matrices are a prefix sum
    // formed on the device (no host temporary, so no allocation and no synchronisation here)
    long maxcells = 0;
    int maxTa = 0, maxTb = 0;
    for (int q = 0; q < n_pairs; ++q) {
        const long Ta = aoff[q + 1] - aoff[q], Tb = boff[q + 1] - boff[q];
        if (Ta * Tb > maxcells) maxcells = Ta * Tb;
        if (Ta > maxTa) maxTa = (int)Ta;
        if (Tb > maxTb) maxTb = (int)Tb;
    }
    hipError_t e = hipMemcpyAsync(d_aoff, aoff, sizeof(int) * (n_pairs + 1), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_boff, boff, sizeof(int) * (n_pairs + 1), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_dtw_offsets, dim3(1), dim3(64), 0, s, d_aoff, d_boff, d_doff, n_pairs);
    DtwArgs g{A, lda, B, ldb, d_aoff, d_boff, d_doff, Dm, path_a, path_b, path_len, total, D};
    if (maxcells > 0) {
        const size_t lds_a = (size_t)DTW_IC * D * sizeof(double);
        if (lds_a > 64 * 1024) return hipErrorInvalidValue;          // (D <= 512 features; the corpus has 25)
        hipLaunchKernelGGL(k_dtw_cost, dim3((unsigned)((maxTb + 255) / 256), (unsigned)((maxTa + DTW_IC - 1) / DTW_IC),
                                            (unsigned)n_pairs),
                           dim3(256), lds_a, s, g);
    }
    const size_t lds = (size_t)3 * (maxTa + 1) * sizeof(double);
    if (lds > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dtw_accumulate),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_dtw_accumulate, dim3(n_pairs), dim3(DTW_THREADS), lds, s, g);
    return hipGetLastError();
}

}  // namespace evc
