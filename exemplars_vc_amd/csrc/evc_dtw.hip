// Dynamic-time-warping alignment of parallel utterance pairs: the step that builds the
// parallel exemplar dictionary before the activation solve - _dtw_alignment(), 01_make_dict_parallel.py:
// 215-228, which calls the third-party `dtw` package with a squared-Euclidean frame distance.
//
// One workgroup per utterance pair (the corpus is 162 pairs: one round on the 256 CUs):
//   k_dtw_cost        C[i][j] = sum_d (a[i][d] - b[j][d])^2, summed left to right without FMA contraction,
//                     exactly the arithmetic of `sum(np.square(x - y))`
//   k_dtw_accumulate  D[i][j] = C[i][j] + min(D[i-1][j-1], D[i][j-1], D[i-1][j]) by a TILED wavefront (round 3): the
//                     matrix is cut into 64 x 64 tiles; inside a tile one wavefront walks the 127 cell diagonals with
//                     lane = row, its neighbours' values arriving by DPP lane shifts (no LDS, no barrier); the tiles of
//                     a tile diagonal are independent and share the workgroup's 16 wavefronts; one workgroup barrier per
//                     TILE diagonal (a pair of 700 x 700 frames: 21 instead of 1399) hands the tiles' last rows and
//                     columns on through LDS.  Round 2 swept cell diagonals with a barrier each: 0.93 us per
//                     diagonal.  The local costs live in a tile-diagonal-major layout (tile, diagonal, lane) so that
//                     every step of a wavefront is one coalesced 512-byte read; the accumulated values never go to
//                     memory: what the trace-back needs is WHICH neighbour was the minimum (the package's rule:
//                     argmin over (diagonal, i-1, j-1), first minimum wins), one byte per cell in the same layout,
//                     and it walks those through LDS, one tile at a time.
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
    double* Dm;             // local costs, tile-diagonal-major (k_dtw_cost -> k_dtw_accumulate)
    unsigned char* dir;     // per cell: which neighbour the trace-back moves to (same layout, one byte per slot)
    int* path_a; int* path_b; int* path_len; double* total;
    int D;
};

// One thread per column j and DTW_IC consecutive rows i: a lane's b-row is a strided (uncoalesced) read, so it is
// read once per DTW_IC cells instead of once per cell (the first version: 3.3 ms for the 162-pair corpus, all of it
// waiting for those reads); the DTW_IC rows of `a` sit in LDS (broadcast reads).  Every cell still sums its D
// squares left to right, without FMA contraction.
constexpr int DTW_IC = 16;

// Tile-diagonal-major layout of a pair's Ta x Tb matrix: 64 x 64 tiles, row-major over (I, J); inside a tile the cell
// (il, jl) sits at [il + jl][il]: 127 diagonals of 64 slots (half of them unused: 2 x the cells).
constexpr int DTW_TILE = 127 * 64;
__host__ __device__ inline long dtw_tiles(long Ta, long Tb) { return ((Ta + 63) / 64) * ((Tb + 63) / 64); }

__global__ __launch_bounds__(256) void k_dtw_cost(DtwArgs g) {
#pragma clang fp contract(off)
    extern __shared__ double sA[];          // [DTW_IC][D], then [DTW_IC][257]: the block's costs on their way out
    double* sC = sA + DTW_IC * g.D;
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
    double acc[DTW_IC];
#pragma unroll
    for (int ii = 0; ii < DTW_IC; ++ii) acc[ii] = 0.0;
    if (j < Tb) {
        const double* b = g.B + (long)(g.boff[pair] + j) * g.ldb;
        for (int d = 0; d < D; ++d) {
            const double bv = b[d];
#pragma unroll
            for (int ii = 0; ii < DTW_IC; ++ii) {
                const double df = sA[ii * D + d] - bv;
                const double sq = df * df;
                acc[ii] = acc[ii] + sq;
            }
        }
    }
    // out through LDS into the tile-diagonal-major layout: the block's 16 rows x 256 columns are 16 rows of 4 tiles;
    // on a tile diagonal d they are <= 16 consecutive slots = one 128-byte run (a thread per slot)
#pragma unroll
    for (int ii = 0; ii < DTW_IC; ++ii) sC[ii * 257 + threadIdx.x] = acc[ii];
    __syncthreads();
    const int ntj = (Tb + 63) / 64, I = i0 >> 6, il0 = i0 & 63;
    double* out = g.Dm + g.doff[pair];
    const int ii = threadIdx.x & 15;
    for (int e = threadIdx.x >> 4; e < 4 * (DTW_IC + 63); e += 16) {        // (tile of the block, diagonal offset)
        const int tj = e / (DTW_IC + 63), dd = e - tj * (DTW_IC + 63);        // cell diagonal d = il0 + dd
        const int jl = dd - ii, jj = 64 * tj + jl, jg = blockIdx.x * 256 + jj;
        if (jl >= 0 && jl < 64 && jg < Tb && i0 + ii < Ta)
            out[((long)I * ntj + (jg >> 6)) * DTW_TILE + (long)(il0 + dd) * 64 + il0 + ii] = sC[ii * 257 + jj];
    }
}

constexpr int DTW_THREADS = 1024, DTW_WAVES = DTW_THREADS / 64;

// lane l <- lane l - 1 (DPP wave_shr:1: one VALU move per 32-bit half, no LDS); lane 0 <- `first`
__device__ __forceinline__ double wave_shr1(double v, double first) {
    const long long b = __double_as_longlong(v), f = __double_as_longlong(first);
    const int lo = __builtin_amdgcn_update_dpp((int)f, (int)b, 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(f >> 32), (int)(b >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// the value lane `l` holds, in every lane (l wave-uniform: v_readlane)
__device__ __forceinline__ double bcast_lane(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// local-cost diagonals of a tile in flight: the costs were written a moment ago by k_dtw_cost on other CUs, so a read
// is a round trip to memory (~2 us); with 6 in flight a step took 340 ns whatever it computed
constexpr int DTW_PF = 32;

__global__ __launch_bounds__(DTW_THREADS) void k_dtw_accumulate(DtwArgs g) {
    // LDS: rowbot[ntj][64] last row of the newest tile of every tile column, colright[nti][64] last column of the
    // newest tile of every tile row, corner[3][ntj + 1] bottom-right cells by tile diagonal (mod 3)
    extern __shared__ double dtw_lds[];
    const int pair = blockIdx.x;
    const int Ta = g.aoff[pair + 1] - g.aoff[pair], Tb = g.boff[pair + 1] - g.boff[pair];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long pbase = (long)g.aoff[pair] + g.boff[pair];   // path buffers: capacity Ta + Tb per pair
    if (Ta <= 0 || Tb <= 0) {
        if (tid == 0) { g.path_len[pair] = 0; if (g.total) g.total[pair] = 0.0; }
        return;
    }
    const double* Dm = g.Dm + g.doff[pair];
    unsigned char* dirm = g.dir + g.doff[pair];
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    const int nti = (Ta + 63) / 64, ntj = (Tb + 63) / 64;
    double* rowbot = dtw_lds;
    double* colright = rowbot + (long)ntj * 64;
    double* corner = colright + (long)nti * 64;             // [3][ntj + 1]
    // borders of the package's D0: inf along the first row and column, 0 at the origin
    for (int e = tid; e < ntj * 64; e += DTW_THREADS) rowbot[e] = inf;
    for (int e = tid; e < nti * 64; e += DTW_THREADS) colright[e] = inf;
    for (int e = tid; e < 3 * (ntj + 1); e += DTW_THREADS) corner[e] = inf;
    __syncthreads();
    if (tid == 0) corner[0] = 0.0;                          // tile (0, 0): D0[0][0] = 0   (diagonal 0 reads slot [0 % 3][0])
    __syncthreads();

    for (int kd = 0; kd < nti + ntj - 1; ++kd) {            // tile diagonals
        const int Ilo = kd - (ntj - 1) > 0 ? kd - (ntj - 1) : 0, Ihi = kd < nti - 1 ? kd : nti - 1;
        for (int I = Ilo + w; I <= Ihi; I += DTW_WAVES) {   // (wave-uniform) the tiles of this diagonal
            const int J = kd - I;
            const double* T = Dm + ((long)I * ntj + J) * DTW_TILE;
            unsigned char* Tdir = dirm + ((long)I * ntj + J) * DTW_TILE;
            const int rows = Ta - 64 * I < 64 ? Ta - 64 * I : 64, cols = Tb - 64 * J < 64 ? Tb - 64 * J : 64;
            // this lane's row: left border value (cell (il, -1)) and the diagonal neighbour of its first cell; the row
            // above the tile sits one column per lane (lane 0 takes column d of it at step d: a v_readlane).  The
            // bottom-right cell of the tile up-left of this one was stored two tile diagonals ago under the slot of
            // THIS diagonal ((kd - 2 + 2) % 3).
            const double cl = colright[I * 64 + lane];                       // D[i][64 J - 1]
            const double cl_up = lane ? colright[I * 64 + lane - 1] : corner[(kd % 3) * (ntj + 1) + J];
            const double rbv = rowbot[J * 64 + lane];                        // D[64 I - 1][64 J + lane]
            double rbn = inf, crv = inf;     // this tile's last row (by column = lane) and last column (by row = lane)
            double cur = inf;                // value of this lane's cell of the previous step: (il, jl - 1)
            double upv = inf;                // `up` of the previous step = cell (il - 1, jl - 1): this step's diagonal neighbour
            double cpf[DTW_PF];
#pragma unroll
            for (int k = 0; k < DTW_PF; ++k) cpf[k] = T[(long)k * 64 + lane];
            const int last_d = rows + cols - 2;
            for (int d0 = 0; d0 <= last_d; d0 += DTW_PF) {
#pragma unroll
                for (int k = 0; k < DTW_PF; ++k) {
                    const int d = d0 + k;
                    const double c = cpf[k];
                    const int dn = d + DTW_PF;
                    cpf[k] = dn <= 126 ? T[(long)dn * 64 + lane] : 0.0;
                    const int jl = d - lane;
                    // the row above: lane - 1's cell of the previous step (a DPP wave shift, no LDS); lane 0 takes the
                    // tile above's last row
                    const double up = wave_shr1(cur, bcast_lane(rbv, d < 64 ? d : 63));
                    const bool first = jl == 0;
                    const double lf = first ? cl : cur;                       // D[i][j-1]
                    const double dg = first ? cl_up : upv;                    // D[i-1][j-1]
                    // min(D0[i,j], D0[i+1,j], D0[i,j+1]) and, for the trace-back, WHICH of them: the package's argmin over
                    // (diagonal, i-1, j-1) - the first minimum wins (dtw.py _traceback)
                    double m = dg;
                    unsigned char tb = 0;
                    if (up < m) { m = up; tb = 1; }
                    if (lf < m) { m = lf; tb = 2; }
                    const bool act = d <= last_d && jl >= 0 && jl < cols && lane < rows;
                    const double v = c + m;
                    if (act) {
                        Tdir[(long)d * 64 + lane] = tb;
                        cur = v;
                        if (jl == 63) crv = v;
                    }
                    // lane 63's cell of this step is column d - 63 of the tile's last row
                    if (d >= 63) rbn = lane == d - 63 ? bcast_lane(cur, 63) : rbn;
                    upv = up;
                }
            }
            if (I == nti - 1 && J == ntj - 1 && lane == rows - 1 && g.total) g.total[pair] = cur;    // D[Ta-1][Tb-1]
            if (rows == 64) rowbot[J * 64 + lane] = rbn;                     // (needed by the tile below: full tiles only)
            if (cols == 64) colright[I * 64 + lane] = crv;
            if (rows == 64 && cols == 64 && lane == 63) corner[((kd + 2) % 3) * (ntj + 1) + J + 1] = crv;
        }
        if (kd == 0 && tid == 0) corner[0] = inf;         // the origin's 0 was for tile (0, 0) only (wave 0 has just used it)
        __syncthreads();
    }
    // trace-back (dtw package _traceback): wavefront 0 walks the path backwards into the end of the buffer.  A walk
    // through global memory is a chain of dependent round trips; so the direction bytes of the tile the path is in are
    // brought into LDS once (8 KiB, one coalesced load) and the steps inside the tile read LDS.  Every lane runs the
    // same (uniform) walk; lane 0 writes.
    __shared__ int s_start;
    const int cap = Ta + Tb;
    int* pa = g.path_a + pbase;
    int* pb = g.path_b + pbase;
    if (w == 0) {
        unsigned char* tl = reinterpret_cast<unsigned char*>(dtw_lds);       // (the border buffers are no longer needed)
        int i = Ta - 1, j = Tb - 1, pos = cap - 1;
        if (lane == 0) { pa[pos] = i; pb[pos] = j; }
        while (i > 0 || j > 0) {
            const int I = i >> 6, J = j >> 6;
            const uint4* src = reinterpret_cast<const uint4*>(dirm + ((long)I * ntj + J) * DTW_TILE);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < DTW_TILE / (64 * 16) + 1; ++k) {
                const int e = k * 64 + lane;
                if (e < DTW_TILE / 16) reinterpret_cast<uint4*>(tl)[e] = src[e];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            while ((i >> 6) == I && (j >> 6) == J && (i > 0 || j > 0)) {
                const int il = i & 63, jl = j & 63;
                const int t = *reinterpret_cast<volatile unsigned char*>(tl + (il + jl) * 64 + il);
                if (t == 0) { --i; --j; } else if (t == 1) { --i; } else { --j; }
                --pos;
                if (lane == 0) { pa[pos] = i; pb[pos] = j; }
            }
        }
        if (lane == 0) {
            s_start = pos;
            g.path_len[pair] = cap - pos;
        }
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

// doff[p] = sum over pairs before p of their tile-diagonal-major matrices' sizes (one wavefront; the corpus has 162 pairs)
__global__ void k_dtw_offsets(const int* __restrict__ aoff, const int* __restrict__ boff, long* __restrict__ doff,
                              int n_pairs) {
    if (threadIdx.x != 0) return;
    long acc = 0;
    doff[0] = 0;
    for (int q = 0; q < n_pairs; ++q) {
        acc += dtw_tiles(aoff[q + 1] - aoff[q], boff[q + 1] - boff[q]) * DTW_TILE;
        doff[q + 1] = acc;
    }
}

size_t dtw_workspace_bytes(const int* aoff, const int* boff, int n_pairs) {
    size_t cells = 0;
    for (int p = 0; p < n_pairs; ++p)
        cells += (size_t)dtw_tiles(aoff[p + 1] - aoff[p], boff[p + 1] - boff[p]) * DTW_TILE;
    return cells * (sizeof(double) + 1) + (size_t)(n_pairs + 1) * (2 * sizeof(int) + sizeof(long)) + 2048;
}

// LDS of k_dtw_accumulate: (nti + ntj) x 64 border values + 3 (ntj + 1) corners; both utterances of a pair at the limit
int dtw_max_frames() { return 64 * 120; }

hipError_t dtw_run(const double* A, long lda, const int* aoff, const double* B, long ldb, const int* boff,
                   int D, int n_pairs, int* path_a, int* path_b, int* path_len, double* total, void* ws,
                   hipStream_t s) {
    char* p = static_cast<char*>(ws);
    int* d_aoff = reinterpret_cast<int*>(p); p += (((size_t)(n_pairs + 1) * sizeof(int)) + 255) & ~size_t(255);
    int* d_boff = reinterpret_cast<int*>(p); p += (((size_t)(n_pairs + 1) * sizeof(int)) + 255) & ~size_t(255);
    long* d_doff = reinterpret_cast<long*>(p); p += (((size_t)(n_pairs + 1) * sizeof(long)) + 255) & ~size_t(255);
    double* Dm = reinterpret_cast<double*>(p);
    size_t cells = 0;
    for (int q = 0; q < n_pairs; ++q) cells += (size_t)dtw_tiles(aoff[q + 1] - aoff[q], boff[q + 1] - boff[q]) * DTW_TILE;
    unsigned char* dir = reinterpret_cast<unsigned char*>(Dm + cells);
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
    DtwArgs g{A, lda, B, ldb, d_aoff, d_boff, d_doff, Dm, dir, path_a, path_b, path_len, total, D};
    if (maxcells > 0) {
        const size_t lds_a = (size_t)DTW_IC * (D + 257) * sizeof(double);
        if (lds_a > 48 * 1024) {                                       // (D <= 512 features; the corpus has 25)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dtw_cost), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_a);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k_dtw_cost, dim3((unsigned)((maxTb + 255) / 256), (unsigned)((maxTa + DTW_IC - 1) / DTW_IC),
                                            (unsigned)n_pairs),
                           dim3(256), lds_a, s, g);
    }
    size_t lds = ((size_t)((maxTa + 63) / 64 + (maxTb + 63) / 64) * 64 + 3 * ((size_t)(maxTb + 63) / 64 + 1)) * sizeof(double);
    if (lds < DTW_TILE + 64) lds = DTW_TILE + 64;        // the trace-back's tile of direction bytes
    if (lds > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dtw_accumulate),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_dtw_accumulate, dim3(n_pairs), dim3(DTW_THREADS), lds, s, g);
    return hipGetLastError();
}


// ----------------------------------------------------------------------------------------------------------------
// Gather of the aligned frames (04_align_n_nmf.py:100-169, align_sp_ap_f0): the dictionary rows are the frames the DTW
// paths name, pair after pair.  Round 3 downloaded the paths and gathered with numpy; here the paths never leave the
// device: an exclusive scan of the path lengths gives every pair its first dictionary row, then one kernel copies rows.
// ----------------------------------------------------------------------------------------------------------------
// row_start[p] = sum of path_len[q], q < p; row_start[n_pairs] = N (one workgroup: n_pairs <= 65535)
__global__ __launch_bounds__(1024) void k_path_scan(const int* __restrict__ path_len, int n_pairs, int* __restrict__ row_start) {
    __shared__ int s_part[1024];
    const int tid = threadIdx.x;
    const int per = (n_pairs + 1023) / 1024;
    int sum = 0;
    for (int i = tid * per; i < min(n_pairs, (tid + 1) * per); ++i) sum += path_len[i];
    s_part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? s_part[tid - o] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = tid ? s_part[tid - 1] : 0;
    for (int i = tid * per; i < min(n_pairs, (tid + 1) * per); ++i) {
        row_start[i] = run;
        run += path_len[i];
    }
    if (tid == 1023) row_start[n_pairs] = s_part[1023];
}

// dst[row_start[p] + k][c] = op(src[src_off[p] + path[pair_off[p] + k]][c * elem_stride]), k < path_len[p].
// One workgroup row of 256 threads walks the columns; blockIdx.y strides over the rows of pair blockIdx.z.
template <typename T>
__global__ __launch_bounds__(256) void k_gather_pairs(const T* __restrict__ src, long ld_src, int elem_stride,
                                                      const int* __restrict__ path, const int* __restrict__ path_len,
                                                      const int* __restrict__ src_off, const int* __restrict__ pair_off,
                                                      const int* __restrict__ row_start, int cols, int op,
                                                      T* __restrict__ dst, long ld_dst) {
    const int p = blockIdx.z;
    const int len = path_len[p];
    const long d0 = row_start[p];
    const int* pp = path + pair_off[p];
    const long s0 = src_off[p];
    for (int k = blockIdx.y; k < len; k += gridDim.y) {
        const T* srow = src + (s0 + pp[k]) * ld_src;
        T* drow = dst + (d0 + k) * ld_dst;
        for (int c = blockIdx.x * 256 + threadIdx.x; c < cols; c += gridDim.x * 256) {
            T v = srow[(long)c * elem_stride];
            if (op == 1) v = v < T(0) ? -v : v;          // |.|  (NaN stays NaN, -0 -> 0 like np.abs)
            drow[c] = v;
        }
    }
}

hipError_t dtw_path_scan(const int* path_len, int n_pairs, int* row_start, hipStream_t s) {
    hipLaunchKernelGGL(k_path_scan, dim3(1), dim3(1024), 0, s, path_len, n_pairs, row_start);
    return hipGetLastError();
}

template <typename T>
hipError_t dtw_gather(const T* src, long ld_src, int elem_stride, const int* path, const int* path_len, const int* src_off,
                      const int* pair_off, const int* row_start, int n_pairs, int cols, int op, T* dst, long ld_dst,
                      hipStream_t s) {
    const unsigned gx = (unsigned)((cols + 255) / 256);
    hipLaunchKernelGGL((k_gather_pairs<T>), dim3(gx > 4 ? 4 : gx, 64, (unsigned)n_pairs), dim3(256), 0, s, src, ld_src,
                       elem_stride, path, path_len, src_off, pair_off, row_start, cols, op, dst, ld_dst);
    return hipGetLastError();
}
template hipError_t dtw_gather<double>(const double*, long, int, const int*, const int*, const int*, const int*, const int*,
                                       int, int, int, double*, long, hipStream_t);
template hipError_t dtw_gather<float>(const float*, long, int, const int*, const int*, const int*, const int*, const int*,
                                      int, int, int, float*, long, hipStream_t);

}  // namespace evc
