// Stand-alone timing harness for k_fused_wide64 (diagnostic build with per-task real-time stamps).
// Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEVC_WIDE_STAMP -o tools/ubench/bin/wide64_bench tools/ubench/wide64_bench.hip
// Run on the GPU box: tools/ubench/bin/wide64_bench [utterances=1] [c=0] [tpw=0] [K=50] [M=513] [N=8192]
// Prints the launch time and, per task kind, the mean length in microseconds of each phase:
//   sweep : wait (dependency) | load (V in, first block staged) | blocks | publish (+ drain + barrier) | tail
//   reduce: wait | sum (loads issued, stores issued) | drain | tail
#include "../../exemplars_vc_amd/csrc/evc_wide64.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace evc;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_fill_utt(int* frame_utt, int* active, double* h0, int Tp, int T_, double v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < Tp) frame_utt[i] = i < T_ ? 0 : -1;
    if (i == 0) { active[0] = 1; h0[0] = v; }
}

int main(int argc, char** argv) {
    const int U = argc > 1 ? atoi(argv[1]) : 1, c_req = argc > 2 ? atoi(argv[2]) : 0, w_req = argc > 3 ? atoi(argv[3]) : 0;
    const int K = argc > 4 ? atoi(argv[4]) : 50, M = argc > 5 ? atoi(argv[5]) : 513, N = argc > 6 ? atoi(argv[6]) : 8192;
    const int T = 688 * U, Mk = round_up(M, 16), Np = round_up(N, 128), Tp = round_up(T, 64);
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const Wide64Layout f = wide64_layout(M, N, T, cus, c_req, w_req);
    printf("M=%d N=%d T=%d K=%d: TPW=%d NB=%d TT=%d G=%d c=%d rmode=%d cus=%d\n", M, N, T, K, f.TPW, f.NB, f.TT,
           f.G, f.c, f.rmode, cus);
    std::vector<double> At((size_t)Np * Mk, 0.0), Xt((size_t)Tp * Mk, 0.0);
    srand(1);
    auto rnd = [] { return (double)((rand() + 1.0) / (RAND_MAX + 2.0)); };
    for (int n = 0; n < N; ++n) {
        double nr = 0;
        for (int m = 0; m < M; ++m) { At[(size_t)n * Mk + m] = rnd() + 1e-3; nr += (double)At[(size_t)n * Mk + m] * At[(size_t)n * Mk + m]; }
        for (int m = 0; m < M; ++m) At[(size_t)n * Mk + m] *= (1.0 / sqrt(nr));
    }
    double xm = 0;
    for (int t = 0; t < T; ++t) {
        for (int m = 0; m < M; ++m) Xt[(size_t)t * Mk + m] = 1e-6;
        for (int k = 0; k < 8; ++k) {
            const int n = rand() % N; const double hv = rnd();
            for (int m = 0; m < M; ++m) Xt[(size_t)t * Mk + m] += At[(size_t)n * Mk + m] * hv;
        }
        for (int m = 0; m < M; ++m) xm += Xt[(size_t)t * Mk + m];
    }
    xm /= (double)T * M;
    double *dAt, *dXt;
    CK(hipMalloc(&dAt, At.size() * 8)); CK(hipMalloc(&dXt, Xt.size() * 8));
    CK(hipMemcpy(dAt, At.data(), At.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dXt, Xt.data(), Xt.size() * 8, hipMemcpyHostToDevice));
    Wide64Buffers b{};
    CK(hipMalloc(&b.Aw, f.aw * 8)); CK(hipMalloc(&b.Xw, f.xw * 8)); CK(hipMalloc(&b.Hw, f.hw * 8)); CK(hipMalloc(&b.Pw, f.hw * 8));
    CK(hipMalloc(&b.Vpart, f.vpart * 8)); CK(hipMalloc(&b.Vsum, f.vsum * 8)); CK(hipMalloc(&b.ctl, wide_ctl_words(f) * 4));
    UttState u{};
    CK(hipMalloc(&u.frame_utt, Tp * 4)); CK(hipMalloc(&u.active, 4)); CK(hipMalloc(&u.h0, 8));
    hipLaunchKernelGGL(k_fill_utt, dim3((Tp + 255) / 256), dim3(256), 0, 0, u.frame_utt, u.active, u.h0, Tp, T, sqrt(xm / N));
    CK(wide_pack_dict(f, dAt, dAt, Mk, Np, b.Aw, 0));
    CK(wide_pack_x(f, dXt, Mk, Tp, b.Xw, 0));
    const long per_it = (long)f.G * f.c * (f.rmode ? 2 : 1), tasks = per_it * (K + 1);
    unsigned long long* dbg;
    CK(hipMalloc(&dbg, tasks * 10 * 8));
    CK(hipMemset(dbg, 0, tasks * 10 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(evc_wide64_dbg), &dbg, sizeof(dbg)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(wide_begin(f, b, 0));
        CK(hipEventRecord(e0, 0));
        CK(wide_iterate(f, b, u, N, T, 0, K + 1, EVC_EPS_ZERO_REPLACE, 2.220446049250313e-16, 0.0, 1, cus, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double fl = (double)(K + 1) * (4.0 * M * N + 3.0 * N) * T;
    printf("launch %.3f ms = %.1f us per iteration, %.1f Tflop/s = %.3f of 78.6\n", best, 1e3 * best / (K + 1),
           fl / best / 1e9, fl / best / 1e9 / 78.6);
    std::vector<unsigned long long> h((size_t)tasks * 10);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double acc[2][5] = {{0}}; long cnt[2] = {0, 0};
    double clk_ticks = 0, clk_us = 0;
    for (long t = per_it * 2; t < tasks; ++t) {          // skip the first two iterations
        const unsigned long long* s = &h[(size_t)t * 10];
        const int kind = (int)(s[6] & 1);
        if (!s[0] || !s[5]) continue;
        ++cnt[kind];
        if (kind == 0 && s[3] > s[2]) { clk_ticks += (double)(s[9] - s[8]); clk_us += (s[3] - s[2]) / 100.0; }
        acc[kind][0] += (s[1] ? s[1] : s[0]) - s[0];
        if (kind == 0) { acc[0][1] += s[2] - (s[1] ? s[1] : s[0]); acc[0][2] += s[3] - s[2]; acc[0][3] += s[4] - s[3]; acc[0][4] += s[5] - s[4]; }
        else { acc[1][1] += s[3] - s[1]; acc[1][2] += s[4] - s[3]; acc[1][3] += s[5] - s[4]; }
    }
    if (cnt[0]) printf("sweep  (%ld tasks): wait %.2f | load %.2f | blocks %.2f | publish %.2f | tail %.2f  us\n", cnt[0],
                       acc[0][0] / cnt[0] / 100, acc[0][1] / cnt[0] / 100, acc[0][2] / cnt[0] / 100, acc[0][3] / cnt[0] / 100, acc[0][4] / cnt[0] / 100);
#ifdef EVC_W64_TIMERS
    {
        double t[7] = {0, 0, 0, 0, 0, 0, 0}; long n = 0;
        for (long tq = per_it * 2; tq < tasks; ++tq) {
            const unsigned long long* q = &h[(size_t)tq * 10];
            if (!q[0] || !q[5] || (q[6] > 1000000)) continue;
            t[0] += q[1]; t[1] += q[4]; t[2] += q[6]; t[3] += q[7]; t[4] += q[8]; t[5] += q[2]; t[6] += q[3]; ++n;
        }
        if (n) printf("D section: first 3.75 positions %.0f, next 8 positions %.0f, last 6.25 positions %.0f cycles\n", t[5] / n, t[6] / n, t[0] / n);
        if (n) printf("cycles per step: D (tail part) %.0f | V' first %.0f | barrier %.0f | V' second + update %.0f | barrier %.0f\n", t[0] / n, t[1] / n, t[2] / n, t[3] / n, t[4] / n);
    }
#endif
    if (clk_us > 0) printf("s_memtime ticks per us inside the block loops: %.1f\n", clk_ticks / clk_us);
    if (cnt[1]) printf("reduce (%ld tasks): wait %.2f | sum %.2f | drain %.2f | tail %.2f  us\n", cnt[1], acc[1][0] / cnt[1] / 100,
                       acc[1][1] / cnt[1] / 100, acc[1][2] / cnt[1] / 100, acc[1][3] / cnt[1] / 100);
    // iteration span: first task start to last task end of one iteration in the middle
    const long itm = K / 2;
    unsigned long long lo = ~0ULL, hi = 0;
    for (long t = per_it * itm; t < per_it * (itm + 1); ++t) { if (h[t * 10] && h[t * 10] < lo) lo = h[t * 10]; if (h[t * 10 + 5] > hi) hi = h[t * 10 + 5]; }
    printf("iteration %ld spans %.2f us (first task start to last task end)\n", itm, (hi - lo) / 100.0);
    return 0;
}
