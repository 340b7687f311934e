// Stand-alone timing harness for k_fused_wide (diagnostic build with per-task real-time stamps).
// Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEVC_WIDE_STAMP -o tools/ubench/bin/wide_bench tools/ubench/wide_bench.hip
// Run on the GPU box: tools/ubench/bin/wide_bench [utterances=1] [c=0] [W=0] [K=150] [M=201] [N=4096] [split=0]
// split > 0: iteration 0 in a launch of its own, then launches of `split` iterations (what the stop checks of a call do);
// a per-iteration table follows (start of the first task, mean wait / blocks / publish of the sweep tasks).
// Prints the launch time and, per task kind, the mean length in microseconds of each phase:
//   sweep : wait (dependency) | load (V in, first block staged) | blocks | publish (+ drain + barrier) | tail
//   reduce: wait | sum (loads issued, stores issued) | drain | tail
#include "../../exemplars_vc_amd/csrc/evc_wide.hip"

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
    const int K = argc > 4 ? atoi(argv[4]) : 150, M = argc > 5 ? atoi(argv[5]) : 201, N = argc > 6 ? atoi(argv[6]) : 4096;
    const int split = argc > 7 ? atoi(argv[7]) : 0;
    const int T = 688 * U, Mk = round_up(M, 16), Np = round_up(N, 128), Tp = round_up(T, 64);
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const WideLayout f = wide_layout(M, N, T, cus, c_req, w_req);
    printf("M=%d N=%d T=%d K=%d: MT=%d W=%d NB=%d TT=%d G=%d c=%d rmode=%d cus=%d\n", M, N, T, K, f.MT, f.W, f.NB, f.TT,
           f.G, f.c, f.rmode, cus);
    std::vector<float> At((size_t)Np * Mk, 0.f), Xt((size_t)Tp * Mk, 0.f);
    srand(1);
    auto rnd = [] { return (float)((rand() + 1.0) / (RAND_MAX + 2.0)); };
    for (int n = 0; n < N; ++n) {
        double nr = 0;
        for (int m = 0; m < M; ++m) { At[(size_t)n * Mk + m] = rnd() + 1e-3f; nr += (double)At[(size_t)n * Mk + m] * At[(size_t)n * Mk + m]; }
        for (int m = 0; m < M; ++m) At[(size_t)n * Mk + m] *= (float)(1.0 / sqrt(nr));
    }
    double xm = 0;
    for (int t = 0; t < T; ++t) {
        for (int m = 0; m < M; ++m) Xt[(size_t)t * Mk + m] = 1e-6f;
        for (int k = 0; k < 8; ++k) {
            const int n = rand() % N; const float hv = rnd();
            for (int m = 0; m < M; ++m) Xt[(size_t)t * Mk + m] += At[(size_t)n * Mk + m] * hv;
        }
        for (int m = 0; m < M; ++m) xm += Xt[(size_t)t * Mk + m];
    }
    xm /= (double)T * M;
    float *dAt, *dXt;
    CK(hipMalloc(&dAt, At.size() * 4)); CK(hipMalloc(&dXt, Xt.size() * 4));
    CK(hipMemcpy(dAt, At.data(), At.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dXt, Xt.data(), Xt.size() * 4, hipMemcpyHostToDevice));
    WideBuffers b{};
    CK(hipMalloc(&b.Aw, f.aw * 4)); CK(hipMalloc(&b.Xw, f.xw * 4)); CK(hipMalloc(&b.Hw, f.hw * 4)); CK(hipMalloc(&b.Pw, f.hw * 4));
    CK(hipMalloc(&b.Vpart, f.vpart * 4)); CK(hipMalloc(&b.Vsum, f.vsum * 4)); CK(hipMalloc(&b.ctl, wide_ctl_words(f) * 4));
    UttState u{};
    CK(hipMalloc(&u.frame_utt, Tp * 4)); CK(hipMalloc(&u.active, 4)); CK(hipMalloc(&u.h0, 8));
    hipLaunchKernelGGL(k_fill_utt, dim3((Tp + 255) / 256), dim3(256), 0, 0, u.frame_utt, u.active, u.h0, Tp, T, sqrt(xm / N));
    CK(wide_pack_dict(f, dAt, dAt, Mk, Np, b.Aw, 0));
    CK(wide_pack_x(f, dXt, Mk, Tp, b.Xw, 0));
    const long per_it = (long)f.G * f.c * (f.rmode ? 2 : 1), tasks = per_it * (K + 1);
    unsigned long long* dbg;
    CK(hipMalloc(&dbg, tasks * 8 * 8));
    CK(hipMemset(dbg, 0, tasks * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(evc_wide_dbg), &dbg, sizeof(dbg)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < (getenv("WB_REPS") ? atoi(getenv("WB_REPS")) : 4); ++rep) {
        CK(wide_begin(f, b, 0));
        CK(hipEventRecord(e0, 0));
        if (split <= 0) {
            CK(wide_iterate(f, b, u, N, T, 0, K + 1, EVC_EPS_ZERO_REPLACE, 1.1920929e-7, 0.0, 1, cus, 0));
        } else {
            // (the stamps are indexed by the ticket of a launch: each launch gets its own part of the buffer)
            for (int it = 0; it < K + 1;) {
                const int n = it == 0 ? 1 : (K + 1 - it < split ? K + 1 - it : split);
                unsigned long long* part = dbg + (size_t)per_it * it * 8;
                CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(evc_wide_dbg), &part, sizeof(part), 0, hipMemcpyHostToDevice, 0));
                CK(wide_iterate(f, b, u, N, T, it, it + n, EVC_EPS_ZERO_REPLACE, 1.1920929e-7, 0.0, 1, cus, 0));
                it += n;
            }
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        if (getenv("WB_VERBOSE")) printf("rep %d: %.3f ms\n", rep, ms);
    }
    const double fl = (double)K * (4.0 * M * N + 3.0 * N) * T;
    printf("launch %.3f ms = %.1f us per iteration, %.1f Tflop/s = %.3f of 157.3\n", best, 1e3 * best / (K + 1),
           fl / best / 1e9, fl / best / 1e9 / 157.3);
    std::vector<unsigned long long> h((size_t)tasks * 8);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double acc[2][5] = {{0}}; long cnt[2] = {0, 0};
    for (long t = per_it * 2; t < tasks; ++t) {          // skip the first two iterations
        const unsigned long long* s = &h[(size_t)t * 8];
        const int kind = (int)(s[6] & 1);
        if (!s[0] || !s[5]) continue;
        ++cnt[kind];
        acc[kind][0] += (s[1] ? s[1] : s[0]) - s[0];
        if (kind == 0) { acc[0][1] += s[2] - (s[1] ? s[1] : s[0]); acc[0][2] += s[3] - s[2]; acc[0][3] += s[4] - s[3]; acc[0][4] += s[5] - s[4]; }
        else { acc[1][1] += s[3] - s[1]; acc[1][2] += s[4] - s[3]; acc[1][3] += s[5] - s[4]; }
    }
    if (cnt[0]) printf("sweep  (%ld tasks): wait %.2f | load %.2f | blocks %.2f | publish %.2f | tail %.2f  us\n", cnt[0],
                       acc[0][0] / cnt[0] / 100, acc[0][1] / cnt[0] / 100, acc[0][2] / cnt[0] / 100, acc[0][3] / cnt[0] / 100, acc[0][4] / cnt[0] / 100);
    if (cnt[1]) printf("reduce (%ld tasks): wait %.2f | sum %.2f | drain %.2f | tail %.2f  us\n", cnt[1], acc[1][0] / cnt[1] / 100,
                       acc[1][1] / cnt[1] / 100, acc[1][2] / cnt[1] / 100, acc[1][3] / cnt[1] / 100);
    // iteration span: first task start to last task end of one iteration in the middle
    const long itm = K / 2;
    unsigned long long lo = ~0ULL, hi = 0;
    for (long t = per_it * itm; t < per_it * (itm + 1); ++t) { if (h[t * 8] && h[t * 8] < lo) lo = h[t * 8]; if (h[t * 8 + 5] > hi) hi = h[t * 8 + 5]; }
    printf("iteration %ld spans %.2f us (first task start to last task end)\n", itm, (hi - lo) / 100.0);
    if (argc > 8) {
        unsigned long long t0 = ~0ULL;
        for (long t = 0; t < per_it; ++t) if (h[t * 8] && h[t * 8] < t0) t0 = h[t * 8];
        printf("it | first start us | last end us | sweep: wait load blocks publish | max blocks | reduce: wait\n");
        for (long it = 0; it <= K; ++it) {
            unsigned long long a0 = ~0ULL, a1 = 0;
            double sw[4] = {0, 0, 0, 0}, rw = 0, mb = 0; long ns = 0, nr = 0;
            for (long t = per_it * it; t < per_it * (it + 1); ++t) {
                const unsigned long long* x = &h[(size_t)t * 8];
                if (!x[0] || !x[5]) continue;
                if (x[0] < a0) a0 = x[0];
                if (x[5] > a1) a1 = x[5];
                const unsigned long long x1 = x[1] ? x[1] : x[0];
                if ((x[6] & 1) == 0) { ++ns; sw[0] += x1 - x[0]; sw[1] += x[2] - x1; sw[2] += x[3] - x[2]; sw[3] += x[4] - x[3]; if (x[3] - x[2] > mb) mb = x[3] - x[2]; }
                else { ++nr; rw += x1 - x[0]; }
            }
            if (!ns) continue;
            printf("%3ld %9.1f %9.1f | %7.1f %6.1f %7.1f %6.1f | %7.1f | %7.1f\n", it, (a0 - t0) / 100.0, (a1 - t0) / 100.0, sw[0] / ns / 100,
                   sw[1] / ns / 100, sw[2] / ns / 100, sw[3] / ns / 100, mb / 100, nr ? rw / nr / 100 : 0.0);
        }
    }
    return 0;
}
