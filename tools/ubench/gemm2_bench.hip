// Stand-alone timing harness for the two contractions of the generic path (k_gemm2 / k_gemm_nt), with ablations.
// Build (repo root):  hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DEVC_G2_ABLATE=n] -o tools/ubench/bin/gemm2_bench[_n] tools/ubench/gemm2_bench.hip
// Run on the GPU box: gemm2_bench [T=11008] [N=4096] [M=201] [f32|f64] [reps=20]
// Prints the average time of  V = H Am^T  (plain, split-K as the library chooses) and of the update contraction.
#include "../../exemplars_vc_amd/csrc/evc_gemm.hip"
#include "../../exemplars_vc_amd/csrc/evc_gemm2.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace evc;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename T> int run(int T_, int N, int M, int reps) {
    const int Mk = round_up(M, 16), Mj = round_up(M, 64), Np = round_up(N, 128), Tp = round_up(T_, sizeof(T) == 4 ? 64 : 128);
    printf("T=%d (Tp %d) N=%d (Np %d) M=%d (Mk %d, Mj %d) %s ablate=%d\n", T_, Tp, N, Np, M, Mk, Mj, sizeof(T) == 4 ? "f32" : "f64", EVC_G2_ABLATE);
    T *H, *H1, *P, *Am, *At, *V, *split;
    int *fu, *act;
    const size_t nsplit = (size_t)(Tp <= 2048 ? 32 : 4) * Tp * Mj;
    CK(hipMalloc(&H, sizeof(T) * Tp * Np)); CK(hipMalloc(&H1, sizeof(T) * Tp * Np)); CK(hipMalloc(&P, sizeof(T) * Tp * Np));
    CK(hipMalloc(&Am, sizeof(T) * Mj * Np)); CK(hipMalloc(&At, sizeof(T) * Np * Mk)); CK(hipMalloc(&V, sizeof(T) * Tp * Mj));
    CK(hipMalloc(&split, sizeof(T) * nsplit)); CK(hipMalloc(&fu, sizeof(int) * Tp)); CK(hipMalloc(&act, sizeof(int)));
    std::vector<T> h((size_t)Tp * Np);
    srand(1);
    for (auto& x : h) x = (T)((rand() + 1.0) / (RAND_MAX + 2.0));
    CK(hipMemcpy(H, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(P, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(Am, h.data(), sizeof(T) * Mj * Np, hipMemcpyHostToDevice));
    CK(hipMemcpy(At, h.data(), sizeof(T) * Np * Mk, hipMemcpyHostToDevice));
    CK(hipMemset(fu, 0, sizeof(int) * Tp));
    const int one = 1;
    CK(hipMemcpy(act, &one, sizeof(int), hipMemcpyHostToDevice));
    MuEpilogue<T> ep{};
    ep.Hin = H; ep.P = P; ep.frame_utt = fu; ep.active = act; ep.ldh = Np; ep.N = N; ep.T_ = T_;
    ep.eps_mode = EVC_EPS_ZERO_REPLACE; ep.eps = (T)1.1920929e-7; ep.l1 = 0; ep.kl = 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) CK(hipEventRecord(e0));
        for (int r = 0; r < (pass ? reps : 2); ++r) {
            int sp = 0;
            CK(gemm_nt<T>(H, Np, Am, Np, V, Mj, Tp, Mj, Np, nullptr, split, nsplit, &sp, Mk));
            if (sp) CK(sum_slabs<T>(split, (long)Tp * Mj, sp, V, nullptr));
        }
        if (pass) { CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
    }
    printf("  V = H Am^T (+ slab sum): %8.1f us  -> %6.1f Tflop/s (algorithmic, M bins)\n", ms / reps * 1e3,
           2.0 * T_ * M * (double)N / (ms / reps * 1e-3) / 1e12);
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) CK(hipEventRecord(e0));
        for (int r = 0; r < (pass ? reps : 2); ++r) CK(gemm_nt_mu<T>(V, Mj, At, Mk, H1, Tp, Np, Mk, ep, nullptr));
        if (pass) { CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
    }
#ifdef EVC_G2_STAMP
    {
        long long* d;
        const size_t n = 2048 * 8 * 8;
        CK(hipMalloc(&d, sizeof(long long) * n));
        CK(hipMemset(d, 0, sizeof(long long) * n));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(evc_g2_dbg), &d, sizeof(d)));
        CK(gemm_nt_mu<T>(V, Mj, At, Mk, H1, Tp, Np, Mk, ep, nullptr));
        CK(hipDeviceSynchronize());
        std::vector<long long> st(n);
        CK(hipMemcpy(st.data(), d, sizeof(long long) * n, hipMemcpyDeviceToHost));
        long long* z = nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(evc_g2_dbg), &z, sizeof(z)));
        // per workgroup (wave 0): main-loop length and epilogue length
        double mainl = 0, epi = 0; int cnt = 0;
        for (int l = 0; l < 2048; ++l) {
            const long long* q = &st[(l * 8) * 8];
            if (!q[0]) continue;
            mainl += q[1] - q[0]; epi += q[2] - q[1]; ++cnt;
        }
        printf("  stamps over %d workgroups (wave 0): main loop %.0f ticks, epilogue %.0f ticks\n", cnt, mainl / cnt, epi / cnt);
    }
#endif
    printf("  update contraction     : %8.1f us  -> %6.1f Tflop/s\n", ms / reps * 1e3,
           2.0 * T_ * M * (double)N / (ms / reps * 1e-3) / 1e12);
    CK(hipDeviceSynchronize());
    return 0;
}

int main(int argc, char** argv) {
    const int T_ = argc > 1 ? atoi(argv[1]) : 11008, N = argc > 2 ? atoi(argv[2]) : 4096, M = argc > 3 ? atoi(argv[3]) : 201;
    const bool f64 = argc > 4 && !strcmp(argv[4], "f64");
    const int reps = argc > 5 ? atoi(argv[5]) : 20;
    return f64 ? run<double>(T_, N, M, reps) : run<float>(T_, N, M, reps);
}
