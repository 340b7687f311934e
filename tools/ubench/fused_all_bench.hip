// Stand-alone timing harness for k_fused_all (diagnostic build with in-kernel s_memtime stamps).
// Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEVC_ALL_TIMING -o /tmp/fused_all_bench tools/ubench/fused_all_bench.hip
// Run on the GPU box: /tmp/fused_all_bench [N=4096] [rounds=2] [iters=100]
// Prints the kernel time and, for the first round, the mean length (shader cycles) of a sweep step, an
// exchange step and the barrier wait that follows each, per half.
#include "../../exemplars_vc_amd/csrc/evc_fused_all.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace evc;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int bin_of_h(int s, int q) { return 16 * (s >> 2) + q + 4 * (s & 3); }

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096;
    const int rounds = argc > 2 ? atoi(argv[2]) : 2;
    const int iters = argc > 3 ? atoi(argv[3]) : 100;
    const int M = 25, msteps = 7, mtiles = 2, msp = 8;
    const int NT = N / 16, C = NT / 32;
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int pairs = cus / C, groups = 2 * pairs;
    const int TT = groups * rounds, T = TT * 16;
    printf("N=%d NT=%d C=%d cus=%d groups=%d TT=%d iters=%d\n", N, NT, C, cus, groups, TT, iters);
    // a consistent NMF problem: positive dictionary with unit columns, X = A H* (sparse H*), H0 constant
    std::vector<double> A((size_t)M * N), X((size_t)M * T);
    srand(1);
    auto rnd = [] { return (rand() + 1.0) / (RAND_MAX + 2.0); };
    for (int n = 0; n < N; ++n) {
        double nr = 0;
        for (int m = 0; m < M; ++m) { A[(size_t)m * N + n] = rnd() + 1e-3; nr += A[(size_t)m * N + n] * A[(size_t)m * N + n]; }
        nr = 1.0 / sqrt(nr);
        for (int m = 0; m < M; ++m) A[(size_t)m * N + n] *= nr;
    }
    for (int t = 0; t < T; ++t) {
        for (int m = 0; m < M; ++m) X[(size_t)m * T + t] = 1e-6;
        for (int k = 0; k < 8; ++k) {
            const int n = rand() % N; const double hv = rnd();
            for (int m = 0; m < M; ++m) X[(size_t)m * T + t] += A[(size_t)m * N + n] * hv;
        }
    }
    double xm = 0; for (double v : X) xm += v; xm /= X.size();
    const double h0 = sqrt(xm / N);
    std::vector<double> A1p((size_t)NT * msp * 64, 0.0), A2p((size_t)NT * mtiles * 4 * 64, 0.0);
    for (long gid = 0; gid < (long)A1p.size(); ++gid) {
        const int e = gid & 1, l = (gid >> 1) & 63, s = 2 * (int)((gid >> 7) % (msp / 2)) + e;
        const long j = (gid >> 7) / (msp / 2);
        const int i = l & 15, bin = bin_of_h(s, l >> 4);
        const long n = 16 * j + 4 * (i & 3) + (i >> 2);
        A1p[gid] = (s < msteps && bin < M) ? A[(size_t)bin * N + n] : 0.0;
    }
    for (long gq = 0; gq < (long)A2p.size(); ++gq) {
        const int e = gq & 1, l = (gq >> 1) & 63, r = 2 * (int)((gq >> 7) & 1) + e, u = (gq >> 8) % mtiles;
        const long j = (gq >> 8) / mtiles;
        const long n = 16 * j + 4 * (l >> 4) + r;
        const int bin = 16 * u + (l & 15);
        A2p[gq] = bin < M ? A[(size_t)bin * N + n] : 0.0;
    }
    std::vector<double> Xp((size_t)TT * msteps * 64), Vp((size_t)TT * 8 * 64, 0.0), Hp((size_t)TT * NT * 256, h0);
    std::vector<double> rowsum(M, 0.0);
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) rowsum[m] += A[(size_t)m * N + n];
    for (long gid = 0; gid < (long)Xp.size(); ++gid) {
        const int l = gid & 63, s = (gid >> 6) % msteps; const long tt = (gid >> 6) / msteps;
        const int bin = bin_of_h(s, l >> 4);
        Xp[gid] = bin < M ? X[(size_t)bin * T + 16 * tt + (l & 15)] : 0.0;
        Vp[(tt * 8 + s) * 64 + l] = bin < M ? h0 * rowsum[bin] : 0.0;
    }
    std::vector<int> fu(T, 0); int one = 1;
    double *dA1, *dA2, *dX, *dH, *dV, *dbuf; int *dfu, *dact, *dcnt; long long* ddbg;
    CK(hipMalloc(&dA1, A1p.size() * 8)); CK(hipMalloc(&dA2, A2p.size() * 8)); CK(hipMalloc(&dX, Xp.size() * 8));
    CK(hipMalloc(&dH, Hp.size() * 8)); CK(hipMalloc(&dV, Vp.size() * 8)); CK(hipMalloc(&dbuf, (size_t)(ALL_SLICE_OFFSET + ALL_SLICE_ELEMS) * 8));
    CK(hipMalloc(&dfu, T * 4)); CK(hipMalloc(&dact, 4)); CK(hipMalloc(&dcnt, 1024 * 4));
    const size_t ndbg = (size_t)cus * 2 * (2 * iters + 1) * 4;
    CK(hipMalloc(&ddbg, ndbg * 8)); CK(hipMemset(ddbg, 0, ndbg * 8)); CK(hipMemset(dcnt, 0, 1024 * 4));
    CK(hipMemcpy(dA1, A1p.data(), A1p.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dA2, A2p.data(), A2p.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dX, Xp.data(), Xp.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dfu, fu.data(), T * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dact, &one, 4, hipMemcpyHostToDevice));
    FusedArgs a{};
    a.A1p = dA1; a.A2p = dA2; a.Xp = dX; a.Hp = reinterpret_cast<f64x2*>(dH); a.Vp = dV; a.err2 = nullptr;
    a.frame_utt = dfu; a.active = dact; a.NT = NT; a.TT = TT; a.N = N; a.T_ = T; a.iters = iters; a.first = 0;
    a.write_err = 0; a.skip_all_live = 0; a.force_live = 0; a.exact_div = 0; a.loss = EVC_LOSS_FROBENIUS;
    a.eps_mode = EVC_EPS_ZERO_REPLACE; a.eps = 1.1920929e-7; a.l1 = 0.0;
    a.M = M; a.spare_q = -1; a.coop_c = C; a.coop_buf = dbuf; a.coop_cnt = dcnt; a.coop_abort = dcnt + 512; a.groups = 0; a.dbg = ddbg;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemcpy(dH, Hp.data(), Hp.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dV, Vp.data(), Vp.size() * 8, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0, 0));
        CK(fused_all_launch(msteps, a, cus, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    int aborted = 0; CK(hipMemcpy(&aborted, dcnt + 512, 4, hipMemcpyDeviceToHost));
    const double sweeps = (double)TT * iters;       // member-sweeps per member index
    printf("kernel %.3f ms  aborted=%d  -> %.3f us per (round-pair step)  frac_f64_peak=%.4f\n", best, aborted,
           best * 1e3 / (rounds * (2.0 * iters + 1)), (double)iters * (4.0 * M * N + 3.0 * N) * T / (best * 1e-3) / 78.6e12);
    (void)sweeps;
    std::vector<long long> dbg(ndbg);
    CK(hipMemcpy(dbg.data(), ddbg, ndbg * 8, hipMemcpyDeviceToHost));
    const int NS = 2 * iters + 1;
    for (int half = 0; half < 2; ++half) {
        double sw = 0, ex = 0, bs = 0, be = 0; long ns = 0, ne = 0;
        for (int b = 0; b < pairs * C; ++b)
            for (int st = 2; st < NS - 2; ++st) {
                const long long* d = &dbg[(((size_t)b * 2 + half) * NS + st) * 4];
                const bool mine = (st & 1) == half;
                if (mine) { sw += d[1] - d[0]; bs += d[2] - d[1]; ++ns; } else { ex += d[1] - d[0]; be += d[2] - d[1]; ++ne; }
            }
        printf("half %d: sweep %.0f cyc (+barrier wait %.0f)   exchange %.0f cyc (+barrier wait %.0f)\n", half,
               sw / ns, bs / ns, ex / ne, be / ne);
    }
    // distribution over workgroups of the exchange length (half 1)
    {
        const int st = 20; double mn = 1e30, mx = 0;
        for (int b = 0; b < pairs * C; ++b) {
            const long long* d = &dbg[(((size_t)b * 2 + 1) * NS + st) * 4];
            const double v = d[1] - d[0]; if (v < mn) mn = v; if (v > mx) mx = v;
        }
        printf("exchange (half 1, step %d): min %.0f max %.0f cycles over workgroups\n", st, mn, mx);
    }
    // clock: step length in cycles vs microseconds
    {
        const long long* d0 = &dbg[((size_t)0 * NS + 2) * 4];
        const long long* d1 = &dbg[((size_t)0 * NS + NS - 3) * 4];
        printf("workgroup 0: %.0f cycles per step\n", (double)(d1[0] - d0[0]) / (NS - 5));
    }
    return 0;
}
