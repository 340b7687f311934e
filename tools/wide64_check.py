"""Quick parity sweep of k_fused_wide64 (float64, 144 < M <= 528) against the oracle: shapes, exemplar-range counts,
eps modes, given H0, stop rule, synthesis; then the C3 shape timed (one utterance and, with `--batch`, sixteen).
Run on the GPU box:  python tools/wide64_check.py [--batch]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import exemplars_vc_amd as evc  # noqa: E402
from oracle import evc_oracle as o  # noqa: E402


def rel(got, want):
    nz = np.abs(want) > 1e-6 * np.abs(want).max()
    return float(np.max(np.abs(got[nz] - want[nz]) / np.abs(want[nz])))


def main():
    bad = 0
    cases = [  # M, N, T, K, c
        (513, 256, 64, 20, 0), (513, 256, 64, 20, 1), (513, 256, 40, 20, 2), (513, 250, 50, 20, 3),
        (513, 1000, 100, 12, 0), (513, 1000, 100, 12, 6), (257, 300, 70, 15, 0), (400, 512, 33, 15, 5),
        (528, 200, 17, 15, 0), (209, 128, 32, 15, 2), (320, 512, 130, 10, 8),
        (201, 256, 64, 20, 0), (201, 300, 50, 15, 2), (145, 128, 40, 10, 1), (208, 200, 33, 12, 3), (201, 1000, 100, 12, 6),
    ]
    for (M, N, T, K, c) in cases:
        p = o.synth_problem(M, N, T, seed=M + N + T)
        A, X = p["A"], p["X"]
        want = o.mu_solve(A, X, np.full((N, T), np.sqrt(X.mean() / N)), K, eps_mode=o.EPS_ZERO_REPLACE,
                          eps=float(np.finfo(np.float64).eps), algo="factored")
        t0 = time.time()
        got, info = evc.solve_activations(A, X, iters=K, eps_mode="zero_replace", init="sklearn", fused_c=c, fused_w=3 if M <= 208 else 4, info=True)
        dt = time.time() - t0
        r = rel(got, want)
        ok = r < 1e-9 and info["kernel"] == "k_fused_wide64"
        bad += not ok
        print(f"M={M} N={N} T={T} K={K} c={c}: kernel={info['kernel']} members={info['members']} "
              f"launches={info['launches']} rel={r:.2e} {dt*1e3:.1f} ms {'ok' if ok else 'FAIL'}", flush=True)
    # pymf semantics (given H0, + eps), convert, bin-major / frame-major
    p = o.synth_problem(513, 512, 90, seed=4)
    A, X, B = p["A"], p["X"], p["B"]
    H0 = np.random.default_rng(0).random((512, 90)) + 1e-4
    want = o.mu_solve(A, X, H0, 30, eps_mode=o.EPS_ADD, eps=1e-9, algo="factored")
    for lay in ("bin_major", "frame_major"):
        tr = (lambda z: z) if lay == "bin_major" else (lambda z: np.ascontiguousarray(z.T))
        H, Y = evc.convert(tr(A), tr(X), tr(B), tr(H0), layout=lay, iters=30, eps_mode="add", fused_w=4)
        H, Y = (H, Y) if lay == "bin_major" else (H.T, Y.T)
        r, ry = rel(H, want), rel(Y, B @ want)
        ok = r < 1e-9 and ry < 1e-9
        bad += not ok
        print(f"pymf {lay}: rel H={r:.2e} Y={ry:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    # stop rule on the sklearn surface
    X_rows, W_rows = np.ascontiguousarray(X.T), np.ascontiguousarray(A.T)
    act, n_ref, _ = o.sklearn_mu_fixed_dictionary(X_rows, W_rows, 150, 1e-3)
    H, info = evc.solve_activations(W_rows, X_rows, layout="frame_major", iters=150, eps_mode="zero_replace",
                                    init="sklearn", check_every=10, stop_rule="sklearn", tol=1e-3, info=True, fused_w=4)
    r = rel(H, act)
    ok = r < 1e-8 and int(info["n_iter"][0]) == n_ref and info["kernel"] == "k_fused_wide64"
    bad += not ok
    print(f"stop rule: n_iter {int(info['n_iter'][0])} vs {n_ref}, rel={r:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    # two utterances with different lengths, stop rule per utterance
    offs = np.array([0, 37, 90], dtype=np.int32)
    H2, info2 = evc.solve_activations(W_rows, X_rows, layout="frame_major", iters=150, eps_mode="zero_replace",
                                      init="sklearn", check_every=10, stop_rule="sklearn", tol=1e-3, info=True,
                                      utt_offsets=offs, fused_w=4)
    for u in range(2):
        a, b = offs[u], offs[u + 1]
        actu, nu, _ = o.sklearn_mu_fixed_dictionary(X_rows[a:b], W_rows, 150, 1e-3)
        r = rel(H2[a:b], actu)
        ok = r < 1e-8 and int(info2["n_iter"][u]) == nu
        bad += not ok
        print(f"utterance {u}: n_iter {int(info2['n_iter'][u])} vs {nu}, rel={r:.2e} {'ok' if ok else 'FAIL'}", flush=True)

    # C3 shape timed
    import torch
    for U in ([1, 16] if "--batch" in sys.argv else [1]):
        M, N, T, K = 513, 8192, 688 * U, 200
        rng = np.random.default_rng(7)
        A = rng.random((M, N)) + 1e-3
        A /= np.linalg.norm(A, axis=0)
        X = A[:, rng.integers(0, N, T)] * rng.random(T) + 1e-6
        Ad, Xd = torch.from_numpy(A).cuda(), torch.from_numpy(X).cuda()
        offs = np.arange(U + 1, dtype=np.int32) * 688
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.time()
            H, info = evc.solve_activations(Ad, Xd, iters=K, eps_mode="zero_replace", init="sklearn", info=True,
                                            utt_offsets=offs if U > 1 else None, fused_w=4)
            torch.cuda.synchronize()
            dt = time.time() - t0
        fl = K * (4.0 * M * N + 3.0 * N) * T
        print(f"C3 x{U}: {info['kernel']} members={info['members']} {dt*1e3:.2f} ms whole call, "
              f"{fl/dt/1e12:.1f} Tflop/s = {fl/dt/1e12/78.6:.3f} of 78.6", flush=True)
    print("FAILED" if bad else "ALL OK", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
