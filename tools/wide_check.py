"""Quick parity sweep of k_fused_wide (float32, 32 < M <= 208) against the oracle: shapes, exemplar-range counts,
wavefronts per workgroup, eps modes, KL, stop rule.  Run on the GPU box:  python tools/wide_check.py"""
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
    cases = [  # M, N, T, K, c, w
        (201, 256, 64, 20, 0, 0), (201, 256, 64, 20, 1, 4), (201, 256, 64, 20, 2, 8), (201, 250, 50, 20, 3, 4),
        (201, 1000, 688, 30, 0, 0), (201, 1000, 688, 30, 6, 4), (201, 1000, 100, 30, 8, 8),
        (40, 300, 100, 25, 0, 0), (64, 512, 130, 25, 4, 0), (100, 512, 130, 25, 5, 0), (150, 200, 33, 25, 0, 8),
        (208, 4096, 688, 20, 0, 0),
    ]
    for (M, N, T, K, c, w) in cases:
        p = o.synth_problem(M, N, T, seed=M + N + T)
        A32, X32 = p["A"].astype(np.float32), p["X"].astype(np.float32)
        want = o.mu_solve(A32.astype(np.float64), X32.astype(np.float64),
                          np.full((N, T), np.sqrt(X32.astype(np.float64).mean() / N)), K, eps_mode=o.EPS_ZERO_REPLACE,
                          eps=float(np.finfo(np.float32).eps), algo="factored")
        t0 = time.time()
        got, info = evc.solve_activations(A32, X32, iters=K, eps_mode="zero_replace", init="sklearn", fused_c=c,
                                          fused_w=w or 4, info=True)
        dt = time.time() - t0
        r = rel(got.astype(np.float64), want)
        ok = r < 2e-3 and info["kernel"] == "k_fused_wide"
        bad += not ok
        print(f"M={M} N={N} T={T} K={K} c={c} w={w}: kernel={info['kernel']} members={info['members']} "
              f"launches={info['launches']} rel={r:.2e} {dt*1e3:.1f} ms {'ok' if ok else 'FAIL'}", flush=True)
    # pymf semantics (given H0, + eps), convert, bin-major / frame-major
    p = o.synth_problem(201, 512, 90, seed=4)
    A32, X32, B32 = (p[k].astype(np.float32) for k in ("A", "X", "B"))
    H0 = (np.random.default_rng(0).random((512, 90)) + 1e-4).astype(np.float32)
    want = o.mu_solve(A32.astype(np.float64), X32.astype(np.float64), H0.astype(np.float64), 30, eps_mode=o.EPS_ADD,
                      eps=1e-9, algo="factored")
    for lay in ("bin_major", "frame_major"):
        tr = (lambda z: z) if lay == "bin_major" else (lambda z: np.ascontiguousarray(z.T))
        H, Y = evc.convert(tr(A32), tr(X32), tr(B32), tr(H0), layout=lay, iters=30, eps_mode="add", fused_w=4)
        H, Y = (H, Y) if lay == "bin_major" else (H.T, Y.T)
        r, ry = rel(H.astype(np.float64), want), rel(Y.astype(np.float64), B32.astype(np.float64) @ want)
        ok = r < 2e-3 and ry < 2e-3
        bad += not ok
        print(f"pymf {lay}: rel H={r:.2e} Y={ry:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    # stop rule + KL on the sklearn surface
    X_rows, W_rows = np.ascontiguousarray(X32.T), np.ascontiguousarray(A32.T)
    act, n_ref, _ = o.sklearn_mu_fixed_dictionary(X_rows.astype(np.float64), W_rows.astype(np.float64), 150, 1e-3)
    H, info = evc.solve_activations(W_rows, X_rows, layout="frame_major", iters=150, eps_mode="zero_replace",
                                    init="sklearn", check_every=10, stop_rule="sklearn", tol=1e-3, info=True, fused_w=4)
    r = rel(H.astype(np.float64), act)
    ok = r < 5e-3 and int(info["n_iter"][0]) == n_ref
    bad += not ok
    print(f"stop rule: n_iter {int(info['n_iter'][0])} vs {n_ref}, rel={r:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    actk, nk, _ = o.sklearn_mu_fixed_dictionary_kl(X_rows.astype(np.float64), W_rows.astype(np.float64), 40, 0.0)
    Hk = evc.solve_activations(W_rows, X_rows, layout="frame_major", iters=40, eps_mode="zero_replace", init="sklearn",
                               loss="kl", fused_w=4)
    r = rel(Hk.astype(np.float64), actk)
    ok = r < 5e-3
    bad += not ok
    print(f"KL: rel={r:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    print("FAILED" if bad else "ALL OK", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
