"""Soak of the task-queue kernels' round-4 scheduling (static schedule, default range counts, stop checks inside a launch)
against the two-contraction path: seeded random shapes and ragged batches of 1 .. 2000 frames, float32 (k_fused_wide,
33 .. 208 bins) and float64 (k_fused_wide64, 145 .. 528 bins), default layout or a forced range count, every eps mode,
stop rule on or off; each case also twice for bitwise repeatability.
    python tools/soak_wide.py [first_seed=100] [count=120]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import exemplars_vc_amd as evc  # noqa: E402
from oracle import evc_oracle as o  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    bad = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        f64 = seed % 3 == 0
        M = int(rng.integers(145, 529)) if f64 else int(rng.integers(33, 209))
        N = int(rng.choice([int(rng.integers(40, 300)), int(rng.integers(300, 1500)), 16 * int(rng.integers(8, 80))]))
        lens = [int(rng.integers(1, 420)) for _ in range(int(rng.integers(1, 6)))]
        T = sum(lens)
        p = o.synth_problem(M, N, T, seed=seed)
        dt = np.float64 if f64 else np.float32
        A, X = np.ascontiguousarray(p["A"].T.astype(dt)), np.ascontiguousarray(p["X"].T.astype(dt))
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        K = int(rng.integers(1, 45))
        kw = dict(layout="frame_major", iters=K, eps_mode=["zero_replace", "add", "clamp", "none"][seed % 4], init="sklearn",
                  utt_offsets=offs, info=True)
        if seed % 2:
            kw.update(check_every=int(rng.integers(2, 6)), stop_rule="sklearn", tol=float(10.0 ** -rng.integers(2, 5)))
        c = int(rng.choice([0, 0, 0, int(rng.integers(1, 12))]))
        force = dict(fused_c=c, fused_w=(3 if M <= 208 else 4) if f64 else 8)
        got, gi = evc.solve_activations(A, X, **force, **kw)
        want, wi = evc.solve_activations(A, X, fused=False, **kw)
        again, _ = evc.solve_activations(A, X, **force, **kw)
        tol = 1e-9 if f64 else 2e-4
        ok = gi["kernel"].startswith("k_fused_wide") and np.array_equal(gi["n_iter"], wi["n_iter"]) and \
            np.allclose(got, want, rtol=tol, atol=tol * 1e-3 * float(np.abs(want).max()))
        same = np.array_equal(got, again)
        if not (ok and same):
            bad += 1
            print(f"seed {seed}: {'f64' if f64 else 'f32'} M={M} N={N} lens={lens} K={K} c={c} kernel={gi['kernel']} "
                  f"n_iter={list(gi['n_iter'])} vs {list(wi['n_iter'])}: close={ok} repeatable={same}", flush=True)
        if (seed - first) % 20 == 19:
            print(f"... {seed - first + 1} cases, {bad} bad", flush=True)
    print("soak:", "FAILED" if bad else "ok", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
