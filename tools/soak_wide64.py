"""Soak of k_fused_wide64 against the two-contraction path: seeded random shapes, ragged batches, every instance, exemplar
ranges, eps modes, both layouts (the differential test of tests/test_gpu_wide64.py over many more seeds).
    python tools/soak_wide64.py [first_seed=100] [count=200]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import exemplars_vc_amd as evc  # noqa: E402
from oracle import evc_oracle as o  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    bad = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        M = int(rng.choice([int(rng.integers(209, 529)), 64 * int(rng.integers(4, 9)) + int(rng.integers(0, 17)), 513, 257]))
        M = min(max(M, 209), 528)
        N = int(rng.choice([int(rng.integers(40, 300)), int(rng.integers(300, 2000)), 16 * int(rng.integers(8, 100))]))
        lens = [int(rng.integers(1, 260)) for _ in range(int(rng.integers(1, 7)))]
        T = sum(lens)
        p = o.synth_problem(M, N, T, seed=seed)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        K = int(rng.integers(1, 25))
        layout = "frame_major" if seed % 2 else "bin_major"
        tr = (lambda a: np.ascontiguousarray(a.T)) if layout == "frame_major" else (lambda a: np.ascontiguousarray(a))
        kw = dict(layout=layout, iters=K, eps_mode=["zero_replace", "add", "clamp", "none"][seed % 4],
                  init=["sklearn", "const"][seed % 2], utt_offsets=offs, exact_div=bool(seed % 3 == 0))
        if kw["init"] == "const":
            kw["init_value"] = 0.21
        if seed % 5 == 1:
            kw.update(l1=0.03)
        c = int(rng.integers(0, 12))
        got = evc.convert(tr(p["A"]), tr(p["X"]), tr(p["B"]), fused_c=c, fused_w=4, **kw)
        want = evc.convert(tr(p["A"]), tr(p["X"]), tr(p["B"]), fused=False, **kw)
        ok = all(np.allclose(g, w, rtol=1e-9, atol=1e-12 * float(np.abs(w).max())) for g, w in zip(got[:2], want[:2]))
        again = evc.convert(tr(p["A"]), tr(p["X"]), tr(p["B"]), fused_c=c, fused_w=4, **kw)
        same = all(np.array_equal(a, b) for a, b in zip(got[:2], again[:2]))
        if not (ok and same):
            bad += 1
            print(f"seed {seed}: M={M} N={N} lens={lens} K={K} c={c} {layout}: close={ok} repeatable={same}", flush=True)
        if (seed - first) % 25 == 24:
            print(f"... {seed - first + 1} cases, {bad} bad", flush=True)
    print("soak:", "FAILED" if bad else "ok", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
