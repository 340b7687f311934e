"""Crossover between k_fused_wide64 and the two-contraction path at the C3 shape (M = 513, N = 8192, float64) by batch
size.  Run on the GPU box:  python tools/tune_wide64.py [K=60]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import exemplars_vc_amd as evc  # noqa: E402
import torch  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    M, N = 513, 8192
    rng = np.random.default_rng(7)
    A = rng.random((M, N)) + 1e-3
    A /= np.linalg.norm(A, axis=0)
    Ad = torch.from_numpy(A).cuda()
    for U in (2, 4, 6, 8, 12, 16, 32, 64):
        T = 688 * U
        X = A[:, rng.integers(0, N, T)] * rng.random(T) + 1e-6
        Xd = torch.from_numpy(X).cuda()
        offs = np.arange(U + 1, dtype=np.int32) * 688
        row = []
        for kw in (dict(), dict(fused_w=4)):
            best = 1e9
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.time()
                H, info = evc.solve_activations(Ad, Xd, iters=K, eps_mode="zero_replace", init="sklearn", info=True,
                                                utt_offsets=offs, **kw)
                torch.cuda.synchronize()
                best = min(best, time.time() - t0)
            fl = K * (4.0 * M * N + 3.0 * N) * T
            row.append(f"{info['kernel']} {best*1e3:8.2f} ms {fl/best/1e12/78.6:.3f}")
        print(f"{U:3d} utterances ({T} frames): " + "   |   ".join(row), flush=True)


if __name__ == "__main__":
    main()
