#!/usr/bin/env python3
"""Where the fused task-queue kernels beat the two contractions: k_fused_wide (float32, M = 64, 201)
and k_fused_wide64 (float64, M = 257, 513) against k_gemm2 / k_gemm_nt over N in {512 ... 16384} and 1 ... 64 utterances
of 688 frames.  GPU box:

    python tools/tune_routing.py [K=20] [f32|f64] [quick] [M,M,...] > gpurun_out/r04/tune_routing.jsonl

One JSON line per (dtype, M, N, utterances): milliseconds and fraction of the matrix peak of both routes, the faster one,
and what the library's default routing picks."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import exemplars_vc_amd as evc  # noqa: E402
import torch  # noqa: E402

PEAK = {"f32": 157.3e12, "f64": 78.6e12}


def timed(Ad, Xd, K, offs, **kw):
    best, info = 1e9, None
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, info = evc.solve_activations(Ad, Xd, iters=K, eps_mode="zero_replace", init="sklearn", info=True,
                                        layout="frame_major", utt_offsets=offs, **kw)
        torch.cuda.synchronize()
        if rep:
            best = min(best, time.perf_counter() - t0)
    return best, info


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    quick = len(sys.argv) > 3 and sys.argv[3] == "quick"      # fewer N and batch sizes (long K)
    only_m = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else None      # e.g. 201 with only = f64
    rng = np.random.default_rng(7)
    for dt, Ms in (("f32", (64, 201)), ("f64", (257, 513))):
        if only and only != dt:
            continue
        npdt = np.float32 if dt == "f32" else np.float64
        for M in (only_m or Ms):
            for N in ((1024, 4096, 16384) if quick else (512, 1024, 4096, 8192, 16384)):
                A = rng.random((N, M)) + 1e-3
                A /= np.linalg.norm(A, axis=1, keepdims=True)
                Ad = torch.from_numpy(A.astype(npdt)).cuda()
                Xall = (A[rng.integers(0, N, 688 * 64)] * rng.random((688 * 64, 1)) + 1e-6).astype(npdt)
                Xd_all = torch.from_numpy(Xall).cuda()
                for U in ((1, 2, 3, 4, 6, 8, 12, 16, 32, 64) if quick else (1, 2, 4, 6, 8, 12, 16, 24, 32, 64)):
                    T = 688 * U
                    if dt == "f64" and N * T * 8 > 6e9:       # H alone beyond 6 GB: skip the corner
                        continue
                    offs = np.arange(U + 1, dtype=np.int32) * 688
                    Xd = Xd_all[:T]
                    tg, ig = timed(Ad, Xd, K, offs, fused=False)
                    tf, if_ = timed(Ad, Xd, K, offs, fused_w=(3 if M <= 208 else 4) if dt == "f64" else 8)
                    _, idef = timed(Ad, Xd, 1, offs)
                    fl = K * (4.0 * M * N + 3.0 * N) * T
                    print(json.dumps({"dtype": dt, "M": M, "N": N, "utterances": U, "frame_tiles": (T + 15) // 16, "K": K,
                                      "two_contractions": {"kernel": ig["kernel"], "ms": tg * 1e3, "frac": fl / tg / PEAK[dt]},
                                      "fused": {"kernel": if_["kernel"], "members": int(if_["members"]), "ms": tf * 1e3,
                                                "frac": fl / tf / PEAK[dt]},
                                      "faster": "fused" if tf < tg else "two_contractions",
                                      "default_route": idef["kernel"]}), flush=True)
                del Ad, Xd_all


if __name__ == "__main__":
    main()
