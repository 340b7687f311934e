#!/usr/bin/env python3
"""The script's default call on its default flow (|Re STFT|, float32, M = 201, N = 4096: tol = 1e-4, a stop test every 10
iterations, <= 150) against the same solve without tests, k_fused_wide, by batch size.  GPU box:
    python tools/bench_default_call_stft.py [utterances ...]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exemplars_vc_amd as evc

M, N, Tu = 201, 4096, 688
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(3)
A = torch.rand(N, M, generator=g, device=dev, dtype=torch.float64) + 1e-3
A /= A.norm(dim=1, keepdim=True)
for U in [int(a) for a in sys.argv[1:]] or [8, 16, 64]:
    Hs = torch.rand(U * Tu, N, generator=g, device=dev, dtype=torch.float64)
    Hs *= (torch.rand(U * Tu, N, generator=g, device=dev, dtype=torch.float64) < 8.0 / N)
    X = (Hs @ A + 1e-6).float().contiguous(); del Hs
    A32 = A.float()
    offs = np.arange(U + 1, dtype=np.int32) * Tu
    H = torch.empty(U * Tu, N, dtype=torch.float32, device=dev)
    row = {"utterances": U}
    for tol in (0.0, 1e-4, 1e-3):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _, info = evc.solve_activations(A32, X, layout="frame_major", iters=150, eps_mode="zero_replace", init="sklearn",
                                            utt_offsets=offs, out=H, check_every=10 if tol > 0 else 0,
                                            stop_rule="sklearn" if tol > 0 else "none", tol=tol, info=True)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        row[f"tol_{tol:g}"] = {"ms": dt * 1e3, "kernel": info["kernel"], "launches": info["launches"],
                               "n_iter_mean": float(info["n_iter"].mean())}
    row["with_tests_over_without"] = row["tol_0"]["ms"] / row["tol_0.0001"]["ms"]
    print(json.dumps(row), flush=True)
