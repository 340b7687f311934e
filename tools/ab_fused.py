#!/usr/bin/env python3
"""A/B timing of the fused-path kernel variants in ONE process, interleaved rounds (GPU box only).
usage: ab_fused.py [N] [utterances] [K] [M]   -> one JSON line per variant (median / min loop ms)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exemplars_vc_amd as evc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
U = int(sys.argv[2]) if len(sys.argv) > 2 else 256
K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
M = int(sys.argv[4]) if len(sys.argv) > 4 else 25
ROUNDS = int(os.environ.get("AB_ROUNDS", "3"))
Tu = 688
T = U * Tu
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(20190131)
A = torch.rand(N, M, generator=g, device=dev, dtype=torch.float64) + 1e-3
A /= A.norm(dim=1, keepdim=True)
Hs = torch.rand(T, N, generator=g, device=dev, dtype=torch.float64)
Hs *= (torch.rand(T, N, generator=g, device=dev, dtype=torch.float64) < (8.0 / N))
X = (Hs @ A + 1e-6).contiguous(); del Hs
offs = np.arange(U + 1, dtype=np.int32) * Tu
H = torch.empty(T, N, dtype=torch.float64, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); e1.record(); torch.cuda.synchronize()

variants = {"all_resident": dict(), "res_or_coop": dict(all_resident=False),
            "res_nocoop": dict(all_resident=False, cooperative=False)}
if os.environ.get("AB_ONLY"):
    variants = {k: v for k, v in variants.items() if k in os.environ["AB_ONLY"].split(",")}
times = {k: [] for k in variants}
ref = None
for r in range(ROUNDS + 1):
    for name, kw in variants.items():
        evc.solve_activations(A, X, layout="frame_major", iters=K, eps_mode="zero_replace", init="sklearn",
                              utt_offsets=offs, out=H, loop_events=(e0, e1), **kw)
        torch.cuda.synchronize()
        if r > 0:
            times[name].append(e0.elapsed_time(e1))
        elif ref is None:
            ref = H[:64].clone()
        else:
            d = float(((H[:64] - ref).abs() / ref.abs().clamp_min(1e-300)).max())
            print(json.dumps({"variant": name, "max_rel_diff_vs_first": d}), flush=True)
fl = K * (4 * M * N + 3 * N) * T
for name, ts in times.items():
    med = float(np.median(ts))
    print(json.dumps({"variant": name, "N": N, "T": T, "K": K, "loop_ms_median": med, "loop_ms_min": min(ts),
                      "tflops_algorithmic": fl / (med * 1e-3) / 1e12, "frac_f64_peak": fl / (med * 1e-3) / 78.6e12}),
          flush=True)
