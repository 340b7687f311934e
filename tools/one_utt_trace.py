#!/usr/bin/env python3
"""One utterance of the wide flows on the default route (what the script's per-utterance `_factorize` call meets):
STFT (float32, M = 201, N = 4096, K = 150) and C3 (float64, M = 513, N = 8192, K = 200), 688 frames.  Prints whole-call
and loop times; under rocprofv3 --kernel-trace, tools/trace_gaps.py shows the kernels and the gaps of the last call.
    python tools/one_utt_trace.py [stft|c3] [utterances=1]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exemplars_vc_amd as evc

which = sys.argv[1] if len(sys.argv) > 1 else "stft"
U = int(sys.argv[2]) if len(sys.argv) > 2 else 1
M, N, K, dt, peak = (201, 4096, 150, torch.float32, 157.3e12) if which == "stft" else (513, 8192, 200, torch.float64, 78.6e12)
T = 688 * U
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(3)
A = (torch.rand(N, M, generator=g, device=dev, dtype=torch.float64) + 1e-3).to(dt)
X = torch.rand(T, M, generator=g, device=dev, dtype=torch.float64).to(dt)
offs = np.arange(U + 1, dtype=np.int32) * 688
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); ev1.record(); torch.cuda.synchronize()
best, bl = 1e9, 1e9
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, info = evc.solve_activations(A, X, layout="frame_major", iters=K, eps_mode="zero_replace", init="sklearn",
                                    utt_offsets=offs, info=True, loop_events=(ev0, ev1))
    torch.cuda.synchronize(); dtc = time.perf_counter() - t0
    if rep:
        best = min(best, dtc); bl = min(bl, ev0.elapsed_time(ev1) * 1e-3)
fl = K * (4.0 * M * N + 3.0 * N) * T
print(json.dumps({"flow": which, "utterances": U, "kernel": info["kernel"], "launches": int(info["launches"]), "call_ms": round(best * 1e3, 3),
                  "loop_ms": round(bl * 1e3, 3), "us_per_iteration": round(bl * 1e6 / K, 2), "frac_loop": round(fl / bl / peak, 4),
                  "frac_call": round(fl / best / peak, 4)}))
