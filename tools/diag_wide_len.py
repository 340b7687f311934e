#!/usr/bin/env python3
"""k_fused_wide: time of ONE launch against its length (iterations), STFT flow.  python tools/diag_wide_len.py [utterances]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exemplars_vc_amd as evc

M, N, Tu = 201, 4096, 688
U = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(3)
A = (torch.rand(N, M, generator=g, device=dev) + 1e-3)
X = torch.rand(U * Tu, M, generator=g, device=dev)
offs = np.arange(U + 1, dtype=np.int32) * Tu
H = torch.empty(U * Tu, N, dtype=torch.float32, device=dev)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); ev1.record(); torch.cuda.synchronize()
prev = None
for n in (0, 1, 2, 3, 5, 10, 20, 30, 50, 75, 100, 150, 300):
    best = 1e9
    for rep in range(3):
        _, info = evc.solve_activations(A, X, layout="frame_major", iters=n, eps_mode="zero_replace", init="sklearn",
                                        utt_offsets=offs, out=H, check_every=0, stop_rule="none", info=True, loop_events=(ev0, ev1))
        torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1))
    print(json.dumps({"utterances": U, "iters": n, "loop_ms": round(best, 3), "launches": info["launches"],
                      "ms_per_added_iter": None if prev is None else round((best - prev[1]) / (n - prev[0]), 4)}), flush=True)
    prev = (n, best)
