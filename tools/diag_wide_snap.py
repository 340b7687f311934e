#!/usr/bin/env python3
"""k_fused_wide, STFT flow at 16 utterances, 150 iterations, residuals recorded every `check_every` iterations, no stop
rule (every variant runs the same iterations): what launches and in-launch checks cost.  (The one-launch-per-check lines of profiles/r04_wide_split_launch.md came from a build with the checks per launch
capped at 1.)  GPU box: python tools/diag_wide_snap.py"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exemplars_vc_amd as evc
M, N, Tu = 201, 4096, 688
U = 16
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(3)
A = (torch.rand(N, M, generator=g, device=dev) + 1e-3)
X = torch.rand(U * Tu, M, generator=g, device=dev)
offs = np.arange(U + 1, dtype=np.int32) * Tu
H = torch.empty(U * Tu, N, dtype=torch.float32, device=dev)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); ev1.record(); torch.cuda.synchronize()
for ce in (0, 50, 10, 30, 75):
    for rep in range(3):
        _, info = evc.solve_activations(A, X, layout="frame_major", iters=150, eps_mode="zero_replace", init="sklearn",
                                        utt_offsets=offs, out=H, check_every=ce, stop_rule="none", info=True, loop_events=(ev0, ev1))
        torch.cuda.synchronize()
    print(json.dumps({"check_every": ce, "loop_ms": round(ev0.elapsed_time(ev1), 3), "launches": info["launches"]}), flush=True)
